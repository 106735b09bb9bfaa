// dev tool (not product): round-3 experiments on the memory skeleton of the resident n = 65536 kernel and on in-place
// streams in general.  Results: profiles/ubench2_*_r03.txt; reading: DESIGN.md sections 4 and 4.2.
//   hipcc -std=c++20 --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench2.hip -o /tmp/ubench2
//   /tmp/ubench2 [all = model + percu | model3 | model3b | model3c | maps | rev | rev2]
//
// model  : one 256-lane workgroup per CU (one wave per SIMD), a transform (256 rows x 2 KiB) at a time, in place.
//          Phase 1 reads 16 column blocks (128-byte row segments), DEPTH blocks in flight, one load per hook point
//          (16 hooks per block, W v_pk_fma between hooks, one LDS exchange + 2 barriers per block); phase 2 writes 16
//          column blocks, one store per hook.  Knobs: DEPTH (blocks in flight), LW / SW (bytes per lane of a load /
//          store: 16 = the half-waves take different rows, a v_permlane32_swap away from the 8-byte layout), W, each
//          stream switched off.  Per-phase s_memtime stamps.  Calibrates to the kernel (0.874 vs 0.865-0.885 ms).
// percu  : what ONE compute unit can read / write when only K of the 256 CUs are active.
// model3 : the same skeleton in groups of 2 / 4 ADJACENT column blocks whose rows are requested alternately
//          (256 / 512 contiguous bytes close in time), loads one group ahead, stores one group behind;
//          model3b: which stream gains, aligned vs straddling pairs, bursts; model3c: all rows of a block needed at
//          its start (as the real kernel needs them).
// maps   : other lane -> (column, row) bijections for the loads and stores of `model`.
// rev(2) : lane order of the accesses of an in-place stream of 64 KiB chunks (ascending, descending on odd rows,
//          descending, rotated), in place / out of place, non-temporal / plain.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

#define CK(x)                                                         \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                        \
    }                                                                 \
  } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int K> using ic = std::integral_constant<int, K>;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}

// one block in flight: 16 complex values per lane, as 16 x 8 B or 8 x 16 B
template <bool W16> struct Blk {
  f2 v[16];
};

// lane geometry.  8-byte accesses: lane = c + 16 t, access e at row t + 16 e, byte c * 8 of the segment.
// 16-byte accesses: inside a wave, lane = j + 8 tl + 32 up; the lane takes columns 2j, 2j+1 of row t + 16 (2 i + up),
// i = 0..7 (lower half-wave the even e, upper half-wave the odd e).
struct Geo {
  int voff8, voff16;
};
// MAP (8-byte accesses only): which (column c, row residue t) a lane takes — any bijection is free in the real kernel
// (the LDS exchanges re-label the lanes): 0 c = l & 15, t = l >> 4; 1 c reversed; 2 rows interleaved over the waves
// (wave w takes t = w, w + 4, w + 8, w + 12); 3 both; 4 t reversed; 5 c and t reversed; 6 c reversed on odd t
template <int MAP = 0> __device__ __forceinline__ Geo geo() {
  int l = threadIdx.x;
  asm volatile("" : "+v"(l));
  Geo g;
  int c = l & 15, t = l >> 4;
  if (MAP == 1 || MAP == 3 || MAP == 5) c = 15 - c;
  if (MAP == 2 || MAP == 3) t = (l >> 6) + 4 * ((l >> 4) & 3);
  if (MAP == 4 || MAP == 5) t = 15 - t;
  if (MAP == 6) c = (t & 1) ? 15 - c : c;
  g.voff8 = t * 2048 + c * 8;
  const int lam = l & 63, w = l >> 6, j = lam & 7, tl = (lam >> 3) & 3, up = lam >> 5;
  g.voff16 = (4 * w + tl + 16 * up) * 2048 + j * 16;
  return g;
}

template <bool W16, int K> __device__ __forceinline__ void load_one(Blk<W16> &b, __amdgpu_buffer_rsrc_t r, const Geo &g) {
  if constexpr (W16) {
    if constexpr (K < 8) {
      u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, g.voff16, K * 65536, 2);
      b.v[2 * K] = __builtin_bit_cast(f2, u2{x.x, x.y});
      b.v[2 * K + 1] = __builtin_bit_cast(f2, u2{x.z, x.w});
    }
  } else {
    b.v[K] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, g.voff8, K * 32768, 2));
  }
}
template <bool W16, int K> __device__ __forceinline__ void store_one(const f2 (&v)[16], __amdgpu_buffer_rsrc_t r, const Geo &g) {
  if constexpr (W16) {
    if constexpr (K < 8) {
      const u2 a = __builtin_bit_cast(u2, v[2 * K]), b = __builtin_bit_cast(u2, v[2 * K + 1]);
      __builtin_amdgcn_raw_buffer_store_b128(u4{a.x, a.y, b.x, b.y}, r, g.voff16, K * 65536, 2);
    }
  } else {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v[K]), r, g.voff8, K * 32768, 2);
  }
}

struct Acc {
  f2 a[8];
};
template <int W> __device__ __forceinline__ void busy(Acc &A) {
  const f2 mm = {0.999f, 1.001f}, cc = {1e-3f, -1e-3f};
#pragma unroll
  for (int i = 0; i < W; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(A.a[i & 7]) : "v"(mm), "v"(cc));
}

// LDS exchange like the kernel's: 8 x b128 written, barrier, 16 x b64 read, (barrier at the next block's start)
__device__ __forceinline__ void xchg(f2 (&v)[16], f2 *sx) {
  int l = threadIdx.x;
  asm volatile("" : "+v"(l));
  const int c = l & 15, t = l >> 4;
  __syncthreads();
  f4 *pw = reinterpret_cast<f4 *>(sx + c * 258 + 16 * t);
#pragma unroll
  for (int i = 0; i < 8; i++) pw[i] = f4{v[2 * i].x, v[2 * i].y, v[2 * i + 1].x, v[2 * i + 1].y};
  __syncthreads();
  const f2 *pr = sx + c * 258 + t;
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = pr[16 * e];
}

// one phase-1 block: consume zone Z (block cb), hooks issue the loads of block cb + DEPTH into the same zone
template <int W, bool LW16, bool XCHG, bool LOAD, bool NOMEM, int... K>
__device__ __forceinline__ void p1_block(Blk<LW16> &z, Acc &A, const char *x, int cbn, const Geo &g, f2 *sx,
                                         std::integer_sequence<int, K...>) {
  f2 cur[16];
#pragma unroll
  for (int e = 0; e < 16; e++) cur[e] = z.v[e];
  if constexpr (LW16) {   // the swap that turns the row-split 16-byte layout into the 8-byte one (cost only)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(cur[2 * i].x), "+v"(cur[2 * i + 1].x));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(cur[2 * i].y), "+v"(cur[2 * i + 1].y));
    }
  }
  const __amdgpu_buffer_rsrc_t r = rsrc(x + cbn * 128);
  auto hook = [&](auto k) {
    constexpr int kk = decltype(k)::value;
    busy<W>(A);
    if constexpr (kk == 7 && XCHG) {
      xchg(cur, sx);
    }
    A.a[kk & 7] += cur[kk];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LOAD && !NOMEM) {
      if constexpr (!LW16) load_one<false, kk>(z, r, g);
      else if constexpr (kk % 2 == 0) load_one<true, kk / 2>(z, r, g);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  (hook(ic<K>()), ...);
}
template <int W, bool SW16, bool XCHG, bool NOMEM, bool PRE, bool LW16, int... K>
__device__ __forceinline__ void p2_block(Acc &A, char *x, int cb, const Geo &g, f2 *sx, f2 (&st)[16],
                                         std::integer_sequence<int, K...>, Blk<LW16> *zn = nullptr, const char *xn = nullptr,
                                         const Geo *gl = nullptr) {
  // st = the previous block's results (parked); this block's arithmetic carries their stores
  const __amdgpu_buffer_rsrc_t r = rsrc(x + cb * 128);
  f2 cur[16];
#pragma unroll
  for (int e = 0; e < 16; e++) cur[e] = A.a[e & 7] + f2{(float)e, 1.f};
  auto hook = [&](auto k) {
    constexpr int kk = decltype(k)::value;
    busy<W>(A);
    if constexpr (kk == 7 && XCHG) xchg(cur, sx);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!NOMEM) {
      if constexpr (!SW16) store_one<false, kk>(st, r, g);
      else if constexpr (kk % 2 == 0) store_one<true, kk / 2>(st, r, g);
    }
    if constexpr (PRE) {
      const __amdgpu_buffer_rsrc_t rn = rsrc(xn);
      if constexpr (!LW16) load_one<false, kk>(*zn, rn, *gl);
      else if constexpr (kk % 2 == 1) load_one<true, kk / 2>(*zn, rn, *gl);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  (hook(ic<K>()), ...);
  if constexpr (SW16) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(cur[2 * i].x), "+v"(cur[2 * i + 1].x));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(cur[2 * i].y), "+v"(cur[2 * i + 1].y));
    }
  }
#pragma unroll
  for (int e = 0; e < 16; e++) st[e] = cur[e];
}

// MODE bits: 1 no loads, 2 no stores (what each stream costs)
template <int DEPTH, int W, bool LW16, bool SW16, bool XCHG, int MODE = 0, int MAPL = 0, int MAPS = 0>
__global__ __launch_bounds__(256) void k_model2(char *data, long mats, unsigned long long *dbg, float *sink) {
  __shared__ f2 sx[16 * 258 + 64];
  Acc A;
#pragma unroll
  for (int i = 0; i < 8; i++) A.a[i] = f2{1.0f + i, 0.5f};
  Blk<LW16> z[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
#pragma unroll
    for (int e = 0; e < 16; e++) z[d].v[e] = f2{0.f, 0.f};
  using S16 = std::make_integer_sequence<int, 16>;
  constexpr bool NOLD = MODE & 1, NOST = MODE & 2;
  unsigned long long c1 = 0, c2 = 0;
  long m = blockIdx.x;
  if (m >= mats) return;
  // prologue: the first DEPTH blocks of the first transform
  {
    const Geo g = geo<MAPL>();
    const char *x = data + m * 524288;
    auto pro = [&](auto zc) {
      constexpr int Z = decltype(zc)::value;
      const __amdgpu_buffer_rsrc_t r = rsrc(x + Z * 128);
      auto one = [&](auto k) {
        if constexpr (!NOLD) load_one<LW16, decltype(k)::value>(z[Z], r, g);
      };
      [&]<int... K>(std::integer_sequence<int, K...>) { (one(ic<K>()), ...); }(S16());
    };
    [&]<int... Z>(std::integer_sequence<int, Z...>) { (pro(ic<Z>()), ...); }(std::make_integer_sequence<int, DEPTH>());
  }
#pragma unroll 1
  for (; m < mats; m += gridDim.x) {
    char *x = data + m * 524288;
    long mn = m + gridDim.x;
    mn = mn < mats ? mn : m;
    const char *xn = data + mn * 524288;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // ---- phase 1
    constexpr int ROUNDS = (16 - DEPTH) / DEPTH;   // rounds of DEPTH blocks that still prefetch
#pragma unroll 1
    for (int rd = 0; rd < ROUNDS; rd++) {
      const Geo g = geo<MAPL>();
      [&]<int... Z>(std::integer_sequence<int, Z...>) {
        (p1_block<W, LW16, XCHG, true, NOLD>(z[Z], A, x, rd * DEPTH + Z + DEPTH, g, sx, S16()), ...);
      }(std::make_integer_sequence<int, DEPTH>());
    }
    // tail: the remaining blocks (ROUNDS * DEPTH .. 15), prefetching while a block DEPTH ahead exists
    {
      const Geo g = geo<MAPL>();
      [&]<int... I>(std::integer_sequence<int, I...>) {
        auto tail = [&](auto i) {
          constexpr int cb = ROUNDS * DEPTH + decltype(i)::value;
          if constexpr (cb < 16) {
            if constexpr (cb + DEPTH < 16) p1_block<W, LW16, XCHG, true, NOLD>(z[cb % DEPTH], A, x, cb + DEPTH, g, sx, S16());
            else p1_block<W, LW16, XCHG, false, NOLD>(z[cb % DEPTH], A, x, 0, g, sx, S16());
          }
        };
        (tail(ic<I>()), ...);
      }(std::make_integer_sequence<int, 2 * DEPTH>());
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    c1 += t1 - t0;
    // ---- phase 2: block k's arithmetic carries block k-1's stores; the last block's stores go out in a burst
    f2 st[16];
#pragma unroll
    for (int e = 0; e < 16; e++) st[e] = A.a[e & 7];
    {
      const Geo g = geo<MAPS>();
      p2_block<W, SW16, XCHG, true, false, LW16>(A, x, 0, g, sx, st, S16());
    }
#pragma unroll 1
    for (int cb = 1; cb < 16 - DEPTH; cb++) {
      const Geo g = geo<MAPS>();
      p2_block<W, SW16, XCHG, NOST, false, LW16>(A, x, cb - 1, g, sx, st, S16());
    }
    {
      // the last DEPTH row blocks also carry the loads of the next transform's first DEPTH column blocks
      const Geo g = geo<MAPS>(), gl = geo<MAPL>();
      [&]<int... Z>(std::integer_sequence<int, Z...>) {
        (p2_block<W, SW16, XCHG, NOST, !NOLD, LW16>(A, x, 16 - DEPTH + Z - 1, g, sx, st, S16(), &z[Z], xn + Z * 128, &gl), ...);
      }(std::make_integer_sequence<int, DEPTH>());
      const __amdgpu_buffer_rsrc_t r = rsrc(x + 15 * 128);
      if constexpr (!NOST) [&]<int... K>(std::integer_sequence<int, K...>) { (store_one<SW16, K>(st, r, g), ...); }(S16());
    }
    c2 += __builtin_amdgcn_s_memtime() - t1;
  }
  if (threadIdx.x == 0) {
    dbg[2 * blockIdx.x] = c1;
    dbg[2 * blockIdx.x + 1] = c2;
  }
  f2 s = A.a[0];
#pragma unroll
  for (int i = 1; i < 8; i++) s += A.a[i];
  if (s.x == 123.456f) *sink = s.y;
}

// ---------------------------------------------------------------- model3
// The same skeleton in GROUPS of GQ adjacent column blocks: a group's 16 GQ loads are issued while the previous
// group is transformed (LPH per hook point, so they are all out after 16 GQ / LPH hooks), its stores while the next
// group is transformed (one per hook).  QL / QS: order inside a group — true: row-major (row e of blocks g = 0..GQ-1
// back to back: GQ adjacent 128-byte segments = GQ x 128 contiguous bytes requested together), false: block-major
// (the same blocks in flight, but adjacent segments a block's length apart in time).
template <int W, int GQ, int LPH, bool QL, bool QS, bool XCHG, int MODE = 0, int POFF = 0, bool EARLY = false>
__global__ __launch_bounds__(256) void k_model3(char *data, long mats, unsigned long long *dbg, float *sink) {
  __shared__ f2 sx[16 * 258 + 64];
  constexpr bool NOLD = MODE & 1, NOST = MODE & 2;
  constexpr int NG = 16 / GQ, NA = 16 * GQ;
  Acc A;
#pragma unroll
  for (int i = 0; i < 8; i++) A.a[i] = f2{1.0f + i, 0.5f};
  f2 cur[GQ][16], nxt[GQ][16], prv[GQ][16];
#pragma unroll
  for (int g = 0; g < GQ; g++)
#pragma unroll
    for (int e = 0; e < 16; e++) cur[g][e] = nxt[g][e] = prv[g][e] = f2{0.f, 0.f};
  unsigned long long c1 = 0, c2 = 0;
  long m = blockIdx.x;
  if (m >= mats) return;
  // access number s of a group -> (block g, row e)
  auto issue_load = [&](auto sc, const char *x, int grp, const Geo &g) {
    constexpr int s_ = decltype(sc)::value;
    if constexpr (s_ < NA && !NOLD) {
      constexpr int gg = QL ? s_ % GQ : s_ / 16, ee = QL ? s_ / GQ : s_ % 16;
      nxt[gg][ee] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rsrc(x + ((grp * GQ + gg + POFF) & 15) * 128), g.voff8, ee * 32768, 2));
    }
  };
  auto issue_store = [&](auto sc, char *x, int grp, const Geo &g) {
    constexpr int s_ = decltype(sc)::value;
    if constexpr (s_ < NA && !NOST) {
      constexpr int gg = QS ? s_ % GQ : s_ / 16, ee = QS ? s_ / GQ : s_ % 16;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, prv[gg][ee]), rsrc(x + ((grp * GQ + gg + POFF) & 15) * 128), g.voff8, ee * 32768, 2);
    }
  };
  {
    const Geo g = geo();
    [&]<int... S>(std::integer_sequence<int, S...>) { (issue_load(ic<S>(), data + m * 524288, 0, g), ...); }(std::make_integer_sequence<int, NA>());
  }
#pragma unroll 1
  for (; m < mats; m += gridDim.x) {
    char *x = data + m * 524288;
    long mn = m + gridDim.x;
    mn = mn < mats ? mn : m;
    const char *xn = data + mn * 524288;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // ---- phase 1 (the last group's hooks fetch the NEXT transform's first group: no branch at the hook points; that
    // group then waits in registers through phase 2, which the model can afford)
#pragma unroll 1
    for (int grp = 0; grp < NG; grp++) {
      const Geo g = geo();
#pragma unroll
      for (int gg = 0; gg < GQ; gg++)
#pragma unroll
        for (int e = 0; e < 16; e++) cur[gg][e] = nxt[gg][e];
      const bool more = grp + 1 < NG;
      const char *xl = more ? x : xn;
      const int gl = more ? grp + 1 : 0;
      [&]<int... H>(std::integer_sequence<int, H...>) {
        auto hook = [&](auto hc) {
          constexpr int h = decltype(hc)::value, gg = h / 16, k = h % 16;
          if constexpr (EARLY && k == 0) {   // the real kernel takes ALL 16 rows of a block out of their landing zone at its start
#pragma unroll
            for (int e = 0; e < 16; e++) A.a[e & 7] += cur[gg][e];
            __builtin_amdgcn_sched_barrier(0);
          }
          busy<W>(A);
          if constexpr (k == 7 && XCHG) xchg(cur[gg], sx);
          if constexpr (!EARLY) A.a[k & 7] += cur[gg][k];
          __builtin_amdgcn_sched_barrier(0);
          [&]<int... I>(std::integer_sequence<int, I...>) { (issue_load(ic<h * LPH + I>(), xl, gl, g), ...); }(std::make_integer_sequence<int, LPH>());
          __builtin_amdgcn_sched_barrier(0);
        };
        (hook(ic<H>()), ...);
      }(std::make_integer_sequence<int, NA>());
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    c1 += t1 - t0;
    // ---- phase 2: group 0 carries no stores (peeled), groups 1.. carry the previous group's, the last group's go out in a burst
    auto p2_group = [&](auto stc, int grp) {
      constexpr bool ST = decltype(stc)::value;
      const Geo g = geo();
      f2 res[GQ][16];
      [&]<int... H>(std::integer_sequence<int, H...>) {
        auto hook = [&](auto hc) {
          constexpr int h = decltype(hc)::value, gg = h / 16, k = h % 16;
          if constexpr (k == 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) res[gg][e] = A.a[e & 7] + f2{(float)e, 1.f};
          }
          busy<W>(A);
          if constexpr (k == 7 && XCHG) xchg(res[gg], sx);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ST) issue_store(ic<h>(), x, grp - 1, g);
          __builtin_amdgcn_sched_barrier(0);
        };
        (hook(ic<H>()), ...);
      }(std::make_integer_sequence<int, NA>());
#pragma unroll
      for (int gg = 0; gg < GQ; gg++)
#pragma unroll
        for (int e = 0; e < 16; e++) prv[gg][e] = res[gg][e];
    };
    p2_group(std::false_type(), 0);
#pragma unroll 1
    for (int grp = 1; grp < NG; grp++) p2_group(std::true_type(), grp);
    {
      const Geo g = geo();
      [&]<int... S>(std::integer_sequence<int, S...>) { (issue_store(ic<S>(), x, NG - 1, g), ...); }(std::make_integer_sequence<int, NA>());
    }
    c2 += __builtin_amdgcn_s_memtime() - t1;
  }
  if (threadIdx.x == 0) {
    dbg[2 * blockIdx.x] = c1;
    dbg[2 * blockIdx.x + 1] = c2;
  }
  f2 s = A.a[0];
#pragma unroll
  for (int i = 1; i < 8; i++) s += A.a[i];
  if (s.x == 123.456f) *sink = s.y;
}

// ---------------------------------------------------------------- rev
// The packed real kernels store half of their bins in DESCENDING lane order (bin M - i of pair i).  Does the lane
// order of an 8-byte access cost anything?  512-lane workgroups, 64 KiB chunks in place: 16 loads v[e] = x[t + 512 e],
// then 16 stores; MODE 0 all ascending, 1 stores: 8 ascending + 8 descending (lane t -> 511 - t), 2 loads: 8 + 8
// descending (the c2r side), 3 both, 4 descending stores as 16 bytes per lane by half the lanes (pairs swapped in)
template <int MODE> __global__ __launch_bounds__(512) void k_rev(f2 *data, long chunks, float *sink) {
  const int t = threadIdx.x;
  for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
    f2 *x = data + c * 8192;
    f2 v[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const bool desc = (MODE == 2 || MODE == 3) && (e & 1);
      v[e] = __builtin_nontemporal_load(x + (desc ? 511 - t : t) + 512 * e);
      v[e].x += 1.0f;   // (an in-place copy of unchanged values is dead code)
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const bool desc = (MODE == 1 || MODE == 3 || MODE == 4) && (e & 1);
      if (MODE == 4 && desc) {
        // pairs of lanes (2k, 2k+1) hold adjacent descending elements: the even lane stores both (16 bytes)
        f2 o = v[e];
        f2 nb;
        nb.x = __shfl_down(o.x, 1);
        nb.y = __shfl_down(o.y, 1);
        if ((t & 1) == 0) __builtin_nontemporal_store(f4{nb.x, nb.y, o.x, o.y}, reinterpret_cast<f4 *>(x + (510 - t) + 512 * e));
      } else {
        __builtin_nontemporal_store(v[e], x + (desc ? 511 - t : t) + 512 * e);
      }
    }
  }
}

// ---------------------------------------------------------------- rev2
// LD / ST: lane order of the loads / stores of a 4 KiB row: 0 ascending, 1 descending on odd rows, 2 descending on
// all rows, 3 wave order rotated by the row (wave w takes segment (w + e) % 8 of row e: a wave's 16 accesses no
// longer fall 4 KiB apart); OOP: out of place; NT: non-temporal
template <int LD, int ST, bool OOP, bool NT> __global__ __launch_bounds__(512) void k_rev2(f2 *dst, f2 *src, long chunks, float *sink) {
  const int t = threadIdx.x;
  auto pos = [&](int mode, int e) {
    if (mode == 1) return (e & 1) ? 511 - t : t;
    if (mode == 2) return 511 - t;
    if (mode == 3) return (t + 64 * e) & 511;
    return t;
  };
  for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
    const f2 *x = src + c * 8192;
    f2 *y = (OOP ? dst : src) + c * 8192;
    f2 v[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const f2 *p = x + pos(LD, e) + 512 * e;
      v[e] = NT ? __builtin_nontemporal_load(p) : *p;
      v[e].x += 1.0f;
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
      f2 *p = y + pos(ST, e) + 512 * e;
      if (NT) __builtin_nontemporal_store(v[e], p);
      else *p = v[e];
    }
  }
}

// ---------------------------------------------------------------- modelB
// Another resident decomposition, N = 16 x 4096 instead of 256 x 256: phase 1 = 16-point transforms over the 16 rows of
// 4096 contiguous elements (no exchange; loads of 2 KiB contiguous per workgroup instruction, rows 32 KiB apart),
// phase 2 = sixteen 4096-point transforms from the keep matrix (3 passes, 2 exchanges each, NO memory traffic),
// phase 3 = one more exchange per block (the transposition that makes the stores contiguous) + fully contiguous
// stores (32 KiB per block).  Same arithmetic as the 256 x 256 form, 48 exchanges instead of 32; but every global
// access is part of a long contiguous run.  What would its memory skeleton deliver?
template <int DEPTH, int W, bool XCHG> __global__ __launch_bounds__(256) void k_modelB(char *data, long mats, unsigned long long *dbg, float *sink) {
  __shared__ f2 sx[16 * 258 + 64];
  Acc A;
#pragma unroll
  for (int i = 0; i < 8; i++) A.a[i] = f2{1.0f + i, 0.5f};
  f2 z[DEPTH][16];
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
#pragma unroll
    for (int e = 0; e < 16; e++) z[d][e] = f2{0.f, 0.f};
  unsigned long long c1 = 0, c2 = 0, c3 = 0;
  long m = blockIdx.x;
  if (m >= mats) return;
  auto voff = [&]() {
    int l = threadIdx.x;
    asm volatile("" : "+v"(l));
    return l * 8;
  };
  auto load_blk = [&](auto zc, const char *x, int j, int vo) {   // block j: rows e = 0..15, 2 KiB at j * 2 KiB of each
    constexpr int Z = decltype(zc)::value;
    const __amdgpu_buffer_rsrc_t r = rsrc(x + j * 2048);
    [&]<int... K>(std::integer_sequence<int, K...>) {
      ((z[Z][K] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, vo, K * 32768, 2))), ...);
    }(std::make_integer_sequence<int, 16>());
  };
  {
    const int vo = voff();
    [&]<int... Z>(std::integer_sequence<int, Z...>) { (load_blk(ic<Z>(), data + m * 524288, Z, vo), ...); }(std::make_integer_sequence<int, DEPTH>());
  }
#pragma unroll 1
  for (; m < mats; m += gridDim.x) {
    char *x = data + m * 524288;
    long mn = m + gridDim.x;
    mn = mn < mats ? mn : m;
    const char *xn = data + mn * 524288;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // ---- phase 1: 16 blocks, one 16-point transform + twiddles each, loads DEPTH blocks ahead (the last DEPTH blocks
    // fetch nothing: the next transform's first blocks are fetched during phase 3)
    auto p1 = [&](auto zc, int j, bool more) {
      constexpr int Z = decltype(zc)::value;
      const int vo = voff();
      f2 cur[16];
#pragma unroll
      for (int e = 0; e < 16; e++) cur[e] = z[Z][e];
      const __amdgpu_buffer_rsrc_t r = rsrc(x + (j + DEPTH) * 2048);
      [&]<int... K>(std::integer_sequence<int, K...>) {
        auto hook = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          busy<W / 2>(A);
          A.a[k & 7] += cur[k];
          __builtin_amdgcn_sched_barrier(0);
          if (more) z[Z][k] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, vo, k * 32768, 2));
          __builtin_amdgcn_sched_barrier(0);
        };
        (hook(ic<K>()), ...);
      }(std::make_integer_sequence<int, 16>());
    };
    constexpr int R1 = 16 / DEPTH;
#pragma unroll 1
    for (int rd = 0; rd < R1; rd++)
      [&]<int... Z>(std::integer_sequence<int, Z...>) { (p1(ic<Z>(), rd * DEPTH + Z, rd + 1 < R1), ...); }(std::make_integer_sequence<int, DEPTH>());
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    c1 += t1 - t0;
    // ---- phase 2: sixteen 4096-point transforms out of the keep matrix: 3 passes and 2 exchanges each, no memory
    f2 cur[16];
#pragma unroll
    for (int e = 0; e < 16; e++) cur[e] = A.a[e & 7];
#pragma unroll 1
    for (int k1 = 0; k1 < 16; k1++) {
#pragma unroll
      for (int pass = 0; pass < 3; pass++) {
#pragma unroll
        for (int h = 0; h < 16; h++) busy<W / 2>(A);
        if (pass < 2 && XCHG) xchg(cur, sx);
      }
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    c2 += t2 - t1;
    // ---- phase 3: 16 blocks: the transposing exchange, then 16 contiguous stores (2 KiB per instruction, 32 KiB per
    // block); the last DEPTH blocks also fetch the next transform's first DEPTH blocks
    auto p3 = [&](auto zc, int blk, bool pre) {
      constexpr int Z = decltype(zc)::value;
      const int vo = voff();
      if (XCHG) xchg(cur, sx);
      const __amdgpu_buffer_rsrc_t r = rsrc(x + blk * 32768), rn = rsrc(xn + Z * 2048);
      [&]<int... K>(std::integer_sequence<int, K...>) {
        auto hook = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          busy<4>(A);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, cur[k] + A.a[k & 7]), r, vo, k * 2048, 2);
          if (pre) z[Z][k] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rn, vo, k * 32768, 2));
          __builtin_amdgcn_sched_barrier(0);
        };
        (hook(ic<K>()), ...);
      }(std::make_integer_sequence<int, 16>());
    };
#pragma unroll 1
    for (int rd = 0; rd < R1; rd++)
      [&]<int... Z>(std::integer_sequence<int, Z...>) { (p3(ic<Z>(), rd * DEPTH + Z, rd + 1 == R1), ...); }(std::make_integer_sequence<int, DEPTH>());
    c3 += __builtin_amdgcn_s_memtime() - t2;
  }
  if (threadIdx.x == 0) {
    dbg[3 * blockIdx.x] = c1;
    dbg[3 * blockIdx.x + 1] = c2;
    dbg[3 * blockIdx.x + 2] = c3;
  }
  f2 s2 = A.a[0];
#pragma unroll
  for (int i = 1; i < 8; i++) s2 += A.a[i];
  if (s2.x == 123.456f) *sink = s2.y;
}

// ---------------------------------------------------------------- percu
// K workgroups (one per CU, K <= CUs), each streaming over its own matrices in 128-byte column blocks
// MODE 1 read only, 2 write only; W16: 16-byte accesses (row-split half-waves)
template <bool W16, int MODE> __global__ __launch_bounds__(256) void k_percu(char *data, long mats_per_wg, float *sink) {
  const Geo g = geo();
  f2 acc = {0.f, 0.f};
  f2 v[16];
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = f2{(float)e, (float)threadIdx.x};
  for (long i = 0; i < mats_per_wg; i++) {
    char *x = data + ((long)blockIdx.x * mats_per_wg + i) * 524288;
    for (int cb = 0; cb < 16; cb++) {
      const __amdgpu_buffer_rsrc_t r = rsrc(x + cb * 128);
      if constexpr (MODE == 1) {
        Blk<W16> b;
#pragma unroll
        for (int e = 0; e < 16; e++) b.v[e] = f2{0.f, 0.f};
        [&]<int... K>(std::integer_sequence<int, K...>) { (load_one<W16, K>(b, r, g), ...); }(std::make_integer_sequence<int, 16>());
#pragma unroll
        for (int e = 0; e < 16; e++) acc += b.v[e];
      } else {
        [&]<int... K>(std::integer_sequence<int, K...>) { (store_one<W16, K>(v, r, g), ...); }(std::make_integer_sequence<int, 16>());
      }
    }
  }
  if (acc.x == 123.456f) *sink = acc.y;
}

static float time_launches(int warm, int reps, const std::function<void()> &launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < warm; i++) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / reps;
}

int main(int argc, char **argv) {
  const char *what = argc > 1 ? argv[1] : "all";
  const bool all = !strcmp(what, "all");
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
  float *sink;
  CK(hipMalloc(&sink, 4096));
  const long mats = 4096;
  char *a;
  CK(hipMalloc(&a, mats * 524288));
  CK(hipMemset(a, 0, mats * 524288));
  unsigned long long *dbg;
  CK(hipMalloc(&dbg, 8192 * 8));
  std::vector<unsigned long long> hdbg(2 * cus);

  if (all || !strcmp(what, "model")) {
    printf("\n[model2] resident skeleton, 4096 transforms in place, one 256-lane workgroup per CU; W = v_pk_fma per hook (16 hooks\n"
           "         per block); ms per pass (median of 4 rounds x 30 launches, configurations interleaved), TB/s = 4 GiB / time,\n"
           "         kcycles per transform and CU in phase 1 / phase 2 (s_memtime)\n");
    struct Shape {
      char name[96];
      std::function<void()> launch;
      std::vector<float> ms;
      double p1, p2;
    };
    std::vector<Shape> shapes;
#define M2(DEPTH, W, LW16, SW16, XCHG, MODE)                                                                              \
  {                                                                                                                       \
    Shape s;                                                                                                              \
    snprintf(s.name, sizeof s.name, "depth %d  W %2d  load %2d B  store %2d B  xchg %d%s", DEPTH, W, LW16 ? 16 : 8,       \
             SW16 ? 16 : 8, (int)XCHG, MODE == 1 ? "  NO LOADS" : MODE == 2 ? "  NO STORES" : MODE == 3 ? "  NO MEMORY" : ""); \
    s.launch = [=] { hipLaunchKernelGGL((k_model2<DEPTH, W, LW16, SW16, XCHG, MODE>), dim3(cus), dim3(256), 0, 0, a, mats, dbg, sink); }; \
    shapes.push_back(s);                                                                                                  \
  }
    // calibration: the kernel as it is (depth 2, 8-byte accesses), with and without each stream
    M2(2, 22, false, false, true, 0) M2(2, 22, false, false, true, 1) M2(2, 22, false, false, true, 2) M2(2, 22, false, false, true, 3)
    // depth
    M2(1, 22, false, false, true, 0) M2(3, 22, false, false, true, 0) M2(4, 22, false, false, true, 0) M2(6, 22, false, false, true, 0)
    // widths
    M2(2, 22, false, true, true, 0) M2(2, 22, true, false, true, 0) M2(2, 22, true, true, true, 0)
    M2(3, 22, false, true, true, 0) M2(3, 22, true, true, true, 0) M2(4, 22, true, true, true, 0) M2(4, 22, false, true, true, 0)
    M2(3, 22, true, true, true, 1) M2(3, 22, true, true, true, 2)
    // no arithmetic: the skeleton alone
    M2(2, 0, false, false, false, 0) M2(4, 0, false, false, false, 0) M2(2, 0, true, true, false, 0) M2(4, 0, true, true, false, 0)
    // lighter / heavier arithmetic
    M2(3, 16, true, true, true, 0) M2(3, 28, true, true, true, 0) M2(2, 16, false, false, true, 0) M2(2, 28, false, false, true, 0)
    for (int round = 0; round < 4; round++)
      for (auto &sh : shapes) {
        sh.ms.push_back(time_launches(6, 30, sh.launch));
        CK(hipMemcpy(hdbg.data(), dbg, 2 * cus * 8, hipMemcpyDeviceToHost));
        double s1 = 0, s2 = 0;
        for (int i = 0; i < cus; i++) {
          s1 += hdbg[2 * i];
          s2 += hdbg[2 * i + 1];
        }
        sh.p1 = s1 / mats * 1e-3;
        sh.p2 = s2 / mats * 1e-3;
      }
    const double by = 2.0 * (double)mats * 524288;
    for (auto &sh : shapes) {
      std::vector<float> v = sh.ms;
      std::sort(v.begin(), v.end());
      printf("%-58s %7.3f ms (min %.3f)  %5.2f TB/s   p1 %5.1f  p2 %5.1f\n", sh.name, v[v.size() / 2], v.front(),
             by / v[v.size() / 2] * 1e-9, sh.p1, sh.p2);
    }
  }


  if (all || !strcmp(what, "model3")) {
    printf("\n[model3] as model2, in groups of GQ adjacent column blocks (loads one group ahead, LPH per hook; stores one group behind);\n"
           "         order inside a group: rows = row-major (GQ x 128 contiguous bytes requested together), blocks = block-major\n");
    struct Shape {
      char name[112];
      std::function<void()> launch;
      std::vector<float> ms;
      double p1, p2;
    };
    std::vector<Shape> shapes;
#define M3(W, GQ, LPH, QL, QS, XCHG, MODE)                                                                                \
  {                                                                                                                       \
    Shape s;                                                                                                              \
    snprintf(s.name, sizeof s.name, "GQ %d  W %2d  loads/hook %d  loads by %-6s stores by %-6s xchg %d%s", GQ, W, LPH,    \
             QL ? "rows" : "blocks", QS ? "rows" : "blocks", (int)XCHG,                                                   \
             MODE == 1 ? "  NO LOADS" : MODE == 2 ? "  NO STORES" : MODE == 3 ? "  NO MEMORY" : "");                      \
    s.launch = [=] { hipLaunchKernelGGL((k_model3<W, GQ, LPH, QL, QS, XCHG, MODE>), dim3(cus), dim3(256), 0, 0, a, mats, dbg, sink); }; \
    shapes.push_back(s);                                                                                                  \
  }
    M3(22, 1, 1, false, false, true, 0) M3(22, 2, 1, false, false, true, 0) M3(22, 2, 1, true, true, true, 0)
    M3(22, 4, 1, false, false, true, 0) M3(22, 4, 1, true, false, true, 0) M3(22, 4, 1, false, true, true, 0) M3(22, 4, 1, true, true, true, 0)
    M3(22, 4, 2, false, false, true, 0) M3(22, 4, 2, true, true, true, 0) M3(22, 4, 4, true, true, true, 0)
    M3(22, 4, 2, true, true, true, 1) M3(22, 4, 2, true, true, true, 2) M3(22, 4, 2, true, true, true, 3)
    M3(0, 4, 2, true, true, false, 0) M3(0, 4, 2, false, false, false, 0) M3(0, 2, 1, true, true, false, 0) M3(0, 1, 1, false, false, false, 0)
    M3(16, 4, 2, true, true, true, 0) M3(28, 4, 2, true, true, true, 0)
    for (int round = 0; round < 4; round++)
      for (auto &sh : shapes) {
        sh.ms.push_back(time_launches(6, 30, sh.launch));
        CK(hipMemcpy(hdbg.data(), dbg, 2 * cus * 8, hipMemcpyDeviceToHost));
        double s1 = 0, s2 = 0;
        for (int i = 0; i < cus; i++) {
          s1 += hdbg[2 * i];
          s2 += hdbg[2 * i + 1];
        }
        sh.p1 = s1 / mats * 1e-3;
        sh.p2 = s2 / mats * 1e-3;
      }
    const double by = 2.0 * (double)mats * 524288;
    for (auto &sh : shapes) {
      std::vector<float> v = sh.ms;
      std::sort(v.begin(), v.end());
      printf("%-86s %7.3f ms (min %.3f)  %5.2f TB/s   p1 %5.1f  p2 %5.1f\n", sh.name, v[v.size() / 2], v.front(),
             by / v[v.size() / 2] * 1e-9, sh.p1, sh.p2);
    }
  }

  if (!strcmp(what, "model3b")) {
    printf("\n[model3b] pairs of column blocks: which stream gains, aligned vs straddling pairs (POFF 1: pairs (1,2), (3,4), ...)\n");
    struct Shape {
      char name[112];
      std::function<void()> launch;
      std::vector<float> ms;
      double p1, p2;
    };
    std::vector<Shape> shapes;
#define M3B(W, GQ, LPH, QL, QS, XCHG, MODE, POFF)                                                                         \
  {                                                                                                                       \
    Shape s;                                                                                                              \
    snprintf(s.name, sizeof s.name, "GQ %d  W %2d  loads/hook %d  loads by %-6s stores by %-6s xchg %d  poff %d", GQ, W, LPH, \
             QL ? "rows" : "blocks", QS ? "rows" : "blocks", (int)XCHG, POFF);                                            \
    s.launch = [=] { hipLaunchKernelGGL((k_model3<W, GQ, LPH, QL, QS, XCHG, MODE, POFF>), dim3(cus), dim3(256), 0, 0, a, mats, dbg, sink); }; \
    shapes.push_back(s);                                                                                                  \
  }
    M3B(22, 2, 1, false, false, true, 0, 0) M3B(22, 2, 1, true, false, true, 0, 0) M3B(22, 2, 1, false, true, true, 0, 0)
    M3B(22, 2, 1, true, true, true, 0, 0) M3B(22, 2, 1, true, true, true, 0, 1) M3B(22, 2, 1, true, false, true, 0, 1)
    M3B(22, 2, 2, true, true, true, 0, 0) M3B(22, 1, 1, false, false, true, 0, 0) M3B(26, 2, 1, true, true, true, 0, 0)
    M3B(26, 2, 1, false, false, true, 0, 0) M3B(26, 1, 1, false, false, true, 0, 0)
    for (int round = 0; round < 5; round++)
      for (auto &sh : shapes) {
        sh.ms.push_back(time_launches(6, 30, sh.launch));
        CK(hipMemcpy(hdbg.data(), dbg, 2 * cus * 8, hipMemcpyDeviceToHost));
        double s1 = 0, s2 = 0;
        for (int i = 0; i < cus; i++) {
          s1 += hdbg[2 * i];
          s2 += hdbg[2 * i + 1];
        }
        sh.p1 = s1 / mats * 1e-3;
        sh.p2 = s2 / mats * 1e-3;
      }
    const double by = 2.0 * (double)mats * 524288;
    for (auto &sh : shapes) {
      std::vector<float> v = sh.ms;
      std::sort(v.begin(), v.end());
      printf("%-86s %7.3f ms (min %.3f)  %5.2f TB/s   p1 %5.1f  p2 %5.1f\n", sh.name, v[v.size() / 2], v.front(),
             by / v[v.size() / 2] * 1e-9, sh.p1, sh.p2);
    }
  }

  if (!strcmp(what, "rev")) {
    printf("\n[rev] 512-lane workgroups, 2 per CU, 64 KiB chunks in place (32768 chunks = 2 GiB), 8-byte accesses; TB/s (read + write)\n");
    const long chunks = 32768;
    const char *names[] = {"all ascending", "stores: 8 asc + 8 desc", "loads: 8 asc + 8 desc", "loads and stores 8 + 8 desc",
                           "desc stores as 16 B by even lanes"};
    float t[5][3];
    for (int round = 0; round < 3; round++) {
      t[0][round] = time_launches(5, 30, [&] { hipLaunchKernelGGL(k_rev<0>, dim3(2 * cus), dim3(512), 0, 0, (f2 *)a, chunks, sink); });
      t[1][round] = time_launches(5, 30, [&] { hipLaunchKernelGGL(k_rev<1>, dim3(2 * cus), dim3(512), 0, 0, (f2 *)a, chunks, sink); });
      t[2][round] = time_launches(5, 30, [&] { hipLaunchKernelGGL(k_rev<2>, dim3(2 * cus), dim3(512), 0, 0, (f2 *)a, chunks, sink); });
      t[3][round] = time_launches(5, 30, [&] { hipLaunchKernelGGL(k_rev<3>, dim3(2 * cus), dim3(512), 0, 0, (f2 *)a, chunks, sink); });
      t[4][round] = time_launches(5, 30, [&] { hipLaunchKernelGGL(k_rev<4>, dim3(2 * cus), dim3(512), 0, 0, (f2 *)a, chunks, sink); });
    }
    for (int m = 0; m < 5; m++)
      printf("%-40s %6.2f %6.2f %6.2f\n", names[m], 2.0 * chunks * 65536 / t[m][0] * 1e-9, 2.0 * chunks * 65536 / t[m][1] * 1e-9,
             2.0 * chunks * 65536 / t[m][2] * 1e-9);
  }

  if (!strcmp(what, "rev2")) {
    printf("\n[rev2] 512-lane workgroups, 2 per CU, 64 KiB chunks (32768 chunks = 2 GiB), 16 x 8-byte loads then 16 x 8-byte stores per lane;\n"
           "       lane order per 4 KiB row: asc, alt (descending on odd rows), desc, rot (wave w takes segment (w + row) %% 8); TB/s (read + write)\n");
    char *b2;
    CK(hipMalloc(&b2, mats * 524288));
    CK(hipMemset(b2, 0, mats * 524288));
    const long chunks = 32768;
    const char *on[] = {"asc", "alt", "desc", "rot"};
#define R2(LD, ST, OOP, NT)                                                                                           \
  {                                                                                                                   \
    float best = 1e9f;                                                                                                \
    for (int r = 0; r < 3; r++)                                                                                       \
      best = std::min(best, time_launches(5, 30, [&] { hipLaunchKernelGGL((k_rev2<LD, ST, OOP, NT>), dim3(2 * cus), dim3(512), 0, 0, (f2 *)b2, (f2 *)a, chunks, sink); })); \
    printf("loads %-4s stores %-4s %-12s %-5s  %6.2f\n", on[LD], on[ST], OOP ? "out of place" : "in place", NT ? "nt" : "plain", 2.0 * chunks * 65536 / best * 1e-9); \
  }
    R2(0, 0, false, true) R2(0, 0, true, true) R2(0, 0, false, false) R2(0, 0, true, false)
    R2(0, 1, false, true) R2(0, 1, true, true) R2(1, 0, false, true) R2(1, 0, true, true) R2(1, 1, false, true) R2(1, 1, true, true)
    R2(0, 2, false, true) R2(2, 0, false, true) R2(2, 2, false, true) R2(2, 2, true, true)
    R2(3, 3, false, true) R2(3, 3, true, true) R2(0, 3, false, true) R2(3, 0, false, true)
    CK(hipFree(b2));
  }

  if (!strcmp(what, "maps")) {
    printf("\n[maps] model2 (depth 2, W 22, 8-byte accesses, exchange on) with other lane -> (column, row) maps for the loads / stores:\n"
           "       0 as the kernel; 1 columns reversed; 2 rows interleaved over the waves; 3 = 1 + 2; 4 rows reversed; 5 = 1 + 4; 6 columns reversed on odd rows\n");
    struct Shape {
      char name[96];
      std::function<void()> launch;
      std::vector<float> ms;
      double p1, p2;
    };
    std::vector<Shape> shapes;
#define MP(ML, MS)                                                                                                        \
  {                                                                                                                       \
    Shape s;                                                                                                              \
    snprintf(s.name, sizeof s.name, "load map %d  store map %d", ML, MS);                                                 \
    s.launch = [=] { hipLaunchKernelGGL((k_model2<2, 22, false, false, true, 0, ML, MS>), dim3(cus), dim3(256), 0, 0, a, mats, dbg, sink); }; \
    shapes.push_back(s);                                                                                                  \
  }
    MP(0, 0) MP(0, 1) MP(1, 0) MP(1, 1) MP(0, 2) MP(2, 0) MP(2, 2) MP(0, 3) MP(3, 3) MP(0, 4) MP(4, 0) MP(0, 5) MP(5, 5) MP(0, 6) MP(6, 0) MP(6, 6)
    MP(1, 2) MP(2, 1) MP(4, 4)
    for (int round = 0; round < 4; round++)
      for (auto &sh : shapes) {
        sh.ms.push_back(time_launches(6, 30, sh.launch));
        CK(hipMemcpy(hdbg.data(), dbg, 2 * cus * 8, hipMemcpyDeviceToHost));
        double s1 = 0, s2 = 0;
        for (int i = 0; i < cus; i++) {
          s1 += hdbg[2 * i];
          s2 += hdbg[2 * i + 1];
        }
        sh.p1 = s1 / mats * 1e-3;
        sh.p2 = s2 / mats * 1e-3;
      }
    const double by = 2.0 * (double)mats * 524288;
    for (auto &sh : shapes) {
      std::vector<float> v = sh.ms;
      std::sort(v.begin(), v.end());
      printf("%-30s %7.3f ms (min %.3f)  %5.2f TB/s   p1 %5.1f  p2 %5.1f\n", sh.name, v[v.size() / 2], v.front(),
             by / v[v.size() / 2] * 1e-9, sh.p1, sh.p2);
    }
  }

  if (!strcmp(what, "model3c")) {
    printf("\n[model3c] pairs by rows when a block needs ALL its rows at its start (as the real kernel does: early = 1)\n");
    struct Shape {
      char name[112];
      std::function<void()> launch;
      std::vector<float> ms;
      double p1, p2;
    };
    std::vector<Shape> shapes;
#define M3C(W, GQ, LPH, QL, QS, EARLY)                                                                                    \
  {                                                                                                                       \
    Shape s;                                                                                                              \
    snprintf(s.name, sizeof s.name, "GQ %d  W %2d  loads/hook %d  loads by %-6s stores by %-6s early %d", GQ, W, LPH,     \
             QL ? "rows" : "blocks", QS ? "rows" : "blocks", (int)EARLY);                                                 \
    s.launch = [=] { hipLaunchKernelGGL((k_model3<W, GQ, LPH, QL, QS, true, 0, 0, EARLY>), dim3(cus), dim3(256), 0, 0, a, mats, dbg, sink); }; \
    shapes.push_back(s);                                                                                                  \
  }
    M3C(22, 1, 1, false, false, false) M3C(22, 1, 1, false, false, true) M3C(22, 2, 1, false, false, true) M3C(22, 2, 1, true, true, true)
    M3C(22, 2, 1, true, false, true) M3C(22, 2, 1, false, true, true) M3C(22, 2, 1, true, true, false) M3C(22, 4, 1, true, true, true)
    for (int round = 0; round < 5; round++)
      for (auto &sh : shapes) {
        sh.ms.push_back(time_launches(6, 30, sh.launch));
        CK(hipMemcpy(hdbg.data(), dbg, 2 * cus * 8, hipMemcpyDeviceToHost));
        double s1 = 0, s2 = 0;
        for (int i = 0; i < cus; i++) {
          s1 += hdbg[2 * i];
          s2 += hdbg[2 * i + 1];
        }
        sh.p1 = s1 / mats * 1e-3;
        sh.p2 = s2 / mats * 1e-3;
      }
    const double by = 2.0 * (double)mats * 524288;
    for (auto &sh : shapes) {
      std::vector<float> v = sh.ms;
      std::sort(v.begin(), v.end());
      printf("%-86s %7.3f ms (min %.3f)  %5.2f TB/s   p1 %5.1f  p2 %5.1f\n", sh.name, v[v.size() / 2], v.front(),
             by / v[v.size() / 2] * 1e-9, sh.p1, sh.p2);
    }
  }

  if (!strcmp(what, "modelB")) {
    printf("\n[modelB] resident skeleton of a 16 x 4096 decomposition (contiguous loads, one more exchange per block, contiguous stores);\n"
           "         kcycles per transform and CU in phase 1 (loads) / 2 (no memory) / 3 (stores)\n");
    std::vector<unsigned long long> h3(3 * cus);
    struct Shape {
      char name[96];
      std::function<void()> launch;
      std::vector<float> ms;
      double p1, p2, p3;
    };
    std::vector<Shape> shapes;
#define MB(DEPTH, W, XCHG)                                                                                                \
  {                                                                                                                       \
    Shape s;                                                                                                              \
    snprintf(s.name, sizeof s.name, "depth %d  W %2d  xchg %d", DEPTH, W, (int)XCHG);                                     \
    s.launch = [=] { hipLaunchKernelGGL((k_modelB<DEPTH, W, XCHG>), dim3(cus), dim3(256), 0, 0, a, mats, dbg, sink); };    \
    shapes.push_back(s);                                                                                                  \
  }
    MB(2, 22, true) MB(4, 22, true) MB(8, 22, true) MB(4, 16, true) MB(4, 0, false) MB(8, 0, false)
    Shape ref;
    snprintf(ref.name, sizeof ref.name, "256 x 256 (model2 depth 2, W 22) for comparison");
    ref.launch = [=] { hipLaunchKernelGGL((k_model2<2, 22, false, false, true, 0>), dim3(cus), dim3(256), 0, 0, a, mats, dbg, sink); };
    shapes.push_back(ref);
    for (int round = 0; round < 4; round++)
      for (auto &sh : shapes) {
        sh.ms.push_back(time_launches(6, 30, sh.launch));
        CK(hipMemcpy(h3.data(), dbg, 3 * cus * 8, hipMemcpyDeviceToHost));
        double s1 = 0, s2 = 0, s3 = 0;
        for (int i = 0; i < cus; i++) {
          s1 += h3[3 * i];
          s2 += h3[3 * i + 1];
          s3 += h3[3 * i + 2];
        }
        sh.p1 = s1 / mats * 1e-3;
        sh.p2 = s2 / mats * 1e-3;
        sh.p3 = s3 / mats * 1e-3;
      }
    const double by = 2.0 * (double)mats * 524288;
    for (auto &sh : shapes) {
      std::vector<float> v = sh.ms;
      std::sort(v.begin(), v.end());
      printf("%-52s %7.3f ms (min %.3f)  %5.2f TB/s   p1 %5.1f  p2 %5.1f  p3 %5.1f\n", sh.name, v[v.size() / 2], v.front(),
             by / v[v.size() / 2] * 1e-9, sh.p1, sh.p2, sh.p3);
    }
  }

  if (all || !strcmp(what, "percu")) {
    printf("\n[percu] K of the %d CUs active (one 256-lane workgroup each), 128-byte column blocks, nt; GB/s per CU\n", cus);
    printf("%-10s %12s %12s %12s %12s\n", "K", "read 8 B", "read 16 B", "write 8 B", "write 16 B");
    for (int K : {8, 32, 64, 128, 256}) {
      const long per = 4096 / 256;   // 16 matrices per workgroup whatever K: same work per CU
      float t[4];
      t[0] = time_launches(3, 10, [&] { hipLaunchKernelGGL((k_percu<false, 1>), dim3(K), dim3(256), 0, 0, a, per, sink); });
      t[1] = time_launches(3, 10, [&] { hipLaunchKernelGGL((k_percu<true, 1>), dim3(K), dim3(256), 0, 0, a, per, sink); });
      t[2] = time_launches(3, 10, [&] { hipLaunchKernelGGL((k_percu<false, 2>), dim3(K), dim3(256), 0, 0, a, per, sink); });
      t[3] = time_launches(3, 10, [&] { hipLaunchKernelGGL((k_percu<true, 2>), dim3(K), dim3(256), 0, 0, a, per, sink); });
      const double by = (double)per * 524288;
      printf("%-10d %12.1f %12.1f %12.1f %12.1f\n", K, by / t[0] * 1e-6, by / t[1] * 1e-6, by / t[2] * 1e-6, by / t[3] * 1e-6);
    }
  }
  return 0;
}
