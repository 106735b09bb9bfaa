// dev tool (not product): round-4 experiments on the memory skeleton of the resident n = 65536 kernel.
//   hipcc -std=c++20 --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench3.hip -o /tmp/ubench3
//   /tmp/ubench3 [model4 | oop | all]
//
// model4 : the resident skeleton in PAIRS of adjacent column blocks (as ubench2's model3, GQ = 2), with two new knobs:
//   * 256-byte segments: the lanes of a wave take (column 0..31 of the pair, 2 rows) instead of (column 0..15 of a
//     block, 4 rows): one wave instruction then asks for 2 x 256 contiguous bytes.  Lane = c + 16 ab + 32 r + 64 w;
//     access (e, s) reads row (4 w + 2 r + s) + 16 e, columns 16 ab + c of the pair.  One v_permlane16_swap per
//     dword (R[e][0] of the odd lane rows <-> R[e][1] of the even ones) turns that into the kernel's compute layout
//     (lane c + 16 t holds rows t + 16 e of ITS column in both blocks of the pair, t = 4 w + 2 r + ab): 32 swaps per
//     pair on the load side, 32 on the store side — the model pays them.
//   * shifted issue: a pair's 32 loads are issued from the second half of the pair two before it to the first half
//     of the pair before it, so that the youngest load is a block's time old when the pair starts (EARLY: a block's
//     16 rows are all consumed at its start, as the real kernel copies them out of their landing zone).
// oop    : every variant with dst != src (a second 2 GiB buffer).
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
template <int K> using ic = std::integral_constant<int, K>;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
struct Acc {
  f2 a[8];
};
template <int W> __device__ __forceinline__ void busy(Acc &A) {
  const f2 mm = {0.999f, 1.001f}, cc = {1e-3f, -1e-3f};
#pragma unroll
  for (int i = 0; i < W; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(A.a[i & 7]) : "v"(mm), "v"(cc));
}
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
struct Geo {
  int v128, v256;
};
__device__ __forceinline__ Geo geo() {
  int l = threadIdx.x;
  asm volatile("" : "+v"(l));
  Geo g;
  g.v128 = (l >> 4) * 2048 + (l & 15) * 8;
  const int w = l >> 6, lam = l & 63, r = lam >> 5;
  g.v256 = (4 * w + 2 * r) * 2048 + (lam & 31) * 8;
  return g;
}
__device__ __forceinline__ void swap16(f2 (&a)[16], f2 (&b)[16]) {
#pragma unroll
  for (int e = 0; e < 16; e++) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a[e].x), "+v"(b[e].x));
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[e].y), "+v"(b[e].y));
  }
}

// LM / SM: 0 a pair's accesses block by block (128-byte segments), 1 row by row (the two 128-byte halves of 256 contiguous
// bytes a hook apart), 2 256-byte segments (pair layout + swaps).  MODE bits: 1 no loads, 2 no stores.
// Pair P lands in zone P % 2 and is copied out of it when it starts (the real kernel's fetch out of its landing zone);
// SHIFT: hooks 0..15 of pair P issue accesses 16..31 of pair P + 1, hooks 16..31 accesses 0..15 of pair P + 2 (into the
// half of zone P % 2 that has just been copied out).
template <int W, int LM, int SM, bool SHIFT, bool EARLY, bool XCHG, int MODE = 0>
__global__ __launch_bounds__(256) void k_model4(const char *src, char *dst, long mats, unsigned long long *dbg, float *sink) {
  __shared__ f2 sx[16 * 258 + 64];
  constexpr bool NOLD = MODE & 1, NOST = MODE & 2;
  Acc A;
#pragma unroll
  for (int i = 0; i < 8; i++) A.a[i] = f2{1.0f + i, 0.5f};
  f2 buf[2][2][16];   // landing zones of two pairs (the model lets the compiler place them)
  f2 par[2][2][16];   // parked results of two pairs
#pragma unroll
  for (int z = 0; z < 2; z++)
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int e = 0; e < 16; e++) buf[z][g][e] = par[z][g][e] = f2{0.f, 0.f};
  unsigned long long c1 = 0, c2 = 0;
  // xcd != 0: the library's XCD-compact assignment (workgroups i, i + 8, ... share an XCD and take adjacent transforms)
  long m = (dbg[8000] && !(gridDim.x & 7)) ? (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  if (m >= mats) return;
  // access number s (0..31) of pair `grp` of the transform at x -> zone Z
  auto issue_load = [&](auto zc, auto sc, const char *x, int grp, const Geo &g) __attribute__((always_inline)) {
    constexpr int Z = decltype(zc)::value, s_ = decltype(sc)::value;
    if constexpr (!NOLD) {
      if constexpr (LM == 2) {
        constexpr int sb = s_ % 2, ee = s_ / 2;
        buf[Z][sb][ee] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rsrc(x + grp * 256), g.v256, ee * 32768 + sb * 2048, 2));
      } else {
        constexpr int gg = LM == 1 ? s_ % 2 : s_ / 16, ee = LM == 1 ? s_ / 2 : s_ % 16;
        buf[Z][gg][ee] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rsrc(x + (grp * 2 + gg) * 128), g.v128, ee * 32768, 2));
      }
    }
  };
  auto issue_store = [&](auto pc, auto sc, char *y, int grp, const Geo &g) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value, s_ = decltype(sc)::value;
    if constexpr (!NOST) {
      if constexpr (SM == 2) {
        constexpr int sb = s_ % 2, ee = s_ / 2;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, par[P][sb][ee]), rsrc(y + grp * 256), g.v256, ee * 32768 + sb * 2048, 2);
      } else {
        constexpr int gg = SM == 1 ? s_ % 2 : s_ / 16, ee = SM == 1 ? s_ / 2 : s_ % 16;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, par[P][gg][ee]), rsrc(y + (grp * 2 + gg) * 128), g.v128, ee * 32768, 2);
      }
    }
  };
  using S32 = std::make_integer_sequence<int, 32>;
  using S16 = std::make_integer_sequence<int, 16>;
  {   // prologue: pair 0 (and the first half of pair 1 when the issue is shifted)
    const Geo g = geo();
    const char *x = src + m * 524288;
    [&]<int... S>(std::integer_sequence<int, S...>) { (issue_load(ic<0>(), ic<S>(), x, 0, g), ...); }(S32());
    if constexpr (SHIFT) [&]<int... S>(std::integer_sequence<int, S...>) { (issue_load(ic<1>(), ic<S>(), x, 1, g), ...); }(S16());
  }
#pragma unroll 1
  for (; m < mats; m += gridDim.x) {
    const char *x = src + m * 524288;
    char *y = dst + m * 524288;
    long mn = m + gridDim.x;
    mn = mn < mats ? mn : m;
    const char *xn = src + mn * 524288;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // ---- phase 1
    auto p1_pair = [&](auto zc, int P) __attribute__((always_inline)) {
      constexpr int Z = decltype(zc)::value;
      const Geo g = geo();
      f2 cur[2][16];
#pragma unroll
      for (int gg = 0; gg < 2; gg++)
#pragma unroll
        for (int e = 0; e < 16; e++) cur[gg][e] = buf[Z][gg][e];
      if constexpr (LM == 2) swap16(cur[0], cur[1]);
      const int T1 = P + 1, T2 = P + 2;
      const char *x1 = T1 < 8 ? x : xn, *x2 = T2 < 8 ? x : xn;
      [&]<int... H>(std::integer_sequence<int, H...>) __attribute__((always_inline)) {
        auto hook = [&](auto hc) __attribute__((always_inline)) {
          constexpr int h = decltype(hc)::value, gg = h / 16, k = h % 16;
          if constexpr (EARLY && k == 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) A.a[e & 7] += cur[gg][e];
            __builtin_amdgcn_sched_barrier(0);
          }
          busy<W>(A);
          if constexpr (k == 7 && XCHG) xchg(cur[gg], sx);
          if constexpr (!EARLY) A.a[k & 7] += cur[gg][k];
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!SHIFT) issue_load(ic<1 - Z>(), ic<h>(), x1, T1 & 7, g);
          else if constexpr (h < 16) issue_load(ic<1 - Z>(), ic<16 + h>(), x1, T1 & 7, g);
          else issue_load(ic<Z>(), ic<h - 16>(), x2, T2 & 7, g);
          __builtin_amdgcn_sched_barrier(0);
        };
        (hook(ic<H>()), ...);
      }(S32());
      if constexpr (XCHG) A.a[0] += cur[0][3] + cur[1][5];
    };
#pragma unroll 1
    for (int pp = 0; pp < 8; pp += 2) {
      p1_pair(ic<0>(), pp);
      p1_pair(ic<1>(), pp + 1);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    c1 += t1 - t0;
    // ---- phase 2: pair P's results are parked in par[P % 2] and stored while pair P + 1 is computed
    auto p2_pair = [&](auto zc, auto stc, int P) __attribute__((always_inline)) {
      constexpr int Z = decltype(zc)::value;
      constexpr bool ST = decltype(stc)::value;
      const Geo g = geo();
      f2 res[2][16];
      [&]<int... H>(std::integer_sequence<int, H...>) __attribute__((always_inline)) {
        auto hook = [&](auto hc) __attribute__((always_inline)) {
          constexpr int h = decltype(hc)::value, gg = h / 16, k = h % 16;
          if constexpr (k == 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) res[gg][e] = A.a[e & 7] + f2{(float)e, 1.f};
          }
          busy<W>(A);
          if constexpr (k == 7 && XCHG) xchg(res[gg], sx);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ST) issue_store(ic<1 - Z>(), ic<h>(), y, P - 1, g);
          __builtin_amdgcn_sched_barrier(0);
        };
        (hook(ic<H>()), ...);
      }(S32());
      if constexpr (SM == 2) swap16(res[0], res[1]);
#pragma unroll
      for (int gg = 0; gg < 2; gg++)
#pragma unroll
        for (int e = 0; e < 16; e++) par[Z][gg][e] = res[gg][e];
    };
    p2_pair(ic<0>(), std::false_type(), 0);
    p2_pair(ic<1>(), std::true_type(), 1);
#pragma unroll 1
    for (int pp = 2; pp < 8; pp += 2) {
      p2_pair(ic<0>(), std::true_type(), pp);
      p2_pair(ic<1>(), std::true_type(), pp + 1);
    }
    {
      const Geo g = geo();
      [&]<int... S>(std::integer_sequence<int, S...>) { (issue_store(ic<1>(), ic<S>(), y, 7, g), ...); }(S32());
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
#pragma unroll
  for (int z = 0; z < 2; z++)
#pragma unroll
    for (int e = 0; e < 16; e++) s += buf[z][0][e] + buf[z][1][e];
  if (s.x == 123.456f) *sink = s.y;
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
  char *a, *b;
  CK(hipMalloc(&a, mats * 524288));
  CK(hipMalloc(&b, mats * 524288));
  CK(hipMemset(a, 0, mats * 524288));
  CK(hipMemset(b, 0, mats * 524288));
  unsigned long long *dbg;
  CK(hipMalloc(&dbg, 8192 * 8));
  CK(hipMemset(dbg, 0, 8192 * 8));
  std::vector<unsigned long long> hdbg(2 * cus);
  struct Shape {
    char name[128];
    std::function<void()> launch;
    std::vector<float> ms;
    double p1, p2;
  };
  std::vector<Shape> shapes;
  const char *mn[3] = {"blocks", "rows", "256 B"};
#define M4(W, LM, SM, SHIFT, EARLY, XCHG, MODE, OOP)                                                                      \
  {                                                                                                                       \
    Shape s;                                                                                                              \
    snprintf(s.name, sizeof s.name, "W %2d  loads %-6s stores %-6s shift %d early %d xchg %d%s%s", W, mn[LM], mn[SM],     \
             (int)SHIFT, (int)EARLY, (int)XCHG, MODE == 1 ? "  NO LOADS" : MODE == 2 ? "  NO STORES" : "",                \
             OOP ? "  OUT OF PLACE" : "");                                                                                \
    char *d_ = OOP ? b : a;                                                                                               \
    s.launch = [=] { hipLaunchKernelGGL((k_model4<W, LM, SM, SHIFT, EARLY, XCHG, MODE>), dim3(cus), dim3(256), 0, 0, a, d_, mats, dbg, sink); }; \
    shapes.push_back(s);                                                                                                  \
  }
#ifdef QUICK
  M4(22, 2, 2, true, true, true, 0, 0)
#else
  if (all || !strcmp(what, "model4")) {
    // the kernel's shape (blocks one after the other), then pairs by rows, then 256-byte segments
    M4(22, 0, 0, false, true, true, 0, 0) M4(22, 0, 0, true, true, true, 0, 0)
    M4(22, 1, 1, false, true, true, 0, 0) M4(22, 1, 1, true, true, true, 0, 0)
    M4(22, 2, 2, false, true, true, 0, 0) M4(22, 2, 2, true, true, true, 0, 0)
    M4(22, 2, 0, true, true, true, 0, 0) M4(22, 0, 2, true, true, true, 0, 0)
    M4(22, 2, 2, true, true, true, 1, 0) M4(22, 2, 2, true, true, true, 2, 0)
    M4(22, 0, 0, true, true, true, 1, 0) M4(22, 0, 0, true, true, true, 2, 0)
    // no arithmetic: the skeletons alone
    M4(0, 0, 0, true, false, false, 0, 0) M4(0, 1, 1, true, false, false, 0, 0) M4(0, 2, 2, true, false, false, 0, 0)
  }
  if (all || !strcmp(what, "oop")) {
    M4(22, 0, 0, true, true, true, 0, 1) M4(22, 1, 1, true, true, true, 0, 1) M4(22, 2, 2, true, true, true, 0, 1)
    M4(0, 0, 0, true, false, false, 0, 1) M4(0, 2, 2, true, false, false, 0, 1)
  }
#endif
  const unsigned long long xcd = getenv("UB_XCD") ? 1 : 0;
  CK(hipMemcpy(dbg + 8000, &xcd, 8, hipMemcpyHostToDevice));
  printf("\ntransform -> workgroup assignment: %s\n", xcd ? "XCD-compact (the library's since round 4)" : "workgroup i takes i, i + G, ... (rounds 1-3)");
  printf("\n[model4] resident skeleton in pairs of column blocks, 4096 transforms, one 256-lane workgroup per CU; W = v_pk_fma per\n"
         "         hook (16 hooks per block); ms per pass (median of 5 rounds x 30 launches, configurations interleaved),\n"
         "         TB/s = 4 GiB / time, kcycles per transform and CU in phase 1 / phase 2\n");
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
    printf("%-92s %7.3f ms (min %.3f)  %5.2f TB/s   p1 %5.1f  p2 %5.1f\n", sh.name, v[v.size() / 2], v.front(),
           by / v[v.size() / 2] * 1e-9, sh.p1, sh.p2);
  }
  return 0;
}
