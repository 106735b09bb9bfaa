// fft_kernels.hip — batched 1-D FFT kernels for gfx950 (MI355X).
//
// Replaces the reference's launch chain reorder + log2(N) x fft (+ conv/iconv)
// (cl_fft.cpp:24-41, 138-151, 178-205) by
//   k_fft_lds    one HBM pass: a transform (n <= 8192) lives in VGPRs + LDS of one
//                workgroup; r2c pack / c2r unpack fused (the two bins of a pair meet in
//                one lane's registers where the pass structure allows);
//   k_fft_4step  n = 2^14..2^16: N1 x N2 decomposition, both phases in ONE persistent
//                kernel, one 512-lane workgroup per CU; the intermediate stays on the CU —
//                two row blocks in LDS, up to ten in the registers of the lanes that
//                computed them, handed over through LDS — all of it for n <= 2^15, 3/4 at
//                n = 2^16 (the rest goes through a 512 KiB scratch slot per workgroup);
//   k_big_cols / k_big_transpose  n = 2^17..2^24, composed with the two above;
//   k_r2c_pack / k_c2r_unpack, k_reorder  stand-alone forms of the reference's
//                conv / iconv / reorder kernels.
#include <cstdlib>


#include "fft_wg.hpp"

#include <type_traits>

namespace clfa {
#ifdef CLFA_ASSIGN_SEARCH
// dev builds only (tools/assign_search.py): the transform -> workgroup assignment of k_fft_lds as a bit permutation:
// [0] bits of (workgroup | iteration << [1]) in use (0: off), [1] log2(grid), [2 + j] source bit of bit j of the transform index
__device__ int g_assign[40];
#endif


typedef float f4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------
// single-workgroup LDS FFT
// ---------------------------------------------------------------------------------

// Global loads of one transform into registers, in the order the first stage wants them:
//   C2C / R2C : v[e] = x[t + T*e]                         (coalesced, T apart)
//   C2R       : v[2k] = x[i], v[2k+1] = x[N-i], i = t + T*k (the pairs of the reference's iconv);
//               pair 0 of lane 0 is (x[0], x[N/2])
// every transform is read once and written once: non-temporal streams (copy kernels on this chip:
// 5.2 TB/s with nt vs 4.95 plain)
// (CLFA_NT_LD / CLFA_NT_ST: tuning switches for A/B builds; the library's choice is the default)
#ifndef CLFA_NT_LD
#define CLFA_NT_LD 1
#endif
#ifndef CLFA_NT_ST
#define CLFA_NT_ST 1
#endif
__device__ __forceinline__ cpx ld_nt(const cpx *p) {
  const unsigned long long raw = CLFA_NT_LD ? __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(p))
                                            : *reinterpret_cast<const unsigned long long *>(p);
  return *reinterpret_cast<const cpx *>(&raw);
}
__device__ __forceinline__ void st_nt(cpx *p, cpx v) {
  if (CLFA_NT_ST) __builtin_nontemporal_store(*reinterpret_cast<unsigned long long *>(&v), reinterpret_cast<unsigned long long *>(p));
  else *reinterpret_cast<unsigned long long *>(p) = *reinterpret_cast<unsigned long long *>(&v);
}

// Transforms owned by a whole workgroup (T >= 256 lanes, one transform per workgroup): the transform's base is
// wave-uniform, so its accesses go through a buffer descriptor — the lane's byte offset in ONE VGPR, everything
// else (the element stride of the pass, the mirrored position of a pair's partner) in the instruction's scalar
// offset.  With flat 64-bit addresses hipcc kept one address pair per access alive (16 pairs = 32 VGPRs for the
// two store streams of the packed real kernels, under a 128-VGPR cap) and rebuilt them every iteration.
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
// (measured, interleaved A/B against flat addressing: r2c + c2r of size 16384 0.213 -> 0.203-0.208 ms; the complex
// transforms, which have one ascending stream each way, lose 2 % at n = 8192 and stay as they were)
// (buffer addressing for the complex n = 8192 kernels was measured per direction as well: inverse 0.780 -> 0.835 ms,
// forward 0.789 -> 0.827 ms per 2 GiB — both lose, although the inverse instantiation carries a 20-byte spill on flat
// addresses)
template <int LOGN, int MODE, bool FWD = true>
constexpr bool kLdsBufAddr = LdsGeom<LOGN>::FPW == 1 && LOGN >= 12 && MODE != MODE_C2C;
struct XferBuf {
  __amdgpu_buffer_rsrc_t r;
  int va;   // t * 8: ascending positions t + c
  int vd;   // (T - t) * 8: descending positions c - t, as vd + (c - T) * 8
};
template <int LOGN> __device__ __forceinline__ XferBuf xfer_buf(const cpx *x, int t) {
  return XferBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<cpx *>(x), 0, 0x7fffffff, 0x00020000), t * 8,
                 (LdsGeom<LOGN>::T - t) * 8};
}
// Cache policy of the packed real kernels' loads: PLAIN loads, non-temporal stores.  Measured (interleaved A/B, steps
// alternating r2c / c2r, 1 GiB): size 16384 0.2054 -> 0.1974 ms (5.23 -> 5.44 TB/s) with both directions' loads plain,
// 0.2012 / 0.2027 with one of them; sizes 8192 and 32768 within 1 %.  (The complex kernels lose 2-9 % with plain
// loads and 3-8 % with plain stores: they keep non-temporal both ways — profiles/ab_cache_policy_r03.txt.)
// The same holds for the persistent four-step kernel (n = 2^14, 2^15: 4.80 -> 4.83, 4.82 -> 4.93 TB/s) and for packed
// real size 65536 (k_rfft_2x<14>: 3.80 -> 3.96 TB/s): plain loads, non-temporal stores.
#ifndef CLFA_NT_LD_4STEP
#define CLFA_NT_LD_4STEP 0
#endif
#ifndef CLFA_NT_LD_R15
#define CLFA_NT_LD_R15 0
#endif
#ifndef CLFA_NT_LD_R2C
#define CLFA_NT_LD_R2C 0
#endif
#ifndef CLFA_NT_LD_C2R
#define CLFA_NT_LD_C2R 0
#endif
template <bool NT> __device__ __forceinline__ cpx ld_buf(const XferBuf &b, int voff, int soff) {
  return __builtin_bit_cast(cpx, __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, NT ? 2 : 0));   // aux 2: non-temporal
}
__device__ __forceinline__ void st_buf(const XferBuf &b, int voff, int soff, cpx v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, v), b.r, voff, soff, CLFA_NT_ST ? 2 : 0);
}
// byte offsets (vector part, scalar part) of position i = pair_index(t, u, q) and of its partner N - i (N / 2 for
// i = 0) of a paired remainder pass (fft_device.hpp); the u = 0 pairs carry lane 0's exceptions in the vector part
template <int LOGN, int LOGE> struct PairOff {
  int vi, si, vj, sj;
};
template <int LOGN, int LOGE> __device__ __forceinline__ PairOff<LOGN, LOGE> pair_off(const XferBuf &b, int t, int u, int q) {
  constexpr int LOGR = pass_rem_logr(LOGN, LOGE), R = 1 << LOGR, T = 1 << (LOGN - LOGE), NB = 1 << (LOGN - LOGR), N = 1 << LOGN;
  PairOff<LOGN, LOGE> o;
  if (u == 0) {
    const int i = pair_index<LOGN, LOGE>(t, 0, q);
    o.vi = i * 8;
    o.vj = (i == 0 ? N / 2 : N - i) * 8;
    o.si = o.sj = 0;
  } else if (q < R / 2) {   // i = t + u T + NB q ascending, partner N - i descending
    o.vi = b.va;
    o.si = (u * T + NB * q) * 8;
    o.vj = b.vd;
    o.sj = (N - NB * q - u * T - T) * 8;
  } else {                  // i = NB (R - q) - (t + u T) descending, partner ascending
    o.vi = b.vd;
    o.si = (NB * (R - q) - u * T - T) * 8;
    o.vj = b.va;
    o.sj = (N - NB * (R - q) + u * T) * 8;
  }
  return o;
}

template <int LOGN, int MODE, bool FWD = true>
__device__ __forceinline__ void lds_fft_load(cpx (&v)[LdsGeom<LOGN>::E], const cpx *x, int t) {
  // No predicates on purpose: callers clamp the transform index instead.  Loads inside
  // exec-masked or even uniform branches make hipcc lose count of them and wait vmcnt(0) at the
  // join, i.e. for the prefetch it has just issued (seen in the ISA); straight-line loads get a
  // counted s_waitcnt vmcnt(N) and stay in flight behind the passes.
  using G = LdsGeom<LOGN>;
  constexpr int N = G::N, E = G::E, T = G::T;
  if constexpr (kLdsBufAddr<LOGN, MODE, FWD>) {
    const XferBuf b = xfer_buf<LOGN>(x, t);
    constexpr bool NT = MODE == MODE_C2C ? CLFA_NT_LD != 0 : (MODE == MODE_C2R ? CLFA_NT_LD_C2R : CLFA_NT_LD_R2C);
    if constexpr (MODE == MODE_C2R) {
#pragma unroll
      for (int k = 0; k < E / 2; k++) {
        if constexpr (pair_ok(LOGN, G::LOGE)) {   // pairs in the order pass_first_paired wants them
          constexpr int R = 1 << pass_rem_logr(LOGN, G::LOGE);
          const auto o = pair_off<LOGN, G::LOGE>(b, t, k / R, k % R);
          v[2 * k] = ld_buf<NT>(b, o.vi, o.si);
          v[2 * k + 1] = ld_buf<NT>(b, o.vj, o.sj);
        } else {
          v[2 * k] = ld_buf<NT>(b, b.va, T * k * 8);
          // partner N - (t + T k); pair 0 of lane 0 is (x[0], x[N/2])
          if (k == 0) v[1] = ld_buf<NT>(b, t == 0 ? (N / 2) * 8 : (N - t) * 8, 0);
          else v[2 * k + 1] = ld_buf<NT>(b, b.vd, (N - T * k - T) * 8);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; e++) v[e] = ld_buf<NT>(b, b.va, T * e * 8);
    }
    return;
  }
  if constexpr (MODE == MODE_C2R) {
#pragma unroll
    for (int k = 0; k < E / 2; k++) {
      int i = t + T * k;
      if constexpr (pair_ok(LOGN, G::LOGE)) {   // pairs in the order pass_first_paired wants them
        constexpr int R = 1 << pass_rem_logr(LOGN, G::LOGE);
        i = pair_index<LOGN, G::LOGE>(t, k / R, k % R);
      }
      v[2 * k] = ld_nt(x + i);
      v[2 * k + 1] = ld_nt(x + (i == 0 ? N / 2 : N - i));
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = ld_nt(x + t + T * e);
  }
}

#ifndef CLFA_LANE_SIGMA
#define CLFA_LANE_SIGMA 1   // n = 8192: the middle passes on permuted lanes (fft_wg.hpp, wg_passes_sigma): conflict-free gathers
#endif
// hipcc pairs neighbouring ds_read_b64 / ds_write_b64 into ds_read2(st64)_b64 / ds_write2_b64, which the LDS serves at half
// the bytes per clock (MI355X_MICROARCH.md, LDS table; fft_resident.hip has the whole story).  Kernels marked CLFA_DS_SINGLE_FN
// are compiled without that pass.  It is a property of the FUNCTION, not of an instantiation, and pays for some
// instantiations only (profiles/ab_ds_single_r05.txt: n = 32768 -2.0 %, packed real 32768 -1.3 .. -1.5 %, real 16384 0 .. -0.9 %;
// complex 8192 +2.1 %, 1024 +2.0 %, real 65536 +3.3 %): k_fft_4step has it, k_rfft_2x exists as one body and two kernels
// (`_s`: single LDS accesses, real size 32768), k_fft_lds and k_cfft_2x stay paired.
#if defined(__HIP_DEVICE_COMPILE__)
#define CLFA_DS_SINGLE_FN __attribute__((target("no-load-store-opt")))
#else
#define CLFA_DS_SINGLE_FN
#endif
template <int LOGN, bool FWD, int MODE, bool SCALE>
__device__ __forceinline__ void fft_lds_body(cpx *__restrict__ data, const cpx *__restrict__ tab_g, const cpx *__restrict__ w2_g,
                                             long batch, long out_off) {
  // out_off: results go to data + out_off (complex elements; 0 = in place, else a disjoint destination: clfa_fft_exec_dev_oop)
  using G = LdsGeom<LOGN>;
  constexpr int N = G::N, E = G::E, T = G::T, WG = G::WG, FPW = G::FPW;
  // twiddles in LDS: half table W_n^k (k < n/2); n = 8192: the lane-addressed tables of LaneTab13
  // (fft_device.hpp: 1280 entries, which keeps the block at 78 KiB so that two workgroups share a CU)
  constexpr bool TWO = kLdsTwoLevel(LOGN);
  constexpr int NTAB = TWO ? kLaneLds : G::HALF;
  __shared__ cpx s_tab[NTAB];
  __shared__ cpx s_x[FPW * G::PADN];
  // packed real size 8192 (n = 4096): the twiddles of the pass that starts at 16 points from a 16 x 16 table (HalfRowTab)
  constexpr bool ROW16 = !TWO && LOGN == 12 && MODE != MODE_C2C;
  __shared__ cpx s_row[ROW16 ? kRow16Lds : 1];

  const int tid = threadIdx.x;
  const int f = FPW == 1 ? 0 : tid / T, t = FPW == 1 ? tid : tid % T;   // (FPW == 1: the base stays provably uniform)
  const long groups = (batch + FPW - 1) / FPW;
  long g = blockIdx.x;
#ifdef CLFA_ASSIGN_SEARCH
  long kk = 0;
  int a_sh[16];   // read once (scalar loads), then SGPRs
  const bool a_on = g_assign[0] && (1u << g_assign[1]) == gridDim.x;
  const int a_lg = g_assign[1];
#pragma unroll
  for (int j = 0; j < 16; j++) a_sh[j] = j < g_assign[0] ? g_assign[2 + j] : 63;
  auto amap = [&](long k) -> long {
    if (!a_on) return blockIdx.x + k * gridDim.x;
    const long vv = blockIdx.x | (k << a_lg);
    long r = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) r |= ((vv >> a_sh[j]) & 1) << j;
    r = r < groups ? r : groups - 1;   // a table that does not fit the launch must not leave the buffer
    return __builtin_amdgcn_readfirstlane((int)r);
  };
  g = amap(0);
  const long per_ = (groups + gridDim.x - 1) / gridDim.x;
#endif
  if (g >= groups) return;   // whole workgroup (uniform): launchers never over-provision the grid
  // the first transform's loads are issued before anything else: they fly while the tables are filled
  cpx v[E], vn[E];
  {
    const long b = g * FPW + f;
    lds_fft_load<LOGN, MODE, FWD>(v, data + (b < batch ? b : batch - 1) * (long)N, t);
  }
  for (int i = tid; i < (TWO ? kLane13Lds : N / 2); i += WG) s_tab[TWO ? lane_lds_index(i) : i] = tab_g[i];
  if constexpr (ROW16) lds_fill_row16<LOGN>(s_row, tab_g, tid, WG);
  cpx *xb = s_x + f * G::PADN;
  // the lane's own twiddle constants: W_8192^t (n = 8192); W_16384^t, ^(2 t), ^(3 t) (n = 16384)
  cpx wl[LOGN == 14 ? 3 : 1];
  wl[0] = mk(1.f, 0.f);
  if constexpr (TWO) {
#pragma unroll
    for (int k = 0; k < (LOGN == 14 ? 3 : 1); k++) wl[k] = tab_g[kLane13Lds + k * T + t];
  }

  // pack / unpack twiddles of this lane's pairs are the same for every transform
  constexpr int NP = (MODE == MODE_C2C) ? 1 : (E / 2 > 0 ? E / 2 : 1);
  // packed real transforms pair bins i, N-i inside the remainder pass when it has two butterflies
  // per lane (fft_device.hpp, pass_last_paired / pass_first_paired): one LDS exchange less
  constexpr bool PAIRED = MODE != MODE_C2C && pair_ok(LOGN, G::LOGE);
  constexpr int RREM = 1 << pass_rem_logr(LOGN, G::LOGE);
  // n = 8192 (M of config 3): the lane's eight pack twiddles W_16384^i, i = t + 512 u and 4096 - i, all
  // derive from ONE lane constant (g0 = w2[t]) times compile-time constants W_32^u, the partners being
  // -+i conj(.) — 2 VGPRs across the batch loop instead of 16 (the kernel runs under a 128-VGPR cap)
  constexpr bool W2LANE = TWO && PAIRED;
  // the pair maps' factors of 1/2 folded away (fft_device.hpp, r2c_pair_prescaled / c2r_pair_halfw): the forward
  // kernel scales its 16 values by 1 / (2N) instead of 1/N, the inverse kernel keeps its pair twiddles halved
  constexpr bool HALFW = MODE == MODE_C2R && PAIRED && LOGN != 14;   // (pair_tw14 carries unscaled constants for lane 0)
  constexpr bool PRESC = MODE == MODE_R2C && PAIRED;
  constexpr float wsc = HALFW ? 0.5f : 1.0f;
  cpx w2r[W2LANE ? 1 : NP];
  if constexpr (W2LANE) {
    w2r[0] = cscale(w2_g[t], wsc);
  } else if constexpr (MODE != MODE_C2C) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
      if constexpr (PAIRED) w2r[k] = cscale(w2_g[pair_index<LOGN, G::LOGE>(t, k / RREM, k % RREM)], wsc);
      else w2r[k] = w2_g[t + T * k];
    }
  }
  // pair k = 2 u + q of the lane (pair_index): q = 0 -> w2[t + 512 u], q = 1 -> w2[4096 - (t + 512 u)]
  // (lane 0, u = 0: w2[2048] = W_8, with the table's sign)
  auto w2_of = [&](int k, int lane) -> cpx {
    if constexpr (W2LANE && LOGN == 14) {
      // pair k = 4 u + q of the lane: w2[i] = W_32768^i, i = pair_index(t, u, q), from the lane constant w2[t]
      return pair_tw14<FWD, 0>(w2r[0], k >> 2, k & 3, lane);
    } else if constexpr (W2LANE) {
      constexpr float c32[4] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f};
      constexpr float s32[4] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f};
      const int u = k >> 1;
      cpx w = w2r[0];
      if (u == 1) w = ctw<FWD>(w, c32[1], s32[1]);
      if (u == 2) w = ctw<FWD>(w, c32[2], s32[2]);
      if (u == 3) w = ctw<FWD>(w, c32[3], s32[3]);
      if (k & 1) {
        w = FWD ? mk(-w.y, -w.x) : mk(w.y, w.x);   // W^(4096 - i) = -i conj(W^i) (forward sign), +i conj (inverse)
        if (k == 1 && lane == 0) w = mk(kC8 * wsc, (FWD ? -kC8 : kC8) * wsc);
      }
      return w;
    } else {
      return w2r[k];
    }
  };
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; e++) asm volatile("" : "+v"(v[e]));
  const int t_invariant = t;
#pragma unroll 1
#ifdef CLFA_ASSIGN_SEARCH
  for (; kk < per_; g = amap(++kk)) {
#else
  for (; g < groups; g += gridDim.x) {
#endif
    // Re-derive the lane index inside the loop through an opaque move: otherwise hipcc hoists every
    // LDS scatter/gather offset and global offset of all passes out of the batch loop, keeps
    // ~100 of them live across it and spills them (seen in the ISA as scratch stores in the
    // prologue and scratch loads in the loop).  Recomputing them costs a few VALU instructions.
    int t = t_invariant;
    asm volatile("" : "+v"(t));
    const auto tab2 = [&]() {
      if constexpr (LOGN == 14) return LaneTab14{s_tab + kRow16Stride * (t & 15), s_tab + kRow16Lds + (t & 255), wl[0], wl[1], wl[2]};
      else return LaneTab13{s_tab + kRow16Stride * (t & 15), s_tab + kRow16Lds + (t & 255), wl[0]};
    }();
    const auto tab1 = [&]() {
      if constexpr (ROW16) return HalfRowTab{s_tab, s_row + kRow16Stride * (t & 15)};
      else return static_cast<const cpx *>(s_tab);
    }();
    const long b = g * FPW + f;
    const bool active = b < batch;
    cpx *x = data + (active ? b : batch - 1) * (long)N;
    // software prefetch: the next transform's loads fly while this one is in the passes.
    // Always issued (index clamped to the last transform) so that it is straight-line code.
    // (every LDS size has it — LdsGeom::PREFETCH; at n = 8192 it fits under the 128-VGPR cap and is worth 5 %)
    if constexpr (G::PREFETCH) {
      long gn = g + gridDim.x;
      gn = gn < groups ? gn : groups - 1;
#ifdef CLFA_ASSIGN_SEARCH
      gn = kk + 1 < per_ ? amap(kk + 1) : g;
#endif
      const long bn = gn * FPW + f;
      lds_fft_load<LOGN, MODE, FWD>(vn, data + (bn < batch ? bn : batch - 1) * (long)N, t);
    }
    if constexpr (MODE == MODE_C2R && PAIRED) {
      // fused reference `iconv` (cl_fft.cpp:192-205) in registers, then the transposed pass chain
      cpx oi[E / 2], oj[E / 2];
#pragma unroll
      for (int k = 0; k < E / 2; k++) {
        const int i = pair_index<LOGN, G::LOGE>(t, k / RREM, k % RREM);
        if constexpr (HALFW) c2r_pair_halfw(v[2 * k], v[2 * k + 1], w2_of(k, t), oi[k], oj[k]);
        else c2r_pair(v[2 * k], v[2 * k + 1], w2_of(k, t), oi[k], oj[k]);
        if (k == 0) {   // lane 0: packed DC/Nyquist, bin N/2 copied through (selects, not a branch)
          const bool z = i == 0;
          oi[0] = mk(z ? v[0].x + v[0].y : oi[0].x, z ? v[0].x - v[0].y : oi[0].y);
          oj[0] = mk(z ? v[1].x : oj[0].x, z ? v[1].y : oj[0].y);
        }
      }
      if constexpr (TWO) pass_first_paired<LOGN, G::LOGE, FWD>(v, t, oi, oj, tab2);
      else pass_first_paired<LOGN, G::LOGE, FWD>(v, t, oi, oj, tab1);
      __syncthreads();
      pass_first_paired_scatter<LOGN, G::LOGE>(v, t, xb);
      __syncthreads();
      constexpr int L1 = pass_last_logns(LOGN, G::LOGE) - G::LOGE;
      if constexpr (TWO) wg_passes_dif_after<LOGN, G::LOGE, L1, FWD>(v, t, tab2, xb);
      else wg_passes_dif_after<LOGN, G::LOGE, L1, FWD>(v, t, tab1, xb);
    } else {
      if constexpr (MODE == MODE_C2R) {
        // fused reference `iconv` (cl_fft.cpp:192-205) on the way in
        __syncthreads();
#pragma unroll
        for (int k = 0; k < E / 2; k++) {
          const int i = t + T * k;
          if (i == 0) {
            xb[0] = mk(v[0].x + v[0].y, v[0].x - v[0].y);
            xb[lds_pad(N / 2)] = v[1];
          } else {
            cpx oi, oj;
            c2r_pair(v[2 * k], v[2 * k + 1], w2r[k], oi, oj);
            xb[lds_pad(i)] = oi;
            xb[lds_pad(N - i)] = oj;
          }
        }
        __syncthreads();
        pass_gather_padded<LOGN, G::LOGE>(v, t, xb);
      }
      constexpr bool PL = MODE == MODE_R2C && PAIRED;
      if constexpr (TWO && CLFA_LANE_SIGMA && FPW == 1) {
        const int ts = lane_sigma(t);
        const auto tab2s = [&]() {
          if constexpr (LOGN == 14) return LaneTab14{s_tab + kRow16Stride * (ts & 15), s_tab + kRow16Lds + (ts & 255), wl[0], wl[1], wl[2]};
          else return LaneTab13{s_tab + kRow16Stride * (ts & 15), s_tab + kRow16Lds + (ts & 255), wl[0]};
        }();
        wg_passes_sigma<LOGN, G::LOGE, 0, FWD, PL>(v, t, ts, tab2, tab2s, xb);
      } else if constexpr (TWO) wg_passes<LOGN, G::LOGE, 0, FWD, PL>(v, t, tab2, xb);
      else wg_passes<LOGN, G::LOGE, 0, FWD, PL>(v, t, tab1, xb);
    }

    if constexpr (SCALE || PRESC) {
      constexpr float inv = (SCALE ? 1.0f / (float)N : 1.0f) * (PRESC ? 0.5f : 1.0f);
#pragma unroll
      for (int e = 0; e < E; e++) v[e] = cscale(v[e], inv);
    }

    // Stores are unconditional as well (same reason as the loads).  Lanes of a ragged last
    // group whose transform index is past the batch were clamped to the LAST transform: they
    // loaded the same input in the same instruction as its owner and store bit-identical output.
    (void)active;
    [[maybe_unused]] XferBuf xo{};
    x += out_off;   // every access from here on is a store of this transform's results
    if constexpr (kLdsBufAddr<LOGN, MODE, FWD>) xo = xfer_buf<LOGN>(x, t);
    if constexpr (MODE == MODE_R2C && PAIRED) {
      // fused reference `conv` (cl_fft.cpp:178-191): both bins of every pair are in this lane's registers
      pairs_visit<LOGN, G::LOGE>(v, t, [&](int k, int i, cpx ci, cpx cj) {
        const int j = i == 0 ? N / 2 : N - i;
        cpx oi, oj;
        r2c_pair_prescaled(ci, cj, w2_of(k, t), oi, oj);   // (ci, cj carry the map's 1/2 already)
        if (k == 0 && i == 0) {   // packed DC/Nyquist; bin N/2 copied through
          oi = mk(ci.x + ci.y, ci.x - ci.y);
          oj = cscale(cj, 2.0f);
        }
        if constexpr (kLdsBufAddr<LOGN, MODE, FWD>) {
          const auto o = pair_off<LOGN, G::LOGE>(xo, t, k / RREM, k % RREM);
          st_buf(xo, o.vi, o.si, oi);
          st_buf(xo, o.vj, o.sj, oj);
        } else {
          st_nt(x + i, oi);
          st_nt(x + j, oj);
        }
      });
    } else if constexpr (MODE == MODE_R2C) {
      // fused reference `conv` (cl_fft.cpp:178-191) on the way out
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; e++) xb[lds_pad(t + T * e)] = v[e];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < E / 2; k++) {
        // branch-free: pair 0 is (bin 0 packed DC/Nyquist, bin N/2 copied through), selected by value
        const int i = t + T * k;
        const int j = i == 0 ? N / 2 : N - i;
        const cpx ci = xb[lds_pad(i)], cj = xb[lds_pad(j)];
        cpx oi, oj;
        r2c_pair(ci, cj, w2r[k], oi, oj);
        if (i == 0) {
          oi = mk((ci.x + ci.y) * .5f, (ci.x - ci.y) * .5f);
          oj = cj;
        }
        if constexpr (kLdsBufAddr<LOGN, MODE, FWD>) {
          st_buf(xo, xo.va, T * k * 8, oi);
          if (k == 0) st_buf(xo, t == 0 ? (N / 2) * 8 : (N - t) * 8, 0, oj);
          else st_buf(xo, xo.vd, (N - T * k - T) * 8, oj);
        } else {
          st_nt(x + i, oi);
          st_nt(x + j, oj);
        }
      }
    } else if constexpr (kLdsBufAddr<LOGN, MODE, FWD>) {
#pragma unroll
      for (int e = 0; e < E; e++) st_buf(xo, xo.va, T * e * 8, v[e]);
    } else {
#pragma unroll
      for (int e = 0; e < E; e++) st_nt(x + t + T * e, v[e]);
    }
    // Consume the prefetch HERE, in straight-line code after the stores: hipcc then waits with an
    // exact s_waitcnt vmcnt(<stores still in flight>).  If the first use were at the loop top, the
    // wait would be merged with the loop-entry path and drain this iteration's stores as well.
    if constexpr (G::PREFETCH) {
#pragma unroll
      for (int e = 0; e < E; e++) {
        asm volatile("" : "+v"(vn[e]));
        v[e] = vn[e];
      }
    } else {
      // no prefetch: load the next transform now (clamped, straight-line)
      long gn = g + gridDim.x;
      gn = gn < groups ? gn : groups - 1;
#ifdef CLFA_ASSIGN_SEARCH
      gn = kk + 1 < per_ ? amap(kk + 1) : g;
#endif
      const long bn = gn * FPW + f;
      lds_fft_load<LOGN, MODE, FWD>(v, data + (bn < batch ? bn : batch - 1) * (long)N, t);
    }
  }
}

template <int LOGN, bool FWD, int MODE, bool SCALE>
__global__ __launch_bounds__(LdsGeom<LOGN>::WG, LdsGeom<LOGN>::MIN_WAVES) void k_fft_lds(cpx *__restrict__ data,
                                                              const cpx *__restrict__ tab_g,
                                                              const cpx *__restrict__ w2_g, long batch, long out_off) {
  fft_lds_body<LOGN, FWD, MODE, SCALE>(data, tab_g, w2_g, batch, out_off);
}

// ---------------------------------------------------------------------------------
// packed real size 65536 (n = 32768 complex): two runs of the 16384-point LDS machinery per transform
// ---------------------------------------------------------------------------------
// n = 2 M, M = 16384: the even and odd complex samples z[2j], z[2j+1] — one 16-byte access per lane — go
// through the 16384-point pass chain one after the other (same 1024 lanes, same exchange buffer); the
// radix-2 step that joins them and the reference's pair map (cl_fft.cpp:178-205) meet in registers
// (fft_device.hpp, rfft2x_fwd_slot / rfft2x_inv_slot): one HBM pass, where the four-step kernel plus the
// stand-alone pack kernel took two.  The inverse runs the transposed network.
__device__ __forceinline__ f4v ld_nt16(const cpx *p) {
  return CLFA_NT_LD_R15 ? __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p)) : *reinterpret_cast<const f4v *>(p);
}
__device__ __forceinline__ cpx ld_r15(const cpx *p) {
  if (CLFA_NT_LD_R15) return ld_nt(p);
  return *p;
}
// Byte offsets (vector part, scalar part) of the four packed bins of slot (u, q) of lane t (rfft2x_pos(i, which),
// i = pair_index<14, 4>(t, u, q)): every one of them is C + j or C - j, j = t + 1024 u, so the lane part is one of TWO
// VGPRs (t * 8, (1024 - t) * 8) and the rest scalar.  With flat addresses each of the 32 accesses of a lane carried
// its own 64-bit address pair — hipcc then issued the inverse kernel's loads four at a time, each group behind a
// full s_waitcnt vmcnt(0): eight exposed memory latencies per transform.  The u = 0 slots carry lane 0's exceptions
// (pair_index, rfft2x_pos) in the vector part.
struct R15Off {
  int v, s;
};
// LOGC: the sub-transforms' length (14: real size 65536, T = 1024 lanes, four pairs per u; 13: real size 32768,
// T = 512 lanes, two pairs per u — pair_index<13, 4>: q = 0 -> i = j, 1 -> 4096 - j)
template <int LOGC, int LOGE> __device__ __forceinline__ R15Off rfft2x_off(const XferBuf &b, int t, int u, int q, int which) {
  constexpr int LOGR = pass_rem_logr(LOGC, LOGE), R = 1 << LOGR, M = 1 << LOGC, NB = M >> LOGR, T = M >> LOGE;
  static_assert(R == 2 || R == 4, "two or four pairs per u");
  if (u == 0) return R15Off{rfft2x_pos<LOGC>(pair_index<LOGC, LOGE>(t, 0, q), which) * 8, 0};
  // i = +j + ci or -j + ci;  position = which 0: i, 1: 2M - i, 2: M - i, 3: M + i
  const bool ineg = q >= R / 2;
  const int ci = R == 4 ? (q == 0 ? 0 : q == 1 ? NB : q == 2 ? 2 * NB : NB) : (q == 0 ? 0 : NB);
  const bool neg = (which == 1 || which == 2) ? !ineg : ineg;                       // sign of j in the position
  const int c = which == 0 ? ci : which == 1 ? 2 * M - ci : which == 2 ? M - ci : M + ci;   // position = c +- j
  return neg ? R15Off{b.vd, (c - u * T - T) * 8} : R15Off{b.va, (c + u * T) * 8};
}
__device__ __forceinline__ void st_nt16(cpx *p, f4v v) {
  if (CLFA_NT_ST) __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(p));
  else *reinterpret_cast<f4v *>(p) = v;
}

// k_rfft_2x<14>: real size 65536, one 1024-lane workgroup per CU (formerly k_rfft_lds15);
// k_rfft_2x<13>: real size 32768, 512 lanes and 71 KiB of LDS — TWO workgroups per CU, which overlap each other's
// memory phases (k_fft_lds<14> with its pair maps puts one 1024-lane workgroup on a CU)
// (the template also instantiates as <11, true | false, ., 3> — real size 8192 on two 2048-point runs with eight points per
// lane, the half table in LDS; measured slower than k_fft_lds<12> in round 4 and without a launcher since)
template <int LOGC, bool FWD, bool SCALE, int LOGE = 4>
__device__ __forceinline__ void rfft_2x_body(cpx *__restrict__ data, const cpx *__restrict__ tab_g, const cpx *__restrict__ w2_g,
                                             long batch, long out_off) {
  constexpr int LOGN = LOGC, E = 1 << LOGE, M = 1 << LOGC, T = M / E, R = 1 << pass_rem_logr(LOGC, LOGE);
  constexpr bool LANE = kLdsTwoLevel(LOGC);   // lane-addressed tables (8192 / 16384 points) or the half table W_M^k
  constexpr int NTAB = LANE ? kLaneLds : M / 2;
  __shared__ cpx s_tab[NTAB];
  __shared__ cpx s_x[lds_padded_size(M)];
  const int tid = threadIdx.x;
  for (int i = tid; i < (LANE ? kLane13Lds : M / 2); i += T) s_tab[LANE ? lane_lds_index(i) : i] = tab_g[i];
  // lane constants kept across the batch loop: W_M^tid and W_4M^tid only (4 VGPRs; the kernel runs under the 128-VGPR
  // cap) — W_M^(2 tid), ^(3 tid) and W_2M^tid are their products
  const cpx wl0 = LANE ? tab_g[kLane13Lds + tid] : mk(1.f, 0.f);
  const cpx h0 = w2_g[tid];   // W_4M^tid (the plan's sign)
  cpx *xb = s_x;
  __syncthreads();
#pragma unroll 1
  for (long b = blockIdx.x; b < batch; b += gridDim.x) {
    int t = tid;   // opaque per iteration: LDS / global offsets are recomputed, not kept live across the loop
    asm volatile("" : "+v"(t));
    const auto tab = [&]() {
      if constexpr (LOGC == 14) {
        const cpx wl1 = cmul(wl0, wl0);
        return LaneTab14{s_tab + kRow16Stride * (t & 15), s_tab + kRow16Lds + (t & 255), wl0, wl1, cmul(wl0, wl1)};
      } else if constexpr (LOGC == 13) {
        return LaneTab13{s_tab + kRow16Stride * (t & 15), s_tab + kRow16Lds + (t & 255), wl0};
      } else {
        return static_cast<const cpx *>(s_tab);
      }
    }();
    const cpx g0 = cmul(h0, h0);   // W_2M^tid
    cpx *x = data + b * (long)(2 * M);
    cpx *xs = x + out_off;   // where the results go (out_off = 0: in place)
    const XferBuf xo{__builtin_amdgcn_make_buffer_rsrc(x, 0, 0x7fffffff, 0x00020000), t * 8, (T - t) * 8};
    cpx va[E], vb[E];
    if constexpr (FWD) {
#pragma unroll
      for (int e = 0; e < E; e++) {
        const f4v q = ld_nt16(x + 2 * (t + T * e));
        va[e] = mk(q.x, q.y);
        vb[e] = mk(q.z, q.w);
      }
      pass_compute<LOGN, LOGE, 0, true>(va, t, tab);
      // staggered: one chain's LDS transfers under the other's passes.  (The middle passes on permuted lanes — fft_wg.hpp,
      // wg_passes_pair_sigma, conflict-free gathers — measured nothing here: size 32768 +1.4 %, 65536 -0.5 %,
      // profiles/ab_lane_sigma_r05.txt; the complex kernel below keeps them for its -0.7 %.)
      wg_passes_pair<LOGN, LOGE, 0, true>(va, vb, t, tab, xb);
      if constexpr (SCALE) {
        constexpr float inv = 1.0f / (float)(2 * M);
#pragma unroll
        for (int e = 0; e < E; e++) {
          va[e] = cscale(va[e], inv);
          vb[e] = cscale(vb[e], inv);
        }
      }
      cpx ai[E / 2], aj[E / 2], bi[E / 2], bj[E / 2];
      pairs_visit<LOGN, LOGE>(va, t, [&](int k, int, cpx ci, cpx cj) {
        ai[k] = ci;
        aj[k] = cj;
      });
      pairs_visit<LOGN, LOGE>(vb, t, [&](int k, int, cpx ci, cpx cj) {
        bi[k] = ci;
        bj[k] = cj;
      });
#pragma unroll
      for (int k = 0; k < E / 2; k++) {
        // (flat addresses for the forward kernel's stores: buffer-addressed they were measured 2 % slower)
        rfft2x_fwd_slot<LOGC>(t, k / R, k % R, pair_index<LOGN, LOGE>(t, k / R, k % R), ai[k], aj[k], bi[k], bj[k], g0, h0,
                              [&](int pos, cpx v) { st_nt(xs + pos, v); });
        __builtin_amdgcn_sched_barrier(0);   // slot by slot: hoisted, the eight slots' twiddles spill
      }
    } else {
      cpx oa[E / 2], pa[E / 2], ob[E / 2], pb[E / 2];
      cpx raw[2 * E];   // all 32 loads of the lane are in flight before the first slot is computed
#pragma unroll
      for (int k = 0; k < E / 2; k++)
#pragma unroll
        for (int w = 0; w < 4; w++) {
          const R15Off o = rfft2x_off<LOGC, LOGE>(xo, t, k / R, k % R, w);
          raw[4 * k + w] = ld_buf<CLFA_NT_LD_R15 != 0>(xo, o.v, o.s);
        }
#pragma unroll
      for (int k = 0; k < E / 2; k++) {
        const int i = pair_index<LOGN, LOGE>(t, k / R, k % R);
        rfft2x_inv_slot<LOGC>(t, k / R, k % R, i, g0, h0, raw[4 * k], raw[4 * k + 1], raw[4 * k + 2], raw[4 * k + 3], oa[k],
                              pa[k], ob[k], pb[k]);
      }
      constexpr int L1 = pass_last_logns(LOGN, LOGE) - LOGE;
      // staggered (fft_wg.hpp, wg_passes_dif_pair): one chain's scatter drains under the other's butterflies
      pass_first_paired<LOGN, LOGE, false>(va, t, oa, pa, tab);
      __syncthreads();
      pass_first_paired_scatter<LOGN, LOGE>(va, t, xb);
      pass_first_paired<LOGN, LOGE, false>(vb, t, ob, pb, tab);
      __syncthreads();
      dif_gather_padded<LOGN, LOGE, L1>(va, t, xb);
      __syncthreads();
      pass_first_paired_scatter<LOGN, LOGE>(vb, t, xb);
      dif_compute<LOGN, LOGE, L1, false>(va, t, tab);
      __syncthreads();
      dif_gather_padded<LOGN, LOGE, L1>(vb, t, xb);
      wg_passes_dif_pair<LOGN, LOGE, L1, false>(va, vb, t, tab, xb);
#pragma unroll
      for (int e = 0; e < E; e++) st_nt16(xs + 2 * (t + T * e), f4v{va[e].x, va[e].y, vb[e].x, vb[e].y});
    }
  }
}

template <int LOGC, bool FWD, bool SCALE, int LOGE = 4>
__global__ __launch_bounds__((1 << LOGC) >> LOGE, 4) void k_rfft_2x(cpx *__restrict__ data, const cpx *__restrict__ tab_g,
                                                                    const cpx *__restrict__ w2_g, long batch, long out_off) {
  rfft_2x_body<LOGC, FWD, SCALE, LOGE>(data, tab_g, w2_g, batch, out_off);
}
// ... with single LDS accesses (real size 32768)
template <int LOGC, bool FWD, bool SCALE, int LOGE = 4>
__global__ __launch_bounds__((1 << LOGC) >> LOGE, 4) CLFA_DS_SINGLE_FN void k_rfft_2x_s(cpx *__restrict__ data, const cpx *__restrict__ tab_g,
                                                                    const cpx *__restrict__ w2_g, long batch, long out_off) {
  rfft_2x_body<LOGC, FWD, SCALE, LOGE>(data, tab_g, w2_g, batch, out_off);
}

hipError_t launch_rfft_lds15(bool fwd, cpx *data, const FftTables &t, long batch, const DeviceInfo &di, hipStream_t s,
                             long out_off) {
  if (batch <= 0) return hipSuccess;
  const int grid = (int)(batch < di.num_cus ? batch : di.num_cus);   // one 1024-lane workgroup per CU
  if (fwd) hipLaunchKernelGGL((k_rfft_2x<14, true, true>), dim3(grid), dim3(1024), 0, s, data, t.half, t.w2, batch, out_off);
  else hipLaunchKernelGGL((k_rfft_2x<14, false, false>), dim3(grid), dim3(1024), 0, s, data, t.half, t.w2, batch, out_off);
  return hipGetLastError();
}
// real size 32768: t.half = the n = 8192 lane tables (kLane13Size), t.w2 = the plan's r2c table (16384 entries)
hipError_t launch_rfft_2x13(bool fwd, cpx *data, const FftTables &t, long batch, const DeviceInfo &di, hipStream_t s,
                            long out_off) {
  if (batch <= 0) return hipSuccess;
  const long cap = 2L * di.num_cus;   // two 512-lane workgroups per CU
  const int grid = (int)(batch < cap ? batch : cap);
  if (fwd) hipLaunchKernelGGL((k_rfft_2x_s<13, true, true>), dim3(grid), dim3(512), 0, s, data, t.half, t.w2, batch, out_off);
  else hipLaunchKernelGGL((k_rfft_2x_s<13, false, false>), dim3(grid), dim3(512), 0, s, data, t.half, t.w2, batch, out_off);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// complex n = 16384 as TWO 8192-point runs (decimation in time: even and odd samples — one 16-byte load per lane
// brings both) through the n = 8192 machinery, staggered through one exchange buffer (wg_passes_pair), and a radix-2
// step in registers: Z[i] = A[i] + W_16384^i B[i], Z[i + 8192] = A[i] - W_16384^i B[i].  512 lanes, 71 KiB of LDS:
// two workgroups share a CU and overlap each other's memory phases — the whole-transform-in-LDS forms (k_fft_lds<14>
// with 1024 lanes, the persistent four-step kernel) put ONE workgroup on a CU, and its load, pass and store phases
// follow one another.  W_16384^(tid + 512 e) = (lane constant W_16384^tid) x (compile-time W_32^e).  (k_cfft_2x<13>.)
// ---------------------------------------------------------------------------------
// (LOGC = 14: n = 32768 on two 16384-point runs, one 1024-lane workgroup per CU — measured against the persistent
// four-step kernel before choosing)
template <int LOGC, bool FWD, bool SCALE>
__global__ __launch_bounds__((1 << LOGC) / 16, 4) void k_cfft_2x(cpx *__restrict__ data, const cpx *__restrict__ tab_g,
                                                                 long batch, long out_off) {
  using G = LdsGeom<LOGC>;
  constexpr int LOGN = LOGC, LOGE = 4, E = 16, M = 1 << LOGC, T = M / E;
  __shared__ cpx s_tab[kLaneLds];
  __shared__ cpx s_x[G::PADN];
  const int tid = threadIdx.x;
  for (int i = tid; i < kLane13Lds; i += T) s_tab[lane_lds_index(i)] = tab_g[i];
  const cpx wl0 = tab_g[kLane13Lds + tid];                                     // W_M^tid
  const cpx h0 = tab_g[(LOGC == 14 ? kLane14Size : kLane13Size) + tid];        // W_2M^tid (forward sign, like every table)
  cpx *xb = s_x;
  __syncthreads();
#pragma unroll 1
  for (long b = blockIdx.x; b < batch; b += gridDim.x) {
    int t = tid;   // opaque per iteration: LDS / global offsets are recomputed, not kept live across the loop
    asm volatile("" : "+v"(t));
    const auto tab_of = [&](int t) {
      if constexpr (LOGC == 14) {
        const cpx wl1 = cmul(wl0, wl0);
        return LaneTab14{s_tab + kRow16Stride * (t & 15), s_tab + kRow16Lds + (t & 255), wl0, wl1, cmul(wl0, wl1)};
      } else {
        return LaneTab13{s_tab + kRow16Stride * (t & 15), s_tab + kRow16Lds + (t & 255), wl0};
      }
    };
    const auto tab = tab_of(t);
    cpx *x = data + b * (long)(2 * M);
    cpx va[E], vb[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
      const f4v q = ld_nt16(x + 2 * (t + T * e));
      va[e] = mk(q.x, q.y);
      vb[e] = mk(q.z, q.w);
    }
    pass_compute<LOGN, LOGE, 0, FWD>(va, t, tab);
    if constexpr (CLFA_LANE_SIGMA) wg_passes_pair_sigma<LOGN, LOGE, 0, FWD, false>(va, vb, t, lane_sigma(t), tab, tab_of(lane_sigma(t)), xb);
    else wg_passes_pair<LOGN, LOGE, 0, FWD, false>(va, vb, t, tab, xb);
    // radix-2 step: position i = t + T e, W_2M^i = W_2M^t W_32^e
    constexpr float c32[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                               0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                               0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                               -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                               -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float s32[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                               0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f,
                               0.98078528040323044913f, 1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                               0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                               0.38268343236508977173f, 0.19509032201612826785f};
    constexpr float inv = SCALE ? 1.0f / (float)(2 * M) : 1.0f;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const cpx w = e == 0 ? h0 : ctw<true>(h0, c32[e], s32[e]);   // W_2M^(t + T e), forward sign
      const cpx p = cmulc<!FWD>(vb[e], w);
      cpx o0 = cadd(va[e], p), o1 = csub(va[e], p);
      if constexpr (SCALE) {
        o0 = cscale(o0, inv);
        o1 = cscale(o1, inv);
      }
      st_nt(x + out_off + t + T * e, o0);
      st_nt(x + out_off + M + t + T * e, o1);
      __builtin_amdgcn_sched_barrier(0);   // element by element: hoisted, the sixteen twiddles spill
    }
  }
}

template <int LOGC>
static hipError_t launch_cfft_2x_n(bool fwd, bool scale, cpx *data, const FftTables &t, long batch, const DeviceInfo &di,
                                   hipStream_t s, long out_off) {
  if (batch <= 0) return hipSuccess;
  constexpr int T = (1 << LOGC) / 16;
  const long cap = (LOGC == 13 ? 2L : 1L) * di.num_cus;   // two 512-lane workgroups per CU, or one of 1024 lanes
  const int grid = (int)(batch < cap ? batch : cap);
  if (fwd && scale) hipLaunchKernelGGL((k_cfft_2x<LOGC, true, true>), dim3(grid), dim3(T), 0, s, data, t.half, batch, out_off);
  else if (fwd) hipLaunchKernelGGL((k_cfft_2x<LOGC, true, false>), dim3(grid), dim3(T), 0, s, data, t.half, batch, out_off);
  else if (!scale) hipLaunchKernelGGL((k_cfft_2x<LOGC, false, false>), dim3(grid), dim3(T), 0, s, data, t.half, batch, out_off);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_cfft_2x13(bool fwd, bool scale, cpx *data, const FftTables &t, long batch, const DeviceInfo &di,
                            hipStream_t s, long out_off) {
  return launch_cfft_2x_n<13>(fwd, scale, data, t, batch, di, s, out_off);
}

// ---------------------------------------------------------------------------------
// n = 8 .. 64 (packed real: .. 256): the same passes, but global memory is touched in workgroup-wide coalesced rows
// ---------------------------------------------------------------------------------
// With T = n/16 < 8 lanes per transform, "lane t owns positions t + T*e" makes a wave's load touch 64
// different cache lines with 8..32 useful bytes each (measured: n = 16 at 0.96 TB/s).  Here the 256
// transforms of a workgroup (one contiguous chunk of 256*E elements) are read in E fully coalesced
// rows of 256 elements, parked in the per-transform padded LDS buffers at their natural positions, and
// picked up from there in the owning lanes' order (pass_gather_padded); results go back the same way.
template <int LOGN, bool FWD, int MODE, bool SCALE>
__global__ __launch_bounds__(256) void k_fft_small(cpx *__restrict__ data, long out_off, const cpx *__restrict__ tab_g,
                                                   const cpx *__restrict__ w2_g, long batch) {
  using G = LdsGeom<LOGN>;
  constexpr int N = G::N, E = G::E, T = G::T, FPW = G::FPW, CHUNK = FPW * N;
  static_assert(G::WG == 256 && CHUNK == 256 * E, "one chunk = E rows of 256 elements");
  __shared__ cpx s_tab[G::HALF];
  __shared__ cpx s_w2[MODE == MODE_C2C ? 1 : N / 2];
  __shared__ cpx s_x[FPW * G::PADN];
  const int tid = threadIdx.x;
  const int f = tid / T, t = tid % T;
  for (int i = tid; i < N / 2; i += 256) s_tab[i] = tab_g[i];
  if constexpr (MODE != MODE_C2C)
    for (int i = tid; i < N / 2; i += 256) s_w2[i] = w2_g[i];
  cpx *xb = s_x + f * G::PADN;
  // element `tid + 256*e` of the chunk: transform (tid >> LOGN) + (256 >> LOGN)*e, position tid & (N-1)
  cpx *park = s_x + (tid >> LOGN) * G::PADN + lds_pad(tid & (N - 1));
  constexpr int PARK_STEP = (256 >> LOGN) * G::PADN;
  const long groups = (batch + FPW - 1) / FPW;
  const long total = batch * (long)N;
  long g = blockIdx.x;
  if (g >= groups) return;
  cpx raw[E];
  auto load_rows = [&](long grp) {
    const long base = grp * CHUNK;
#pragma unroll
    for (int e = 0; e < E; e++) {
      long idx = base + tid + 256 * e;
      idx = idx < total ? idx : total - 1;   // ragged last group: clamped, straight-line
      raw[e] = ld_nt(data + idx);
    }
  };
  // the reference's pair maps (cl_fft.cpp:178-205) in place on the natural-order LDS copy: lane t of a
  // transform owns pairs i = t + T*k (and their partners n - i); pair 0 is the packed DC/Nyquist bin
  auto pair_map = [&]() {
#pragma unroll
    for (int k = 0; k < E / 2; k++) {
      const int i = t + T * k, j = i == 0 ? N / 2 : N - i;
      const cpx ci = xb[lds_pad(i)], cj = xb[lds_pad(j)];
      cpx oi, oj;
      if constexpr (MODE == MODE_R2C) r2c_pair(ci, cj, s_w2[i], oi, oj);
      else c2r_pair(ci, cj, s_w2[i], oi, oj);
      if (k == 0) {   // selects, not a branch
        const bool z = i == 0;
        const float h = MODE == MODE_R2C ? .5f : 1.f;
        oi = mk(z ? (ci.x + ci.y) * h : oi.x, z ? (ci.x - ci.y) * h : oi.y);
        oj = mk(z ? cj.x : oj.x, z ? cj.y : oj.y);
      }
      xb[lds_pad(i)] = oi;
      xb[lds_pad(j)] = oj;
    }
  };
  load_rows(g);
#pragma unroll
  for (int e = 0; e < E; e++) asm volatile("" : "+v"(raw[e]));
  __syncthreads();
#pragma unroll 1
  for (; g < groups; g += gridDim.x) {
    cpx v[E];
#pragma unroll
    for (int e = 0; e < E; e++) park[e * PARK_STEP] = raw[e];
    {  // the next chunk's rows fly behind this one's passes
      long gn = g + gridDim.x;
      load_rows(gn < groups ? gn : groups - 1);
    }
    __syncthreads();
    if constexpr (MODE == MODE_C2R) {
      pair_map();
      __syncthreads();
    }
    pass_gather_padded<LOGN, G::LOGE>(v, t, xb);
    wg_passes<LOGN, G::LOGE, 0, FWD>(v, t, s_tab, xb);
    if constexpr (SCALE) {
#pragma unroll
      for (int e = 0; e < E; e++) v[e] = cscale(v[e], 1.0f / (float)N);
    }
    __syncthreads();   // every lane is done with the exchange buffer
    dif_scatter_padded<LOGN, G::LOGE>(v, t, xb);
    __syncthreads();
    if constexpr (MODE == MODE_R2C) {
      pair_map();
      __syncthreads();
    }
    const long base = g * CHUNK;
    const bool full = base + CHUNK <= total;   // uniform
    if (full) {
#pragma unroll
      for (int e = 0; e < E; e++) st_nt(data + out_off + base + tid + 256 * e, park[e * PARK_STEP]);
    } else {
#pragma unroll
      for (int e = 0; e < E; e++)
        if (base + tid + 256 * e < total) data[out_off + base + tid + 256 * e] = park[e * PARK_STEP];
    }
    __syncthreads();   // the parked results are out before the next chunk is parked
#pragma unroll
    for (int e = 0; e < E; e++) asm volatile("" : "+v"(raw[e]));
  }
}

// How many workgroups of the persistent grids share a CU.  NOT "as many as fit": these kernels keep the next transform's
// loads in flight behind the current one's passes, so one workgroup per CU already covers the memory latency, and every
// further one only adds concurrent streams for the memory controllers to interleave.  Chosen per size and packing from
// interleaved A/Bs on random data, directions alternating (profiles/wgs_per_cu_r05.txt; one / two / three / all that fit):
// n = 1024 at one workgroup per CU 6.02 TB/s, at the three that fit 5.37; n = 16 .. 2048 and 8192 -4 .. -11 % of the time;
// n = 8 and n = 4096 like two.  The packed real kernels' pair maps stall between barriers, so most of them want company:
// sizes 8, 16, 64 .. 512, 2048, 4096 take two, sizes 32 and 1024 one (-2 .. -22 % against what fits at 2 GiB per launch),
// sizes 8192 and 16384 stay.
// (CLFA_WGS_TABLE 0: round 4's grids; CLFA_WGS_C / CLFA_WGS_RF + CLFA_WGS_RI: one figure for every size, for A/B builds.)
// All of this holds for batches that STREAM from HBM: up to about twice the 256 MiB Infinity Cache the same A/B reads the
// other way (n = 1024: 16 MiB +21 %, 256 MiB +2 %, 512 MiB -5 %, 1 GiB -10 %; n = 64 still +8 % at 768 MiB, -4 % at 1 GiB),
// so the table applies from 1 GiB of transforms per launch and smaller batches keep every workgroup that fits.
#ifndef CLFA_WGS_TABLE
#define CLFA_WGS_TABLE 1
#endif
static inline bool streaming_batch(long batch, int logn) { return (batch << logn) >= (1L << 27); }   // 8-byte samples: 1 GiB
template <int LOGN, int MODE> constexpr int wgs_per_cu() {
  if (!CLFA_WGS_TABLE) return 64;
  if (MODE == MODE_C2C) {
#if defined(CLFA_WGS_C)
    return CLFA_WGS_C;
#endif
    return LOGN == 3 || LOGN == 12 ? 2 : (LOGN >= 4 && LOGN <= 13) ? 1 : 64;
  }
#if defined(CLFA_WGS_RF)
  return MODE == MODE_R2C ? CLFA_WGS_RF : CLFA_WGS_RI;
#endif
  switch (LOGN) {   // packed real size 2^(LOGN + 1), at 2 GiB per launch (the second table of profiles/wgs_per_cu_r05.txt)
    // (size 16384: one or two, forward or inverse, within 1 %; size 8192: -2 % at best; size 2048 with one: -7 % at 2 GiB,
    // but behind two in a sweep at 1 GiB — two is never behind)
    case 4: case 9: return 1;
    case 2: case 3: case 5: case 6: case 7: case 8: case 10: case 11: return 2;
    default: return 64;
  }
}

template <int LOGN, bool FWD, int MODE, bool SCALE>
static hipError_t launch_small_one(cpx *data, const FftTables &t, long batch, const DeviceInfo &di, hipStream_t s, long out_off) {
  using G = LdsGeom<LOGN>;
  long groups = (batch + G::FPW - 1) / G::FPW;
  static int occ = 0;
  if (occ == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fft_small<LOGN, FWD, MODE, SCALE>, 256, 0) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      nb = 1;
    }
    occ = nb;
  }
  long cap = (long)di.num_cus * (streaming_batch(batch, LOGN) && wgs_per_cu<LOGN, MODE>() < occ ? wgs_per_cu<LOGN, MODE>() : occ);
  int grid = (int)(groups < cap ? groups : cap);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_fft_small<LOGN, FWD, MODE, SCALE>), dim3(grid), dim3(256), 0, s, data, out_off, t.half, t.w2, batch);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// n = 2 and n = 4, complex: a copy kernel with a butterfly in it
// ---------------------------------------------------------------------------------
// Every lane moves 16 bytes (two complex samples) per access, lanes in address order — the access shape of a plain copy.
// n = 2: the lane holds the whole transform.  n = 4: lanes 2k and 2k + 1 hold (x0, x1) and (x2, x3) and read each other's
// pair through the DPP lane crossbar (quad_perm [1, 0, 3, 2]: four v_mov_dpp, no LDS); the even lane leaves with (X0, X1),
// the odd one with (X2, X3), so the stores are in address order as well.  The reference's two stages (cl_fft.cpp:24-41 on
// bit-reversed input): s0 = x0 + x2, d0 = x0 - x2, s1 = x1 + x3, d1 = x1 - x3; X0 = s0 + s1, X2 = s0 - s1,
// X1 = d0 + w d1, X3 = d0 - w d1, w = -i forward, +i inverse (exact in every rounding).  A workgroup moves contiguous
// runs of 256 * UNROLL pieces, UNROLL accesses of a lane in flight at once; a ragged tail is clamped on the load side and
// predicated on the store side (the lanes of a pair are always both inside or both outside: a transform is 32 bytes).
#ifndef CLFA_TINY
#define CLFA_TINY 1
#endif
__device__ __forceinline__ float dpp_swap1(float v) {   // lane l <- lane l ^ 1
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
template <int LOGN, bool FWD, bool SCALE, int UNROLL>
__global__ __launch_bounds__(256) void k_fft_tiny(cpx *__restrict__ data, long out_off, long total16) {
  static_assert(LOGN == 1 || LOGN == 2, "n = 2 or 4");
  constexpr float sc = SCALE ? 1.0f / (float)(1 << LOGN) : 1.0f;
  constexpr long TILE = 256 * UNROLL;   // 16-byte pieces a workgroup moves per iteration: one contiguous run
  const bool odd = threadIdx.x & 1;
#pragma unroll 1
  for (long i0 = (long)blockIdx.x * TILE + threadIdx.x; i0 < total16; i0 += (long)gridDim.x * TILE) {
    f4v q[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      long i = i0 + u * 256;
      i = i < total16 ? i : total16 - 1;   // ragged tail: clamped, the result is not stored
      q[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(data + 2 * i));
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const f4v m = q[u];
      f4v r;
      if constexpr (LOGN == 1) {
        r = f4v{(m.x + m.z) * sc, (m.y + m.w) * sc, (m.x - m.z) * sc, (m.y - m.w) * sc};
      } else {
        const f4v o = f4v{dpp_swap1(m.x), dpp_swap1(m.y), dpp_swap1(m.z), dpp_swap1(m.w)};
        // (x0, x1) = even lane's pair, (x2, x3) = odd lane's: sums are symmetric, differences change sign in the odd lane
        const float s0x = m.x + o.x, s0y = m.y + o.y, s1x = m.z + o.z, s1y = m.w + o.w;
        const float d0x = odd ? o.x - m.x : m.x - o.x, d0y = odd ? o.y - m.y : m.y - o.y;
        const float d1x = odd ? o.z - m.z : m.z - o.z, d1y = odd ? o.w - m.w : m.w - o.w;
        // w d1, w = -i (forward): (d1y, -d1x); +i (inverse): (-d1y, d1x)
        const float wx = FWD ? d1y : -d1y, wy = FWD ? -d1x : d1x;
        r = f4v{(odd ? s0x - s1x : s0x + s1x) * sc, (odd ? s0y - s1y : s0y + s1y) * sc,
                (odd ? d0x - wx : d0x + wx) * sc, (odd ? d0y - wy : d0y + wy) * sc};
      }
      const long i = i0 + u * 256;
      if (i < total16) st_nt16(data + out_off + 2 * i, r);
    }
  }
}
// Two accesses per lane in flight and four workgroups per CU: 5.9-6.1 TB/s in place (the chip's plain copy); one or eight
// workgroups per CU, or four / eight accesses per lane, 3.3-5.8 (profiles/tiny_r05.txt).
template <int LOGN, bool FWD, bool SCALE>
static hipError_t launch_tiny_one(cpx *data, long batch, const DeviceInfo &di, hipStream_t s, long out_off) {
  constexpr int UNROLL = 2;
  const long total16 = batch << (LOGN - 1);   // 16-byte pieces
  const long want = (total16 + 256 * UNROLL - 1) / (256 * UNROLL);
  const long cap = 4L * di.num_cus;
  hipLaunchKernelGGL((k_fft_tiny<LOGN, FWD, SCALE, UNROLL>), dim3((int)(want < cap ? want : cap)), dim3(256), 0, s, data, out_off, total16);
  return hipGetLastError();
}

template <int LOGN, bool FWD, int MODE, bool SCALE>
static hipError_t launch_lds_one(cpx *data, const FftTables &t, long batch, const DeviceInfo &di,
                                 hipStream_t s, long out_off) {
  using G = LdsGeom<LOGN>;
  long groups = (batch + G::FPW - 1) / G::FPW;
  // persistent grid: exactly the workgroups that are resident at once (occupancy x CUs), each
  // grid-striding over many transforms, so the LDS twiddle table is loaded once per workgroup
  // and every transform but the first is software-prefetched
  static int occ = 0;  // per instantiation
  if (occ == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fft_lds<LOGN, FWD, MODE, SCALE>, G::WG, 0) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      nb = 1;
    }
    occ = nb;
  }
  long cap = (long)di.num_cus * (streaming_batch(batch, LOGN) && wgs_per_cu<LOGN, MODE>() < occ ? wgs_per_cu<LOGN, MODE>() : occ);
  int grid = (int)(groups < cap ? groups : cap);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_fft_lds<LOGN, FWD, MODE, SCALE>), dim3(grid), dim3(G::WG), 0, s, data, t.half, t.w2, batch, out_off);
  return hipGetLastError();
}

template <int LOGN>
static hipError_t launch_lds_n(bool fwd, int mode, bool scale, cpx *data, const FftTables &t, long batch,
                               const DeviceInfo &di, hipStream_t s, long out_off) {
#define CLFA_CASE(F, M, S)                                                                                     \
  if (fwd == F && mode == M && scale == S) {                                                                   \
    if constexpr (CLFA_TINY && LOGN <= 2 && M == MODE_C2C) return launch_tiny_one<LOGN, F, S>(data, batch, di, s, out_off); \
    /* sub-64-byte rows per transform (and the packed real transforms up to 256 bins, whose pair maps  */     \
    /* store 8-byte pieces): coalesced staging through LDS                                              */     \
    if constexpr (LOGN >= 2 && (LOGN <= 6 || (M != MODE_C2C && LOGN <= 8)))                                     \
      return launch_small_one<LOGN, F, M, S>(data, t, batch, di, s, out_off);                                  \
    else return launch_lds_one<LOGN, F, M, S>(data, t, batch, di, s, out_off);                                 \
  }
  CLFA_CASE(true, MODE_C2C, true)
  CLFA_CASE(true, MODE_C2C, false)
  CLFA_CASE(false, MODE_C2C, false)
  CLFA_CASE(true, MODE_R2C, true)
  CLFA_CASE(false, MODE_C2R, false)
#undef CLFA_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_fft_lds(int logn, bool fwd, int mode, bool scale, cpx *data, const FftTables &t,
                          long batch, const DeviceInfo &di, hipStream_t s, long out_off) {
  if (batch <= 0) return hipSuccess;
  switch (logn) {
#define CLFA_N(L) \
  case L:         \
    return launch_lds_n<L>(fwd, mode, scale, data, t, batch, di, s, out_off);
    CLFA_N(1) CLFA_N(2) CLFA_N(3) CLFA_N(4) CLFA_N(5) CLFA_N(6) CLFA_N(7) CLFA_N(8) CLFA_N(9) CLFA_N(10)
    CLFA_N(11) CLFA_N(12) CLFA_N(13)
#undef CLFA_N
    default:
      return hipErrorInvalidValue;
  }
}

const char *name_fft_lds(int logn, bool, int mode) {
  if (CLFA_TINY && logn <= 2 && mode == MODE_C2C) return "k_fft_tiny";
  if (logn >= 2 && (logn <= 6 || (mode != MODE_C2C && logn <= 8))) return "k_fft_small";
  return "k_fft_lds";
}

// ---------------------------------------------------------------------------------
// four-step FFT for n = 2^14 .. 2^16
// ---------------------------------------------------------------------------------

int fourstep_split(int logn, int *l1, int *l2, int *loglo) {
  if (logn < 14 || logn > 16) return -1;
  *l1 = logn / 2;
  *l2 = logn - *l1;
  *loglo = logn / 2;
  return 0;
}

#ifndef CLFA_4STEP_RRB
#define CLFA_4STEP_RRB 5   // register-resident row blocks per slice of the default n = 65536 kernel
#endif
template <int LOGN> struct FourGeom {
  static constexpr int LOGN1 = LOGN / 2, LOGN2 = LOGN - LOGN1;
  static constexpr int N = 1 << LOGN, N1 = 1 << LOGN1, N2 = 1 << LOGN2;
  static constexpr int LOGLO = LOGN / 2, LO = 1 << LOGLO, HI = 1 << (LOGN - LOGLO);
  static constexpr int SLICE = 256;                 // lanes per slice
  static constexpr int T1 = N1 / 16, C1 = SLICE / T1;   // lanes per column FFT, columns per slice
  static constexpr int T2 = N2 / 16, R2 = SLICE / T2;   // lanes per row FFT, rows per slice
  static constexpr int S2 = lds_padded_size(N2) | 1;    // odd row stride in LDS
  static constexpr int SL = (N1 * C1 > R2 * S2) ? N1 * C1 : R2 * S2;  // exchange elements per slice
  static constexpr int NCB = N2 / C1, NRB = N1 / R2;    // column blocks, row blocks per transform
  static constexpr int TABS = N1 / 2 + N2 / 2 + LO + HI;
  // rows of the intermediate kept in LDS instead of the scratch (k_fft_4step, KL rows): row stride
  // N2 + 16 elements puts the 4 rows a wave touches on 2 x 32 banks (the 2 passes 512 B need anyway)
  static constexpr int RS = N2 + 16;
};

// Wave-uniform base pointers kept in SGPR pairs.  A global access whose address is
// (uniform 64-bit base) + (32-bit lane offset) uses the saddr form  global_load v, v_off, s[b:b+1]:
// no 64-bit VALU add with carry (and its hazard nops) per access.  The base goes through an opaque
// SGPR integer so that hipcc cannot fold the lane part into it, and comes back as a global-memory
// (address space 1) pointer so that the access is not demoted to a flat one.
typedef __attribute__((address_space(1))) unsigned long long *gptr;
typedef const __attribute__((address_space(1))) unsigned long long *gcptr;
__device__ __forceinline__ gptr sgpr_base(const cpx *p) {
  unsigned long long b = reinterpret_cast<unsigned long long>(p);
  asm volatile("" : "+s"(b));
  return reinterpret_cast<gptr>(b);
}
// streaming mode: 0 plain, 1 non-temporal, 2 system scope (sc0 sc1), 3 agent scope (sc1: bypasses the CU's L1)
template <int SM> __device__ __forceinline__ cpx ld_g(gcptr p) {
  unsigned long long raw;
  if constexpr (SM == 1 && CLFA_NT_LD_4STEP) raw = __builtin_nontemporal_load(p);
  else if constexpr (SM == 1) raw = *p;
  else if constexpr (SM == 2) raw = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else if constexpr (SM == 3) raw = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else raw = *p;
  return *reinterpret_cast<const cpx *>(&raw);
}
template <int SM> __device__ __forceinline__ void st_g(gptr p, cpx v) {
  const unsigned long long raw = *reinterpret_cast<const unsigned long long *>(&v);
  if constexpr (SM == 1 && CLFA_NT_ST) __builtin_nontemporal_store(raw, p);
  else if constexpr (SM == 1) *p = raw;
  else if constexpr (SM == 2) __hip_atomic_store(p, raw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else *p = raw;
}

// phase 1 of one slice: column block cb of `src` (N1 x N2, row-major) ->
// N1-point FFT down the columns, times W_N^(n2*k1), stored to dst[k1][n2].
// streaming mode of the input loads / output stores: 0 plain, 1 non-temporal, 2 system scope (sc0 sc1)
template <int LOGN, int SM>
__device__ __forceinline__ void four_load1(cpx (&v)[16], const cpx *__restrict__ src, int cb, int l) {
  using G = FourGeom<LOGN>;
  const unsigned col = (unsigned)l % G::C1, tf = ((unsigned)l & (G::SLICE - 1)) / G::C1;
  const unsigned lane = tf * G::N2 + col;
  const cpx *base = src + cb * G::C1;   // cb is wave-uniform
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = ld_g<SM>(sgpr_base(base + (long)(G::T1 * e) * G::N2) + lane);
}
// A lane's results of one register-resident row block over the column blocks of its slice (8 values
// for every n: n = 65536 has 1 row set per block x 8 column blocks, 32768 2 x 4, 16384 4 x 2).  A native
// vector so that hipcc indexes it with the uniform loop counter through s_set_gpr_idx (an array would
// go to scratch memory).
typedef float vkeep __attribute__((ext_vector_type(16)));
struct NoKeep {};
// KL > 0: rows k1 < KL of the result (the first KL / R2 row blocks) stay in LDS (`rows`, stride RS) and
// never reach the scratch; NE > 0: the next NE row blocks stay in the lane's own registers
// (`keep[block]`, element it * EB + eb for the slice's it-th column block) until phase 2 hands them
// over through LDS
template <int LOGN, bool FWD, int KL = 0, int NE = 0, class Keep = NoKeep, class Tab = const cpx *>
__device__ __forceinline__ void four_body1(cpx (&v)[16], cpx *__restrict__ dst, int cb, int l, const Tab &tab1,
                                           const cpx *tlo, const cpx *thi, cpx *sx, cpx *rows = nullptr,
                                           Keep *keep = nullptr, int it = 0) {
  using G = FourGeom<LOGN>;
  const int col = (unsigned)l % G::C1, tf = ((unsigned)l & (G::SLICE - 1)) / G::C1;
  const int n2 = cb * G::C1 + col;
  pass_compute<G::LOGN1, 4, 0, FWD>(v, tf, tab1);
  __syncthreads();
  pass_scatter<G::LOGN1, 4, 0>(v, tf, [&](int p, cpx val) { sx[p * G::C1 + col] = val; });
  __syncthreads();
  pass_gather<G::LOGN1, 4>(v, tf, [&](int p) { return sx[p * G::C1 + col]; });
  pass_compute<G::LOGN1, 4, 4, FWD>(v, tf, tab1);
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k1 = tf + G::T1 * e;
    const int ex = n2 * k1;  // < N
    const cpx o = cmulc<!FWD>(v[e], cmul(tlo[ex & (G::LO - 1)], thi[ex >> G::LOGLO]));
    // row k1 = tf + T1*e belongs to row block e / EB (EB row sets per block): all decided at compile time
    constexpr int EB = G::R2 / G::T1;
    const int blk = e / EB, eb = e % EB;
    if (blk < KL / G::R2) {
      rows[k1 * G::RS + n2] = o;
    } else if (blk < KL / G::R2 + NE) {
      if constexpr (NE > 0) {
        keep[blk - KL / G::R2][2 * (it * EB + eb)] = o.x;
        keep[blk - KL / G::R2][2 * (it * EB + eb) + 1] = o.y;
      }
    } else {
      st_g<0>(sgpr_base(dst + (long)(G::T1 * e) * G::N2 + cb * G::C1) + (unsigned)(tf * G::N2 + col), o);
    }
  }
}
template <int LOGN, bool FWD, int SM>
__device__ __forceinline__ void four_phase1(const cpx *__restrict__ src, cpx *__restrict__ dst, int cb, int l,
                                            const cpx *tab1, const cpx *tlo, const cpx *thi, cpx *sx) {
  cpx v[16];
  four_load1<LOGN, SM>(v, src, cb, l);
  four_body1<LOGN, FWD>(v, dst, cb, l, tab1, tlo, thi, sx);
}

// 8-byte load that bypasses the CU's vector L1 (global_load_dwordx2 ... sc1): data another
// CU of the same XCD has stored is served from the shared L2 (MI355X_MICROARCH.md, workgroup
// dispatch & inter-workgroup visibility)
__device__ __forceinline__ cpx ld_sc1(const cpx *p) {
  unsigned long long raw = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
  return *reinterpret_cast<cpx *>(&raw);
}

template <int LOGN, bool SC1>
__device__ __forceinline__ void four_load2(cpx (&v)[16], const cpx *__restrict__ src, int rb, int l) {
  using G = FourGeom<LOGN>;
  const int tf = (unsigned)l % G::T2, row = ((unsigned)l & (G::SLICE - 1)) / G::T2;
  const gcptr p = sgpr_base(src + (long)(rb * G::R2) * G::N2) + (unsigned)(row * G::N2 + tf);   // rb is wave-uniform
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = ld_g<SC1 ? 3 : 0>(p + G::T2 * e);
}
// the same row block out of the LDS-resident rows
template <int LOGN>
__device__ __forceinline__ void four_load2_rows(cpx (&v)[16], const cpx *rows, int rb, int l) {
  using G = FourGeom<LOGN>;
  const int tf = l % G::T2, row = l / G::T2;
  const cpx *p = rows + (rb * G::R2 + row) * G::RS + tf;
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = p[G::T2 * e];
}
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// `between` runs between the block's two barriers (after every wave has passed the first one): the
// register-resident row blocks are handed over there at no extra barrier
template <int LOGN, bool FWD, bool SCALE, int SM, class Tab = const cpx *, class Hook = NoHook>
__device__ __forceinline__ void four_body2(cpx (&v)[16], cpx *__restrict__ dst, int rb, int l, const Tab &tab2,
                                           cpx *sx, unsigned *read_done = nullptr, Hook between = Hook()) {
  using G = FourGeom<LOGN>;
  {
    const int tf = l % G::T2, row = l / G::T2;
    pass_compute<G::LOGN2, 4, 0, FWD>(v, tf, tab2);
    __syncthreads();
    cpx *xr = sx + row * G::S2;
    pass_scatter_padded<G::LOGN2, 4, 0>(v, tf, xr);
    between();
    __syncthreads();
    // every lane has consumed its loads from the scratch: the slot may be reused
    if (read_done != nullptr && l == 0)
      (void)__hip_atomic_fetch_add(read_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // in the XCD's L2
  }
  // the last pass runs with rows on the fast lane index so that the transposed
  // store below is contiguous across lanes
  // (masked: the lane index passes through an opaque move in the callers; its range has to be visible
  // for the 32-bit lane offsets of the saddr addressing)
  const int row = (unsigned)l % G::R2, tf = ((unsigned)l & (G::SLICE - 1)) / G::R2;
  const cpx *xr = sx + row * G::S2;
  pass_gather_padded<G::LOGN2, 4>(v, tf, xr);
  pass_compute<G::LOGN2, 4, 4, FWD>(v, tf, tab2);
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k2 = tf + G::T2 * e;
    cpx o = v[e];
    if constexpr (SCALE) o = cscale(o, 1.0f / (float)G::N);
    (void)k2;
    st_g<SM>(sgpr_base(dst + (long)(G::T2 * e) * G::N1 + rb * G::R2) + (unsigned)(tf * G::N1 + row), o);
  }
}
// phase 2 of one slice: row block rb of `src` (rows k1, contiguous n2) ->
// N2-point FFT along each row -> dst[k1 + N1*k2] (natural order of the result)
template <int LOGN, bool FWD, bool SCALE, int SM, bool SC1 = false>
__device__ __forceinline__ void four_phase2(const cpx *__restrict__ src, cpx *__restrict__ dst, int rb, int l,
                                            const cpx *tab2, cpx *sx, unsigned *read_done = nullptr) {
  cpx v[16];
  four_load2<LOGN, SC1>(v, src, rb, l);
  four_body2<LOGN, FWD, SCALE, SM>(v, dst, rb, l, tab2, sx, read_done);
}

// The first row block of every slice — rows k1 < KL = NSLICE * R2, 1/8 of the
// intermediate for n = 65536 — stays in LDS between the phases instead of going through the scratch
template <int LOGN, bool FWD, bool SCALE>
__global__ __launch_bounds__(512) CLFA_DS_SINGLE_FN void k_fft_4step(cpx *__restrict__ data, cpx *__restrict__ scratch,
                                                           const cpx *__restrict__ tabs_g, long batch, long out_off) {
  using G = FourGeom<LOGN>;
  constexpr int NSLICE = 2;        // two 256-lane slices per workgroup
  constexpr bool NT = true;        // non-temporal input loads / output stores
  constexpr bool ROWS = true, PF = true;
  constexpr int KL = NSLICE * G::R2;
  // ... and the next RRB row blocks of every slice in registers (16 VGPRs per block; the 512-lane
  // workgroup has 256 per lane): 5 of the remaining 7 for n = 65536, all of them for 32768 (3) and
  // 16384 (1), whose scratch is then never touched
  constexpr int RRB = !(ROWS && NSLICE == 2) ? 0 : LOGN == 16 ? CLFA_4STEP_RRB : G::NRB / NSLICE - 1;
  constexpr int NE = RRB * NSLICE;
  constexpr int EB = G::R2 / G::T1, NIT = G::NCB / NSLICE;
  static_assert(NE == 0 || EB * NIT == 8, "a row block is 8 values per lane");
  __shared__ cpx s_tabs[G::TABS];
  __shared__ cpx s_x[NSLICE * G::SL];
  __shared__ cpx s_rows[ROWS ? KL * G::RS : 1];
  // full W_N1 / W_N2 tables for the pass twiddles of the prefetching form (no half-table sign logic)
  __shared__ cpx s_full[PF ? G::N1 + G::N2 : 1];
  const int tid = threadIdx.x;
  for (int i = tid; i < G::TABS; i += 256 * NSLICE) s_tabs[i] = tabs_g[i];
  const cpx *tlo = s_tabs + G::N1 / 2 + G::N2 / 2, *thi = tlo + G::LO;
  if constexpr (PF) {
    for (int i = tid; i < G::N1 + G::N2; i += 256 * NSLICE) {
      const bool second = i >= G::N1;
      const int k = second ? i - G::N1 : i, h = (second ? G::N2 : G::N1) / 2;
      const cpx w = tabs_g[(second ? G::N1 / 2 : 0) + (k & (h - 1))];
      s_full[i] = (k & h) ? mk(-w.x, -w.y) : w;
    }
  }
  const FullTab ftab1{s_full}, ftab2{s_full + G::N1};
  // the slice index is wave-uniform (a slice is 4 whole waves): say so, so that block indices and the
  // pointers derived from them stay in SGPRs
  const int slice = __builtin_amdgcn_readfirstlane(tid / G::SLICE), l = tid % G::SLICE;
  cpx *sx = s_x + slice * G::SL;
  cpx *mid = scratch + (long)blockIdx.x * G::N;
  __syncthreads();

  // prefetching form: the first column block of a transform is loaded behind the last row block of the
  // previous one (`vnext`), so that only the very first load of the workgroup is exposed
  cpx vnext[16];
  if constexpr (PF) four_load1<LOGN, NT ? 1 : 0>(vnext, data + xcd_first(blockIdx.x, gridDim.x) * G::N, slice, l);
#pragma unroll 1
  for (long b = xcd_first(blockIdx.x, gridDim.x); b < batch; b += gridDim.x) {
    cpx *x = data + b * (long)G::N;
    {
      // software-prefetched form: the next block's loads fly behind the current block's passes.
      // The last block of each phase is peeled so that every prefetch is straight-line code
      // (counted s_waitcnt, see k_fft_lds), and consumed at the end of the iteration.
      cpx v[16], vn[16];
      vkeep keep[NE > 0 ? NE : 1];
      int it = 0;   // the slice's column-block counter: uniform, indexes `keep`
      // consumed before the loop: otherwise the wait for these loads is merged into the loop header,
      // where it turns into vmcnt(0) on the back edge too and drains every iteration's scratch stores
#pragma unroll
      for (int e = 0; e < 16; e++) {
        asm volatile("" : "+v"(vnext[e]));
        v[e] = vnext[e];
      }
#pragma unroll 1
      for (int cb = slice; cb + NSLICE < G::NCB; cb += NSLICE) {
        int lo_ = l;   // opaque per iteration (see above)
        asm volatile("" : "+v"(lo_));
        four_load1<LOGN, NT ? 1 : 0>(vn, x, cb + NSLICE, lo_);
        four_body1<LOGN, FWD, KL, NE>(v, mid, cb, lo_, ftab1, tlo, thi, sx, s_rows, keep, it);
        it++;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          asm volatile("" : "+v"(vn[e]));
          v[e] = vn[e];
        }
      }
      {
        int lo_ = l;
        asm volatile("" : "+v"(lo_));
        four_body1<LOGN, FWD, KL, NE>(v, mid, G::NCB - NSLICE + slice, lo_, ftab1, tlo, thi, sx, s_rows, keep,
                                      G::NCB / NSLICE - 1);
      }
      __syncthreads();
      if constexpr (ROWS) four_load2_rows<LOGN>(v, s_rows, slice, l);
      else four_load2<LOGN, false>(v, mid, slice, l);
#pragma unroll
      for (int e = 0; e < 16; e++) asm volatile("" : "+v"(v[e]));
      int rb0 = slice;
      if constexpr (NE > 0) {
        // Row blocks 1..RRB of each slice (rows 32.., alternating between the slices): every lane hands
        // its register-resident results over through the LDS rows the previous blocks have just left.
        // Block r+1 is dumped between the two barriers of block r-1's passes: at the first of them every
        // wave has already taken block r out of those rows (its load precedes the passes in program
        // order), the second publishes the dump — no barrier of its own except for the first block.
        auto dump = [&](auto rc) {
          constexpr int r = decltype(rc)::value;
          const int col = l % G::C1, tf = l / G::C1;
#pragma unroll
          for (int q = 0; q < NSLICE; q++) {
#pragma unroll
            for (int eb = 0; eb < EB; eb++) {
              cpx *pr = s_rows + (q * G::R2 + tf + G::T1 * eb) * G::RS + slice * G::C1 + col;
#pragma unroll
              for (int j = 0; j < NIT; j++)
                pr[j * NSLICE * G::C1] = mk(keep[NSLICE * r + q][2 * (j * EB + eb)], keep[NSLICE * r + q][2 * (j * EB + eb) + 1]);
            }
          }
        };
        __syncthreads();
        dump(std::integral_constant<int, 0>());
        __syncthreads();
        auto round = [&](auto rc) {
          constexpr int r = decltype(rc)::value;
          int lo_ = l;
          asm volatile("" : "+v"(lo_));
          four_load2_rows<LOGN>(vn, s_rows, slice, lo_);
          if constexpr (r + 1 < RRB) {
            four_body2<LOGN, FWD, SCALE, NT ? 1 : 0>(v, x + out_off, slice + NSLICE * r, lo_, ftab2, sx, nullptr,
                                                     [&]() { dump(std::integral_constant<int, r + 1>()); });
          } else {
            four_body2<LOGN, FWD, SCALE, NT ? 1 : 0>(v, x + out_off, slice + NSLICE * r, lo_, ftab2, sx);
          }
#pragma unroll
          for (int e = 0; e < 16; e++) v[e] = vn[e];
        };
        round(std::integral_constant<int, 0>());
        if constexpr (RRB > 1) round(std::integral_constant<int, 1>());
        if constexpr (RRB > 2) round(std::integral_constant<int, 2>());
        if constexpr (RRB > 3) round(std::integral_constant<int, 3>());
        if constexpr (RRB > 4) round(std::integral_constant<int, 4>());
        if constexpr (RRB > 5) round(std::integral_constant<int, 5>());
        static_assert(RRB <= 6, "unrolled by hand up to 6 rounds");
        rb0 = slice + NSLICE * RRB;
      }
#pragma unroll 1
      for (int rb = rb0; rb + NSLICE < G::NRB; rb += NSLICE) {
        int lo_ = l;
        asm volatile("" : "+v"(lo_));
        four_load2<LOGN, false>(vn, mid, rb + NSLICE, lo_);
        four_body2<LOGN, FWD, SCALE, NT ? 1 : 0>(v, x + out_off, rb, lo_, ftab2, sx);
#pragma unroll
        for (int e = 0; e < 16; e++) {
          asm volatile("" : "+v"(vn[e]));
          v[e] = vn[e];
        }
      }
      {
        int lo_ = l;
        asm volatile("" : "+v"(lo_));
        // the next transform's first column block (index clamped to the last transform: straight-line loads)
        long bn = b + gridDim.x;
        bn = bn < batch ? bn : batch - 1;
        four_load1<LOGN, NT ? 1 : 0>(vnext, data + bn * (long)G::N, slice, lo_);
        four_body2<LOGN, FWD, SCALE, NT ? 1 : 0>(v, x + out_off, G::NRB - NSLICE + slice, lo_, ftab2, sx);
      }
      __syncthreads();
    }
  }
}

// Small batches (fewer transforms than resident workgroups): one workgroup per column block,
// then one per row block — two launches, N2/C1 workgroups per transform, instead of one
// persistent workgroup walking all 32 blocks of its transform serially (86 us for batch 1).
template <int LOGN, bool FWD>
__global__ __launch_bounds__(256) void k_fft_4step_cols(const cpx *__restrict__ data, cpx *__restrict__ scratch,
                                                        const cpx *__restrict__ tabs_g) {
  using G = FourGeom<LOGN>;
  __shared__ cpx s_tabs[G::TABS];
  __shared__ cpx s_x[G::SL];
  const int tid = threadIdx.x;
  for (int i = tid; i < G::TABS; i += 256) s_tabs[i] = tabs_g[i];
  const cpx *tab1 = s_tabs, *tlo = s_tabs + G::N1 / 2 + G::N2 / 2, *thi = tlo + G::LO;
  __syncthreads();
  const long b = blockIdx.y;
  four_phase1<LOGN, FWD, 0>(data + b * (long)G::N, scratch + b * (long)G::N, blockIdx.x, tid, tab1, tlo, thi, s_x);
}
template <int LOGN, bool FWD, bool SCALE>
__global__ __launch_bounds__(256) void k_fft_4step_rows(cpx *__restrict__ data, const cpx *__restrict__ scratch,
                                                        const cpx *__restrict__ tabs_g) {
  using G = FourGeom<LOGN>;
  __shared__ cpx s_tabs[G::TABS];
  __shared__ cpx s_x[G::SL];
  const int tid = threadIdx.x;
  for (int i = tid; i < G::TABS; i += 256) s_tabs[i] = tabs_g[i];
  const cpx *tab2 = s_tabs + G::N1 / 2;
  __syncthreads();
  const long b = blockIdx.y;
  four_phase2<LOGN, FWD, SCALE, 0>(scratch + b * (long)G::N, data + b * (long)G::N, blockIdx.x, tid, tab2, s_x);
}

int fourstep_grid(const DeviceInfo &di) { return di.num_cus; }   // one 512-lane workgroup per CU

template <int LOGN, bool FWD, bool SCALE>
static hipError_t launch_4step_v(cpx *data, cpx *scratch, const FftTables &t, long batch, const DeviceInfo &di,
                                 hipStream_t s, long out_off) {
  int grid = fourstep_grid(di);
  if (batch * 4 <= grid && batch <= 65535) {
    // few transforms: spread each over its column / row blocks (scratch holds `grid` transforms)
    using G = FourGeom<LOGN>;
    hipLaunchKernelGGL((k_fft_4step_cols<LOGN, FWD>), dim3(G::NCB, (unsigned)batch), dim3(256), 0, s, data, scratch, t.four);
    hipLaunchKernelGGL((k_fft_4step_rows<LOGN, FWD, SCALE>), dim3(G::NRB, (unsigned)batch), dim3(256), 0, s, data + out_off,
                       scratch, t.four);
    return hipGetLastError();
  }
  if constexpr (LOGN == 16) {
    // n = 65536: the resident kernel (fft_resident.hip); `scratch` provides its per-workgroup slots
    return launch_fft_res16(FWD, SCALE, data, data + out_off, scratch, t.res16, batch, di, s);
  } else {
    if (batch < grid) grid = (int)batch;
    hipLaunchKernelGGL((k_fft_4step<LOGN, FWD, SCALE>), dim3(grid), dim3(512), 0, s, data, scratch, t.four, batch, out_off);
    return hipGetLastError();
  }
}

template <int LOGN>
static hipError_t launch_4step_n(bool fwd, bool scale, cpx *data, cpx *scratch, const FftTables &t, long batch,
                                 const DeviceInfo &di, hipStream_t s, long out_off) {
  if (fwd && scale) return launch_4step_v<LOGN, true, true>(data, scratch, t, batch, di, s, out_off);
  if (fwd && !scale) return launch_4step_v<LOGN, true, false>(data, scratch, t, batch, di, s, out_off);
  if (!fwd && !scale) return launch_4step_v<LOGN, false, false>(data, scratch, t, batch, di, s, out_off);
  return hipErrorInvalidValue;
}

hipError_t launch_fft_4step(int logn, bool fwd, bool scale, cpx *data, cpx *scratch, const FftTables &t, long batch,
                            const DeviceInfo &di, hipStream_t s, long out_off) {
  if (batch <= 0) return hipSuccess;
  switch (logn) {
    case 14: return launch_4step_n<14>(fwd, scale, data, scratch, t, batch, di, s, out_off);
    case 15: return launch_4step_n<15>(fwd, scale, data, scratch, t, batch, di, s, out_off);
    case 16: return launch_4step_n<16>(fwd, scale, data, scratch, t, batch, di, s, out_off);
    default: return hipErrorInvalidValue;
  }
}

const char *name_fft_4step(int logn) { return logn == 16 ? "k_fft_res16" : "k_fft_4step"; }


// ---------------------------------------------------------------------------------
// n = 2^17 .. 2^24: beyond the reference's reach (its stage kernel overflows int32 above 65536,
// cl_fft.cpp:32) — an extension, composed from the kernels above
// ---------------------------------------------------------------------------------
// Above 2^22 (two passes up to there, see k_big2_*): n = N1 x N2, N1 = 128, 256 (columns), N2 = 65536 (rows):
//   1. k_big_cols: N1-point FFT down 16..128 adjacent columns of data[n1][n2], times W_n^(n2 k1),
//      to scratch[k1][n2]                                                       (16 B/sample)
//   2. the batched row kernel of this file over the n-contiguous rows of scratch, N1 * batch of them,
//      in place: k_fft_lds (N2 <= 8192, 16 B/sample) or the four-step kernel (32 B/sample)
//   3. k_big_transpose: scratch[k1][k2] -> data[k2 * N1 + k1] (natural order), times 1/n for forward
//      plans                                                                     (16 B/sample)
// Twiddles W_n^e: big_tw() below.

#ifndef CLFA_BIG2_MAX
#define CLFA_BIG2_MAX 22   // the largest two-pass size (20: round 4's three passes for 2^21 and 2^22, for A/B builds)
#endif
constexpr int kBig2MaxLog = CLFA_BIG2_MAX;
#ifndef CLFA_BIG2_ODD_UP
#define CLFA_BIG2_ODD_UP 1   // n = 2^21: the 2048-point factor in the rows (0) or in the columns (1: 2.04 -> 2.30 TB/s)
#endif

int big_split(int logn, BigGeom *g) {
  if (logn <= kMaxLog || logn > kBigMaxLog) return -1;
  g->logn = logn;
  if (logn <= kBig2MaxLog) {   // two passes, N1 x N2 with both <= 2048 (k_big2_cols / k_big2_rows)
    g->logn1 = (logn + (logn == 21 ? CLFA_BIG2_ODD_UP : 0)) / 2;
    g->logn2 = logn - g->logn1;
  } else {            // three passes
    g->logn2 = logn == 21 ? 13 : logn - 8;
    g->logn1 = logn - g->logn2;
  }
  g->two_run = true;   // (the one-run form of the 1024-point blocks lost its A/B by 1.4-6 % and left the library in round 4)
  return 0;
}

#ifndef CLFA_BIG_XCD
#define CLFA_BIG_XCD 1   // the column / row block a workgroup of the two-pass kernels takes: XCD-compact (xcd_first) or blockIdx.x
#endif
#define CLFA_BIGX ((int)(CLFA_BIG_XCD ? xcd_first(blockIdx.x, gridDim.x) : blockIdx.x))
// W_n^e between the passes: e = e0 + 128 e1 + 16384 e2 from three tables of 128, 128 and n / 16384 entries (each rounded from
// double) in LDS, two multiplies: rms error of the factor 3.9e-8 where the two-table form it replaces had 3.4e-8 (a first table
// of W_n^e0 - 1, applied as a + a d, is no better: 4.0e-8), against 2.5-4e-7 of a whole transform.
// (Rounds 2-4 read two tables of 4096 and n / 4096 entries from global memory, one multiply: 64 more vector-memory instructions
// per lane and block than the 64 that move the data — without them the column passes run 12-34 % faster,
// profiles/big_two_pass_r05.txt.)
constexpr int kBigTwFixed = 256;   // entries of the first two tables
__device__ __forceinline__ void big_tw_fill(cpx *s_tw, const cpx *tw_g, int ntw, int tid, int lanes) {
  for (int i = tid; i < ntw; i += lanes) s_tw[i] = tw_g[i];
}
__device__ __forceinline__ cpx big_tw(const cpx *s_tw, int ex) {
  return cmul(cmul(s_tw[ex & 127], s_tw[128 + ((ex >> 7) & 127)]), s_tw[kBigTwFixed + (ex >> 14)]);
}
template <int LOGN1, bool FWD>
__global__ __launch_bounds__(256) void k_big_cols(const cpx *__restrict__ data, cpx *__restrict__ scratch,
                                                  const cpx *__restrict__ tabs_g, int logn2, int ntw) {
  constexpr int N1 = 1 << LOGN1, T1 = N1 / 16, C1 = 256 / T1;
  __shared__ cpx s_tab1[N1 / 2];
  __shared__ cpx s_tw[kBigTwFixed + (1 << (kBigMaxLog - 14))];
  __shared__ cpx s_x[N1 * C1];
  const int tid = threadIdx.x;
  for (int i = tid; i < N1 / 2; i += 256) s_tab1[i] = tabs_g[i];
  big_tw_fill(s_tw, tabs_g + N1 / 2, ntw, tid, 256);
  const int col = tid % C1, tf = tid / C1;
  const int n2 = CLFA_BIGX * C1 + col;
  const long base = ((long)blockIdx.y << (LOGN1 + logn2)) + n2;
  cpx v[16];
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = ld_nt(data + base + ((long)(tf + T1 * e) << logn2));
  __syncthreads();
  pass_compute<LOGN1, 4, 0, FWD>(v, tf, s_tab1);
  pass_scatter<LOGN1, 4, 0>(v, tf, [&](int p, cpx val) { s_x[p * C1 + col] = val; });
  __syncthreads();
  pass_gather<LOGN1, 4>(v, tf, [&](int p) { return s_x[p * C1 + col]; });
  pass_compute<LOGN1, 4, 4, FWD>(v, tf, s_tab1);
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k1 = tf + T1 * e;
    const int ex = n2 * k1;  // < n <= 2^24
    scratch[base + ((long)k1 << logn2)] = cmulc<!FWD>(v[e], big_tw(s_tw, ex));
  }
}

template <int LOGN1, bool SCALE>
__global__ __launch_bounds__(256) void k_big_transpose(const cpx *__restrict__ scratch, cpx *__restrict__ data,
                                                       int logn2, float inv_n) {
  constexpr int N1 = 1 << LOGN1, TK1 = N1 < 64 ? N1 : 64, TK2 = 4096 / TK1;
  __shared__ cpx tile[TK1 * (TK2 + 1)];
  const int tid = threadIdx.x;
  const int k2_0 = CLFA_BIGX * TK2, k1_0 = blockIdx.y * TK1;
  const long tbase = (long)blockIdx.z << (LOGN1 + logn2);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int i = tid + 256 * r, k1 = i / TK2, k2 = i % TK2;
    tile[k1 * (TK2 + 1) + k2] = scratch[tbase + ((long)(k1_0 + k1) << logn2) + k2_0 + k2];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int i = tid + 256 * r, k2 = i / TK1, k1 = i % TK1;
    cpx o = tile[k1 * (TK2 + 1) + k2];
    if constexpr (SCALE) o = cscale(o, inv_n);
    st_nt(data + tbase + ((long)(k2_0 + k2) << LOGN1) + k1_0 + k1, o);
  }
}

template <int LOGN1>
static hipError_t launch_big_n1(const BigGeom &g, bool fwd, bool scale, cpx *data, cpx *out, cpx *scratch, cpx *scratch2,
                                const cpx *bigtabs, const FftTables &sub, long batch, const DeviceInfo &di,
                                hipStream_t s) {
  constexpr int N1 = 1 << LOGN1, T1 = N1 / 16, C1 = 256 / T1, TK1 = N1 < 64 ? N1 : 64, TK2 = 4096 / TK1;
  const int n2 = 1 << g.logn2;
  const dim3 gc(n2 / C1, (unsigned)batch), gt(n2 / TK2, N1 / TK1, (unsigned)batch);
  if (fwd) hipLaunchKernelGGL((k_big_cols<LOGN1, true>), gc, dim3(256), 0, s, data, scratch, bigtabs, g.logn2, kBigTwFixed + (1 << (g.logn > 14 ? g.logn - 14 : 0)));
  else hipLaunchKernelGGL((k_big_cols<LOGN1, false>), gc, dim3(256), 0, s, data, scratch, bigtabs, g.logn2, kBigTwFixed + (1 << (g.logn > 14 ? g.logn - 14 : 0)));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (g.logn2 <= kLdsMaxLog) e = launch_fft_lds(g.logn2, fwd, MODE_C2C, false, scratch, sub, batch * N1, di, s, 0);
  else e = launch_fft_4step(g.logn2, fwd, false, scratch, scratch2, sub, batch * N1, di, s, 0);
  if (e != hipSuccess) return e;
  const float inv_n = 1.0f / (float)(1L << g.logn);
  if (scale) hipLaunchKernelGGL((k_big_transpose<LOGN1, true>), gt, dim3(256), 0, s, scratch, out, g.logn2, inv_n);
  else hipLaunchKernelGGL((k_big_transpose<LOGN1, false>), gt, dim3(256), 0, s, scratch, out, g.logn2, inv_n);
  return hipGetLastError();
}

// ---- n = 2^17 .. 2^22 in TWO passes (32 B/sample): both factors <= 2048, so a block of 16 columns
// (pass 1) or 16 rows (pass 2) of one transform fits the LDS of a CU (128-139 KiB, one workgroup of
// N1 resp. N2 lanes per CU) and both passes move 128-byte segments:
//   k_big2_cols: data[n1][16 columns] -> N1-point FFTs, times W_n^(n2 k1) -> scratch[k1][n2]
//   k_big2_rows: scratch[16 rows k1][n2] -> N2-point FFTs -> data[k2 * N1 + k1] (natural order)
template <int LOGN1, int LOGNS, bool FWD>
__device__ __forceinline__ void col_passes(cpx (&v)[16], int tf, const cpx *tab, cpx *sx, int col) {
  pass_compute<LOGN1, 4, LOGNS, FWD>(v, tf, tab);
  constexpr int LOGR = pass_logr(LOGN1, 4, LOGNS);
  if constexpr (LOGNS + LOGR < LOGN1) {
    __syncthreads();
    pass_scatter<LOGN1, 4, LOGNS>(v, tf, [&](int p, cpx val) { sx[p * 16 + col] = val; });
    __syncthreads();
    pass_gather<LOGN1, 4>(v, tf, [&](int p) { return sx[p * 16 + col]; });
    col_passes<LOGN1, LOGNS + LOGR, FWD>(v, tf, tab, sx, col);
  }
}
template <int LOGN1, bool FWD>
__global__ __launch_bounds__(1 << LOGN1) void k_big2_cols(const cpx *__restrict__ data, cpx *__restrict__ scratch,
                                                          const cpx *__restrict__ tabs_g, int logn2, int ntw) {
  constexpr int N1 = 1 << LOGN1, T1 = N1 / 16;
  __shared__ cpx s_tab1[N1 / 2];
  __shared__ cpx s_tw[kBigTwFixed + (1 << (2 * LOGN1 + 1 - 14))];   // n <= 2^(2 LOGN1 + 1)
  __shared__ cpx s_x[N1 * 16];
  const int tid = threadIdx.x;
  for (int i = tid; i < N1 / 2; i += N1) s_tab1[i] = tabs_g[i];
  big_tw_fill(s_tw, tabs_g + N1 / 2, ntw, tid, N1);
  const int col = tid % 16, tf = tid / 16;
  const int n2 = CLFA_BIGX * 16 + col;
  const long base = ((long)blockIdx.y << (LOGN1 + logn2)) + n2;
  cpx v[16];
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = ld_nt(data + base + ((long)(tf + T1 * e) << logn2));
  __syncthreads();
  col_passes<LOGN1, 0, FWD>(v, tf, s_tab1, s_x, col);
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k1 = tf + T1 * e;
    const int ex = n2 * k1;  // < n <= 2^19
    scratch[base + ((long)k1 << logn2)] = cmulc<!FWD>(v[e], big_tw(s_tw, ex));
  }
}

// N1 = 1024 in the two-run form of k_cfft_2x: the block's 16 columns x 1024 rows as two
// 512-point runs per column (even / odd rows) through ONE 64 KiB exchange buffer and a radix-2 step in registers —
// 512 lanes and half the LDS, so two workgroups share a CU where the one-run form (128 KiB) leaves one
// (LOGC = 10, round 5: N1 = 2048 as two 1024-point runs, 1024 lanes and 140 KiB of LDS, one workgroup per CU — what puts
// n = 2^21 and 2^22 on two passes of 128-byte segments: 1.70 -> 2.30 / 2.12 TB/s algorithmic.  Blocks of 8 columns, the other
// way to fit 2048 rows into LDS, copy at 3.8-4.1 TB/s against 5.3 for 16; a persistent form of both kernels that loads the
// next block under the stores of this one measured slower (profiles/big_two_pass_r05.txt).)
template <int LOGC, bool FWD>
__global__ __launch_bounds__(1 << LOGC, LOGC == 9 ? 4 : 1) void k_big2_cols_2x(const cpx *__restrict__ data, cpx *__restrict__ scratch,
                                                         const cpx *__restrict__ tabs_g, int logn2, int ntw) {
  constexpr int M = 1 << LOGC, TC = M / 16;   // M-point runs, TC lanes per column
  __shared__ cpx s_tabh[M / 2];   // W_M^k
  __shared__ cpx s_tabj[M];       // W_2M^k, k < M (the radix-2 step)
  __shared__ cpx s_tw[kBigTwFixed + (1 << (2 * (LOGC + 1) - 14))];   // n <= 2^(2 (LOGC + 1))
  __shared__ cpx s_x[M * 16];
  const int tid = threadIdx.x;
  s_tabj[tid] = tabs_g[tid];
  if (tid < M / 2) s_tabh[tid] = tabs_g[2 * tid];
  big_tw_fill(s_tw, tabs_g + M, ntw, tid, M);
  const int col = tid % 16, tf = tid / 16;
  const int n2 = CLFA_BIGX * 16 + col;
  const long base = ((long)blockIdx.y << (LOGC + 1 + logn2)) + n2;
  cpx va[16], vb[16];
#pragma unroll
  for (int e = 0; e < 16; e++) {
    va[e] = ld_nt(data + base + ((long)(2 * (tf + TC * e)) << logn2));
    vb[e] = ld_nt(data + base + ((long)(2 * (tf + TC * e) + 1) << logn2));
  }
  __syncthreads();
  col_passes<LOGC, 0, FWD>(va, tf, s_tabh, s_x, col);
  __syncthreads();
  col_passes<LOGC, 0, FWD>(vb, tf, s_tabh, s_x, col);
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k = tf + TC * e;
    const cpx p = cmulc<!FWD>(vb[e], s_tabj[k]);
    const cpx o0 = cadd(va[e], p), o1 = csub(va[e], p);
    const int ex0 = n2 * k, ex1 = n2 * (k + M);  // < n <= 2^22
    scratch[base + ((long)k << logn2)] = cmulc<!FWD>(o0, big_tw(s_tw, ex0));
    scratch[base + ((long)(k + M) << logn2)] = cmulc<!FWD>(o1, big_tw(s_tw, ex1));
  }
}

// all passes but the last with the lanes of a row adjacent (tf fast); the last one with the 16 rows on
// the fast lane index, so that the transposed store is 128-byte segments
template <int LOGN2, int LOGNS, bool FWD>
__device__ __forceinline__ void row_passes(cpx (&v)[16], int l, const cpx *tab, cpx *sx) {
  constexpr int T2 = (1 << LOGN2) / 16, S2 = lds_padded_size(1 << LOGN2) | 1;
  constexpr int LOGR = pass_logr(LOGN2, 4, LOGNS), NEXT = LOGNS + LOGR;
  const int tf = l % T2, row = l / T2;
  pass_compute<LOGN2, 4, LOGNS, FWD>(v, tf, tab);
  __syncthreads();
  pass_scatter_padded<LOGN2, 4, LOGNS>(v, tf, sx + row * S2);
  __syncthreads();
  if constexpr (NEXT + pass_logr(LOGN2, 4, NEXT) < LOGN2) {
    pass_gather_padded<LOGN2, 4>(v, tf, sx + row * S2);
    row_passes<LOGN2, NEXT, FWD>(v, l, tab, sx);
  } else {
    const int row2 = l % 16, tf2 = l / 16;
    pass_gather_padded<LOGN2, 4>(v, tf2, sx + row2 * S2);
    pass_compute<LOGN2, 4, NEXT, FWD>(v, tf2, tab);
  }
}
template <int LOGN2, bool FWD, bool SCALE>
__global__ __launch_bounds__(1 << LOGN2) void k_big2_rows(const cpx *__restrict__ scratch, cpx *__restrict__ data,
                                                          const cpx *__restrict__ tab_g, int logn1, float inv_n) {
  constexpr int N2 = 1 << LOGN2, T2 = N2 / 16, S2 = lds_padded_size(N2) | 1;
  __shared__ cpx s_tab2[N2 / 2];
  __shared__ cpx s_x[16 * S2];
  const int l = threadIdx.x;
  for (int i = l; i < N2 / 2; i += N2) s_tab2[i] = tab_g[i];
  const long tbase = (long)blockIdx.y << (LOGN2 + logn1);
  cpx v[16];
  {
    const int tf = l % T2, row = l / T2;
    const cpx *p = scratch + tbase + ((long)(CLFA_BIGX * 16 + row) << LOGN2) + tf;
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = p[T2 * e];
  }
  __syncthreads();
  row_passes<LOGN2, 0, FWD>(v, l, s_tab2, s_x);
  const int row2 = l % 16, tf2 = l / 16;
  cpx *dst = data + tbase + CLFA_BIGX * 16 + row2;
#pragma unroll
  for (int e = 0; e < 16; e++) {
    cpx o = v[e];
    if constexpr (SCALE) o = cscale(o, inv_n);
    st_nt(dst + ((long)(tf2 + T2 * e) << logn1), o);
  }
}

// N2 = 1024 in the two-run form: 16 rows x 1024 points as two 512-point runs per row (one 16-byte load per lane brings
// an even and an odd sample), the last pass with the rows on the fast lane index as above, radix-2 step in registers
template <int LOGC, bool FWD, bool SCALE>
__global__ __launch_bounds__(1 << LOGC, LOGC == 9 ? 4 : 1) void k_big2_rows_2x(const cpx *__restrict__ scratch, cpx *__restrict__ data,
                                                         const cpx *__restrict__ tab_g, int logn1, float inv_n) {
  constexpr int M = 1 << LOGC, TC = M / 16, S2 = lds_padded_size(M) | 1;
  __shared__ cpx s_tabh[M / 2];   // W_M^k
  __shared__ cpx s_tabj[M];       // W_2M^k, k < M
  __shared__ cpx s_x[16 * S2];
  const int l = threadIdx.x;
  s_tabj[l] = tab_g[l];
  if (l < M / 2) s_tabh[l] = tab_g[2 * l];
  const long tbase = (long)blockIdx.y << (LOGC + 1 + logn1);
  cpx va[16], vb[16];
  {
    const int tf = l % TC, row = l / TC;
    const cpx *p = scratch + tbase + ((long)(CLFA_BIGX * 16 + row) << (LOGC + 1)) + 2 * tf;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const f4v q = *reinterpret_cast<const f4v *>(p + 2 * TC * e);
      va[e] = mk(q.x, q.y);
      vb[e] = mk(q.z, q.w);
    }
  }
  __syncthreads();
  row_passes<LOGC, 0, FWD>(va, l, s_tabh, s_x);
  __syncthreads();
  row_passes<LOGC, 0, FWD>(vb, l, s_tabh, s_x);
  const int row2 = l % 16, tf2 = l / 16;
  cpx *dst = data + tbase + CLFA_BIGX * 16 + row2;
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k = tf2 + TC * e;
    const cpx p = cmulc<!FWD>(vb[e], s_tabj[k]);
    cpx o0 = cadd(va[e], p), o1 = csub(va[e], p);
    if constexpr (SCALE) {
      o0 = cscale(o0, inv_n);
      o1 = cscale(o1, inv_n);
    }
    st_nt(dst + ((long)k << logn1), o0);
    st_nt(dst + ((long)(k + M) << logn1), o1);
  }
}

template <int LOGN1>
static hipError_t launch_big2_cols(const BigGeom &g, bool fwd, const cpx *data, cpx *scratch, const cpx *bigtabs,
                                   long batch, hipStream_t s) {
  const dim3 grid((1 << g.logn2) / 16, (unsigned)batch);
  if (fwd) hipLaunchKernelGGL((k_big2_cols<LOGN1, true>), grid, dim3(1 << LOGN1), 0, s, data, scratch, bigtabs, g.logn2, kBigTwFixed + (1 << (g.logn > 14 ? g.logn - 14 : 0)));
  else hipLaunchKernelGGL((k_big2_cols<LOGN1, false>), grid, dim3(1 << LOGN1), 0, s, data, scratch, bigtabs, g.logn2, kBigTwFixed + (1 << (g.logn > 14 ? g.logn - 14 : 0)));
  return hipGetLastError();
}
template <int LOGN2>
static hipError_t launch_big2_rows(const BigGeom &g, bool fwd, bool scale, const cpx *scratch, cpx *data,
                                   const cpx *half2, long batch, hipStream_t s) {
  const dim3 grid((1 << g.logn1) / 16, (unsigned)batch);
  const float inv_n = 1.0f / (float)(1L << g.logn);
  if (fwd && scale) hipLaunchKernelGGL((k_big2_rows<LOGN2, true, true>), grid, dim3(1 << LOGN2), 0, s, scratch, data, half2, g.logn1, inv_n);
  else if (fwd) hipLaunchKernelGGL((k_big2_rows<LOGN2, true, false>), grid, dim3(1 << LOGN2), 0, s, scratch, data, half2, g.logn1, inv_n);
  else hipLaunchKernelGGL((k_big2_rows<LOGN2, false, false>), grid, dim3(1 << LOGN2), 0, s, scratch, data, half2, g.logn1, inv_n);
  return hipGetLastError();
}
template <int LOGC>
static hipError_t launch_big2_cols_2x(const BigGeom &g, bool fwd, const cpx *data, cpx *scratch, const cpx *bigtabs, long batch, hipStream_t s) {
  const dim3 grid((1 << g.logn2) / 16, (unsigned)batch);
  if (fwd) hipLaunchKernelGGL((k_big2_cols_2x<LOGC, true>), grid, dim3(1 << LOGC), 0, s, data, scratch, bigtabs, g.logn2, kBigTwFixed + (1 << (g.logn > 14 ? g.logn - 14 : 0)));
  else hipLaunchKernelGGL((k_big2_cols_2x<LOGC, false>), grid, dim3(1 << LOGC), 0, s, data, scratch, bigtabs, g.logn2, kBigTwFixed + (1 << (g.logn > 14 ? g.logn - 14 : 0)));
  return hipGetLastError();
}
template <int LOGC>
static hipError_t launch_big2_rows_2x(const BigGeom &g, bool fwd, bool scale, const cpx *scratch, cpx *out, const cpx *half2, long batch, hipStream_t s) {
  const dim3 grid((1 << g.logn1) / 16, (unsigned)batch);
  const float inv_n = 1.0f / (float)(1L << g.logn);
  if (fwd && scale) hipLaunchKernelGGL((k_big2_rows_2x<LOGC, true, true>), grid, dim3(1 << LOGC), 0, s, scratch, out, half2, g.logn1, inv_n);
  else if (fwd) hipLaunchKernelGGL((k_big2_rows_2x<LOGC, true, false>), grid, dim3(1 << LOGC), 0, s, scratch, out, half2, g.logn1, inv_n);
  else hipLaunchKernelGGL((k_big2_rows_2x<LOGC, false, false>), grid, dim3(1 << LOGC), 0, s, scratch, out, half2, g.logn1, inv_n);
  return hipGetLastError();
}
static hipError_t launch_fft_big2(const BigGeom &g, bool fwd, bool scale, cpx *data, cpx *out, cpx *scratch, const cpx *bigtabs,
                                  const FftTables &sub, long batch, hipStream_t s) {
  hipError_t e;
  switch (g.logn1) {
    case 8: e = launch_big2_cols<8>(g, fwd, data, scratch, bigtabs, batch, s); break;
    case 9: e = launch_big2_cols<9>(g, fwd, data, scratch, bigtabs, batch, s); break;
    case 10: e = launch_big2_cols_2x<9>(g, fwd, data, scratch, bigtabs, batch, s); break;    // 1024-point columns as two 512-point runs (two workgroups per CU)
    case 11: e = launch_big2_cols_2x<10>(g, fwd, data, scratch, bigtabs, batch, s); break;   // 2048-point columns as two 1024-point runs
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  switch (g.logn2) {
    case 9: return launch_big2_rows<9>(g, fwd, scale, scratch, out, sub.half, batch, s);
    case 10: return launch_big2_rows_2x<9>(g, fwd, scale, scratch, out, sub.half, batch, s);
    case 11: return launch_big2_rows_2x<10>(g, fwd, scale, scratch, out, sub.half, batch, s);
    default: return hipErrorInvalidValue;
  }
}

// scratch: `batch` transforms (the caller chunks); scratch2: the four-step workspace when N2 > 8192
hipError_t launch_fft_big(const BigGeom &g, bool fwd, bool scale, cpx *data, cpx *out, cpx *scratch, cpx *scratch2,
                          const cpx *bigtabs, const FftTables &sub, long batch, const DeviceInfo &di, hipStream_t s) {
  if (batch <= 0) return hipSuccess;
  if (batch > 65535) return hipErrorInvalidValue;
  if (g.logn <= kBig2MaxLog) return launch_fft_big2(g, fwd, scale, data, out, scratch, bigtabs, sub, batch, s);
  switch (g.logn1) {
    case 5: return launch_big_n1<5>(g, fwd, scale, data, out, scratch, scratch2, bigtabs, sub, batch, di, s);
    case 6: return launch_big_n1<6>(g, fwd, scale, data, out, scratch, scratch2, bigtabs, sub, batch, di, s);
    case 7: return launch_big_n1<7>(g, fwd, scale, data, out, scratch, scratch2, bigtabs, sub, batch, di, s);
    case 8: return launch_big_n1<8>(g, fwd, scale, data, out, scratch, scratch2, bigtabs, sub, batch, di, s);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------
// stand-alone pack / unpack / reorder
// ---------------------------------------------------------------------------------

// reference conv (cl_fft.cpp:178-191) over a batch; thread per pair
__global__ __launch_bounds__(256) void k_r2c_pack(cpx *__restrict__ data, const cpx *__restrict__ w2, int m,
                                                  long total_pairs, long out_off) {
  const int hp = m / 2;
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total_pairs; g += (long)gridDim.x * 256) {
    long b = g / hp;
    int i = (int)(g % hp);
    const cpx *c = data + b * m;
    cpx *o = data + b * m + out_off;   // out_off = 0: in place
    if (i == 0) {
      cpx z = c[0];
      o[0] = mk((z.x + z.y) * .5f, (z.x - z.y) * .5f);
      if (out_off) o[hp] = c[hp];   // the bin the reference never visits (cl_fft.cpp:278) travels as it is
    } else {
      cpx oi, oj;
      r2c_pair(c[i], c[m - i], w2[i], oi, oj);
      o[i] = oi;
      o[m - i] = oj;
    }
  }
}
// reference iconv (cl_fft.cpp:192-205)
__global__ __launch_bounds__(256) void k_c2r_unpack(cpx *__restrict__ data, const cpx *__restrict__ w2, int m,
                                                    long total_pairs, long out_off) {
  const int hp = m / 2;
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total_pairs; g += (long)gridDim.x * 256) {
    long b = g / hp;
    int i = (int)(g % hp);
    const cpx *c = data + b * m;
    cpx *o = data + b * m + out_off;
    if (i == 0) {
      cpx z = c[0];
      o[0] = mk(z.x + z.y, z.x - z.y);
      if (out_off) o[hp] = c[hp];
    } else {
      cpx oi, oj;
      c2r_pair(c[i], c[m - i], w2[i], oi, oj);
      o[i] = oi;
      o[m - i] = oj;
    }
  }
}

static int grid_for(long items) {
  long g = (items + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  if (g < 1) g = 1;
  return (int)g;
}

hipError_t launch_r2c_pack(cpx *data, const cpx *w2, int m, long batch, hipStream_t s, long out_off) {
  long pairs = batch * (m / 2);
  if (pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_r2c_pack, dim3(grid_for(pairs)), dim3(256), 0, s, data, w2, m, pairs, out_off);
  return hipGetLastError();
}
hipError_t launch_c2r_unpack(cpx *data, const cpx *w2, int m, long batch, hipStream_t s, long out_off) {
  long pairs = batch * (m / 2);
  if (pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_c2r_unpack, dim3(grid_for(pairs)), dim3(256), 0, s, data, w2, m, pairs, out_off);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// arbitrary lengths (extension): Bluestein's chirp-z over the power-of-two kernels
// ---------------------------------------------------------------------------------
// The reference only knows powers of two (its callers pad, opcode.cpp:30-35).  For any other n,
//   X[k] = w[k] * sum_j (x[j] w[j]) conj(w)[k - j],   w[j] = exp(-+ i pi j^2 / n),
// a circular convolution of length m = 2^ceil(log2(2n - 1)): pre-multiply and zero-pad into the workspace,
// m-point forward transform (scaled 1/m), times B = DFT_m(conj(w) wrapped), m-point inverse transform
// (unscaled), post-multiply (and 1/n for forward plans).  Tables w, B come from the host in double.
__global__ __launch_bounds__(256) void k_blue_pre(const cpx *__restrict__ x, const cpx *__restrict__ w, cpx *__restrict__ a,
                                                  int n, int m, long total) {
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const long b = g / m;
    const int j = (int)(g - b * m);
    a[g] = j < n ? cmul(x[b * n + j], w[j]) : mk(0.f, 0.f);
  }
}
__global__ __launch_bounds__(256) void k_blue_mul(cpx *__restrict__ a, const cpx *__restrict__ bt, int m, long total) {
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total; g += (long)gridDim.x * 256) a[g] = cmul(a[g], bt[g % m]);
}
__global__ __launch_bounds__(256) void k_blue_post(const cpx *__restrict__ a, const cpx *__restrict__ w, cpx *__restrict__ x,
                                                   int n, int m, float scale, long total) {
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const long b = g / n;
    const int k = (int)(g - b * n);
    x[g] = cscale(cmul(a[b * m + k], w[k]), scale);
  }
}
// Bluestein in ONE launch for m <= 8192 (n <= 4096): a transform's chirp multiply, m-point forward transform, multiply by
// the chirp's spectrum, m-point inverse transform and second chirp multiply all happen in the registers + LDS of one
// workgroup, on the pass chains of k_fft_lds (same tables: the m-point plan's) — one read and one write of the data where
// the composed form (pre, plan, mul, plan, post) makes five passes over a zero-padded copy.  In place (x == y) is fine:
// a transform is read completely before any of it is written.
template <int LOGM>
__global__ __launch_bounds__(LdsGeom<LOGM>::WG, LdsGeom<LOGM>::MIN_WAVES) void k_blue_lds(const cpx *__restrict__ x, cpx *__restrict__ y,
                                                              const cpx *__restrict__ w, const cpx *__restrict__ bt,
                                                              const cpx *__restrict__ tab_g, int n, float scale, long batch) {
  using G = LdsGeom<LOGM>;
  constexpr int M = G::N, E = G::E, T = G::T, WG = G::WG, FPW = G::FPW;
  static_assert(G::LOGE == 4 && LOGM >= 8 && LOGM <= 13, "16 points per lane");
  constexpr bool TWO = kLdsTwoLevel(LOGM);
  __shared__ cpx s_tab[TWO ? kLaneLds : G::HALF];
  __shared__ cpx s_x[FPW * G::PADN];
  const int tid = threadIdx.x;
  const int f = FPW == 1 ? 0 : tid / T, t0 = FPW == 1 ? tid : tid % T;
  for (int i = tid; i < (TWO ? kLane13Lds : M / 2); i += WG) s_tab[TWO ? lane_lds_index(i) : i] = tab_g[i];
  cpx wl = mk(1.f, 0.f);
  if constexpr (TWO) wl = tab_g[kLane13Lds + t0];
  cpx *xb = s_x + f * G::PADN;
  const long groups = (batch + FPW - 1) / FPW;
  __syncthreads();
#pragma unroll 1
  for (long g = blockIdx.x; g < groups; g += gridDim.x) {
    int t = t0;   // opaque per iteration (see k_fft_lds)
    asm volatile("" : "+v"(t));
    const auto tab = [&]() {
      if constexpr (TWO) return LaneTab13{s_tab + kRow16Stride * (t & 15), s_tab + kRow16Lds + (t & 255), wl};
      else return static_cast<const cpx *>(s_tab);
    }();
    long b = g * FPW + f;
    b = b < batch ? b : batch - 1;   // lanes of a ragged last group redo the last transform (identical stores)
    const cpx *xi = x + b * (long)n;
    cpx v[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int j = t + T * e;
      v[e] = j < n ? cmul(xi[j], w[j]) : mk(0.f, 0.f);
    }
    wg_passes<LOGM, 4, 0, true>(v, t, tab, xb);
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = cmul(v[e], bt[t + T * e]);
    wg_passes<LOGM, 4, 0, false>(v, t, tab, xb);
    cpx *yo = y + b * (long)n;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int k = t + T * e;
      if (k < n) yo[k] = cscale(cmul(v[e], w[k]), scale);
    }
  }
}
template <int LOGM>
static hipError_t launch_blue_lds_m(const cpx *x, cpx *y, const cpx *w, const cpx *bt, const cpx *tab, int n, float scale,
                                    long batch, const DeviceInfo &di, hipStream_t s) {
  using G = LdsGeom<LOGM>;
  const long groups = (batch + G::FPW - 1) / G::FPW;
  static int occ = 0;
  if (occ == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_blue_lds<LOGM>, G::WG, 0) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      nb = 1;
    }
    occ = nb;
  }
  const long cap = (long)di.num_cus * occ;
  const int grid = (int)(groups < cap ? groups : cap);
  hipLaunchKernelGGL((k_blue_lds<LOGM>), dim3(grid < 1 ? 1 : grid), dim3(G::WG), 0, s, x, y, w, bt, tab, n, scale, batch);
  return hipGetLastError();
}
bool blue_lds_ok(int m) { return m >= 256 && m <= 8192; }
// x -> y (may be equal), batch transforms of n points; w = chirp (n), bt = its padded spectrum (m), tab = the m-point plan's
// LDS table (FftTables::half); scale = the plan's output factor times 1 / m
hipError_t launch_blue_lds(int m, const cpx *x, cpx *y, const cpx *w, const cpx *bt, const cpx *tab, int n, float scale,
                           long batch, const DeviceInfo &di, hipStream_t s) {
  if (batch <= 0) return hipSuccess;
  switch (m) {
    case 256: return launch_blue_lds_m<8>(x, y, w, bt, tab, n, scale, batch, di, s);
    case 512: return launch_blue_lds_m<9>(x, y, w, bt, tab, n, scale, batch, di, s);
    case 1024: return launch_blue_lds_m<10>(x, y, w, bt, tab, n, scale, batch, di, s);
    case 2048: return launch_blue_lds_m<11>(x, y, w, bt, tab, n, scale, batch, di, s);
    case 4096: return launch_blue_lds_m<12>(x, y, w, bt, tab, n, scale, batch, di, s);
    case 8192: return launch_blue_lds_m<13>(x, y, w, bt, tab, n, scale, batch, di, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_blue_pre(const cpx *x, const cpx *w, cpx *a, int n, int m, long batch, hipStream_t s) {
  const long total = batch * m;
  hipLaunchKernelGGL(k_blue_pre, dim3(grid_for(total)), dim3(256), 0, s, x, w, a, n, m, total);
  return hipGetLastError();
}
hipError_t launch_blue_mul(cpx *a, const cpx *bt, int m, long batch, hipStream_t s) {
  const long total = batch * m;
  hipLaunchKernelGGL(k_blue_mul, dim3(grid_for(total)), dim3(256), 0, s, a, bt, m, total);
  return hipGetLastError();
}
hipError_t launch_blue_post(const cpx *a, const cpx *w, cpx *x, int n, int m, float scale, long batch, hipStream_t s) {
  const long total = batch * n;
  hipLaunchKernelGGL(k_blue_post, dim3(grid_for(total)), dim3(256), 0, s, a, w, x, n, m, scale, total);
  return hipGetLastError();
}

// reference reorder (cl_fft.cpp:24-27): out[k] = in[bitrev(k)].  The table of
// cl_fft.cpp:96-101 is exactly the log2(n)-bit reversal, computed here with
// v_bfrev_b32 instead of a table read.
__global__ __launch_bounds__(256) void k_reorder(cpx *__restrict__ out, const cpx *__restrict__ in, int logn,
                                                 long total) {
  const unsigned mask = (1u << logn) - 1u;
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    unsigned k = (unsigned)g & mask;
    unsigned r = __brev(k) >> (32 - logn);
    out[g] = in[(g - k) + r];
  }
}

hipError_t launch_reorder(cpx *out, const cpx *in, int logn, long batch, hipStream_t s) {
  long total = batch << logn;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_reorder, dim3(grid_for(total)), dim3(256), 0, s, out, in, logn, total);
  return hipGetLastError();
}

}  // namespace clfa

#ifdef CLFA_ASSIGN_SEARCH
extern "C" __attribute__((visibility("default"))) int clfa_debug_set_assign(const int *h, int n) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(clfa::g_assign), h, (size_t)n * 4);
}
#endif
