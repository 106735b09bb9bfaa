// fft_wg.hpp — workgroup-level pieces shared by fft_kernels.hip and conv_kernels.hip:
// the LDS-exchanged pass chain, its geometry, and the reference's r2c/c2r pair maps.
#pragma once
#include "internal.hpp"

namespace clfa {

// PAIRLAST: the last pass is pass_last_paired — the caller picks the pairs up with pairs_visit
// instead of storing positions tid + T*e.
template <int LOGN, int LOGE, int LOGNS, bool FWD, bool PAIRLAST = false, class Tab>
__device__ __forceinline__ void wg_passes(cpx (&v)[1 << LOGE], int t, const Tab &tab, cpx *xb) {
  pass_compute<LOGN, LOGE, LOGNS, FWD>(v, t, tab);
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  if constexpr (LOGNS + LOGR < LOGN) {
    __syncthreads();  // everybody is done reading the previous exchange
    pass_scatter_padded<LOGN, LOGE, LOGNS>(v, t, xb);
    __syncthreads();
    constexpr int NEXT = LOGNS + LOGR;
    if constexpr (PAIRLAST && NEXT + pass_logr(LOGN, LOGE, NEXT) == LOGN) {
      pass_last_paired<LOGN, LOGE, FWD>(v, t, tab, xb);
    } else {
      pass_gather_padded<LOGN, LOGE>(v, t, xb);
      wg_passes<LOGN, LOGE, NEXT, FWD, PAIRLAST>(v, t, tab, xb);
    }
  }
}

// The same chain with the MIDDLE passes worked by permuted lanes (T >= 512 lanes: n = 8192, 16384).  A gather reads
// positions L + T e, L = the lane's logical index; with L = tid the two 16-lane halves of a ds_read_b64 group read
// neighbouring 16-blocks of the padded buffer, 34 dwords apart, and lane 31 falls on lane 0's banks — every gather costs
// twice its LDS cycles (MI355X_MICROARCH.md: 32 lanes over 64 banks).  With L = sigma(tid) — bits 4 and 8 swapped — the
// halves are 256 positions = 544 dwords = 32 (mod 64) apart: conflict-free.  Which lane works which butterfly is free
// between the first pass (global loads: L = tid) and the last (global stores: L = tid), so the passes in between run on
// L = sigma(tid) with the twiddle table of that lane (`tabs`); the scatters see sixteen consecutive logical lanes per
// 16-lane write group either way.  Results are the plain chain's, value for value (tests/cpp/emulate_engine.cpp).
template <int LOGN, int LOGE, int LOGNS, bool FWD, bool PAIRLAST = false, class Tab>
__device__ __forceinline__ void wg_passes_sigma(cpx (&v)[1 << LOGE], int t, int ts, const Tab &tab, const Tab &tabs, cpx *xb) {
  static_assert(LOGN - LOGE >= 9, "needs bit 8 of the lane index");
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS), NEXT = LOGNS + LOGR;
  constexpr bool FIRST = LOGNS == 0, LAST = NEXT == LOGN;
  // the first and the last pass belong to lane t, the others to lane ts
  pass_compute<LOGN, LOGE, LOGNS, FWD>(v, (FIRST || LAST) ? t : ts, (FIRST || LAST) ? tab : tabs);
  if constexpr (!LAST) {
    constexpr bool NEXT_LAST = NEXT + pass_logr(LOGN, LOGE, NEXT) == LOGN;
    __syncthreads();  // everybody is done reading the previous exchange
    pass_scatter_padded<LOGN, LOGE, LOGNS>(v, FIRST ? t : ts, xb);
    __syncthreads();
    if constexpr (PAIRLAST && NEXT_LAST) {
      pass_last_paired<LOGN, LOGE, FWD>(v, t, tab, xb);
    } else {
      pass_gather_padded<LOGN, LOGE>(v, NEXT_LAST ? t : ts, xb);
      wg_passes_sigma<LOGN, LOGE, NEXT, FWD, PAIRLAST>(v, t, ts, tab, tabs, xb);
    }
  }
}

// the transposed chain from the pass (2^LOGNS, radix 2^LOGE) downwards; v holds what the pass
// before it (a bigger LOGNS) has just computed, i.e. positions tid + T*e
template <int LOGN, int LOGE, int LOGNS, bool FWD, class Tab>
__device__ __forceinline__ void wg_passes_dif_after(cpx (&v)[1 << LOGE], int t, const Tab &tab, cpx *xb) {
  dif_gather_padded<LOGN, LOGE, LOGNS>(v, t, xb);
  dif_compute<LOGN, LOGE, LOGNS, FWD>(v, t, tab);
  if constexpr (LOGNS > 0) {
    __syncthreads();
    dif_scatter_padded<LOGN, LOGE>(v, t, xb);
    __syncthreads();
    wg_passes_dif_after<LOGN, LOGE, LOGNS - LOGE, FWD>(v, t, tab, xb);
  }
}

// TWO independent transforms through ONE exchange buffer, staggered (packed real size 65536: the even and the odd
// half of a transform, k_rfft_2x; two runs of a complex transform, k_cfft_2x): while one transform's values are on their way through LDS the other's pass is
// computed.  Run one after the other, every wave of the workgroup does the same thing at the same time — all of them
// compute, then all of them wait for their scattered writes to drain (16 ds_write_b64 per lane move at a third of the
// LDS read rate), then all of them gather — and the chains, not the memory, bound the kernel.
// Same passes, same number of barriers as two wg_passes calls.
// Precondition: va holds the OUTPUT of pass LOGNS (computed), vb its INPUT (not yet computed).  PAIRLAST: the last
// pass is pass_last_paired (packed real transforms); otherwise both end as wg_passes does, at positions tid + T*e.
template <int LOGN, int LOGE, int LOGNS, bool FWD, bool PAIRLAST = true, class Tab>
__device__ __forceinline__ void wg_passes_pair(cpx (&va)[1 << LOGE], cpx (&vb)[1 << LOGE], int t, const Tab &tab, cpx *xb) {
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS), NEXT = LOGNS + LOGR;
  static_assert(NEXT < LOGN, "at least one more pass");
  constexpr bool LAST = NEXT + pass_logr(LOGN, LOGE, NEXT) == LOGN;   // the pass after this one is the last
  __syncthreads();   // everybody is done reading the previous exchange
  pass_scatter_padded<LOGN, LOGE, LOGNS>(va, t, xb);
  pass_compute<LOGN, LOGE, LOGNS, FWD>(vb, t, tab);          // under a's LDS writes
  __syncthreads();
  if constexpr (LAST && PAIRLAST) {
    pass_last_paired<LOGN, LOGE, FWD>(va, t, tab, xb, [&]() {
      __syncthreads();   // a's gather is complete in every wave
      pass_scatter_padded<LOGN, LOGE, LOGNS>(vb, t, xb);     // ... and b's writes run under a's last butterflies
    });
    __syncthreads();
    pass_last_paired<LOGN, LOGE, FWD>(vb, t, tab, xb);
  } else {
    pass_gather_padded<LOGN, LOGE>(va, t, xb);
    __syncthreads();
    pass_scatter_padded<LOGN, LOGE, LOGNS>(vb, t, xb);
    pass_compute<LOGN, LOGE, NEXT, FWD>(va, t, tab);         // under b's LDS writes
    __syncthreads();
    pass_gather_padded<LOGN, LOGE>(vb, t, xb);
    if constexpr (LAST) pass_compute<LOGN, LOGE, NEXT, FWD>(vb, t, tab);
    else wg_passes_pair<LOGN, LOGE, NEXT, FWD, PAIRLAST>(va, vb, t, tab, xb);
  }
}

// ... with the middle passes on permuted lanes (wg_passes_sigma above): the pass that starts at 2^LOGNS belongs to lane
// t if it is the first or the last of the chain, else to lane ts = lane_sigma(t) with that lane's tables `tabs`
template <int LOGN, int LOGE, int LOGNS, bool FWD, bool PAIRLAST = true, class Tab>
__device__ __forceinline__ void wg_passes_pair_sigma(cpx (&va)[1 << LOGE], cpx (&vb)[1 << LOGE], int t, int ts, const Tab &tab,
                                                     const Tab &tabs, cpx *xb) {
  static_assert(LOGN - LOGE >= 9, "needs bit 8 of the lane index");
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS), NEXT = LOGNS + LOGR;
  static_assert(NEXT < LOGN, "at least one more pass");
  constexpr bool LAST = NEXT + pass_logr(LOGN, LOGE, NEXT) == LOGN;   // the pass after this one is the last
  const int tc = LOGNS == 0 ? t : ts, tn = LAST ? t : ts;            // owners of this pass and of the next
  const Tab &tabc = LOGNS == 0 ? tab : tabs, &tabn = LAST ? tab : tabs;
  __syncthreads();   // everybody is done reading the previous exchange
  pass_scatter_padded<LOGN, LOGE, LOGNS>(va, tc, xb);
  pass_compute<LOGN, LOGE, LOGNS, FWD>(vb, tc, tabc);          // under a's LDS writes
  __syncthreads();
  if constexpr (LAST && PAIRLAST) {
    pass_last_paired<LOGN, LOGE, FWD>(va, t, tab, xb, [&]() {
      __syncthreads();   // a's gather is complete in every wave
      pass_scatter_padded<LOGN, LOGE, LOGNS>(vb, tc, xb);      // ... and b's writes run under a's last butterflies
    });
    __syncthreads();
    pass_last_paired<LOGN, LOGE, FWD>(vb, t, tab, xb);
  } else {
    pass_gather_padded<LOGN, LOGE>(va, tn, xb);
    __syncthreads();
    pass_scatter_padded<LOGN, LOGE, LOGNS>(vb, tc, xb);
    pass_compute<LOGN, LOGE, NEXT, FWD>(va, tn, tabn);         // under b's LDS writes
    __syncthreads();
    pass_gather_padded<LOGN, LOGE>(vb, tn, xb);
    if constexpr (LAST) pass_compute<LOGN, LOGE, NEXT, FWD>(vb, tn, tabn);
    else wg_passes_pair_sigma<LOGN, LOGE, NEXT, FWD, PAIRLAST>(va, vb, t, ts, tab, tabs, xb);
  }
}

// ... and the transposed chains from the pass (2^LOGNS, radix 2^LOGE) downwards.  Precondition: va has been through
// dif_compute<LOGNS>; vb has been gathered for that pass (dif_gather_padded<LOGNS>) but not computed yet.
template <int LOGN, int LOGE, int LOGNS, bool FWD, class Tab>
__device__ __forceinline__ void wg_passes_dif_pair(cpx (&va)[1 << LOGE], cpx (&vb)[1 << LOGE], int t, const Tab &tab, cpx *xb) {
  if constexpr (LOGNS > 0) {
    __syncthreads();
    dif_scatter_padded<LOGN, LOGE>(va, t, xb);
    dif_compute<LOGN, LOGE, LOGNS, FWD>(vb, t, tab);           // under a's LDS writes
    __syncthreads();
    dif_gather_padded<LOGN, LOGE, LOGNS - LOGE>(va, t, xb);
    __syncthreads();
    dif_scatter_padded<LOGN, LOGE>(vb, t, xb);
    dif_compute<LOGN, LOGE, LOGNS - LOGE, FWD>(va, t, tab);    // under b's LDS writes
    __syncthreads();
    dif_gather_padded<LOGN, LOGE, LOGNS - LOGE>(vb, t, xb);
    wg_passes_dif_pair<LOGN, LOGE, LOGNS - LOGE, FWD>(va, vb, t, tab, xb);
  } else {
    dif_compute<LOGN, LOGE, 0, FWD>(vb, t, tab);
  }
}

template <int LOGN> struct LdsGeom {
  static constexpr int N = 1 << LOGN;
  // 16 points per lane.  (32 points per lane with radix-32 passes was measured for n = 8192:
  // slower — 4.08 vs 4.38 TB/s — so the radix-32 butterfly stays in fft_device.hpp unused here.)
  static constexpr int LOGE = cmin(4, LOGN);
  // n = 8192 is capped at 128 VGPRs so that two 512-lane workgroups share a CU (LDS 71 KiB each thanks
  // to the two-level twiddle table); with packed arithmetic the register prefetch fits under the cap
  // (28 bytes of spill in c2c / r2c) and is worth 5 % (c2c 4.99 -> 5.22 TB/s, interleaved A/B)
  static constexpr bool PREFETCH = LOGN <= 14;
  static constexpr int MIN_WAVES = LOGN >= 13 ? 4 : 1;
  static constexpr int E = 1 << LOGE;
  static constexpr int T = N / E;                       // lanes per transform
  static constexpr int WG = T >= 256 ? T : 256;         // threads per workgroup
  static constexpr int FPW = WG / T;                    // transforms per workgroup
  static constexpr int PADN = lds_padded_size(N);
  static constexpr int HALF = N / 2 > 0 ? N / 2 : 1;
};

}  // namespace clfa
