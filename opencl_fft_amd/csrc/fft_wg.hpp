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
