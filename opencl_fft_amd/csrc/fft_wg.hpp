// fft_wg.hpp — workgroup-level pieces shared by fft_kernels.hip and conv_kernels.hip:
// the LDS-exchanged pass chain, its geometry, and the reference's r2c/c2r pair maps.
#pragma once
#include "internal.hpp"

namespace clfa {

template <int LOGN, int LOGE, int LOGNS, bool FWD, class Tab>
__device__ __forceinline__ void wg_passes(cpx (&v)[1 << LOGE], int t, const Tab &tab, cpx *xb) {
  pass_compute<LOGN, LOGE, LOGNS, FWD>(v, t, tab);
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  if constexpr (LOGNS + LOGR < LOGN) {
    __syncthreads();  // everybody is done reading the previous exchange
    pass_scatter_padded<LOGN, LOGE, LOGNS>(v, t, xb);
    __syncthreads();
    pass_gather_padded<LOGN, LOGE>(v, t, xb);
    wg_passes<LOGN, LOGE, LOGNS + LOGR, FWD>(v, t, tab, xb);
  }
}

template <int LOGN> struct LdsGeom {
  static constexpr int N = 1 << LOGN;
  // 16 points per lane.  (32 points per lane with radix-32 passes was measured for n = 8192:
  // slower — 4.08 vs 4.38 TB/s — so the radix-32 butterfly stays in fft_device.hpp unused here.)
  static constexpr int LOGE = cmin(4, LOGN);
  // n = 8192 runs WITHOUT the register prefetch but capped at 128 VGPRs, so that two 512-lane
  // workgroups share a CU (LDS 71 KiB each thanks to the two-level twiddle table)
  static constexpr bool PREFETCH = LOGN <= 12;
  static constexpr int MIN_WAVES = LOGN >= 13 ? 4 : 1;
  static constexpr int E = 1 << LOGE;
  static constexpr int T = N / E;                       // lanes per transform
  static constexpr int WG = T >= 256 ? T : 256;         // threads per workgroup
  static constexpr int FPW = WG / T;                    // transforms per workgroup
  static constexpr int PADN = lds_padded_size(N);
  static constexpr int HALF = N / 2 > 0 ? N / 2 : 1;
};

// reference conv kernel, cl_fft.cpp:178-191 (pair i, M-i; bin M/2 not visited)
__device__ __forceinline__ void r2c_pair(cpx ci, cpx cjraw, cpx w, cpx &oi, cpx &oj) {
  cpx cj = cconj(cjraw);
  cpx e = cscale(cadd(ci, cj), .5f);
  cpx d = csub(cj, ci);
  cpx o = cscale(mk(-d.y, d.x), .5f);
  cpx p = cmul(w, o);
  oi = cadd(e, p);
  oj = cconj(csub(e, p));
}
// reference iconv kernel, cl_fft.cpp:192-205
__device__ __forceinline__ void c2r_pair(cpx ci, cpx cjraw, cpx w, cpx &oi, cpx &oj) {
  cpx cj = cconj(cjraw);
  cpx e = cscale(cadd(ci, cj), .5f);
  cpx d = csub(ci, cj);
  cpx o = cscale(mk(-d.y, d.x), .5f);
  cpx p = cmul(w, o);
  oi = cadd(e, p);
  oj = cconj(csub(e, p));
}

}  // namespace clfa
