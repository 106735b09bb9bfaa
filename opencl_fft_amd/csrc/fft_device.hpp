// fft_device.hpp — register-level FFT building blocks for gfx950 (CDNA4).
//
// Design (not a translation of the reference's one-radix-2-stage-per-launch
// kernels, cl_fft.cpp:29-41): a transform of n = 2^LOGN points is computed by
// T = n/E cooperating lanes, each holding E = 2^LOGE (<=16) points in VGPRs,
// as a Stockham autosort sequence of radix-16 passes (plus one remainder pass
// of radix 2/4/8).  Lane `tid` always owns positions  tid + T*e  (e = 0..E-1)
// on entry to every pass and after the last one, so
//   * the first pass loads and the last pass stores are coalesced (lanes walk
//     consecutive addresses, registers are T apart),
//   * results come out in natural order: no bit-reversal gather pass exists
//     (the reference's `reorder` kernel, cl_fft.cpp:24-27, is fused away),
//   * between passes data cross lanes through LDS only.
// Twiddles W_n^k are looked up in a table rounded from double exactly like the
// reference's (cl_fft.cpp:86-91), never from fast-math sin/cos.
//
// The header is also host-compilable (CLFA_HD) so tests/cpp/emulate_engine.cpp
// can run the same pass code on the CPU, lane by lane.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CLFA_HD __host__ __device__ __forceinline__
#else
#define CLFA_HD inline
#endif

namespace clfa {

struct alignas(8) cpx {
  float x, y;
};

CLFA_HD cpx mk(float x, float y) { cpx r; r.x = x; r.y = y; return r; }
CLFA_HD cpx cadd(cpx a, cpx b) { return mk(a.x + b.x, a.y + b.y); }
CLFA_HD cpx csub(cpx a, cpx b) { return mk(a.x - b.x, a.y - b.y); }
CLFA_HD cpx cmul(cpx a, cpx b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
CLFA_HD cpx cscale(cpx a, float s) { return mk(a.x * s, a.y * s); }
CLFA_HD cpx cconj(cpx a) { return mk(a.x, -a.y); }
// multiply by W_4^1: -i for a forward transform, +i for an inverse one
template <bool FWD> CLFA_HD cpx rot4(cpx a) { return FWD ? mk(a.y, -a.x) : mk(-a.y, a.x); }
// constant twiddle (c, -s) forward / (c, +s) inverse
template <bool FWD> CLFA_HD cpx ctw(cpx a, float c, float s) {
  return FWD ? mk(a.x * c + a.y * s, a.y * c - a.x * s) : mk(a.x * c - a.y * s, a.y * c + a.x * s);
}

constexpr float kC8 = 0.70710678118654752440f;   // cos(pi/4)
constexpr float kC16 = 0.92387953251128675613f;  // cos(pi/8)
constexpr float kS16 = 0.38268343236508977173f;  // sin(pi/8)

// ---- in-register DFTs on v[u + U*t], t = 0..R-1; result q lands in v[u + U*q] ----

template <int U, int E, bool FWD> CLFA_HD void dft2(cpx (&v)[E], int u) {
  cpx a = v[u], b = v[u + U];
  v[u] = cadd(a, b);
  v[u + U] = csub(a, b);
}

// natural-order 4-point DFT of (a0,a1,a2,a3) -> (y0,y1,y2,y3)
template <bool FWD> CLFA_HD void bf4(cpx &a0, cpx &a1, cpx &a2, cpx &a3) {
  cpx s02 = cadd(a0, a2), d02 = csub(a0, a2);
  cpx s13 = cadd(a1, a3), d13 = rot4<FWD>(csub(a1, a3));
  a0 = cadd(s02, s13);
  a2 = csub(s02, s13);
  a1 = cadd(d02, d13);
  a3 = csub(d02, d13);
}

template <int U, int E, bool FWD> CLFA_HD void dft4(cpx (&v)[E], int u) {
  bf4<FWD>(v[u], v[u + U], v[u + 2 * U], v[u + 3 * U]);
}

// 8 = 4 x 2: t = 2a + b, q = q0 + 4*q1
template <int U, int E, bool FWD> CLFA_HD void dft8(cpx (&v)[E], int u) {
  cpx x[8];
#pragma unroll
  for (int t = 0; t < 8; t++) x[t] = v[u + U * t];
  bf4<FWD>(x[0], x[2], x[4], x[6]);  // b = 0: x[2*q0]
  bf4<FWD>(x[1], x[3], x[5], x[7]);  // b = 1: x[2*q0 + 1]
  x[3] = ctw<FWD>(x[3], kC8, kC8);   // W_8^1
  x[5] = rot4<FWD>(x[5]);            // W_8^2
  x[7] = ctw<FWD>(x[7], -kC8, kC8);  // W_8^3
#pragma unroll
  for (int q0 = 0; q0 < 4; q0++) {
    cpx a = x[2 * q0], b = x[2 * q0 + 1];
    v[u + U * q0] = cadd(a, b);
    v[u + U * (q0 + 4)] = csub(a, b);
  }
}

// 16 = 4 x 4: t = 4a + b, q = q0 + 4*q1
template <int U, int E, bool FWD> CLFA_HD void dft16(cpx (&v)[E], int u) {
  cpx x[16];
#pragma unroll
  for (int t = 0; t < 16; t++) x[t] = v[u + U * t];
#pragma unroll
  for (int b = 0; b < 4; b++) bf4<FWD>(x[b], x[4 + b], x[8 + b], x[12 + b]);  // -> x[4*q0 + b]
  // W_16^(b*q0)
  x[4 + 1] = ctw<FWD>(x[4 + 1], kC16, kS16);    // 1
  x[4 + 2] = ctw<FWD>(x[4 + 2], kC8, kC8);      // 2
  x[4 + 3] = ctw<FWD>(x[4 + 3], kS16, kC16);    // 3
  x[8 + 1] = ctw<FWD>(x[8 + 1], kC8, kC8);      // 2
  x[8 + 2] = rot4<FWD>(x[8 + 2]);               // 4
  x[8 + 3] = ctw<FWD>(x[8 + 3], -kC8, kC8);     // 6
  x[12 + 1] = ctw<FWD>(x[12 + 1], kS16, kC16);  // 3
  x[12 + 2] = ctw<FWD>(x[12 + 2], -kC8, kC8);   // 6
  x[12 + 3] = ctw<FWD>(x[12 + 3], -kC16, -kS16);// 9
#pragma unroll
  for (int q0 = 0; q0 < 4; q0++) {
    bf4<FWD>(x[4 * q0], x[4 * q0 + 1], x[4 * q0 + 2], x[4 * q0 + 3]);  // -> q1
#pragma unroll
    for (int q1 = 0; q1 < 4; q1++) v[u + U * (q0 + 4 * q1)] = x[4 * q0 + q1];
  }
}

// 32 = 16 x 2: t = 2a + b, q = q0 + 16*q1.  cos/sin(2 pi k/32), k = 1..15, as float literals.
constexpr float kC32[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                            0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                            0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                            -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                            -0.92387953251128675613f, -0.98078528040323044913f};
constexpr float kS32[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                            0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f,
                            0.98078528040323044913f, 1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                            0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                            0.38268343236508977173f, 0.19509032201612826785f};
template <int U, int E, bool FWD> CLFA_HD void dft32(cpx (&v)[E], int u) {
  cpx x[32];
#pragma unroll
  for (int t = 0; t < 32; t++) x[t] = v[u + U * t];
  dft16<2, 32, FWD>(x, 0);  // b = 0: inputs x[2a], results q0 at x[2*q0]
  dft16<2, 32, FWD>(x, 1);  // b = 1: x[2a+1] -> x[2*q0+1]
#pragma unroll
  for (int q0 = 0; q0 < 16; q0++) {
    cpx a = x[2 * q0], b = x[2 * q0 + 1];
    if (q0 > 0) b = ctw<FWD>(b, kC32[q0], kS32[q0]);  // W_32^q0
    v[u + U * q0] = cadd(a, b);
    v[u + U * (q0 + 16)] = csub(a, b);
  }
}

template <int LOGR, int U, int E, bool FWD> CLFA_HD void dft(cpx (&v)[E], int u) {
  if constexpr (LOGR == 1) dft2<U, E, FWD>(v, u);
  else if constexpr (LOGR == 2) dft4<U, E, FWD>(v, u);
  else if constexpr (LOGR == 3) dft8<U, E, FWD>(v, u);
  else if constexpr (LOGR == 4) dft16<U, E, FWD>(v, u);
  else if constexpr (LOGR == 5) dft32<U, E, FWD>(v, u);
}

// ---- Stockham pass schedule ---------------------------------------------------

constexpr int cmin(int a, int b) { return a < b ? a : b; }
// radix (log2) of the pass that starts with sub-transform length 2^LOGNS
constexpr int pass_logr(int LOGN, int LOGE, int LOGNS) { return cmin(LOGE, LOGN - LOGNS); }

// padded LDS position: one pad element per 16 keeps the stride-16 scatter of
// the first exchange and the contiguous gathers conflict-free (ds_*_b64)
CLFA_HD int lds_pad(int p) { return p + (p >> 4); }
constexpr int lds_padded_size(int n) { return n + (n >> 4) + 1; }

// Half-table twiddle lookup: tab holds W_n^k for k in [0, n/2) (forward sign);
// W_n^(k+n/2) = -W_n^k; inverse = conjugate.
template <int LOGN, bool FWD, class Tab> CLFA_HD cpx tw_lookup(const Tab &tab, int k) {
  if constexpr (LOGN == 0) return mk(1.f, 0.f);
  constexpr int half = (1 << LOGN) >> 1;
  cpx w = tab[k & (half - 1)];
  if (k & half) w = mk(-w.x, -w.y);
  if (!FWD) w.y = -w.y;
  return w;
}

// Two-level twiddle table: W_n^k = hi[k >> LOGLO] * lo[k & (2^LOGLO - 1)], hi[j] = W_n^(j * 2^LOGLO),
// lo[j] = W_n^j, both rounded from double.  n/2^LOGLO + 2^LOGLO entries instead of n/2: it lets
// the 8192-point kernel keep two workgroups per CU.  One extra complex multiply per lookup.
template <int LOGLO> struct TwoLevelTab {
  const cpx *hi;
  const cpx *lo;
};
template <int LOGN, bool FWD, int LOGLO> CLFA_HD cpx tw_lookup(const TwoLevelTab<LOGLO> &tab, int k) {
  cpx w = cmul(tab.hi[k >> LOGLO], tab.lo[k & ((1 << LOGLO) - 1)]);
  if (!FWD) w.y = -w.y;
  return w;
}

// One pass on the registers of lane `tid`: input twiddles then U butterflies.
template <int LOGN, int LOGE, int LOGNS, bool FWD, class Tab>
CLFA_HD void pass_compute(cpx (&v)[1 << LOGE], int tid, const Tab &tab) {
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  constexpr int E = 1 << LOGE, R = 1 << LOGR, U = E / R, T = 1 << (LOGN - LOGE), NS = 1 << LOGNS;
  if constexpr (LOGNS > 0) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      int jm = (tid + u * T) & (NS - 1);
#pragma unroll
      for (int t = 1; t < R; t++) {
        int k = (jm * t) << (LOGN - LOGNS - LOGR);
        v[u + U * t] = cmul(v[u + U * t], tw_lookup<LOGN, FWD>(tab, k));
#if defined(__HIP_DEVICE_COMPILE__)
        // with 32 points per lane hipcc otherwise hoists every table read of the pass ahead of
        // the multiplies (100+ live VGPRs of twiddles) and spills: fence the scheduler every 4
        if constexpr (E == 32) {
          if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
#endif
      }
    }
  }
#pragma unroll
  for (int u = 0; u < U; u++) dft<LOGR, U, E, FWD>(v, u);
}

// Scatter the outputs of the pass that started at 2^LOGNS into the exchange
// buffer; `st(pos, value)` receives natural (unpadded) positions.
template <int LOGN, int LOGE, int LOGNS, class St>
CLFA_HD void pass_scatter(const cpx (&v)[1 << LOGE], int tid, St st) {
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  constexpr int E = 1 << LOGE, R = 1 << LOGR, U = E / R, T = 1 << (LOGN - LOGE), NS = 1 << LOGNS;
#pragma unroll
  for (int u = 0; u < U; u++) {
    int j = tid + u * T;
    int base = ((j >> LOGNS) << (LOGNS + LOGR)) + (j & (NS - 1));
#pragma unroll
    for (int q = 0; q < R; q++) st(base + (q << LOGNS), v[u + U * q]);
  }
}

// Gather lane-owned positions tid + T*e
template <int LOGN, int LOGE, class Ld> CLFA_HD void pass_gather(cpx (&v)[1 << LOGE], int tid, Ld ld) {
  constexpr int E = 1 << LOGE, T = 1 << (LOGN - LOGE);
#pragma unroll
  for (int e = 0; e < E; e++) v[e] = ld(tid + T * e);
}

// The same two operations on a buffer in the padded layout lds_pad(), with the padding
// arithmetic hoisted: lds_pad(b + k*s) == lds_pad(b) + k*(s + s/16) whenever s is a multiple
// of 16, and == lds_pad(b) + k*s when b is a multiple of 16 and k*s < 16.  Every position is then
// `one lane-dependent base + compile-time constant`, which the compiler folds into the
// ds_read/ds_write offset field instead of a shift and an add per element.
template <int LOGN, int LOGE, int LOGNS>
CLFA_HD void pass_scatter_padded(const cpx (&v)[1 << LOGE], int tid, cpx *xb) {
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  constexpr int E = 1 << LOGE, R = 1 << LOGR, U = E / R, T = 1 << (LOGN - LOGE), NS = 1 << LOGNS;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int j = tid + u * T;
    const int base = ((j >> LOGNS) << (LOGNS + LOGR)) + (j & (NS - 1));
    if constexpr (NS >= 16) {
      cpx *p = xb + lds_pad(base);
#pragma unroll
      for (int q = 0; q < R; q++) p[q * (NS + NS / 16)] = v[u + U * q];
    } else if constexpr (NS == 1 && R == 16) {
      cpx *p = xb + lds_pad(base);   // base = 16 j
#pragma unroll
      for (int q = 0; q < R; q++) p[q] = v[u + U * q];
    } else {
#pragma unroll
      for (int q = 0; q < R; q++) xb[lds_pad(base + (q << LOGNS))] = v[u + U * q];
    }
  }
}
template <int LOGN, int LOGE> CLFA_HD void pass_gather_padded(cpx (&v)[1 << LOGE], int tid, const cpx *xb) {
  constexpr int E = 1 << LOGE, T = 1 << (LOGN - LOGE);
  if constexpr (T >= 16) {
    const cpx *p = xb + lds_pad(tid);
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = p[e * (T + T / 16)];
  } else {
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = xb[lds_pad(tid + T * e)];
  }
}

}  // namespace clfa
