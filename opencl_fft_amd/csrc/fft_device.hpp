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
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CLFA_HD __host__ __device__ __forceinline__
#else
#define CLFA_HD inline
#endif

namespace clfa {

// ---- which transform a workgroup of a persistent kernel starts with (it then steps by the grid size) ----------------
// Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md: blocks i and i + 8 share one).  With
// "workgroup i takes transform i" every XCD works on every eighth transform of the window of gridDim.x transforms the
// chip is in; here XCD x takes the x-th CONTIGUOUS eighth of that window (workgroup i = x + 8 c starts at
// x * (G / 8) + c), so that the workgroups behind one L2 stream through one compact address range.  Same windows,
// same step, any batch; measured on the resident n = 65536 kernel: 0.871 -> 0.831 ms per 4096 transforms
// (profiles/r04_assignment.txt; tools/res16_probe.hip PROBE_PERM sweeps the other assignments).
#ifndef CLFA_XCD_MAP
#define CLFA_XCD_MAP 1
#endif
CLFA_HD long xcd_first(unsigned i, unsigned grid) {
  if (!CLFA_XCD_MAP || (grid & 7)) return i;
  return (long)(i & 7) * (grid >> 3) + (i >> 3);
}


// A complex value is a native 2-float vector on the device (and wherever clang compiles this
// header): it lives in an aligned VGPR pair from the 8-byte load to the 8-byte store, and the
// arithmetic below is CDNA's packed fp32 (v_pk_add/mul/fma_f32: both halves per instruction, the
// same issue slot as a scalar op).  The multiplies and the +-i rotations use the instructions'
// operand swizzles (op_sel: which half of a source feeds each half of the result; neg_lo/neg_hi)
// through inline asm, because hipcc materialises such swizzles as v_mov / v_xor instead of folding
// them: a complex multiply is 2 instructions instead of 4, add/sub (also with one operand times
// +-i) 1 instead of 2.  g++ (tests/cpp/emulate_engine.cpp) sees a plain struct and scalar code.
#if defined(__clang__)
typedef float cpx __attribute__((ext_vector_type(2)));
#define CLFA_VEC 1
#else
struct alignas(8) cpx {
  float x, y;
};
#define CLFA_VEC 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define CLFA_PK 1
#else
#define CLFA_PK 0
#endif

CLFA_HD cpx mk(float x, float y) { cpx r; r.x = x; r.y = y; return r; }
#if CLFA_VEC
CLFA_HD cpx cadd(cpx a, cpx b) { return a + b; }
CLFA_HD cpx csub(cpx a, cpx b) { return a - b; }
CLFA_HD cpx cscale(cpx a, float s) { return a * s; }
#else
CLFA_HD cpx cadd(cpx a, cpx b) { return mk(a.x + b.x, a.y + b.y); }
CLFA_HD cpx csub(cpx a, cpx b) { return mk(a.x - b.x, a.y - b.y); }
CLFA_HD cpx cscale(cpx a, float s) { return mk(a.x * s, a.y * s); }
#endif
CLFA_HD cpx cconj(cpx a) { return mk(a.x, -a.y); }
// a * b, or a * conj(b).  The two instructions sit in ONE asm statement: hipcc pads every statement
// whose result the next instruction reads with an s_nop (it cannot see that this is a plain VALU
// dependency), which cost one issue slot per complex multiply.
template <bool CONJ = false> CLFA_HD cpx cmulc(cpx a, cpx b) {
#if CLFA_PK
  cpx r;
  if (!CONJ) {
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"                                                // (ax bx, ax by)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"                  // (-ay by, ay bx) + t
        : "=&v"(r) : "v"(a), "v"(b));
  } else {
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"                                   // (ax bx, -ax by)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"                                 // (ay by, ay bx) + t
        : "=&v"(r) : "v"(a), "v"(b));
  }
  return r;
#else
  return CONJ ? mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y) : mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
#endif
}
// two independent products in one statement, interleaved (mul, mul, fma, fma): no dependent
// back-to-back issue and one statement boundary for four instructions
template <bool CONJ = false> CLFA_HD void cmulc2(cpx &r0, cpx &r1, cpx a0, cpx b0, cpx a1, cpx b1) {
#if CLFA_PK
  cpx x, y;
  if (!CONJ) {
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %4, %5 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(x), "=&v"(y) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
  } else {
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %4, %5 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=&v"(x), "=&v"(y) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
  }
  r0 = x;
  r1 = y;
#else
  r0 = cmulc<CONJ>(a0, b0);
  r1 = cmulc<CONJ>(a1, b1);
#endif
}
CLFA_HD cpx cmul(cpx a, cpx b) { return cmulc<false>(a, b); }
// the same without inline asm, for loops hipcc has to unroll with a run-time trip count: HIP treats
// every asm statement as convergent, and a loop with a convergent operation is not given a remainder loop
CLFA_HD cpx cmul_plain(cpx a, cpx b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// multiply by W_4^1: -i for a forward transform, +i for an inverse one
template <bool FWD> CLFA_HD cpx rot4(cpx a) { return FWD ? mk(a.y, -a.x) : mk(-a.y, a.x); }
// x + rot4(y), x - rot4(y) in one instruction each
template <bool FWD> CLFA_HD cpx add_rot(cpx x, cpx y) {
#if CLFA_PK
  cpx r;
  if (FWD) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));  // (x.x + y.y, x.y - y.x)
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(x), "v"(y));      // (x.x - y.y, x.y + y.x)
  return r;
#else
  return cadd(x, rot4<FWD>(y));
#endif
}
template <bool FWD> CLFA_HD cpx sub_rot(cpx x, cpx y) { return add_rot<!FWD>(x, y); }
// constant twiddle (c, -s) forward / (c, +s) inverse: a*c + rot4(a)*s (one statement, see cmulc)
template <bool FWD> CLFA_HD cpx ctw(cpx a, float c, float s) {
#if CLFA_PK
  const cpx k = mk(c, s);   // compile-time constants: an SGPR pair
  cpx r;
  if (FWD)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"                                                 // (ax c, ay c)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"                   // (ay s, -ax s) + t
        : "=&v"(r) : "v"(a), "s"(k));
  else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"                   // (-ay s, ax s) + t
        : "=&v"(r) : "v"(a), "s"(k));
  return r;
#else
  return FWD ? mk(a.x * c + a.y * s, a.y * c - a.x * s) : mk(a.x * c - a.y * s, a.y * c + a.x * s);
#endif
}
// two constant twiddles, interleaved
template <bool FWD> CLFA_HD void ctw2(cpx &a0, float c0, float s0, cpx &a1, float c1, float s1) {
#if CLFA_PK
  const cpx k0 = mk(c0, s0), k1 = mk(c1, s1);
  cpx x, y;
  if (FWD)
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %4, %5 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(x), "=&v"(y) : "v"(a0), "s"(k0), "v"(a1), "s"(k1));
  else
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %4, %5 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=&v"(x), "=&v"(y) : "v"(a0), "s"(k0), "v"(a1), "s"(k1));
  a0 = x;
  a1 = y;
#else
  a0 = ctw<FWD>(a0, c0, s0);
  a1 = ctw<FWD>(a1, c1, s1);
#endif
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
  cpx s13 = cadd(a1, a3), d13 = csub(a1, a3);
  a0 = cadd(s02, s13);
  a2 = csub(s02, s13);
  a1 = add_rot<FWD>(d02, d13);
  a3 = sub_rot<FWD>(d02, d13);
}
// the same with rot4(a2) in place of a2 (the W_16^4 input of dft16)
template <bool FWD> CLFA_HD void bf4_rot2(cpx &a0, cpx &a1, cpx &a2, cpx &a3) {
  cpx s02 = add_rot<FWD>(a0, a2), d02 = sub_rot<FWD>(a0, a2);
  cpx s13 = cadd(a1, a3), d13 = csub(a1, a3);
  a0 = cadd(s02, s13);
  a2 = csub(s02, s13);
  a1 = add_rot<FWD>(d02, d13);
  a3 = sub_rot<FWD>(d02, d13);
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
  x[7] = ctw<FWD>(x[7], -kC8, kC8);  // W_8^3
  v[u] = cadd(x[0], x[1]);
  v[u + U * 4] = csub(x[0], x[1]);
  v[u + U * 1] = cadd(x[2], x[3]);
  v[u + U * 5] = csub(x[2], x[3]);
  v[u + U * 2] = add_rot<FWD>(x[4], x[5]);   // W_8^2 = rot4, folded into the butterfly
  v[u + U * 6] = sub_rot<FWD>(x[4], x[5]);
  v[u + U * 3] = cadd(x[6], x[7]);
  v[u + U * 7] = csub(x[6], x[7]);
}

// 16 = 4 x 4: t = 4a + b, q = q0 + 4*q1
template <int U, int E, bool FWD> CLFA_HD void dft16(cpx (&v)[E], int u) {
  cpx x[16];
#pragma unroll
  for (int t = 0; t < 16; t++) x[t] = v[u + U * t];
#pragma unroll
  for (int b = 0; b < 4; b++) bf4<FWD>(x[b], x[4 + b], x[8 + b], x[12 + b]);  // -> x[4*q0 + b]
  // W_16^(b*q0), two per statement (W_16^4 = rot4 is folded into bf4_rot2 below)
  ctw2<FWD>(x[4 + 1], kC16, kS16, x[4 + 2], kC8, kC8);        // 1, 2
  ctw2<FWD>(x[4 + 3], kS16, kC16, x[8 + 1], kC8, kC8);        // 3, 2
  ctw2<FWD>(x[8 + 3], -kC8, kC8, x[12 + 1], kS16, kC16);      // 6, 3
  ctw2<FWD>(x[12 + 2], -kC8, kC8, x[12 + 3], -kC16, -kS16);   // 6, 9
#pragma unroll
  for (int q0 = 0; q0 < 4; q0++) {
    if (q0 == 2) bf4_rot2<FWD>(x[8], x[9], x[10], x[11]);
    else bf4<FWD>(x[4 * q0], x[4 * q0 + 1], x[4 * q0 + 2], x[4 * q0 + 3]);  // -> q1
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
// radix (log2) of the pass that starts with sub-transform length 2^LOGNS: 2^LOGE, except the last
// ("remainder") pass, which starts at 2^pass_last_logns and has radix 2^pass_rem_logr
constexpr int pass_logr(int LOGN, int LOGE, int LOGNS) { return cmin(LOGE, LOGN - LOGNS); }
constexpr int pass_rem_logr(int LOGN, int LOGE) { return (LOGN - 1) % LOGE + 1; }
constexpr int pass_last_logns(int LOGN, int LOGE) { return LOGN - pass_rem_logr(LOGN, LOGE); }

// padded LDS position: one pad element per 16 keeps the stride-16 scatter of
// the first exchange and the contiguous gathers conflict-free (ds_*_b64)
CLFA_HD int lds_pad(int p) { return p + (p >> 4); }
constexpr int lds_padded_size(int n) { return n + (n >> 4) + 1; }

// Twiddle multiplies: v * W_n^k for a forward transform, v * conj(W_n^k) for an inverse one (the
// conjugation rides on the multiply's operand modifiers).
// Half table: tab holds W_n^k for k in [0, n/2) (forward sign); W_n^(k+n/2) = -W_n^k.
template <int LOGN, bool FWD, class Tab> CLFA_HD cpx cmul_tw(cpx v, const Tab &tab, int k) {
  if constexpr (LOGN == 0) return v;
  constexpr int half = (1 << LOGN) >> 1;
  cpx r = cmulc<!FWD>(v, tab[k & (half - 1)]);
  if (k & half) r = mk(-r.x, -r.y);
  return r;
}
// Two-level twiddle table: W_n^k = hi[k >> LOGLO] * lo[k & (2^LOGLO - 1)], hi[j] = W_n^(j * 2^LOGLO),
// lo[j] = W_n^j, both rounded from double.  n/2^LOGLO + 2^LOGLO entries instead of n/2: it lets
// the 8192-point kernel keep two workgroups per CU.  One extra complex multiply per lookup.
template <int LOGLO> struct TwoLevelTab {
  const cpx *hi;
  const cpx *lo;
};
// full table W_n^k, k < n (forward sign): no half-table sign logic (3-4 VALU per lookup); used where LDS allows
struct FullTab {
  const cpx *p;
};
template <int LOGN, bool FWD> CLFA_HD cpx cmul_tw(cpx v, const FullTab &tab, int k) { return cmulc<!FWD>(v, tab.p[k]); }
template <int LOGN, bool FWD, int LOGLO> CLFA_HD cpx cmul_tw(cpx v, const TwoLevelTab<LOGLO> &tab, int k) {
  return cmulc<!FWD>(v, cmul(tab.hi[k >> LOGLO], tab.lo[k & ((1 << LOGLO) - 1)]));
}

// ---- n = 8192: twiddles addressed by lane constants ------------------------------------------------
// The two-level table costs ~6 integer instructions and two LDS reads per twiddle (a third of the
// 8192-point kernel's instruction stream was index arithmetic).  Every pass's twiddles are powers of
// ONE value per lane, so for the pass structure 16 x 16 x 16 x 2 (16 points per lane, 512 lanes):
//   pass 2 (NS = 16)   W_256^(jm t), jm = tid & 15: row jm of a 16 x 16 table — one base address per
//                      lane, the 15 reads use immediate offsets;
//   pass 3 (NS = 256)  W_4096^(jm t), jm = tid & 255: the powers t = 1..15 of W_4096^jm as products
//                      of four exact table values W_4096^(jm), ^(2 jm), ^(4 jm), ^(8 jm) (at most three
//                      multiplications deep);
//   pass 4 (radix 2)   W_8192^(tid + 512 u) = W_8192^tid * W_16^u: a lane constant times compile-time
//                      constants.
// No integer arithmetic at all; the LDS table has 1280 entries.
struct LaneTab13 {
  const cpx *row16;   // LDS: W_256^((tid & 15) t), t = 0..15
  const cpx *s256;    // LDS: [256 k] = W_4096^(2^k (tid & 255)), k = 0..3 (exponent taken mod 4096)
  cpx w;              // W_8192^tid (forward sign, like every table)
};
constexpr float kC16t[8] = {1.0f, 0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f, 0.0f,
                            -0.38268343236508977173f, -0.70710678118654752440f, -0.92387953251128675613f};
constexpr float kS16t[8] = {0.0f, 0.38268343236508977173f, 0.70710678118654752440f, 0.92387953251128675613f, 1.0f,
                            0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f};
// W_8192^(tid + 512 u), forward sign
template <int U8> CLFA_HD cpx lane_w13(const LaneTab13 &tab) {
  if constexpr (U8 == 0) return tab.w;
  else return ctw<true>(tab.w, kC16t[U8], kS16t[U8]);
}
// the twiddles of the pass that starts at 2^LOGNS on elements v[u + U t], t >= 1 (DIT: before the
// butterflies; transposed chain: after them) — forward: times W, inverse: times conj(W)
template <int LOGNS, bool FWD> CLFA_HD void lane_tw13(cpx (&v)[16], const LaneTab13 &tab) {
  if constexpr (LOGNS == 4) {
    v[1] = cmulc<!FWD>(v[1], tab.row16[1]);
#pragma unroll
    for (int t = 2; t < 16; t += 2) cmulc2<!FWD>(v[t], v[t + 1], v[t], tab.row16[t], v[t + 1], tab.row16[t + 1]);
  } else if constexpr (LOGNS == 8) {
    const cpx s1 = tab.s256[0], s2 = tab.s256[256], s4 = tab.s256[512], s8 = tab.s256[768];
    cpx p3, p5, p6, p7;
    cmulc2(p3, p5, s1, s2, s1, s4);
    cmulc2(p6, p7, s2, s4, p3, s4);
    cmulc2<!FWD>(v[1], v[2], v[1], s1, v[2], s2);
    cmulc2<!FWD>(v[3], v[4], v[3], p3, v[4], s4);
    cmulc2<!FWD>(v[5], v[6], v[5], p5, v[6], p6);
    cmulc2<!FWD>(v[7], v[8], v[7], p7, v[8], s8);
    // t = 8 + r: (v * P_r) * s8 — as many multiplications as forming P_(8+r) first, fewer registers
    cmulc2<!FWD>(v[9], v[10], v[9], s1, v[10], s2);
    cmulc2<!FWD>(v[11], v[12], v[11], p3, v[12], s4);
    cmulc2<!FWD>(v[13], v[14], v[13], p5, v[14], p6);
    v[15] = cmulc<!FWD>(v[15], p7);
    cmulc2<!FWD>(v[9], v[10], v[9], s8, v[10], s8);
    cmulc2<!FWD>(v[11], v[12], v[11], s8, v[12], s8);
    cmulc2<!FWD>(v[13], v[14], v[13], s8, v[14], s8);
    v[15] = cmulc<!FWD>(v[15], s8);
  } else {
    static_assert(LOGNS == 12, "passes of the 8192-point transform");
    cmulc2<!FWD>(v[8], v[9], v[8], lane_w13<0>(tab), v[9], lane_w13<1>(tab));
    cmulc2<!FWD>(v[10], v[11], v[10], lane_w13<2>(tab), v[11], lane_w13<3>(tab));
    cmulc2<!FWD>(v[12], v[13], v[12], lane_w13<4>(tab), v[13], lane_w13<5>(tab));
    cmulc2<!FWD>(v[14], v[15], v[14], lane_w13<6>(tab), v[15], lane_w13<7>(tab));
  }
}
// paired radix-2 pass (pass_last_paired / pass_first_paired, NB = 4096): element u + 8 belongs to
// butterfly j = tid + 512 u (twiddle W^j), element u + 4 + 8 to its partner NB - j (twiddle
// W^(4096 - j) = -conj(W^j)); lane 0's u = 0 pairs butterflies 0 and NB/2 = 2048 (W^2048 = -i)
template <bool FWD> CLFA_HD void lane_tw13_paired(cpx (&v)[16], int tid, const LaneTab13 &tab) {
  const cpx w0 = lane_w13<0>(tab), w1 = lane_w13<1>(tab), w2 = lane_w13<2>(tab), w3 = lane_w13<3>(tab);
  cpx q0 = mk(-w0.x, w0.y);
  const cpx q1 = mk(-w1.x, w1.y), q2 = mk(-w2.x, w2.y), q3 = mk(-w3.x, w3.y);
  if (tid == 0) q0 = mk(0.f, -1.f);
  cmulc2<!FWD>(v[8], v[9], v[8], w0, v[9], w1);
  cmulc2<!FWD>(v[10], v[11], v[10], w2, v[11], w3);
  cmulc2<!FWD>(v[12], v[13], v[12], q0, v[13], q1);
  cmulc2<!FWD>(v[14], v[15], v[14], q2, v[15], q3);
}

// ---- n = 16384: the same idea for the pass structure 16 x 16 x 16 x 4 (16 points per lane, 1024 lanes) ----
//   passes 2 and 3    as for n = 8192 (row16, s256 addressed by tid & 15, tid & 255);
//   pass 4 (radix 4)  butterfly j = tid + 1024 u (u < 4), element t (1..3): W_16384^(j t) =
//                     W_16384^(t tid) * W_16^(u t): three lane constants (exact table values) times
//                     compile-time constants.
struct LaneTab14 {
  const cpx *row16;   // LDS: W_256^((tid & 15) t), t = 0..15
  const cpx *s256;    // LDS: [256 k] = W_4096^(2^k (tid & 255)), k = 0..3
  cpx w1, w2, w3;     // W_16384^(tid), ^(2 tid), ^(3 tid) (forward sign)
};
#ifndef CLFA_ROW16_STRIDE
#define CLFA_ROW16_STRIDE 18
#endif
constexpr int kRow16StrideDev = CLFA_ROW16_STRIDE;
// Half table plus the 16 x 16 table W_256^(j t) for the pass that starts at 16 points: that pass's twiddles are
// W_n^(j t n / 256) = W_256^(j t) whatever n is, j = tid & 15 — read from the half table they sit t n / 128 dwords apart
// between neighbouring lanes (n = 4096: 32 t, two bank groups for sixteen addresses, an 8-way conflict on every read);
// row j of the small table, rows kRow16StrideDev (18) entries apart, is conflict-free and needs no index arithmetic.  The
// entries are the half table's own values (lds_fill_row16), so results do not change by a bit.  Used by the packed real
// kernels of size 8192 (n = 4096: -5 %); for the complex kernels and the smaller sizes it measured between -1.3 and +2.2 %
// (profiles/ab_small_twiddle_tables_r05.txt) and is not used.
struct HalfRowTab {
  const cpx *half;    // LDS: W_n^k, k < n / 2
  const cpx *row16;   // LDS: row (tid & 15) of W_256^(j t)
};
template <class Tab> struct has_row16 : std::false_type {};
template <> struct has_row16<HalfRowTab> : std::true_type {};
template <int LOGN, bool FWD> CLFA_HD cpx cmul_tw(cpx v, const HalfRowTab &tab, int k) { return cmul_tw<LOGN, FWD>(v, tab.half, k); }
// fills the 16 x 16 table from the half table W_n^k (k < n / 2, forward sign) of an n-point transform, n >= 256
template <int LOGN> CLFA_HD void lds_fill_row16(cpx *row, const cpx *half_g, int tid, int nthreads) {
  static_assert(LOGN >= 8, "the pass that starts at 16 points has radix 16 from n = 256 on");
  constexpr int N = 1 << LOGN;
  for (int i = tid; i < 256; i += nthreads) {
    const int e = ((i >> 4) * (i & 15) * (N / 256)) & (N - 1);
    const cpx w = half_g[e & (N / 2 - 1)];
    row[(i >> 4) * kRow16StrideDev + (i & 15)] = e & (N / 2) ? mk(-w.x, -w.y) : w;
  }
}
template <class Tab> struct is_lane_tab : std::false_type {};
template <> struct is_lane_tab<LaneTab13> : std::true_type {};
template <> struct is_lane_tab<LaneTab14> : std::true_type {};
// cos / sin(2 pi k / 16), k = 0..9
constexpr float kC16u[10] = {1.0f, 0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f, 0.0f,
                             -0.38268343236508977173f, -0.70710678118654752440f, -0.92387953251128675613f, -1.0f,
                             -0.92387953251128675613f};
constexpr float kS16u[10] = {0.0f, 0.38268343236508977173f, 0.70710678118654752440f, 0.92387953251128675613f, 1.0f,
                             0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f, 0.0f,
                             -0.38268343236508977173f};
template <int LOGNS, bool FWD> CLFA_HD void lane_tw14(cpx (&v)[16], const LaneTab14 &tab) {
  if constexpr (LOGNS == 4 || LOGNS == 8) {
    lane_tw13<LOGNS, FWD>(v, LaneTab13{tab.row16, tab.s256, tab.w1});
  } else {
    static_assert(LOGNS == 12, "passes of the 16384-point transform");
    // element u + 4 t: (v * W^(t tid)) * W_16^(u t)
    cmulc2<!FWD>(v[4], v[5], v[4], tab.w1, v[5], tab.w1);
    cmulc2<!FWD>(v[6], v[7], v[6], tab.w1, v[7], tab.w1);
    cmulc2<!FWD>(v[8], v[9], v[8], tab.w2, v[9], tab.w2);
    cmulc2<!FWD>(v[10], v[11], v[10], tab.w2, v[11], tab.w2);
    cmulc2<!FWD>(v[12], v[13], v[12], tab.w3, v[13], tab.w3);
    cmulc2<!FWD>(v[14], v[15], v[14], tab.w3, v[15], tab.w3);
    ctw2<FWD>(v[5], kC16u[1], kS16u[1], v[6], kC16u[2], kS16u[2]);     // u t = 1, 2
    ctw2<FWD>(v[7], kC16u[3], kS16u[3], v[9], kC16u[2], kS16u[2]);     // 3, 2
    ctw2<FWD>(v[10], kC16u[4], kS16u[4], v[11], kC16u[6], kS16u[6]);   // 4, 6
    ctw2<FWD>(v[13], kC16u[3], kS16u[3], v[14], kC16u[6], kS16u[6]);   // 3, 6
    v[15] = ctw<FWD>(v[15], kC16u[9], kS16u[9]);                       // 9
  }
}
// paired radix-4 pass (pass_last_paired / pass_first_paired, NB = 4096): slot u + 4 t (u = 0, 1) belongs to
// butterfly j = tid + 1024 u (twiddle W^(j t)), slot u + 2 + 4 t to its partner NB - j: W^((4096 - j) t) =
// (-i)^t conj(W^(j t)); lane 0's u = 0 pairs butterflies 0 and NB / 2 = 2048 (W^(2048 t) = W_8^t)
template <bool FWD> CLFA_HD void lane_tw14_paired(cpx (&v)[16], int tid, const LaneTab14 &tab) {
  const cpx a1 = tab.w1, a2 = tab.w2, a3 = tab.w3;                         // u = 0: W^(t tid)
  const cpx b1 = ctw<true>(tab.w1, kC16u[1], kS16u[1]), b2 = ctw<true>(tab.w2, kC16u[2], kS16u[2]),
            b3 = ctw<true>(tab.w3, kC16u[3], kS16u[3]);                      // u = 1: times W_16^t
  cpx p1 = mk(-a1.y, -a1.x), p2 = mk(-a2.x, a2.y), p3 = mk(a3.y, a3.x);   // partners of u = 0
  const cpx q1 = mk(-b1.y, -b1.x), q2 = mk(-b2.x, b2.y), q3 = mk(b3.y, b3.x);
  if (tid == 0) {
    p1 = mk(kC8, -kC8);
    p2 = mk(0.f, -1.f);
    p3 = mk(-kC8, -kC8);
  }
  cmulc2<!FWD>(v[4], v[5], v[4], a1, v[5], b1);
  cmulc2<!FWD>(v[6], v[7], v[6], p1, v[7], q1);
  cmulc2<!FWD>(v[8], v[9], v[8], a2, v[9], b2);
  cmulc2<!FWD>(v[10], v[11], v[10], p2, v[11], q2);
  cmulc2<!FWD>(v[12], v[13], v[12], a3, v[13], b3);
  cmulc2<!FWD>(v[14], v[15], v[14], p3, v[15], q3);
}
// the lane tables' pass twiddles, by table type
template <int LOGNS, bool FWD> CLFA_HD void lane_tw(cpx (&v)[16], const LaneTab13 &tab) { lane_tw13<LOGNS, FWD>(v, tab); }
template <int LOGNS, bool FWD> CLFA_HD void lane_tw(cpx (&v)[16], const LaneTab14 &tab) { lane_tw14<LOGNS, FWD>(v, tab); }
template <bool FWD> CLFA_HD void lane_tw_paired(cpx (&v)[16], int tid, const LaneTab13 &tab) { lane_tw13_paired<FWD>(v, tid, tab); }
template <bool FWD> CLFA_HD void lane_tw_paired(cpx (&v)[16], int tid, const LaneTab14 &tab) { lane_tw14_paired<FWD>(v, tid, tab); }

// lane permutation of the middle passes of the 512- / 1024-lane chains (fft_wg.hpp, wg_passes_sigma): bits 4 and 8 swapped
CLFA_HD int lane_sigma(int t) { return (t & ~0x110) | ((t & 0x10) << 4) | ((t & 0x100) >> 4); }

// One pass on the registers of lane `tid`: input twiddles then U butterflies.
template <int LOGN, int LOGE, int LOGNS, bool FWD, class Tab>
CLFA_HD void pass_compute(cpx (&v)[1 << LOGE], int tid, const Tab &tab) {
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  constexpr int E = 1 << LOGE, R = 1 << LOGR, U = E / R, T = 1 << (LOGN - LOGE), NS = 1 << LOGNS;
  if constexpr (LOGNS > 0 && is_lane_tab<Tab>::value) {
    static_assert((LOGN == 13 || LOGN == 14) && LOGE == 4, "lane tables exist for 8192 and 16384 points");
    lane_tw<LOGNS, FWD>(v, tab);
  } else if constexpr (LOGNS == 4 && LOGR == 4 && LOGE == 4 && has_row16<Tab>::value) {
    lane_tw13<4, FWD>(v, LaneTab13{tab.row16, nullptr, mk(1.f, 0.f)});
  } else if constexpr (LOGNS > 0) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      int jm = (tid + u * T) & (NS - 1);
#pragma unroll
      for (int t = 1; t < R; t++) {
        int k = (jm * t) << (LOGN - LOGNS - LOGR);
        v[u + U * t] = cmul_tw<LOGN, FWD>(v[u + U * t], tab, k);
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

// ---- the transposed chain (decimation in frequency) ---------------------------------------------
// The DFT matrix is symmetric, so the transposes of the passes above, run in the opposite order
// (remainder pass FIRST), compute the same transform.  Transposed pass (NS, R): butterfly
// j = tid + u*T gathers its inputs from where the pass above scatters to, runs the R-point DFT and
// multiplies OUTPUT q by W_(NS*R)^((j mod NS) * q); output q belongs at position j + (n/R)*q, so a
// lane owns positions tid + T*e on EXIT from every pass and the final stores are coalesced.  Same
// butterflies, same twiddle count, same LDS patterns (reads and writes swapped).  It exists for the
// inverse packed-real transform, whose pair map sits on the INPUT side and needs the small-radix
// pass there (pass_first_paired below).
template <int LOGN, int LOGE, int LOGNS, bool FWD, class Tab>
CLFA_HD void dif_compute(cpx (&v)[1 << LOGE], int tid, const Tab &tab) {
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  constexpr int E = 1 << LOGE, R = 1 << LOGR, U = E / R, T = 1 << (LOGN - LOGE), NS = 1 << LOGNS;
#pragma unroll
  for (int u = 0; u < U; u++) dft<LOGR, U, E, FWD>(v, u);
  if constexpr (LOGNS > 0 && is_lane_tab<Tab>::value) {
    lane_tw<LOGNS, FWD>(v, tab);
  } else if constexpr (LOGNS == 4 && LOGR == 4 && LOGE == 4 && has_row16<Tab>::value) {
    lane_tw13<4, FWD>(v, LaneTab13{tab.row16, nullptr, mk(1.f, 0.f)});
  } else if constexpr (LOGNS > 0) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      int jm = (tid + u * T) & (NS - 1);
#pragma unroll
      for (int q = 1; q < R; q++) {
        int k = (jm * q) << (LOGN - LOGNS - LOGR);
        v[u + U * q] = cmul_tw<LOGN, FWD>(v[u + U * q], tab, k);
      }
    }
  }
}
// inputs of the transposed pass (NS, R) from the padded buffer (mirror of pass_scatter_padded)
template <int LOGN, int LOGE, int LOGNS>
CLFA_HD void dif_gather_padded(cpx (&v)[1 << LOGE], int tid, const cpx *xb) {
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  constexpr int E = 1 << LOGE, R = 1 << LOGR, U = E / R, T = 1 << (LOGN - LOGE), NS = 1 << LOGNS;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int j = tid + u * T;
    const int base = ((j >> LOGNS) << (LOGNS + LOGR)) + (j & (NS - 1));
    if constexpr (NS >= 16) {
      const cpx *p = xb + lds_pad(base);
#pragma unroll
      for (int t = 0; t < R; t++) v[u + U * t] = p[t * (NS + NS / 16)];
    } else if constexpr (NS == 1 && R == 16) {
      const cpx *p = xb + lds_pad(base);   // base = 16 j
#pragma unroll
      for (int t = 0; t < R; t++) v[u + U * t] = p[t];
    } else {
#pragma unroll
      for (int t = 0; t < R; t++) v[u + U * t] = xb[lds_pad(base + (t << LOGNS))];
    }
  }
}
// outputs to the lane-owned positions tid + T*e (mirror of pass_gather_padded)
template <int LOGN, int LOGE> CLFA_HD void dif_scatter_padded(const cpx (&v)[1 << LOGE], int tid, cpx *xb) {
  constexpr int E = 1 << LOGE, T = 1 << (LOGN - LOGE);
  if constexpr (T >= 16) {
    cpx *p = xb + lds_pad(tid);
#pragma unroll
    for (int e = 0; e < E; e++) p[e * (T + T / 16)] = v[e];
  } else {
#pragma unroll
    for (int e = 0; e < E; e++) xb[lds_pad(tid + T * e)] = v[e];
  }
}

// ---- packed real transforms: the reference's pair maps, and passes that pair in registers ------

// reference conv kernel, cl_fft.cpp:178-191 (pair i, M-i; bin M/2 not visited)
CLFA_HD void r2c_pair(cpx ci, cpx cjraw, cpx w, cpx &oi, cpx &oj) {
  cpx cj = cconj(cjraw);
  cpx e = cscale(cadd(ci, cj), .5f);
  cpx d = csub(cj, ci);
  cpx o = cscale(mk(-d.y, d.x), .5f);
  cpx p = cmul(w, o);
  oi = cadd(e, p);
  oj = cconj(csub(e, p));
}
// reference iconv kernel, cl_fft.cpp:192-205
CLFA_HD void c2r_pair(cpx ci, cpx cjraw, cpx w, cpx &oi, cpx &oj) {
  cpx cj = cconj(cjraw);
  cpx e = cscale(cadd(ci, cj), .5f);
  cpx d = csub(ci, cj);
  cpx o = cscale(mk(-d.y, d.x), .5f);
  cpx p = cmul(w, o);
  oi = cadd(e, p);
  oj = cconj(csub(e, p));
}

// The same maps with their factors of 1/2 folded away (powers of two: bit-identical results).  r2c: the caller has
// scaled both inputs by 1/2 (together with its 1/N); c2r: the pair twiddle comes halved (wh = w / 2).
CLFA_HD void r2c_pair_prescaled(cpx ci, cpx cjraw, cpx w, cpx &oi, cpx &oj) {
  cpx cj = cconj(cjraw);
  cpx e = cadd(ci, cj);
  cpx d = csub(cj, ci);
  cpx p = cmul(w, mk(-d.y, d.x));
  oi = cadd(e, p);
  oj = cconj(csub(e, p));
}
CLFA_HD void c2r_pair_halfw(cpx ci, cpx cjraw, cpx wh, cpx &oi, cpx &oj) {
  cpx cj = cconj(cjraw);
  cpx e = cscale(cadd(ci, cj), .5f);
  cpx d = csub(ci, cj);
  cpx p = cmul(wh, mk(-d.y, d.x));
  oi = cadd(e, p);
  oj = cconj(csub(e, p));
}

// The pair maps combine bins i and M-i of an M-point complex transform.  In the remainder pass
// (radix R < 2^LOGE, U = E/R >= 2 butterflies per lane, NB = M/R butterflies in all) butterfly j
// touches positions j + NB*t, and M - (j + NB*t) = (NB - j) + NB*(R-1-t): a lane that takes
// butterflies j AND NB - j holds both halves of R pairs in its own registers, so the LDS round
// trip that the pair map otherwise needs (write every bin, barrier, read the partners) disappears.
// Lane `tid` takes j = tid + u*T (u < U/2, so j < NB/2) in register slots u + U*t and NB - j in
// slots u + U/2 + U*t.  Butterflies 0 and NB/2 pair within themselves: lane 0 takes both as its
// u = 0 pair and permutes its registers so that the same straight-line pair code applies.
// Pair (u, q) is handed out as (k = u*R + q, i = the reference's loop index, partner M - i, or
// M/2 for i = 0: the packed DC/Nyquist bin and the bin the reference never visits).
constexpr bool pair_ok(int LOGN, int LOGE) { return LOGN > LOGE && pass_rem_logr(LOGN, LOGE) < LOGE; }

template <int LOGN, int LOGE> CLFA_HD int pair_index(int tid, int u, int q) {
  constexpr int LOGR = pass_rem_logr(LOGN, LOGE), R = 1 << LOGR, T = 1 << (LOGN - LOGE), NB = 1 << (LOGN - LOGR);
  const int j = tid + u * T;
  if (j == 0) return q < R / 2 ? NB * q : NB / 2 + NB * (q - R / 2);
  return q < R / 2 ? j + NB * q : NB * (R - q) - j;
}

// lane 0, u = 0: slots as the straight-line pair code expects them <- butterflies 0 (m) and NB/2 (p)
template <int R> CLFA_HD void pair_perm_lane0(bool lane0, cpx (&m)[R], cpx (&p)[R]) {
  cpx nm[R], np[R];
  nm[0] = m[0];
  np[R - 1] = m[R / 2];
#pragma unroll
  for (int q = 1; q < R / 2; q++) {
    nm[q] = m[q];
    np[R - 1 - q] = m[R - q];
  }
#pragma unroll
  for (int q = 0; q < R / 2; q++) {
    np[R / 2 - 1 - q] = p[q];
    nm[R / 2 + q] = p[R - 1 - q];
  }
#pragma unroll
  for (int q = 0; q < R; q++) {
    m[q] = lane0 ? nm[q] : m[q];
    p[q] = lane0 ? np[q] : p[q];
  }
}
// and back: butterflies 0 (m) and NB/2 (p) <- slots as the pair code fills them
template <int R> CLFA_HD void pair_unperm_lane0(bool lane0, cpx (&m)[R], cpx (&p)[R]) {
  cpx am[R], ap[R];
  am[0] = m[0];
  am[R / 2] = p[R - 1];
#pragma unroll
  for (int q = 1; q < R / 2; q++) {
    am[q] = m[q];
    am[R - q] = p[R - 1 - q];
  }
#pragma unroll
  for (int q = 0; q < R / 2; q++) {
    ap[q] = p[R / 2 - 1 - q];
    ap[R - 1 - q] = m[R / 2 + q];
  }
#pragma unroll
  for (int q = 0; q < R; q++) {
    m[q] = lane0 ? am[q] : m[q];
    p[q] = lane0 ? ap[q] : p[q];
  }
}

// Forward: the LAST (remainder) pass of an M-point transform, butterflies paired as above; gathers
// from the padded exchange buffer.  Afterwards slot u + U*q holds Z[j + NB*q] and slot
// u + U/2 + U*q holds Z[jp + NB*q], jp = NB - j (NB/2 for j = 0).
struct NoMid {
  CLFA_HD void operator()() const {}
};
// mid(): called between the gather from the exchange buffer and the arithmetic (wg_passes_pair: the other
// transform's scatter goes there, so that its LDS transfer runs under this pass's butterflies)
template <int LOGN, int LOGE, bool FWD, class Tab, class Mid = NoMid>
CLFA_HD void pass_last_paired(cpx (&v)[1 << LOGE], int tid, const Tab &tab, const cpx *xb, const Mid &mid = Mid()) {
  static_assert(pair_ok(LOGN, LOGE), "needs a remainder pass with two butterflies per lane");
  constexpr int LOGR = pass_rem_logr(LOGN, LOGE), R = 1 << LOGR, E = 1 << LOGE, U = E / R;
  constexpr int T = 1 << (LOGN - LOGE), NB = 1 << (LOGN - LOGR);
#pragma unroll
  for (int u = 0; u < U / 2; u++) {
    const int j = tid + u * T;
    const int jp = j == 0 ? NB / 2 : NB - j;
    if constexpr (NB >= 16) {   // lane base + constant (see pass_scatter_padded)
      const cpx *pm = xb + lds_pad(j), *pp = xb + lds_pad(jp);
#pragma unroll
      for (int t = 0; t < R; t++) {
        v[u + U * t] = pm[t * (NB + NB / 16)];
        v[u + U / 2 + U * t] = pp[t * (NB + NB / 16)];
      }
    } else {
#pragma unroll
      for (int t = 0; t < R; t++) {
        v[u + U * t] = xb[lds_pad(j + NB * t)];
        v[u + U / 2 + U * t] = xb[lds_pad(jp + NB * t)];
      }
    }
    if constexpr (!is_lane_tab<Tab>::value) {   // (table twiddles go with the gather; mid() then follows both)
#pragma unroll
      for (int t = 1; t < R; t++) {
        v[u + U * t] = cmul_tw<LOGN, FWD>(v[u + U * t], tab, j * t);
        v[u + U / 2 + U * t] = cmul_tw<LOGN, FWD>(v[u + U / 2 + U * t], tab, jp * t);
      }
    }
  }
  mid();
  if constexpr (is_lane_tab<Tab>::value) lane_tw_paired<FWD>(v, tid, tab);
#pragma unroll
  for (int u = 0; u < U; u++) dft<LOGR, U, E, FWD>(v, u);
}

// ... and its pairs: f(k, i, Z[i], Z[M - i]) (for i = 0: Z[0], Z[M/2]), k = u*R + q
template <int LOGN, int LOGE, class F> CLFA_HD void pairs_visit(const cpx (&v)[1 << LOGE], int tid, F f) {
  constexpr int LOGR = pass_rem_logr(LOGN, LOGE), R = 1 << LOGR, E = 1 << LOGE, U = E / R;
#pragma unroll
  for (int u = 0; u < U / 2; u++) {
    cpx m[R], p[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
      m[q] = v[u + U * q];
      p[q] = v[u + U / 2 + U * q];
    }
    if (u == 0) pair_perm_lane0<R>(tid == 0, m, p);
#pragma unroll
    for (int q = 0; q < R; q++) {
      const int i = pair_index<LOGN, LOGE>(tid, u, q);
      if (q < R / 2) f(u * R + q, i, m[q], p[R - 1 - q]);
      else f(u * R + q, i, p[R - 1 - q], m[q]);
    }
  }
}

// Inverse: the FIRST pass of the transposed chain (the remainder pass, NS = NB).  oi[k] / oj[k] are
// the pair map's results for positions i / M - i of pair k (pair_index); runs the radix-R butterflies
// with their output twiddles and scatters to positions j + NB*q, jp + NB*q of the padded buffer.
// Continue with dif_gather_padded / dif_compute at LOGNS = pass_last_logns - LOGE.
template <int LOGN, int LOGE, bool FWD, class Tab>
CLFA_HD void pass_first_paired(cpx (&v)[1 << LOGE], int tid, const cpx (&oi)[(1 << LOGE) / 2],
                               const cpx (&oj)[(1 << LOGE) / 2], const Tab &tab) {
  static_assert(pair_ok(LOGN, LOGE), "needs a remainder pass with two butterflies per lane");
  constexpr int LOGR = pass_rem_logr(LOGN, LOGE), R = 1 << LOGR, E = 1 << LOGE, U = E / R;
  constexpr int T = 1 << (LOGN - LOGE), NB = 1 << (LOGN - LOGR);
#pragma unroll
  for (int u = 0; u < U / 2; u++) {
    cpx m[R], p[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
      if (q < R / 2) {
        m[q] = oi[u * R + q];
        p[R - 1 - q] = oj[u * R + q];
      } else {
        p[R - 1 - q] = oi[u * R + q];
        m[q] = oj[u * R + q];
      }
    }
    if (u == 0) pair_unperm_lane0<R>(tid == 0, m, p);
#pragma unroll
    for (int q = 0; q < R; q++) {
      v[u + U * q] = m[q];
      v[u + U / 2 + U * q] = p[q];
    }
  }
#pragma unroll
  for (int u = 0; u < U; u++) dft<LOGR, U, E, FWD>(v, u);
  if constexpr (is_lane_tab<Tab>::value) {
    lane_tw_paired<FWD>(v, tid, tab);
  } else {
#pragma unroll
    for (int u = 0; u < U / 2; u++) {
      const int j = tid + u * T;
      const int jp = j == 0 ? NB / 2 : NB - j;
#pragma unroll
      for (int q = 1; q < R; q++) {
        v[u + U * q] = cmul_tw<LOGN, FWD>(v[u + U * q], tab, j * q);
        v[u + U / 2 + U * q] = cmul_tw<LOGN, FWD>(v[u + U / 2 + U * q], tab, jp * q);
      }
    }
  }
}
template <int LOGN, int LOGE> CLFA_HD void pass_first_paired_scatter(const cpx (&v)[1 << LOGE], int tid, cpx *xb) {
  constexpr int LOGR = pass_rem_logr(LOGN, LOGE), R = 1 << LOGR, E = 1 << LOGE, U = E / R;
  constexpr int T = 1 << (LOGN - LOGE), NB = 1 << (LOGN - LOGR);
#pragma unroll
  for (int u = 0; u < U / 2; u++) {
    const int j = tid + u * T;
    const int jp = j == 0 ? NB / 2 : NB - j;
    if constexpr (NB >= 16) {
      cpx *pm = xb + lds_pad(j), *pp = xb + lds_pad(jp);
#pragma unroll
      for (int q = 0; q < R; q++) {
        pm[q * (NB + NB / 16)] = v[u + U * q];
        pp[q * (NB + NB / 16)] = v[u + U / 2 + U * q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < R; q++) {
        xb[lds_pad(j + NB * q)] = v[u + U * q];
        xb[lds_pad(jp + NB * q)] = v[u + U / 2 + U * q];
      }
    }
  }
}


// ---- n = 32768 packed real (size 65536) on two 16384-point sub-transforms --------------------------
// Z (n = 2 M points, M = 16384) splits by decimation in time into A = FFT_M(z[2j]) and B = FFT_M(z[2j+1]):
// Z[i] = A[i] + W_2M^i B[i], Z[i + M] = A[i] - W_2M^i B[i].  The paired remainder pass leaves A[i], A[M - i],
// B[i], B[M - i] of eight pairs in one lane, and those four values give Z[i], Z[M + i], Z[M - i], Z[2M - i]:
// exactly the two pairs (i, 2M - i) and (M - i, M + i) of the reference's pair maps (cl_fft.cpp:178-205).
// So the whole packed real transform of size 65536 runs through the 16384-point LDS machinery twice, with
// the radix-2 step and the pair map in registers; the inverse is the transposed network (split, then the
// transposed chains).  All twiddles W_P^i (P = 2M for the radix-2 step, 4M for the pair maps) of the lane's
// pair (u, q), i = pair_index<14, 4>(lane, u, q), derive from ONE lane constant c0 = W_P^lane:
//   j = lane + 1024 u:  q = 0 -> i = j, 1 -> j + 4096, 2 -> 8192 - j, 3 -> 4096 - j
//   (lane 0, u = 0: i = 0, 4096, 2048, 6144).  S = 0: P = 32768, S = 1: P = 65536.  Sign: c0's (the plan's).
template <bool FWD, int S> CLFA_HD cpx pair_tw14(cpx c0, int u, int q, int lane) {
  static_assert(S == 0 || S == 1, "");
  // W_P^1024 (u = 1), W_P^4096 (q = 1, 3), W_P^2048 / W_P^6144 (lane 0)
  constexpr float cu = S == 0 ? 0.98078528040323044913f : 0.99518472667219688624f;   // cos(2 pi / 32), cos(2 pi / 64)
  constexpr float su = S == 0 ? 0.19509032201612826785f : 0.09801714032956060199f;
  constexpr float cq = S == 0 ? kC8 : kC16, sq = S == 0 ? kC8 : kS16;                 // W_8, W_16
  constexpr float cuq = S == 0 ? 0.55557023301960222474f : 0.88192126434835502971f;  // W_32^5, W_64^5
  constexpr float suq = S == 0 ? 0.83146961230254523708f : 0.47139673682599764856f;
  constexpr float c2k = S == 0 ? kC16 : 0.98078528040323044913f, s2k = S == 0 ? kS16 : 0.19509032201612826785f;   // W_P^2048
  constexpr float c6k = S == 0 ? kS16 : 0.83146961230254523708f, s6k = S == 0 ? kC16 : 0.55557023301960222474f;   // W_P^6144
  if (u == 1 && q == 1) return ctw<FWD>(c0, cuq, suq);
  cpx z = c0;
  if (u == 1) z = ctw<FWD>(z, cu, su);
  if (q == 0) return z;
  if (q == 1) return ctw<FWD>(z, cq, sq);
  cpx w;
  if (q == 2) {   // W_P^8192 conj(z): S = 0: -+i conj(z); S = 1: W_8 conj(z)
    if (S == 0) w = FWD ? mk(-z.y, -z.x) : mk(z.y, z.x);
    else w = ctw<FWD>(mk(z.x, -z.y), kC8, kC8);
    if (u == 0 && lane == 0) w = mk(c2k, FWD ? -s2k : s2k);
  } else {        // W_P^4096 conj(z)
    w = ctw<FWD>(mk(z.x, -z.y), cq, sq);
    if (u == 0 && lane == 0) w = mk(c6k, FWD ? -s6k : s6k);
  }
  return w;
}
// The same for 8192-point sub-transforms (packed real size 32768, M = 8192): i = pair_index<13, 4>(lane, u, q),
//   j = lane + 512 u (u = 0..3):  q = 0 -> i = j, 1 -> 4096 - j   (lane 0, u = 0: i = 0, 2048).
// S = 0: P = 16384, S = 1: P = 32768.
template <bool FWD, int S> CLFA_HD cpx pair_tw13(cpx c0, int u, int q, int lane) {
  static_assert(S == 0 || S == 1, "");
  // W_P^(512 u): W_32^u (S = 0), W_64^u (S = 1)
  constexpr float cu[2][4] = {{1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f},
                              {1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f}};
  constexpr float su[2][4] = {{0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f},
                              {0.0f, 0.09801714032956060199f, 0.19509032201612826785f, 0.29028467725446236764f}};
  cpx z = c0;
  if (u == 1) z = ctw<FWD>(z, cu[S][1], su[S][1]);
  if (u == 2) z = ctw<FWD>(z, cu[S][2], su[S][2]);
  if (u == 3) z = ctw<FWD>(z, cu[S][3], su[S][3]);
  if (q == 0) return z;
  cpx w;   // W_P^4096 conj(z): S = 0: -+i conj(z); S = 1: W_8 conj(z)
  if (S == 0) w = FWD ? mk(-z.y, -z.x) : mk(z.y, z.x);
  else w = ctw<FWD>(mk(z.x, -z.y), kC8, kC8);
  if (u == 0 && lane == 0) w = S == 0 ? mk(kC8, FWD ? -kC8 : kC8) : mk(kC16, FWD ? -kS16 : kS16);   // W_P^2048
  return w;
}
// (LOGC = 11 with EIGHT points per lane — passes 8 x 8 x 8 x 4, 256 lanes — has the pair structure of LOGC = 14 scaled by
// 1/8: four pairs per u, NB = M / 4, and only u = 0; every constant of pair_tw14 is a ratio of NB or of the lane-0
// positions to P, so it applies unchanged)
template <int LOGC, bool FWD, int S> CLFA_HD cpx pair_tw2x(cpx c0, int u, int q, int lane) {
  static_assert(LOGC == 11 || LOGC == 13 || LOGC == 14, "sub-transforms of 2048, 8192 or 16384 points");
  if constexpr (LOGC == 13) return pair_tw13<FWD, S>(c0, u, q, lane);
  else return pair_tw14<FWD, S>(c0, u, q, lane);
}
// forward: slot (u, q) of lane `lane`: A[i], A[M - i], B[i], B[M - i] (for i = 0: A[0], A[M / 2], ...) ->
// st(position, packed spectrum value) x 4.  g0 = W_2M^lane, h0 = W_4M^lane (the r2c table's entries 2 lane, lane).
// M = 2^LOGC is the length of the sub-transforms.
template <int LOGC, class St> CLFA_HD void rfft2x_fwd_slot(int lane, int u, int q, int i, cpx ai, cpx aj, cpx bi, cpx bj,
                                                          cpx g0, cpx h0, St st) {
  constexpr int M = 1 << LOGC;
  const bool first = i == 0;
  const cpx g = pair_tw2x<LOGC, true, 0>(g0, u, q, lane), h = pair_tw2x<LOGC, true, 1>(h0, u, q, lane);
  const cpx gp = first ? mk(0.f, -1.f) : mk(-g.x, g.y);       // W_2M^(M - i) = -conj(g);  W_2M^(M / 2) = -i
  const cpx hp = first ? mk(kC8, -kC8) : mk(-h.y, -h.x);      // W_4M^(M - i) = -i conj(h);  W_4M^(M / 2) = W_8
  const cpx t1 = cmul(g, bi), t2 = cmul(gp, bj);
  const cpx zi_a = cadd(ai, t1), zp_a = cadd(aj, t2);
  cpx zi_b = csub(ai, t1), zp_b = csub(aj, t2);               // Z[M + i], Z[2M - i]
  if (first) {                                                // i = 0: pair 1 is (Z[0], Z[M]), pair 2 (Z[M/2], Z[3M/2])
    const cpx s = zi_b;
    zi_b = zp_b;
    zp_b = s;
  }
  cpx o1a, o1b, o2a, o2b;
  r2c_pair(zi_a, zp_b, h, o1a, o1b);
  r2c_pair(zp_a, zi_b, hp, o2a, o2b);
  if (first) {   // packed DC / Nyquist; bin n / 2 copied through (cl_fft.cpp:178-191 never visits it)
    o1a = mk((zi_a.x + zi_a.y) * .5f, (zi_a.x - zi_a.y) * .5f);
    o1b = zp_b;
  }
  const int i2 = first ? M / 2 : i;
  st(i, o1a);
  st(first ? M : 2 * M - i, o1b);
  st(M - i2, o2a);
  st(M + i2, o2b);
}
// positions of the slot's four packed bins: pair (i, 2M - i) and pair (M - i, M + i); the slot with i = 0 (lane 0's
// first) holds (0, M) and (M / 2, 3M / 2)
template <int LOGC> CLFA_HD int rfft2x_pos(int i, int which) {
  constexpr int M = 1 << LOGC;
  const bool first = i == 0;
  const int i2 = first ? M / 2 : i;
  return which == 0 ? i : which == 1 ? (first ? M : 2 * M - i) : which == 2 ? M - i2 : M + i2;
}
// inverse: the four packed bins of the slot (x1a .. x2b at rfft2x_pos(i, 0 .. 3)) -> inputs of the two transposed
// chains: oa / ob = value at sub-position i, pa / pb = value at its partner M - i (pass_first_paired's oi[k] / oj[k])
template <int LOGC> CLFA_HD void rfft2x_inv_slot(int lane, int u, int q, int i, cpx g0, cpx h0, cpx x1a, cpx x1b, cpx x2a,
                                                 cpx x2b, cpx &oa, cpx &pa, cpx &ob, cpx &pb) {
  const bool first = i == 0;
  const cpx g = pair_tw2x<LOGC, false, 0>(g0, u, q, lane), h = pair_tw2x<LOGC, false, 1>(h0, u, q, lane);
  const cpx gp = first ? mk(0.f, 1.f) : mk(-g.x, g.y);
  const cpx hp = first ? mk(kC8, kC8) : mk(h.y, h.x);
  cpx z1a, z1b, z2a, z2b;
  c2r_pair(x1a, x1b, h, z1a, z1b);
  c2r_pair(x2a, x2b, hp, z2a, z2b);
  if (first) {
    z1a = mk(x1a.x + x1a.y, x1a.x - x1a.y);
    z1b = z2b;      // Z[3M/2] pairs with Z[M/2] below ...
    z2b = x1b;      // ... and Z[M] (copied through) with Z[0]
  }
  oa = cadd(z1a, z2b);
  ob = cmul(csub(z1a, z2b), g);
  pa = cadd(z2a, z1b);
  pb = cmul(csub(z2a, z1b), gp);
}

}  // namespace clfa
