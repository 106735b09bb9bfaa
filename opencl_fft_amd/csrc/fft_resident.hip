// fft_resident.hip — n = 65536 complex, one HBM pass, nothing but the input and the output ever
// leaves the compute unit.
//
// Replaces the reference's reorder + 16 stage launches for N = 65536 (cl_fft.cpp:24-41, 138-151).
//
// The transform is N1 x N2 = 256 x 256 (four-step): phase 1 = 256-point FFTs down the columns of the
// row-major input times W_N^(n2 k1), phase 2 = 256-point FFTs along the rows, stored transposed
// (X[k1 + 256 k2]).  512 KiB per transform do not fit the 160 KiB of LDS — but they do fit the compute
// unit: ONE 256-lane workgroup per CU (one wave per SIMD) owns the whole 512-entry register file
// of every lane (256 arch VGPRs + 256 accumulation VGPRs).
//
// Lane l = c + 16 t.  Phase 1, column block cb (16 columns, 128-byte row segments): the lane loads
// rows t + 16 e of column n2 = 16 cb + c, and after the two radix-16 passes (one LDS exchange) it
// holds Z[k1 = t + 16 e][n2].  Phase 2, row block rb (16 rows): the lane that works on row
// k1 = 16 rb + t at positions n2 = c + 16 e needs exactly Z[16 rb + t][16 e + c], e = 0..15 — the
// values this very lane produced for e = rb in column blocks cb = 0..15.  So the intermediate never
// changes lanes: every lane keeps a private 16 x 16 matrix keep[rb][cb] of complex values
//   rb 0..2   in LDS          (lane-private spill area, 400 bytes per lane)
//   rb 3      in a 32 KiB per-workgroup global slot (L2-resident: 8 MiB for the whole chip) — the one
//             sixteenth that the register file cannot take next to the working registers
//   rb 4..7   in arch VGPRs   (four 32-float vectors, written through s_set_gpr_idx)
//   rb 8..15  in AGPRs        (a[32 (rb-8) + 2 cb], written / read by v_accvgpr moves with literal
//                              register numbers inside a uniform switch: nothing the compiler
//                              allocates lives in the accumulation file)
// and no hand-over between the phases exists at all.  Fabric traffic is the algorithmic 16 bytes per
// sample plus the slot's 0.5 (written through) and at most 0.5 (read back, normally from L2).
//
// Global accesses are buffer_load/store_dwordx2 ... offen nt with the row offsets e * 32 KiB in
// SGPRs: one instruction per access, no address arithmetic.  Both LDS exchanges write 8 x b128 and
// read 16 x b64, conflict-free (layouts below).  The four-step twiddles W_N^(n2 (t + 16 e)) =
// b * s^e come from one two-level lookup (b), four exact table values s, s^2, s^4, s^8 and a
// product tree (15 complex multiplies for 16 values); the forward 1/N rides on the table of b.
#include <hip/hip_runtime.h>

#include <utility>

#include "internal.hpp"

namespace clfa {
namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

constexpr int kN = 65536;
// row blocks per storage class, in this order: rb 0..2 LDS, rb 3 global slot, rb 4..7 VGPR, rb 8..15 AGPR
constexpr int kLdsBlk = 3, kGlbBlk = 1, kVgprBlk = 4, kAgprBlk = 8;
constexpr int kVgprFirst = kLdsBlk + kGlbBlk, kAgprFirst = kVgprFirst + kVgprBlk;
static_assert(kAgprFirst + kAgprBlk == 16, "16 row blocks");
constexpr int kSpillStride = 400;   // bytes per lane: 3 x 128 + 16 (36 dwords mod 64: b64 / b128 conflict-free)
constexpr int kXA = 258;            // phase-1 exchange: element (column c, position p) at c * 258 + p
constexpr int kXB = 290;            // phase-2 exchange: element (row r, position p) at r * 290 + p + 2 (p / 16)
constexpr int kXSize = 16 * kXB;
// table blob (host: fill_res16_tables): [tw 16x16 | lo 256 | hi 256 | S 4x256]
constexpr int kTabTw = 0, kTabLo = 256, kTabHi = 512, kTabS = 768, kTabSize = 1792;

// ---- AGPR file, addressed by literal register numbers ------------------------------------------
template <int I> __device__ __forceinline__ void acc_write(float v) {
  asm volatile("v_accvgpr_write_b32 a[%0], %1" ::"n"(I), "v"(v));
}
template <int I> __device__ __forceinline__ float acc_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(I));
  return v;
}
// the kernel's descriptor has to allocate all 256 AGPRs: name them as clobbered once
#define CLFA_A10(p) "a" #p "0", "a" #p "1", "a" #p "2", "a" #p "3", "a" #p "4", "a" #p "5", "a" #p "6", "a" #p "7", "a" #p "8", "a" #p "9"
__device__ __forceinline__ void acc_claim_all() {
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", CLFA_A10(1), CLFA_A10(2), CLFA_A10(3),
               CLFA_A10(4), CLFA_A10(5), CLFA_A10(6), CLFA_A10(7), CLFA_A10(8), CLFA_A10(9), CLFA_A10(10), CLFA_A10(11),
               CLFA_A10(12), CLFA_A10(13), CLFA_A10(14), CLFA_A10(15), CLFA_A10(16), CLFA_A10(17), CLFA_A10(18),
               CLFA_A10(19), CLFA_A10(20), CLFA_A10(21), CLFA_A10(22), CLFA_A10(23), CLFA_A10(24), "a250", "a251", "a252",
               "a253", "a254", "a255");
}
#undef CLFA_A10
// column block CB: element e = 8 + J of the lane's results goes to a[32 J + 2 CB]
template <int CB, int... J> __device__ __forceinline__ void acc_deposit(const cpx (&o)[16], std::integer_sequence<int, J...>) {
  ((acc_write<32 * J + 2 * CB>(o[kAgprFirst + J].x), acc_write<32 * J + 2 * CB + 1>(o[kAgprFirst + J].y)), ...);
}
// row block 8 + RB: a[32 RB + 2 e] -> v[e]
template <int RB, int... E> __device__ __forceinline__ void acc_fetch(cpx (&v)[16], std::integer_sequence<int, E...>) {
  ((v[E].x = acc_read<32 * RB + 2 * E>(), v[E].y = acc_read<32 * RB + 2 * E + 1>()), ...);
}

// ---- global accesses ----------------------------------------------------------------------------
// raw buffer descriptor over one transform (base wave-uniform: it stays in SGPRs)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t res_rsrc(const cpx *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<cpx *>(base), 0, 0x7fffffff, 0x00020000);
}
// PROBE (tools/res16_probe.hip only; the library instantiates 0): timing experiments that leave parts
// of the kernel out — 1 no global loads, 2 no global stores, 4 no barriers, 8 no slot traffic,
// 16 per-phase clock stamps into `dbg`
enum { kProbeNoLoad = 1, kProbeNoStore = 2, kProbeNoBarrier = 4, kProbeNoSlot = 8, kProbeStamps = 16 };
template <int PROBE> __device__ __forceinline__ void res_barrier() {
  if constexpr (!(PROBE & kProbeNoBarrier)) __syncthreads();
}
// 16 rows 16 apart (32 KiB), lane offset `voff` bytes; non-temporal (aux 2)
template <int PROBE = 0> __device__ __forceinline__ void res_load(cpx (&v)[16], __amdgpu_buffer_rsrc_t r, int voff) {
#pragma unroll
  for (int e = 0; e < 16; e++) {
    if constexpr (PROBE & kProbeNoLoad) {
      asm volatile("" : "+v"(v[e]));
    } else {
      const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(r, voff, e * 32768, 2);
      v[e] = __builtin_bit_cast(cpx, raw);
    }
  }
}
template <int PROBE = 0> __device__ __forceinline__ void res_store(const cpx (&v)[16], __amdgpu_buffer_rsrc_t r, int voff) {
#pragma unroll
  for (int e = 0; e < 16; e++) {
    if constexpr (PROBE & kProbeNoStore) {
      cpx t = v[e];
      asm volatile("" : "+v"(t));
    } else {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v[e]), r, voff, e * 32768, 2);
    }
  }
}

__device__ __forceinline__ f4 pack2(cpx a, cpx b) { return f4{a.x, a.y, b.x, b.y}; }

// second radix-16 pass of a 256-point transform: inputs times W_256^(t j) (row t of the table), then
// the butterflies
template <bool FWD> __device__ __forceinline__ void twiddled_dft16(cpx (&v)[16], const cpx *tw_row) {
  const f4 *pt = reinterpret_cast<const f4 *>(tw_row);
  {
    const f4 w = pt[0];
    v[1] = cmulc<!FWD>(v[1], mk(w.z, w.w));
  }
#pragma unroll
  for (int i = 1; i < 8; i++) {
    const f4 w = pt[i];
    cmulc2<!FWD>(v[2 * i], v[2 * i + 1], v[2 * i], mk(w.x, w.y), v[2 * i + 1], mk(w.z, w.w));
  }
  dft16<1, 16, FWD>(v, 0);
}

struct ResLane {
  int c, t;          // lane = c + 16 t
  int voff;          // byte offset of the lane inside a column / row block of global memory
  cpx *xa_w;         // phase-1 exchange: 16 consecutive elements written (b128)
  const cpx *xa_r;   //   ... elements 16 e apart read
  cpx *xb_w;         // phase-2 exchange
  const cpx *xb_r;
  const cpx *tw_row; // W_256^(t j), j = 0..15
  char *spill;       // lane-private LDS rows
  int slot_off;      // byte offset of the lane in one [cb] row of the workgroup's global slot
};

// ---- phase 1: one column block ------------------------------------------------------------------
// v: rows t + 16 e of column n2 = 16 cb + c (already loaded) -> o[e] = Z[t + 16 e][n2]
template <bool FWD, int PROBE = 0> __device__ __forceinline__ void res_col_block(cpx (&v)[16], const ResLane &L, int cb, const cpx *s_tab, cpx *s_x) {
  dft16<1, 16, FWD>(v, 0);
  res_barrier<PROBE>();   // the previous block's readers are done with the exchange buffer
  {
    f4 *pw = reinterpret_cast<f4 *>(L.xa_w);
#pragma unroll
    for (int i = 0; i < 8; i++) pw[i] = pack2(v[2 * i], v[2 * i + 1]);
  }
  res_barrier<PROBE>();
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = L.xa_r[16 * e];
  twiddled_dft16<FWD>(v, L.tw_row);
  // four-step twiddles W_N^(n2 (t + 16 e)) = b * s^e,  b = W_N^(n2 t),  s = W_4096^n2
  const int n2 = cb * 16 + L.c;
  const int m = n2 * L.t;   // < 4096
  const cpx b = cmul(s_tab[kTabLo + (m & 255)], s_tab[kTabHi + (m >> 8)]);
  const cpx *ps = s_tab + kTabS + n2;
  const cpx s1 = ps[0], s2 = ps[256], s4 = ps[512], s8 = ps[768];
  // product tree in halves of four (T_r = b s^r, U_r = T_r s^8), two products per statement
  cpx T[4], U[4];
  T[0] = b;
  cmulc2(T[1], T[2], b, s1, b, s2);
  cmulc2(T[3], U[0], T[1], s2, b, s8);
  cmulc2(U[1], U[2], T[1], s8, T[2], s8);
  U[3] = cmul(T[3], s8);
#pragma unroll
  for (int r = 0; r < 4; r++) cmulc2<!FWD>(v[r], v[r + 8], v[r], T[r], v[r + 8], U[r]);
  cmulc2(T[0], T[1], T[0], s4, T[1], s4);
  cmulc2(T[2], T[3], T[2], s4, T[3], s4);
  cmulc2(U[0], U[1], T[0], s8, T[1], s8);
  cmulc2(U[2], U[3], T[2], s8, T[3], s8);
#pragma unroll
  for (int r = 0; r < 4; r++) cmulc2<!FWD>(v[r + 4], v[r + 12], v[r + 4], T[r], v[r + 12], U[r]);
}

// o[e] -> keep[e][cb]
template <int PROBE = 0>
__device__ __forceinline__ void res_deposit(const cpx (&o)[16], const ResLane &L, int cb, f32x32 (&K)[kVgprBlk],
                                            __amdgpu_buffer_rsrc_t slot) {
  {
    cpx *ps = reinterpret_cast<cpx *>(L.spill + cb * 8);
#pragma unroll
    for (int e = 0; e < kLdsBlk; e++) ps[e * 16] = o[e];
  }
  // the one row block that does not fit the CU: [cb][lane] in the workgroup's 32 KiB slot (L2-resident)
  if constexpr (!(PROBE & kProbeNoSlot))
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[kLdsBlk]), slot, L.slot_off, cb * 2048, 0);
#pragma unroll
  for (int j = 0; j < kVgprBlk; j++) {
    K[j][2 * cb] = o[kVgprFirst + j].x;
    K[j][2 * cb + 1] = o[kVgprFirst + j].y;
  }
  using S8 = std::make_integer_sequence<int, kAgprBlk>;
  switch (cb) {
#define CLFA_C(c) case c: acc_deposit<c>(o, S8()); break;
    CLFA_C(0) CLFA_C(1) CLFA_C(2) CLFA_C(3) CLFA_C(4) CLFA_C(5) CLFA_C(6) CLFA_C(7)
    CLFA_C(8) CLFA_C(9) CLFA_C(10) CLFA_C(11) CLFA_C(12) CLFA_C(13) CLFA_C(14)
#undef CLFA_C
    default: acc_deposit<15>(o, S8()); break;
  }
}

// keep[rb][e] -> v[e]
template <int RB> __device__ __forceinline__ void res_fetch_static(cpx (&v)[16], const ResLane &L, const f32x32 (&K)[kVgprBlk]) {
  if constexpr (RB < kLdsBlk) {
    const f4 *pf = reinterpret_cast<const f4 *>(L.spill + RB * 128);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const f4 w = pf[i];
      v[2 * i] = mk(w.x, w.y);
      v[2 * i + 1] = mk(w.z, w.w);
    }
  } else if constexpr (RB < kVgprFirst) {
    static_assert(RB >= kVgprFirst || RB < kLdsBlk, "the global slot's block is fetched by res_slot_load");
  } else if constexpr (RB < kAgprFirst) {
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = mk(K[RB - kVgprFirst][2 * e], K[RB - kVgprFirst][2 * e + 1]);
  } else {
    acc_fetch<RB - kAgprFirst>(v, std::make_integer_sequence<int, 16>());
  }
}
// (rb = 3, the global slot's block: `slot_v`, loaded at the start of phase 2)
__device__ __forceinline__ void res_fetch(cpx (&v)[16], const ResLane &L, int rb, const f32x32 (&K)[kVgprBlk],
                                          const cpx (&slot_v)[16]) {
  switch (rb) {
#define CLFA_C(r) case r: res_fetch_static<r>(v, L, K); break;
    CLFA_C(0) CLFA_C(1) CLFA_C(2)
    case kLdsBlk:
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = slot_v[e];
      break;
    CLFA_C(4) CLFA_C(5) CLFA_C(6) CLFA_C(7)
    CLFA_C(8) CLFA_C(9) CLFA_C(10) CLFA_C(11) CLFA_C(12) CLFA_C(13) CLFA_C(14)
#undef CLFA_C
    default: res_fetch_static<15>(v, L, K); break;
  }
}

// ---- phase 2: one row block ---------------------------------------------------------------------
// v[e] = Z[16 rb + t][c + 16 e] -> X[16 rb + c + 256 (t + 16 e)] left in v[e] (lane = row c, k2 = t + 16 e)
template <bool FWD, int PROBE = 0> __device__ __forceinline__ void res_row_block(cpx (&v)[16], const ResLane &L) {
  dft16<1, 16, FWD>(v, 0);
  res_barrier<PROBE>();
  {
    f4 *pw = reinterpret_cast<f4 *>(L.xb_w);
#pragma unroll
    for (int i = 0; i < 8; i++) pw[i] = pack2(v[2 * i], v[2 * i + 1]);
  }
  res_barrier<PROBE>();
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = L.xb_r[18 * e];
  twiddled_dft16<FWD>(v, L.tw_row);
}

}  // namespace

// slots: one 32 KiB slot per workgroup (the single row block that does not fit the CU)
template <bool FWD, bool SCALE, int PROBE = 0>
__global__ __launch_bounds__(256) void k_fft_res16(cpx *__restrict__ data, cpx *__restrict__ slots,
                                                   const cpx *__restrict__ tabs_g, long batch,
                                                   unsigned long long *__restrict__ dbg = nullptr) {
  __shared__ __attribute__((aligned(16))) cpx s_tab[kTabSize];
  __shared__ __attribute__((aligned(16))) cpx s_x[kXSize];
  __shared__ __attribute__((aligned(16))) char s_spill[256 * kSpillStride];
  acc_claim_all();
  const int tid = threadIdx.x;
  for (int i = tid; i < kTabSize; i += 256) {
    cpx w = tabs_g[i];
    if (SCALE && i >= kTabLo && i < kTabHi) w = cscale(w, 1.0f / (float)kN);   // exact: a power of two
    s_tab[i] = w;
  }
  ResLane L;
  L.c = tid & 15;
  L.t = tid >> 4;
  L.voff = L.t * 2048 + L.c * 8;
  L.xa_w = s_x + L.c * kXA + 16 * L.t;
  L.xa_r = s_x + L.c * kXA + L.t;
  L.xb_w = s_x + L.t * kXB + 18 * L.c;
  L.xb_r = s_x + L.c * kXB + L.t;
  L.tw_row = s_tab + kTabTw + 16 * L.t;
  L.spill = s_spill + tid * kSpillStride;
  L.slot_off = tid * 8;
  const __amdgpu_buffer_rsrc_t slot = res_rsrc(slots + (long)blockIdx.x * 4096);
  __syncthreads();

  f32x32 K[kVgprBlk];
#pragma unroll
  for (int j = 0; j < kVgprBlk; j++) K[j] = 0.f;
  cpx v[16], vn[16];
  long b = blockIdx.x;
  unsigned long long clk1 = 0, clk2 = 0;
  res_load<PROBE>(v, res_rsrc(data + b * (long)kN), L.voff);
#pragma unroll 1
  for (; b < batch; b += gridDim.x) {
    cpx *x = data + b * (long)kN;
    unsigned long long t0 = 0;
    if constexpr (PROBE & kProbeStamps) t0 = __builtin_amdgcn_s_memtime();
    // ---- phase 1: column blocks two at a time (v and vn change roles), the next block's loads in flight
#pragma unroll 1
    for (int cb = 0; cb < 14; cb += 2) {
      res_load<PROBE>(vn, res_rsrc(x + (cb + 1) * 16), L.voff);
      res_col_block<FWD, PROBE>(v, L, cb, s_tab, s_x);
      res_deposit<PROBE>(v, L, cb, K, slot);
      res_load<PROBE>(v, res_rsrc(x + (cb + 2) * 16), L.voff);
      res_col_block<FWD, PROBE>(vn, L, cb + 1, s_tab, s_x);
      res_deposit<PROBE>(vn, L, cb + 1, K, slot);
    }
    res_load<PROBE>(vn, res_rsrc(x + 15 * 16), L.voff);
    res_col_block<FWD, PROBE>(v, L, 14, s_tab, s_x);
    res_deposit<PROBE>(v, L, 14, K, slot);
    res_col_block<FWD, PROBE>(vn, L, 15, s_tab, s_x);
    res_deposit<PROBE>(vn, L, 15, K, slot);
    // ---- phase 2: row blocks; the slot's block comes back behind the first three (agent-scope
    // loads: they bypass this CU's L1, which may still hold the previous transform's lines)
    if constexpr (PROBE & kProbeStamps) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      clk1 += t1 - t0;
      t0 = t1;
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
      if constexpr (PROBE & kProbeNoSlot) asm volatile("" : "+v"(vn[e]));
      else vn[e] = __builtin_bit_cast(cpx, __builtin_amdgcn_raw_buffer_load_b64(slot, L.slot_off, e * 2048, 16));
    }
#pragma unroll 1
    for (int rb = 0; rb < 15; rb++) {
      res_fetch(v, L, rb, K, vn);
      res_row_block<FWD, PROBE>(v, L);
      res_store<PROBE>(v, res_rsrc(x + rb * 16), L.voff);
    }
    // last row block: the next transform's first column block is loaded behind it (index clamped to
    // the last transform so that the loads are straight-line code)
    long bn = b + gridDim.x;
    bn = bn < batch ? bn : batch - 1;
    res_load<PROBE>(vn, res_rsrc(data + bn * (long)kN), L.voff);
    res_fetch_static<15>(v, L, K);
    res_row_block<FWD, PROBE>(v, L);
    res_store<PROBE>(v, res_rsrc(x + 15 * 16), L.voff);
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = vn[e];
    if constexpr (PROBE & kProbeStamps) clk2 += __builtin_amdgcn_s_memtime() - t0;
  }
  if constexpr (PROBE & kProbeStamps) {
    if (tid == 0) {
      dbg[2 * blockIdx.x] = clk1;
      dbg[2 * blockIdx.x + 1] = clk2;
    }
  }
}

hipError_t launch_fft_res16(bool fwd, bool scale, cpx *data, cpx *slots, const cpx *tabs, long batch,
                            const DeviceInfo &di, hipStream_t s) {
  if (batch <= 0) return hipSuccess;
  const int grid = (int)(batch < di.num_cus ? batch : di.num_cus);
  if (fwd && scale) hipLaunchKernelGGL((k_fft_res16<true, true>), dim3(grid), dim3(256), 0, s, data, slots, tabs, batch);
  else if (fwd) hipLaunchKernelGGL((k_fft_res16<true, false>), dim3(grid), dim3(256), 0, s, data, slots, tabs, batch);
  else if (!scale) hipLaunchKernelGGL((k_fft_res16<false, false>), dim3(grid), dim3(256), 0, s, data, slots, tabs, batch);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace clfa
