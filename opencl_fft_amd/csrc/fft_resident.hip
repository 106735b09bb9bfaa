// fft_resident.hip — n = 65536 complex in one HBM pass: the intermediate of the four-step transform
// stays on the compute unit.
//
// Replaces the reference's reorder + 16 stage launches for N = 65536 (cl_fft.cpp:24-41, 138-151).
//
// The transform is N1 x N2 = 256 x 256 (four-step): phase 1 = 256-point FFTs down the columns of the
// row-major input times W_N^(n2 k1), phase 2 = 256-point FFTs along the rows, stored transposed
// (X[k1 + 256 k2]).  512 KiB per transform do not fit the 160 KiB of LDS — but they (almost) fit the
// compute unit: ONE 256-lane workgroup per CU (one wave per SIMD) owns the whole 512-entry register
// file of every lane (256 arch VGPRs + 256 accumulation VGPRs).
//
// Lane l = c + 16 t.  Phase 1, column block cb (16 columns, 128-byte row segments): the lane takes
// rows t + 16 e of column n2 = 16 cb + c, and after the two radix-16 passes (one LDS exchange) it
// holds Z[k1 = t + 16 e][n2].  Phase 2, row block rb (16 rows): the lane that works on row
// k1 = 16 rb + t at positions n2 = c + 16 e needs exactly Z[16 rb + t][16 e + c], e = 0..15 — the
// values this very lane produced for e = rb in column blocks cb = 0..15.  So the intermediate never
// changes lanes: every lane keeps a private 16 x 16 matrix keep[rb][cb] of complex values
//   rb 0..2   in LDS          (lane-private spill area, 400 bytes per lane)
//   rb 3      in a 32 KiB per-workgroup global slot (8 MiB for the whole chip, read back from L2) — the
//             one sixteenth that the register file cannot take next to the working registers
//   rb 4..7   in arch VGPRs   (four 32-float vectors, written through s_set_gpr_idx)
//   rb 8..15  in AGPRs        (a[32 (rb-8) + 2 cb], moved by v_accvgpr_* with literal register
//                              numbers inside a uniform switch)
// and no hand-over between the phases exists at all.  Fabric traffic is the algorithmic 16 bytes per
// sample plus the slot's 0.5 (written through) and at most 0.5 (read back).
//
// What one wave per SIMD costs, and what the kernel does about it (tools/res16_probe.hip measures every
// item; profiles/res16_probe_r02.txt):
//   * nothing else runs while the wave waits, so a block's 16 loads / stores cannot be issued back
//     to back (each then waits ~80 cycles for a queue slot): they ride along the arithmetic, one per
//     hook point (16 per block, ~20 instructions apart) — the loads of the column block TWO ahead in
//     phase 1, the stores of the PREVIOUS row block in phase 2;
//   * those loads need somewhere to land that hipcc does not touch (it copies "its" registers
//     whenever it likes, also while a load is still pending on them): even column blocks land in AGPR
//     columns of the keep matrix that are still empty, odd ones in v[224:255] — the kernel is compiled
//     with amdgpu_num_vgpr(224), so the register allocator never sees those; the same registers park
//     a row block's results in phase 2 until their stores have been issued;
//   * hipcc counts none of that traffic: every wait is an explicit s_waitcnt vmcnt(N), N = the asm
//     loads issued after the awaited ones.
// tools/check_isa.py audits the code object (no compiler-generated AGPR moves, no compiler instruction
// on v[224:255], no scratch); tests/test_abi_cpu.py runs it on every build.
//
// Global accesses are buffer_load/store_dwordx2 ... offen nt with the row offsets e * 32 KiB in
// SGPRs: one instruction per access, no address arithmetic.  Both LDS exchanges write 8 x b128 and
// read 16 x SINGLE b64 (CLFA_DS_SINGLE below: paired into ds_read2_b64, as hipcc would, every read is a 2-way bank
// conflict), conflict-free under MI355X_MICROARCH.md's lane-group rules (layouts below; rocprofv3 round 5:
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.29 -> 0.05, what is left is the spill deposit).  The four-step twiddles W_N^(n2 (t + 16 e)) =
// b * s^e come from one two-level lookup (b), four exact table values s, s^2, s^4, s^8 and a
// product tree (15 complex multiplies for 16 values); the forward 1/N rides on the table of b.
//
// The same kernel carries the packed real transforms of size 131072 (the largest of the reference's range,
// cl_fft.cpp:208-211, 267-296) in one pass as well: template flags R2C (the reference's `conv` pair map inside phase 2) and
// C2R (`iconv` inside phase 1) — sections "packed real transforms ... forward / inverse" below.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "internal.hpp"

namespace clfa {
namespace {

// cache policy of the streams (tuning switches for A/B builds; the library's choice is the default):
// CLFA_LDNT / CLFA_STNT = the modifier string of the asm accesses, CLFA_ST_AUX = the same policy as the aux
// operand of the store builtin (bit 0 sc0, bit 1 nt, bit 4 sc1)
#ifndef CLFA_LDNT
#define CLFA_LDNT " nt"
#endif
#ifndef CLFA_STNT
#define CLFA_STNT " nt"
#define CLFA_ST_AUX 2
#endif
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

constexpr int kN = 65536;
// row blocks per storage class, in this order: rb 0..2 LDS, rb 3 global slot, rb 4..7 VGPR, rb 8..15 AGPR
constexpr int kLdsBlk = 3, kGlbBlk = 1, kVgprBlk = 4, kAgprBlk = 8;
constexpr int kVgprFirst = kLdsBlk + kGlbBlk, kAgprFirst = kVgprFirst + kVgprBlk;
static_assert(kAgprFirst + kAgprBlk == 16, "16 row blocks");
// bytes per lane: 3 x 128 + 16 = 100 dwords.  The b128 reads of phase 2 (16-lane groups over 64 banks: 36 l mod 64) are
// conflict-free; the three ds_write_b64 of a deposit (16 contiguous lanes over 32 banks: 4 l mod 32) pair lanes l, l + 8 —
// 8 LDS-array cycles against the 6 the instruction takes to hand its operands over anyway: 2 cycles per write.  A stride
// that serves both (2 x odd dwords) would turn the reads into 16 x b64 for nothing measurable.
constexpr int kSpillStride = 400;
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
               "a253", "a254", "a255",
               // ... and the landing registers v[224:255] (kept out of hipcc's hands by amdgpu_num_vgpr(224))
               "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237",
               "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251",
               "v252", "v253", "v254", "v255");
}
#undef CLFA_A10
// Two AGPR column pairs double as landing zones for the even column blocks' loads while they are
// still empty: Z0 = columns 14, 15 (blocks 0, 4, 8, 12), Z1 = columns 12, 13 (blocks 2, 6, 10, 14).
// They fall free in the order Z0 (block 12 taken out), Z1 (block 14 taken out), so the keep matrix's
// columns 12..15 are stored swapped: logical column c lives in physical column acc_col(c).
constexpr int kZone0 = 14, kZone1 = 12;
// INV (the packed real inverse kernel, whose phase 1 takes the column blocks in the order 7, 8, 6, 9, ... 0, 15): the
// zones are the columns of the blocks deposited last there — Z0 = blocks 0, 15, Z1 = blocks 1, 14
template <bool INV = false> constexpr int acc_col(int c) {
  if (INV) return c == 0 ? 14 : c == 15 ? 15 : c == 1 ? 12 : c == 14 ? 13 : c - 2;
  return c < 12 ? c : c ^ 2;
}
// column block CB: element e = 8 + J of the lane's results goes to a[32 J + 2 acc_col(CB)]
template <int CB, bool INV, int... J> __device__ __forceinline__ void acc_deposit(const cpx (&o)[16], std::integer_sequence<int, J...>) {
  ((acc_write<32 * J + 2 * acc_col<INV>(CB)>(o[kAgprFirst + J].x), acc_write<32 * J + 2 * acc_col<INV>(CB) + 1>(o[kAgprFirst + J].y)), ...);
}
// row block 8 + RB: a[32 RB + 2 acc_col(e)] -> v[e]
template <int RB, bool INV, int... E> __device__ __forceinline__ void acc_fetch(cpx (&v)[16], std::integer_sequence<int, E...>) {
  ((v[E].x = acc_read<32 * RB + 2 * acc_col<INV>(E)>(), v[E].y = acc_read<32 * RB + 2 * acc_col<INV>(E) + 1>()), ...);
}

// ---- global accesses ----------------------------------------------------------------------------
// raw buffer descriptor over one transform (base wave-uniform: it stays in SGPRs)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t res_rsrc(const cpx *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<cpx *>(base), 0, 0x7fffffff, 0x00020000);
}
// PROBE (tools/res16_probe.hip only; the library instantiates 0): timing experiments that leave parts
// of the kernel out — 1 no global loads, 2 no global stores, 4 no barriers, 8 no slot traffic,
// 16 per-phase clock stamps into `dbg`
//   32 loads issued but never waited for, 64 no arithmetic (exchanges, barriers and memory traffic only),
//   128 phase 1 only (phase 2 skipped), 256 each workgroup starts its address sequence at another
//   column block (blockIdx rotates the 128-byte column offset; results are then garbage)
enum { kProbeNoLoad = 1, kProbeNoStore = 2, kProbeNoBarrier = 4, kProbeNoSlot = 8, kProbeStamps = 16,
       kProbeNoWait = 32, kProbeNoMath = 64, kProbePhase1Only = 128, kProbeRotate = 256, kProbeGridSync = 512,
       kProbeSlots = 1024, kProbePack = 2048, kProbeNoXchg = 4096 };
//   4096 no LDS exchange at all (no ds_write / ds_read / barriers between the two passes of a block: garbage results) —
//   the upper bound of what hiding the exchange behind arithmetic could buy
//   2048 a third phase: the workgroup re-reads its own transform (pairs i, n - i, 8-byte accesses) and writes it back —
//   the memory behaviour of a pair map fused behind phase 2 (what would real size 131072 cost in one launch?)
//   1024 time slots: every workgroup starts phase k no earlier than its own start + S[k] (slot lengths in 10 ns
//   ticks at dbg[2048], dbg[2049]): read and write phases aligned chip-wide without any communication
//   512 a grid-wide barrier at every phase boundary (counter at dbg[1024]; the stamps exclude the wait):
//   what perfectly aligned read and write phases would be worth
template <int PROBE> __device__ __forceinline__ void res_barrier() {
  if constexpr (!(PROBE & kProbeNoBarrier)) __syncthreads();
}
// 16 rows 16 apart (32 KiB), lane offset `voff` bytes; non-temporal (aux 2)
template <int PROBE = 0> __device__ __forceinline__ void res_store(const cpx (&v)[16], __amdgpu_buffer_rsrc_t r, int voff) {
#pragma unroll
  for (int e = 0; e < 16; e++) {
    if constexpr (PROBE & kProbeNoStore) {
      cpx t = v[e];
      asm volatile("" : "+v"(t));
    } else {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v[e]), r, voff, e * 32768, CLFA_ST_AUX);
    }
  }
}

// ---- loads the compiler does not see ---------------------------------------------------------------
// Phase 1 keeps TWO column blocks in flight (64 KiB per CU: one block ahead is latency-bound, see
// DESIGN.md), and there are no 32 spare VGPRs for the second one.  It lands in the accumulation
// registers of the keep matrix's columns cb and cb + 1, which are still empty while block cb waits
// (rows e < 8 -> a[32 e + 2 cb], rows e >= 8 -> a[32 (e - 8) + 2 (cb + 1)]); blocks with odd cb land
// in reserved VGPRs (below).  hipcc counts neither kind (all are asm), so the waits are explicit: s_waitcnt vmcnt(N) with
// N = the asm loads issued after the awaited ones (compiler-issued stores in between only make the
// wait stronger).  The s_nop 4 covers SALU-written descriptor / offset SGPRs read by VMEM.
template <int COL, int E0> __device__ __forceinline__ void res_load_acc8(__amdgpu_buffer_rsrc_t r, int voff) {
#define CLFA_LD "buffer_load_dwordx2 a[%c"
  asm volatile("s_nop 4\n\t"
               "buffer_load_dwordx2 a[%c2:%c3], %0, %1, %18 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 a[%c4:%c5], %0, %1, %19 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 a[%c6:%c7], %0, %1, %20 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 a[%c8:%c9], %0, %1, %21 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 a[%c10:%c11], %0, %1, %22 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 a[%c12:%c13], %0, %1, %23 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 a[%c14:%c15], %0, %1, %24 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 a[%c16:%c17], %0, %1, %25 offen" CLFA_LDNT
               :
               : "v"(voff), "s"(r), "n"(0 * 32 + 2 * COL), "n"(0 * 32 + 2 * COL + 1), "n"(1 * 32 + 2 * COL),
                 "n"(1 * 32 + 2 * COL + 1), "n"(2 * 32 + 2 * COL), "n"(2 * 32 + 2 * COL + 1), "n"(3 * 32 + 2 * COL),
                 "n"(3 * 32 + 2 * COL + 1), "n"(4 * 32 + 2 * COL), "n"(4 * 32 + 2 * COL + 1), "n"(5 * 32 + 2 * COL),
                 "n"(5 * 32 + 2 * COL + 1), "n"(6 * 32 + 2 * COL), "n"(6 * 32 + 2 * COL + 1), "n"(7 * 32 + 2 * COL),
                 "n"(7 * 32 + 2 * COL + 1), "s"((E0 + 0) * 32768), "s"((E0 + 1) * 32768), "s"((E0 + 2) * 32768),
                 "s"((E0 + 3) * 32768), "s"((E0 + 4) * 32768), "s"((E0 + 5) * 32768), "s"((E0 + 6) * 32768),
                 "s"((E0 + 7) * 32768)
               : "memory");
#undef CLFA_LD
}
// one column block (at `base`) -> AGPR columns ZC, ZC + 1, all 16 loads at once
template <int ZC> __device__ __forceinline__ void res_load_acc(const cpx *base, int voff) {
  const __amdgpu_buffer_rsrc_t r = res_rsrc(base);
  res_load_acc8<ZC, 0>(r, voff);
  res_load_acc8<ZC + 1, 8>(r, voff);
}
// ... and back out, once its wait has passed
template <int CB, int... E> __device__ __forceinline__ void acc_fetch_raw(cpx (&v)[16], std::integer_sequence<int, E...>) {
  ((v[E].x = acc_read<32 * E + 2 * CB>(), v[E].y = acc_read<32 * E + 2 * CB + 1>()), ...);
  ((v[8 + E].x = acc_read<32 * E + 2 * CB + 2>(), v[8 + E].y = acc_read<32 * E + 2 * CB + 3>()), ...);
}
// Blocks with odd cb land in v[224:255].  The kernel is compiled with amdgpu_num_vgpr(224), so hipcc
// allocates v0..v223 only and never reads, copies or spills a register with a load still pending on it
// (with compiler-allocated destinations it did: it moved them ahead of the wait).
template <int E0> __device__ __forceinline__ void res_load_land8(__amdgpu_buffer_rsrc_t r, int voff) {
  asm volatile("s_nop 4\n\t"
               "buffer_load_dwordx2 v[%c2:%c3], %0, %1, %18 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 v[%c4:%c5], %0, %1, %19 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 v[%c6:%c7], %0, %1, %20 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 v[%c8:%c9], %0, %1, %21 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 v[%c10:%c11], %0, %1, %22 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 v[%c12:%c13], %0, %1, %23 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 v[%c14:%c15], %0, %1, %24 offen" CLFA_LDNT "\n\t"
               "buffer_load_dwordx2 v[%c16:%c17], %0, %1, %25 offen" CLFA_LDNT
               :
               : "v"(voff), "s"(r), "n"(224 + 2 * E0), "n"(225 + 2 * E0), "n"(226 + 2 * E0), "n"(227 + 2 * E0),
                 "n"(228 + 2 * E0), "n"(229 + 2 * E0), "n"(230 + 2 * E0), "n"(231 + 2 * E0), "n"(232 + 2 * E0),
                 "n"(233 + 2 * E0), "n"(234 + 2 * E0), "n"(235 + 2 * E0), "n"(236 + 2 * E0), "n"(237 + 2 * E0),
                 "n"(238 + 2 * E0), "n"(239 + 2 * E0), "s"((E0 + 0) * 32768), "s"((E0 + 1) * 32768),
                 "s"((E0 + 2) * 32768), "s"((E0 + 3) * 32768), "s"((E0 + 4) * 32768), "s"((E0 + 5) * 32768),
                 "s"((E0 + 6) * 32768), "s"((E0 + 7) * 32768)
               : "memory");
}
__device__ __forceinline__ void res_load_land(const cpx *base, int voff) {
  const __amdgpu_buffer_rsrc_t r = res_rsrc(base);
  res_load_land8<0>(r, voff);
  res_load_land8<8>(r, voff);
}
// ... and out of the landing registers (after the wait)
__device__ __forceinline__ void res_land_fetch(cpx (&v)[16]) {
  asm volatile("v_mov_b64 %0, v[224:225]\n\tv_mov_b64 %1, v[226:227]\n\tv_mov_b64 %2, v[228:229]\n\tv_mov_b64 %3, v[230:231]\n\t"
               "v_mov_b64 %4, v[232:233]\n\tv_mov_b64 %5, v[234:235]\n\tv_mov_b64 %6, v[236:237]\n\tv_mov_b64 %7, v[238:239]"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]));
  asm volatile("v_mov_b64 %0, v[240:241]\n\tv_mov_b64 %1, v[242:243]\n\tv_mov_b64 %2, v[244:245]\n\tv_mov_b64 %3, v[246:247]\n\t"
               "v_mov_b64 %4, v[248:249]\n\tv_mov_b64 %5, v[250:251]\n\tv_mov_b64 %6, v[252:253]\n\tv_mov_b64 %7, v[254:255]"
               : "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
}
// waits for the asm loads: N = the asm loads issued after the awaited ones
template <int N> __device__ __forceinline__ void res_wait_vm() {
  static_assert(N == 0 || N == 1 || N == 2 || N == 16 || N == 32 || N == 48, "");
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if constexpr (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
}

__device__ __forceinline__ f4 pack2(cpx a, cpx b) { return f4{a.x, a.y, b.x, b.y}; }

// ---- loads interleaved with the arithmetic ---------------------------------------------------------
// One wave per SIMD cannot afford to issue a block's 16 loads back to back: with the memory pipeline
// saturated every load instruction then waits ~80 cycles for a queue slot, and nothing else runs on
// that SIMD meanwhile (measured: the loads cost the same whether or not anything waits for their
// data, profiles/res16_probe_r02.txt).  So a column block's code has 16 hook points, ~20 instructions
// apart, and each issues ONE load of the block two ahead.  so[e] = e * 32 KiB, pinned in SGPRs.
template <int K> using ic = std::integral_constant<int, K>;
template <int... I, class F> __device__ __forceinline__ void static_for_(std::integer_sequence<int, I...>, F &&f) { (f(ic<I>()), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_(std::make_integer_sequence<int, N>(), f); }
struct HookNone {
  template <int K> __device__ __forceinline__ void operator()(ic<K>) const {}
};
// KEEP: a plain (cached) load instead of the streaming one — the mirrored loads of the packed real inverse kernel touch
// every line twice, one pair apart (15 of its 16 columns, then the last), and the second touch should find it in L2
template <int CB, bool KEEP = false> struct HookAcc {   // -> AGPR columns CB, CB + 1 (a landing zone)
  __amdgpu_buffer_rsrc_t r;
  int voff;
  const int (&so)[16];
  template <int K> __device__ __forceinline__ void operator()(ic<K>) const {
    constexpr int lo = 32 * (K & 7) + 2 * (K < 8 ? CB : CB + 1);
    // K == 0: the descriptor's SGPRs may be fresh from SALU — 5 wait states before VMEM reads them, in the SAME asm
    // statement as the load (between two statements hipcc may re-materialise the descriptor)
#define CLFA_LD_ACC(PRE, POL) \
  asm volatile(PRE "buffer_load_dwordx2 a[%c2:%c3], %0, %1, %4 offen" POL ::"v"(voff), "s"(r), "n"(lo), "n"(lo + 1), "s"(so[K]) : "memory")
    if constexpr (K == 0 && KEEP) CLFA_LD_ACC("s_nop 4\n\t", "");
    else if constexpr (K == 0) CLFA_LD_ACC("s_nop 4\n\t", CLFA_LDNT);
    else if constexpr (KEEP) CLFA_LD_ACC("", "");
    else CLFA_LD_ACC("", CLFA_LDNT);
#undef CLFA_LD_ACC
  }
};
template <bool KEEP = false> struct HookLandT {   // -> landing registers v[224:255]
  __amdgpu_buffer_rsrc_t r;
  int voff;
  const int (&so)[16];
  template <int K> __device__ __forceinline__ void operator()(ic<K>) const {
#define CLFA_LD_LAND(PRE, POL) \
  asm volatile(PRE "buffer_load_dwordx2 v[%c2:%c3], %0, %1, %4 offen" POL ::"v"(voff), "s"(r), "n"(224 + 2 * K), "n"(225 + 2 * K), "s"(so[K]) : "memory")
    if constexpr (K == 0 && KEEP) CLFA_LD_LAND("s_nop 4\n\t", "");
    else if constexpr (K == 0) CLFA_LD_LAND("s_nop 4\n\t", CLFA_LDNT);
    else if constexpr (KEEP) CLFA_LD_LAND("", "");
    else CLFA_LD_LAND("", CLFA_LDNT);
#undef CLFA_LD_LAND
  }
};
using HookLand = HookLandT<false>;
// block 15 of phase 1 has nothing left to prefetch: its hooks bring the global slot's row block back
// (columns 0..14; column 15 is still in the lane's registers then) into v[224:253]; agent scope (sc1): the
// loads bypass this CU's L1, which may still hold the previous transform's lines
struct HookSlot {
  __amdgpu_buffer_rsrc_t r;
  int voff;
  const int (&so)[16];
  template <int K> __device__ __forceinline__ void operator()(ic<K>) const {
    if constexpr (K < 15) {
      int off;
      asm volatile("s_lshr_b32 %0, %3, 4\n\ts_nop 4\n\tbuffer_load_dwordx2 v[%c4:%c5], %1, %2, %0 offen sc1"
                   : "=&s"(off)
                   : "v"(voff), "s"(r), "s"(so[K]), "n"(224 + 2 * K), "n"(225 + 2 * K)
                   : "memory", "scc");
    }
  }
};
// phase 2: a row block's 16 stores ride along the NEXT block's arithmetic, out of the landing registers
// (idle in phase 2), where res_stage() has put the block's results
struct HookStore {
  __amdgpu_buffer_rsrc_t r;
  int voff;
  const int (&so)[16];
  template <int K> __device__ __forceinline__ void operator()(ic<K>) const {
    if constexpr (K == 0)
      asm volatile("s_nop 4\n\tbuffer_store_dwordx2 v[%c2:%c3], %0, %1, %4 offen" CLFA_STNT ::"v"(voff), "s"(r), "n"(224 + 2 * K), "n"(225 + 2 * K), "s"(so[K]) : "memory");
    else
      asm volatile("buffer_store_dwordx2 v[%c2:%c3], %0, %1, %4 offen" CLFA_STNT ::"v"(voff), "s"(r), "n"(224 + 2 * K), "n"(225 + 2 * K), "s"(so[K]) : "memory");
  }
};
__device__ __forceinline__ void res_stage(const cpx (&v)[16]) {
  asm volatile("v_mov_b64 v[224:225], %0\n\tv_mov_b64 v[226:227], %1\n\tv_mov_b64 v[228:229], %2\n\tv_mov_b64 v[230:231], %3\n\t"
               "v_mov_b64 v[232:233], %4\n\tv_mov_b64 v[234:235], %5\n\tv_mov_b64 v[236:237], %6\n\tv_mov_b64 v[238:239], %7"
               ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
  asm volatile("v_mov_b64 v[240:241], %0\n\tv_mov_b64 v[242:243], %1\n\tv_mov_b64 v[244:245], %2\n\tv_mov_b64 v[246:247], %3\n\t"
               "v_mov_b64 v[248:249], %4\n\tv_mov_b64 v[250:251], %5\n\tv_mov_b64 v[252:253], %6\n\tv_mov_b64 v[254:255], %7"
               ::"v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
}
__device__ __forceinline__ void res_stage_one15(cpx o) { asm volatile("v_mov_b64 v[254:255], %0" ::"v"(o)); }

// hook-point maps: eight local points of dft16_h -> global hook numbers (-1: none)
struct HookMap {
  int p[8];
};
constexpr HookMap kMapColA{{0, -1, 1, -1, 2, -1, 3, -1}};      // column block, first pass: hooks 0..3
constexpr HookMap kMapColB{{8, -1, 9, -1, 10, -1, 11, -1}};    // ... second pass: hooks 8..11 (4..7: twiddles, 12..15: four-step)
constexpr HookMap kMapRowC{{0, 1, 2, -1, 3, 4, 5, -1}};        // row block, first pass: hooks 0..5
constexpr HookMap kMapRowD{{10, 11, 12, -1, 13, 14, 15, -1}};  // ... second pass: hooks 10..15 (6..9: twiddles)
#ifndef CLFA_RES16_PIN_HOOKS
#define CLFA_RES16_PIN_HOOKS 1
#endif
// a hook stays where it is written: without the fences hipcc lets the arithmetic drift around the asm
// statements and the loads end up in clusters of four
template <int G, class H> __device__ __forceinline__ void hook_at(const H &hook) {
  if constexpr (G >= 0 && !std::is_same<H, HookNone>::value) {
    if (CLFA_RES16_PIN_HOOKS) __builtin_amdgcn_sched_barrier(0);
    hook(ic<G>());
    if (CLFA_RES16_PIN_HOOKS) __builtin_amdgcn_sched_barrier(0);
  }
}
// dft16 of fft_device.hpp with eight hook points
struct NoTail {
  __device__ __forceinline__ void operator()() const {}
};
// `tail` runs after hook point P6, ahead of the last two butterflies (res_col_block issues its table lookups there)
template <bool FWD, class H, int P0, int P1, int P2, int P3, int P4, int P5, int P6, int P7, class T = NoTail>
__device__ __forceinline__ void dft16_hp(cpx (&v)[16], const H &hook, const T &tail = T()) {
  bf4<FWD>(v[0], v[4], v[8], v[12]);
  hook_at<P0>(hook);
  bf4<FWD>(v[1], v[5], v[9], v[13]);
  hook_at<P1>(hook);
  bf4<FWD>(v[2], v[6], v[10], v[14]);
  hook_at<P2>(hook);
  bf4<FWD>(v[3], v[7], v[11], v[15]);
  ctw2<FWD>(v[4 + 1], kC16, kS16, v[4 + 2], kC8, kC8);
  hook_at<P3>(hook);
  ctw2<FWD>(v[4 + 3], kS16, kC16, v[8 + 1], kC8, kC8);
  ctw2<FWD>(v[8 + 3], -kC8, kC8, v[12 + 1], kS16, kC16);
  hook_at<P4>(hook);
  ctw2<FWD>(v[12 + 2], -kC8, kC8, v[12 + 3], -kC16, -kS16);
  cpx x[16];
#pragma unroll
  for (int t = 0; t < 16; t++) x[t] = v[t];
  bf4<FWD>(x[0], x[1], x[2], x[3]);
  hook_at<P5>(hook);
  bf4<FWD>(x[4], x[5], x[6], x[7]);
  hook_at<P6>(hook);
  tail();
  bf4_rot2<FWD>(x[8], x[9], x[10], x[11]);
  hook_at<P7>(hook);
  bf4<FWD>(x[12], x[13], x[14], x[15]);
#pragma unroll
  for (int q0 = 0; q0 < 4; q0++)
#pragma unroll
    for (int q1 = 0; q1 < 4; q1++) v[q0 + 4 * q1] = x[4 * q0 + q1];
}
// The last four butterflies of a row block's second pass with their results written straight into the landing registers
// (v[224 + 2 k] for result k): the block's results are stored from there along the next block, and the 16 moves of
// res_stage() are 3 % of this kernel's VALU instructions — which one wave per SIMD pays in full.  Result k = q0 + 4 q1 of
// butterfly q0 overwrites landing register k only after hook point k has issued its store (hooks 0..13 precede the first
// of these butterflies, 14 follows the first, 15 the second; butterfly q0 writes k = q0, q0 + 4, q0 + 8, q0 + 12).
#define CLFA_PLUS ""
#define CLFA_MINUS " neg_lo:[0,1] neg_hi:[0,1]"
#define CLFA_ROTF " op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"   /* x + (-i) y */
#define CLFA_ROTI " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"   /* x + (+i) y */
#define CLFA_BF4_LAND(M02A, M02B, MR1, MR3)                                                              \
  asm volatile("v_pk_add_f32 %0, %4, %6" M02A "\n\t"                                                     \
               "v_pk_add_f32 %1, %4, %6" M02B "\n\t"                                                     \
               "v_pk_add_f32 %2, %5, %7\n\t"                                                             \
               "v_pk_add_f32 %3, %5, %7" CLFA_MINUS "\n\t"                                               \
               "v_pk_add_f32 v[%c8:%c9], %0, %2\n\t"                                                     \
               "v_pk_add_f32 v[%c10:%c11], %1, %3" MR1 "\n\t"                                            \
               "v_pk_add_f32 v[%c12:%c13], %0, %2" CLFA_MINUS "\n\t"                                     \
               "v_pk_add_f32 v[%c14:%c15], %1, %3" MR3                                                   \
               : "=&v"(s02), "=&v"(d02), "=&v"(s13), "=&v"(d13)                                          \
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(224 + 2 * Q0), "n"(225 + 2 * Q0), "n"(232 + 2 * Q0), \
                 "n"(233 + 2 * Q0), "n"(240 + 2 * Q0), "n"(241 + 2 * Q0), "n"(248 + 2 * Q0), "n"(249 + 2 * Q0))
template <bool FWD, bool ROT2, int Q0> __device__ __forceinline__ void bf4_land(cpx a0, cpx a1, cpx a2, cpx a3) {
  cpx s02, d02, s13, d13;
  if constexpr (FWD && !ROT2) CLFA_BF4_LAND(CLFA_PLUS, CLFA_MINUS, CLFA_ROTF, CLFA_ROTI);
  if constexpr (!FWD && !ROT2) CLFA_BF4_LAND(CLFA_PLUS, CLFA_MINUS, CLFA_ROTI, CLFA_ROTF);
  if constexpr (FWD && ROT2) CLFA_BF4_LAND(CLFA_ROTF, CLFA_ROTI, CLFA_ROTF, CLFA_ROTI);
  if constexpr (!FWD && ROT2) CLFA_BF4_LAND(CLFA_ROTI, CLFA_ROTF, CLFA_ROTI, CLFA_ROTF);
}
#undef CLFA_BF4_LAND
// dft16_hp whose results end in the landing registers (nothing is left in v)
template <bool FWD, class H, int P0, int P1, int P2, int P3, int P4, int P5, int P6, int P7>
__device__ __forceinline__ void dft16_hp_land(cpx (&v)[16], const H &hook) {
  bf4<FWD>(v[0], v[4], v[8], v[12]);
  hook_at<P0>(hook);
  bf4<FWD>(v[1], v[5], v[9], v[13]);
  hook_at<P1>(hook);
  bf4<FWD>(v[2], v[6], v[10], v[14]);
  hook_at<P2>(hook);
  bf4<FWD>(v[3], v[7], v[11], v[15]);
  ctw2<FWD>(v[4 + 1], kC16, kS16, v[4 + 2], kC8, kC8);
  hook_at<P3>(hook);
  ctw2<FWD>(v[4 + 3], kS16, kC16, v[8 + 1], kC8, kC8);
  ctw2<FWD>(v[8 + 3], -kC8, kC8, v[12 + 1], kS16, kC16);
  hook_at<P4>(hook);
  ctw2<FWD>(v[12 + 2], -kC8, kC8, v[12 + 3], -kC16, -kS16);
  static_assert(P4 >= 13 && P5 == 14 && P6 == 15 && P7 < 0, "landing register k is free once hook k has issued its store");
  bf4_land<FWD, false, 0>(v[0], v[1], v[2], v[3]);
  hook_at<P5>(hook);
  bf4_land<FWD, false, 1>(v[4], v[5], v[6], v[7]);
  hook_at<P6>(hook);
  bf4_land<FWD, true, 2>(v[8], v[9], v[10], v[11]);
  bf4_land<FWD, false, 3>(v[12], v[13], v[14], v[15]);
}
#define CLFA_DFT16_H(FWD, v, hook, M) \
  dft16_hp<FWD, decltype(hook), M.p[0], M.p[1], M.p[2], M.p[3], M.p[4], M.p[5], M.p[6], M.p[7]>(v, hook)
#define CLFA_DFT16_HT(FWD, v, hook, M, tail) \
  dft16_hp<FWD, decltype(hook), M.p[0], M.p[1], M.p[2], M.p[3], M.p[4], M.p[5], M.p[6], M.p[7], decltype(tail)>(v, hook, tail)

// build switches of the second passes (A/B; the library's choice is the default)
#ifndef CLFA_TW_AHEAD
#define CLFA_TW_AHEAD 1     // the second pass's twiddle rows are read one group (two b128) ahead of their use: the hook
#endif                      // points are scheduling fences, and a read issued right before its use costs one wave per
                            // SIMD the whole LDS latency, three times per block
#ifndef CLFA_FS_AHEAD
#define CLFA_FS_AHEAD 1     // the four-step twiddle lookups are issued before the second pass's last butterflies
#endif
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

// second pass of a block: inputs times W_256^(t j) (row t of the table); hooks H0 .. H0 + 3 after the four groups
template <bool FWD, int H0, bool AHEAD, class H> __device__ __forceinline__ void res_tw_rows(cpx (&v)[16], const ResLane &L, const H &hook) {
  const f4 *pt = reinterpret_cast<const f4 *>(L.tw_row);
  if constexpr (AHEAD) {
    f4 wa = pt[0], wb = pt[1], na = pt[2], nb = pt[3];
    static_for<4>([&](auto G) {
      constexpr int g = decltype(G)::value;
      if constexpr (g == 0) v[1] = cmulc<!FWD>(v[1], mk(wa.z, wa.w));
      else cmulc2<!FWD>(v[4 * g], v[4 * g + 1], v[4 * g], mk(wa.x, wa.y), v[4 * g + 1], mk(wa.z, wa.w));
      cmulc2<!FWD>(v[4 * g + 2], v[4 * g + 3], v[4 * g + 2], mk(wb.x, wb.y), v[4 * g + 3], mk(wb.z, wb.w));
      wa = na;
      wb = nb;
      if constexpr (g < 2) {   // the group after next, issued ahead of the fence
        na = pt[2 * g + 4];
        nb = pt[2 * g + 5];
      }
      hook_at<H0 + g>(hook);
    });
  } else {
    {
      const f4 w = pt[0];
      v[1] = cmulc<!FWD>(v[1], mk(w.z, w.w));
    }
#pragma unroll
    for (int i = 1; i < 8; i++) {
      const f4 w = pt[i];
      cmulc2<!FWD>(v[2 * i], v[2 * i + 1], v[2 * i], mk(w.x, w.y), v[2 * i + 1], mk(w.z, w.w));
      if (i == 1) hook_at<H0>(hook);
      if (i == 3) hook_at<H0 + 1>(hook);
      if (i == 5) hook_at<H0 + 2>(hook);
      if (i == 7) hook_at<H0 + 3>(hook);
    }
  }
}

// ---- phase 1: one column block ------------------------------------------------------------------
// v: rows t + 16 e of column n2 = 16 cb + c (already loaded) -> o[e] = Z[t + 16 e][n2]
template <bool FWD, int PROBE = 0, class H = HookNone>
__device__ __forceinline__ void res_col_block(cpx (&v)[16], const ResLane &L, int cb, const cpx *s_tab, cpx *s_x,
                                              const H &hook = H()) {
  if constexpr (!(PROBE & kProbeNoMath)) CLFA_DFT16_H(FWD, v, hook, kMapColA);
  if constexpr (!(PROBE & kProbeNoXchg)) {
  res_barrier<PROBE>();   // the previous block's readers are done with the exchange buffer
  {
    f4 *pw = reinterpret_cast<f4 *>(L.xa_w);
#pragma unroll
    for (int i = 0; i < 8; i++) pw[i] = pack2(v[2 * i], v[2 * i + 1]);
  }
  res_barrier<PROBE>();
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = L.xa_r[16 * e];
  }
  if constexpr (PROBE & kProbeNoMath) return;
  // second pass: inputs times W_256^(t j) (row t of the table), then the butterflies
  res_tw_rows<FWD, 4, CLFA_TW_AHEAD>(v, L, hook);
  // four-step twiddles W_N^(n2 (t + 16 e)) = b * s^e,  b = W_N^(n2 t),  s = W_4096^n2
  const int n2 = cb * 16 + L.c;
  const int m = n2 * L.t;   // < 4096
  const cpx *ps = s_tab + kTabS + n2;
  cpx blo, bhi, s1, s2, s4, s8;
  auto lookups = [&]() {
    blo = s_tab[kTabLo + (m & 255)], bhi = s_tab[kTabHi + (m >> 8)];
    s1 = ps[0], s2 = ps[256], s4 = ps[512], s8 = ps[768];
  };
  if constexpr (CLFA_FS_AHEAD) {
    // the six reads go out ahead of the pass's last two butterflies (a fence keeps them there)
    auto tail = [&]() {
      lookups();
      __builtin_amdgcn_sched_barrier(0);
    };
    CLFA_DFT16_HT(FWD, v, hook, kMapColB, tail);
  } else {
    CLFA_DFT16_H(FWD, v, hook, kMapColB);
    lookups();
  }
  const cpx b = cmul(blo, bhi);
  // product tree in halves of four (T_r = b s^r, U_r = T_r s^8), two products per statement
  cpx T[4], U[4];
  T[0] = b;
  cmulc2(T[1], T[2], b, s1, b, s2);
  cmulc2(T[3], U[0], T[1], s2, b, s8);
  hook_at<12>(hook);
  cmulc2(U[1], U[2], T[1], s8, T[2], s8);
  U[3] = cmul(T[3], s8);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    cmulc2<!FWD>(v[r], v[r + 8], v[r], T[r], v[r + 8], U[r]);
    if (r == 1) hook_at<13>(hook);
  }
  cmulc2(T[0], T[1], T[0], s4, T[1], s4);
  cmulc2(T[2], T[3], T[2], s4, T[3], s4);
  hook_at<14>(hook);
  cmulc2(U[0], U[1], T[0], s8, T[1], s8);
  cmulc2(U[2], U[3], T[2], s8, T[3], s8);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    cmulc2<!FWD>(v[r + 4], v[r + 12], v[r + 4], T[r], v[r + 12], U[r]);
    if (r == 1) hook_at<15>(hook);
  }
}

// o[e] -> keep[e][cb]
// LAST (column block 15): the slot's element goes straight to its landing register (the rest of that row
// block is on its way there, HookSlot)
template <int PROBE = 0, bool LAST = false, bool INV = false>
__device__ __forceinline__ void res_deposit(const cpx (&o)[16], const ResLane &L, int cb, f32x32 (&K)[kVgprBlk],
                                            __amdgpu_buffer_rsrc_t slot) {
  {
    cpx *ps = reinterpret_cast<cpx *>(L.spill + cb * 8);
#pragma unroll
    for (int e = 0; e < kLdsBlk; e++) ps[e * 16] = o[e];
  }
  // the one row block that does not fit the CU: [cb][lane] in the workgroup's 32 KiB slot (L2-resident)
  if constexpr (LAST) res_stage_one15(o[kLdsBlk]);
  else if constexpr (!(PROBE & kProbeNoSlot))
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[kLdsBlk]), slot, L.slot_off, cb * 2048, 0);
#pragma unroll
  for (int j = 0; j < kVgprBlk; j++) {
    K[j][2 * cb] = o[kVgprFirst + j].x;
    K[j][2 * cb + 1] = o[kVgprFirst + j].y;
  }
  using S8 = std::make_integer_sequence<int, kAgprBlk>;
  switch (cb) {
#define CLFA_C(c) case c: acc_deposit<c, INV>(o, S8()); break;
    CLFA_C(0) CLFA_C(1) CLFA_C(2) CLFA_C(3) CLFA_C(4) CLFA_C(5) CLFA_C(6) CLFA_C(7)
    CLFA_C(8) CLFA_C(9) CLFA_C(10) CLFA_C(11) CLFA_C(12) CLFA_C(13) CLFA_C(14)
#undef CLFA_C
    default: acc_deposit<15, INV>(o, S8()); break;
  }
}

// keep[rb][e] -> v[e]
template <int RB, bool INV = false> __device__ __forceinline__ void res_fetch_static(cpx (&v)[16], const ResLane &L, const f32x32 (&K)[kVgprBlk]) {
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
    acc_fetch<RB - kAgprFirst, INV>(v, std::make_integer_sequence<int, 16>());
  }
}
// ---- phase 2: one row block ---------------------------------------------------------------------
// v[e] = Z[16 rb + t][c + 16 e] -> X[16 rb + c + 256 (t + 16 e)] left in v[e] (lane = row c, k2 = t + 16 e)
// LAND: the results go straight into the landing registers (dft16_hp_land) instead of v
// TWA: the twiddle rows read a group ahead (CLFA_TW_AHEAD; the packed real forward kernel has no registers for it)
template <bool FWD, int PROBE = 0, bool LAND = false, bool TWA = true, class H = HookNone>
__device__ __forceinline__ void res_row_block(cpx (&v)[16], const ResLane &L, const H &hook = H()) {
  if constexpr (!(PROBE & kProbeNoMath)) CLFA_DFT16_H(FWD, v, hook, kMapRowC);
  if constexpr (!(PROBE & kProbeNoXchg)) {
  res_barrier<PROBE>();
  {
    f4 *pw = reinterpret_cast<f4 *>(L.xb_w);
#pragma unroll
    for (int i = 0; i < 8; i++) pw[i] = pack2(v[2 * i], v[2 * i + 1]);
  }
  res_barrier<PROBE>();
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = L.xb_r[18 * e];
  }
  if constexpr (PROBE & kProbeNoMath) return;
  res_tw_rows<FWD, 6, CLFA_TW_AHEAD && TWA>(v, L, hook);
  if constexpr (LAND) dft16_hp_land<FWD, H, kMapRowD.p[0], kMapRowD.p[1], kMapRowD.p[2], kMapRowD.p[3], kMapRowD.p[4], kMapRowD.p[5], kMapRowD.p[6], kMapRowD.p[7]>(v, hook);
  else CLFA_DFT16_H(FWD, v, hook, kMapRowD);
}

// phase 1, one column block: wait for its data (N younger asm loads), take it out of its landing zone
// (ZC >= 0: AGPR columns ZC, ZC + 1; ZC < 0: v[224:255]), transform it with the loads of the block two
// ahead riding along (-> AGPR columns NZ, NZ + 1, or the landing registers for NZ < 0; none if !LOAD)
template <bool FWD, int PROBE, int ZC, int NZ, bool LOAD, int N = 16, bool LAST = false>
__device__ __forceinline__ void res_phase1_block(cpx (&v)[16], const ResLane &L, const cpx *x, int cb, int rot,
                                                 const int (&so)[16], f32x32 (&K)[kVgprBlk], __amdgpu_buffer_rsrc_t slot,
                                                 const cpx *s_tab, cpx *s_x) {
  if constexpr (!(PROBE & (kProbeNoLoad | kProbeNoWait))) res_wait_vm<N>();
  if constexpr (ZC >= 0) acc_fetch_raw<ZC>(v, std::make_integer_sequence<int, 8>());
  else res_land_fetch(v);
  if constexpr (LOAD && !(PROBE & kProbeNoLoad)) {
    const __amdgpu_buffer_rsrc_t r = res_rsrc(x + ((cb + 2 + rot) & 15) * 16);
    if constexpr (NZ >= 0) res_col_block<FWD, PROBE>(v, L, cb, s_tab, s_x, HookAcc<NZ>{r, L.voff, so});
    else res_col_block<FWD, PROBE>(v, L, cb, s_tab, s_x, HookLand{r, L.voff, so});
  } else if constexpr (LAST && !(PROBE & kProbeNoSlot)) {
    res_col_block<FWD, PROBE>(v, L, cb, s_tab, s_x, HookSlot{slot, L.slot_off, so});
  } else {
    res_col_block<FWD, PROBE>(v, L, cb, s_tab, s_x);
  }
  res_deposit<PROBE, LAST>(v, L, cb, K, slot);
}


// ---- packed real transforms of size 2 kN = 131072, forward (R2C): the reference's `conv` pair map (cl_fft.cpp:178-191)
// inside phase 2, so that the transform still crosses HBM once.
//
// The map combines bins i and M - i (M = kN).  With i = 16 q + c + 256 (t + 16 e) (row block q, lane (c, t), register
// e) the partner is 16 (15 - q) + (16 - c) + 256 (15 - t) + 4096 (15 - e): row block 15 - q, and — if that block is
// worked through MIRRORED lane maps (the lane reads row (16 - c) mod 16 of the exchange and the first pass's output
// 15 - t) — the same lane's register 15 - e.  So phase 2 runs the row blocks in pairs A = q, B = 15 - q (q = 0..7),
// B mirrored, and the map is register-to-register in every lane with c != 0.  The lanes c = 0 (rows k1 = 16 rb) pair
// one block further: A_q's with B_(q-1)'s (still in the same lane), which is why B's results stay parked for one
// more block (in the AGPR row that block 15 has left free) and are completed there before their stores are issued;
// rows k1 = 0 (in A_0) and k1 = 128 (in B_7) pair within themselves, across the 16 lanes c = 0, through 2 KiB of LDS.
// Pair twiddles W_2M^i = W_2M^(16 q + c) * W_512^t * W_32^e: two lookups (the first 256 entries of the plan's w2 table
// and every 256th) and compile-time constants.  The map's 1/2 rides on the 1/N of the table (r2c_pair_prescaled).

// build switches of the packed real variants (A/B and debugging; the library's choice is the default)
#ifndef CLFA_C2R_WAIT
#define CLFA_C2R_WAIT 1      // 0: every counted wait of the two packed real variants is vmcnt(0) (tools/check_waits.py)
#endif
#ifndef CLFA_C2R_KEEP_A
#define CLFA_C2R_KEEP_A 1    // its natural loads cached as well: per 1024 transforms all streaming 0.265 ms, the mirrored
#endif                       // loads cached 0.253, all cached 0.245 (profiles/rfft131072_fused_r04.txt)
constexpr bool kC2rWait = CLFA_C2R_WAIT;
constexpr bool kC2rKeepA = CLFA_C2R_KEEP_A;
constexpr int kTabPair = kTabSize;   // [W_2M^k, k < 256 | W_512^t, t < 16]
constexpr int kTabSizeR = kTabSize + 272;
constexpr int kParkAcc = 224;        // B' results parked in a[224:255] (keep row 15's registers, fetched first)
constexpr int kSlotAcc = 192;        // the slot's row block lands in a[192:223] (keep row 14's, free after pair 1)


template <int BASE> __device__ __forceinline__ void acc_fetch_flat(cpx (&v)[16]) {
  static_for<16>([&](auto E) { v[decltype(E)::value] = mk(acc_read<BASE + 2 * decltype(E)::value>(), acc_read<BASE + 2 * decltype(E)::value + 1>()); });
}
template <int BASE> __device__ __forceinline__ void acc_park_flat(const cpx (&v)[16]) {
  static_for<16>([&](auto E) { acc_write<BASE + 2 * decltype(E)::value>(v[decltype(E)::value].x); acc_write<BASE + 2 * decltype(E)::value + 1>(v[decltype(E)::value].y); });
}
struct HookStoreAcc {   // the parked B' block, out of a[kParkAcc ...]
  __amdgpu_buffer_rsrc_t r;
  int voff;
  const int (&so)[16];
  template <int K> __device__ __forceinline__ void operator()(ic<K>) const {
    if constexpr (K == 0)
      asm volatile("s_nop 4\n\tbuffer_store_dwordx2 a[%c2:%c3], %0, %1, %4 offen" CLFA_STNT ::"v"(voff), "s"(r), "n"(kParkAcc + 2 * K), "n"(kParkAcc + 1 + 2 * K), "s"(so[K]) : "memory");
    else
      asm volatile("buffer_store_dwordx2 a[%c2:%c3], %0, %1, %4 offen" CLFA_STNT ::"v"(voff), "s"(r), "n"(kParkAcc + 2 * K), "n"(kParkAcc + 1 + 2 * K), "s"(so[K]) : "memory");
  }
};
struct HookSlotAcc {   // the global slot's row block (all 16 columns) -> a[kSlotAcc ...]; sc1 as in HookSlot
  __amdgpu_buffer_rsrc_t r;
  int voff;
  const int (&so)[16];
  template <int K> __device__ __forceinline__ void operator()(ic<K>) const {
    int off;
    asm volatile("s_lshr_b32 %0, %3, 4\n\ts_nop 4\n\tbuffer_load_dwordx2 a[%c4:%c5], %1, %2, %0 offen sc1"
                 : "=&s"(off)
                 : "v"(voff), "s"(r), "s"(so[K]), "n"(kSlotAcc + 2 * K), "n"(kSlotAcc + 1 + 2 * K)
                 : "memory", "scc");
  }
};
template <class A, class B> struct Hook2 {
  A a;
  B b;
  template <int K> __device__ __forceinline__ void operator()(ic<K> k) const {
    a(k);
    b(k);
  }
};
// all 16 stores of the landing registers at once (the last A' block of a transform)
__device__ __forceinline__ void res_store_land(__amdgpu_buffer_rsrc_t r, int voff, const int (&so)[16]) {
  const HookStore h{r, voff, so};
  static_for<16>([&](auto Kc) { h(Kc); });
}
// a block's 16 stores out of compiler registers, with the pinned row offsets (the builtin of res_store() would make
// hipcc hold a second copy of the 15 offsets in SGPRs, which this variant of the kernel does not have)
__device__ __forceinline__ void res_store_so(const cpx (&v)[16], __amdgpu_buffer_rsrc_t r, int voff, const int (&so)[16]) {
  // (the descriptor's SGPRs may be fresh from SALU: the wait states sit in the first store's own statement)
  asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, %3 offen" CLFA_STNT ::"v"(v[0]), "v"(voff), "s"(r), "s"(so[0]) : "memory");
#pragma unroll
  for (int e = 1; e < 16; e++)
    asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" CLFA_STNT ::"v"(v[e]), "v"(voff), "s"(r), "s"(so[e]) : "memory");
}
// W_2M^i of the lane's register e: base * W_32^e
template <int E, bool FWD = true> __device__ __forceinline__ cpx pair_tw_e(cpx base) {
  if constexpr (E == 0) return base;
  else return ctw<FWD>(base, kC32[E], kS32[E]);
}
// r2c_pair_prescaled (fft_device.hpp) in six packed instructions: the conjugations and the rotation by i ride on the
// operand modifiers.  One wave per SIMD pays for every instruction in full, so the map is written out here.
__device__ __forceinline__ void r2c_pair6(cpx a, cpx b, cpx w, cpx &oi, cpx &oj) {
  cpx e, r, x, y;
  asm("v_pk_add_f32 %0, %4, %5 neg_hi:[0,1]\n\t"                                 // e = a + conj(b)
      "v_pk_add_f32 %1, %4, %5 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"    // r = i (conj(b) - a) = (a.y + b.y, b.x - a.x)
      "v_pk_mul_f32 %2, %6, %1 op_sel_hi:[0,1]\n\t"                              // x = w r
      "v_pk_fma_f32 %2, %6, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_add_f32 %3, %0, %2 neg_lo:[0,1] neg_hi:[1,0]\n\t"                    // y = conj(e - x)
      "v_pk_add_f32 %2, %0, %2"                                                   // x = e + x
      : "=&v"(e), "=&v"(r), "=&v"(x), "=&v"(y)
      : "v"(a), "v"(b), "v"(w));
  oi = x;
  oj = y;
}
// ... with the A value in (and the result back into) the landing register pair of register E: no moves
template <int E> __device__ __forceinline__ void r2c_pair6_land(cpx &b, cpx w) {
  cpx e, r, y;
  asm volatile("v_pk_add_f32 %0, v[%c5:%c6], %3 neg_hi:[0,1]\n\t"
               "v_pk_add_f32 %1, v[%c5:%c6], %3 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
               "v_pk_mul_f32 v[%c5:%c6], %4, %1 op_sel_hi:[0,1]\n\t"
               "v_pk_fma_f32 v[%c5:%c6], %4, %1, v[%c5:%c6] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
               "v_pk_add_f32 %2, %0, v[%c5:%c6] neg_lo:[0,1] neg_hi:[1,0]\n\t"
               "v_pk_add_f32 v[%c5:%c6], %0, v[%c5:%c6]"
               : "=&v"(e), "=&v"(r), "=&v"(y)
               : "v"(b), "v"(w), "n"(224 + 2 * E), "n"(225 + 2 * E));
  b = y;
}
// lanes c != 0, end of pair q: the A block (raw, in the landing registers) against the B block (raw, in v); A' stays
// in the landing registers, B' in v
__device__ __forceinline__ void res_pair_map(cpx (&v)[16], cpx base) {
  static_for<16>([&](auto E) {
    constexpr int e = decltype(E)::value;
    r2c_pair6_land<e>(v[15 - e], pair_tw_e<e>(base));
    // (every two pairs a fence: hipcc otherwise piles up all 16 twiddles and spills — into AGPRs, this kernel's own)
    if (e & 1) __builtin_amdgcn_sched_barrier(0);
  });
}
// lanes c = 0, pair q >= 1, after the A block: its rows k1 = 16 q pair with the previous B block's k1 = 16 (16 - q),
// parked raw in a[kParkAcc ...] of these lanes; both are finished here
__device__ __forceinline__ void res_pair_patch_c0(cpx (&v)[16], cpx base) {
  static_for<16>([&](auto E) {
    constexpr int e = decltype(E)::value, pe = kParkAcc + 2 * (15 - e);
    const cpx bq = mk(acc_read<pe>(), acc_read<pe + 1>());
    cpx oi, oj;
    r2c_pair6(v[e], bq, pair_tw_e<e>(base), oi, oj);
    v[e] = oi;
    asm volatile("" : "+v"(v[e]));
    acc_write<pe>(oj.x);
    acc_write<pe + 1>(oj.y);
    if (e & 1) __builtin_amdgcn_sched_barrier(0);
  });
}
// lanes c = 0 of a block whose row pairs within itself (k1 = 0: natural lanes, k2 = t + 16 e; k1 = 128: mirrored
// lanes, k2 = (15 - t) + 16 e): the partners are in other lanes c = 0 -> through s_c0[t][e].  Every lane computes
// its own 16 results (each pair twice, by both of its lanes).  ROW0 has the reference's two exceptions: bin 0 packs
// DC / Nyquist, bin M/2 is left as the complex transform made it (cl_fft.cpp:178-191 starts at i = 1 and never
// reaches M/2).  Called by all lanes (barrier inside).
template <bool ROW0> __device__ __forceinline__ void res_pair_self_row(cpx (&v)[16], int c, int t, cpx base, cpx *s_c0) {
  if (c == 0) {
#pragma unroll
    for (int e = 0; e < 16; e++) s_c0[t * 16 + e] = v[e];
  }
  __syncthreads();
  if (c == 0) {
    static_for<16>([&](auto E) {
      constexpr int e = decltype(E)::value;
      int idx;
      if constexpr (ROW0) {
        const int k2 = (256 - (t + 16 * e)) & 255;
        idx = (k2 & 15) * 16 + (k2 >> 4);
      } else {
        idx = (15 - t) * 16 + (15 - e);
      }
      const cpx ci = v[e], zp = s_c0[idx];
      cpx oi, oj;
      r2c_pair6(ci, zp, pair_tw_e<e>(base), oi, oj);
      if constexpr (ROW0 && e == 0) {
        if (t == 0) oi = mk(ci.x + ci.y, ci.x - ci.y);
      }
      if constexpr (ROW0 && e == 8) {
        if (t == 0) oi = cscale(ci, 2.0f);
      }
      v[e] = oi;
      asm volatile("" : "+v"(v[e]));
      if (e & 1) __builtin_amdgcn_sched_barrier(0);
    });
  }
}


// ---- packed real transforms of size 2 kN = 131072, inverse (C2R): the reference's `iconv` pair map (cl_fft.cpp:192-205)
// inside phase 1.
//
// The input index has the structure of the forward kernel's output: i = 16 cb + c + 256 (t + 16 e) pairs with column
// 16 - c of column block 15 - cb, row (15 - t) + 16 (15 - e).  Phase 1 takes the column blocks in pairs A = q natural,
// B = 15 - q loaded through mirrored lanes (q = 7 .. 0): the map is register-to-register (A in the lane's registers,
// B in the landing registers), and B un-mirrors itself in its own exchange — the lane writes its first-pass results to
// column slot 16 - c at position 16 (15 - t) and the natural lanes read them.  The lanes c = 0 have loaded column 0 of
// block 16 - q (the partners of their A column): it belongs to the NEXT pair's B block, so its first-pass results go
// to a copy buffer that the next B block's lanes c = 0 read instead of slot 0 (loads do not mind the detour; the
// forward kernel's stores did, profiles/rfft131072_fused_r04.txt).  Columns 0 (in A_0) and 128 (block 8's, loaded
// separately before the first pair) pair within themselves across the 16 lanes c = 0.
// The map's two factors 1/2 ride on the four-step twiddle table (x 0.5); the untouched bins 0 and M/2 are doubled.
// Both blocks of the next pair are loaded along the A block (two loads per hook point, into the two AGPR zones; the
// B data then move to the landing registers); the last pair's B block comes through the landing registers directly.
__device__ __forceinline__ void c2r_pair6(cpx a, cpx b, cpx w, cpx &oi, cpx &oj) {
  cpx e, r, x, y;
  asm("v_pk_add_f32 %0, %4, %5 neg_hi:[0,1]\n\t"                                                // e = a + conj(b)
      "v_pk_add_f32 %1, %4, %5 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[1,1] neg_hi:[0,1]\n\t"      // r = i (a - conj(b))
      "v_pk_mul_f32 %2, %6, %1 op_sel_hi:[0,1]\n\t"                                             // x = w r
      "v_pk_fma_f32 %2, %6, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_add_f32 %3, %0, %2 neg_lo:[0,1] neg_hi:[1,0]\n\t"                                   // y = conj(e - x)
      "v_pk_add_f32 %2, %0, %2"                                                                // x = e + x
      : "=&v"(e), "=&v"(r), "=&v"(x), "=&v"(y)
      : "v"(a), "v"(b), "v"(w));
  oi = x;
  oj = y;
}
// ... with the B value in (and its result back into) the landing register pair VB
template <int VB> __device__ __forceinline__ void c2r_pair6_land(cpx &a, cpx w) {
  cpx e, r, x;
  asm volatile("v_pk_add_f32 %0, %3, v[%c5:%c6] neg_hi:[0,1]\n\t"
               "v_pk_add_f32 %1, %3, v[%c5:%c6] op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[1,1] neg_hi:[0,1]\n\t"
               "v_pk_mul_f32 %2, %4, %1 op_sel_hi:[0,1]\n\t"
               "v_pk_fma_f32 %2, %4, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
               "v_pk_add_f32 v[%c5:%c6], %0, %2 neg_lo:[0,1] neg_hi:[1,0]\n\t"
               "v_pk_add_f32 %2, %0, %2"
               : "=&v"(e), "=&v"(r), "=&v"(x)
               : "v"(a), "v"(w), "n"(VB), "n"(VB + 1));
  a = x;
}
// start of a pair: the A block (raw, in v) against the B block (raw, in the landing registers), both finished in place
__device__ __forceinline__ void res_unpair_map(cpx (&v)[16], cpx base) {
  static_for<16>([&](auto E) {
    constexpr int e = decltype(E)::value;
    c2r_pair6_land<224 + 2 * (15 - e)>(v[e], pair_tw_e<e, false>(base));
    if (e & 1) __builtin_amdgcn_sched_barrier(0);
  });
}
// lanes c = 0 of a column that pairs within itself (COL0: n2 = 0, natural lanes, row n1 = t + 16 e pairs with 256 - n1,
// rows 0 and 128 are the reference's untouched bins 0 and M/2; else n2 = 128, mirrored lanes, n1 = (15 - t) + 16 e pairs
// with 255 - n1).  Every lane computes its own 16 values.  Called by all lanes (barrier inside).
template <bool COL0> __device__ __forceinline__ void res_unpair_self_col(cpx (&v)[16], int c, int t, cpx base, cpx *s_c0) {
  if (c == 0) {
#pragma unroll
    for (int e = 0; e < 16; e++) s_c0[t * 16 + e] = v[e];
  }
  __syncthreads();
  if (c == 0) {
    static_for<16>([&](auto E) {
      constexpr int e = decltype(E)::value;
      int idx;
      if constexpr (COL0) {
        const int n1 = (256 - (t + 16 * e)) & 255;
        idx = (n1 & 15) * 16 + (n1 >> 4);
      } else {
        idx = (15 - t) * 16 + (15 - e);
      }
      const cpx ci = v[e], zp = s_c0[idx];
      cpx oi, oj;
      c2r_pair6(ci, zp, pair_tw_e<e, false>(base), oi, oj);
      if constexpr (COL0 && e == 0) {
        if (t == 0) oi = mk(2.0f * (ci.x + ci.y), 2.0f * (ci.x - ci.y));
      }
      if constexpr (COL0 && e == 8) {
        if (t == 0) oi = cscale(ci, 2.0f);
      }
      v[e] = oi;
      asm volatile("" : "+v"(v[e]));
      if (e & 1) __builtin_amdgcn_sched_barrier(0);
    });
  }
}
// zone Z1 (AGPR columns 12, 13) -> landing registers
__device__ __forceinline__ void res_zone1_to_land() {
  static_for<16>([&](auto E) {
    constexpr int e = decltype(E)::value, src = 32 * (e & 7) + 2 * (kZone1 + (e >> 3));
    asm volatile("v_accvgpr_read_b32 v[%c0], a[%c2]\n\tv_accvgpr_read_b32 v[%c1], a[%c3]" ::"n"(224 + 2 * e), "n"(225 + 2 * e), "n"(src), "n"(src + 1));
  });
}
template <class H> __device__ __forceinline__ void res_issue_all(const H &h) {
  static_for<16>([&](auto Kc) { h(Kc); });
}

}  // namespace

#ifdef CLFA_RES16_PROBE
// probe only: all workgroups meet (monotonic counter; bounded spin)
__device__ __forceinline__ void res_probe_grid_sync(unsigned long long *dbg, unsigned &epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned *cnt = reinterpret_cast<unsigned *>(dbg + 1024);
    epoch += gridDim.x;
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; spin < 2000000; spin++) {
      if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch) break;
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}
#endif

// slots: one 32 KiB slot per workgroup (the single row block that does not fit the CU)
// R2C: packed real transforms of size 2 kN, forward — the same transform with the reference's pair map inside phase 2
// (above); w2_g = the plan's pair twiddles W_2M^i (cl_fft.cpp:233-238), M entries
// C2R: ... inverse — the pair map inside phase 1 (above)
// CLFA_DS_SINGLE: hipcc's load / store optimiser pairs the exchanges' sixteen ds_read_b64 into eight ds_read2_b64, which the
// LDS serves per 16 contiguous lanes over 32 banks at half the rate (MI355X_MICROARCH.md, LDS table): in the layouts above —
// made for the single reads, 2 x 32 lanes over 64 banks — every access is then a 2-way conflict, four times the LDS cycles
// (rocprofv3 round 4: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.29).  The pass is off for this kernel.
#ifndef CLFA_DS_SINGLE
#define CLFA_DS_SINGLE 1
#endif
#if CLFA_DS_SINGLE && defined(__HIP_DEVICE_COMPILE__)
#define CLFA_RES16_TARGET __attribute__((target("no-load-store-opt")))
#else
#define CLFA_RES16_TARGET
#endif
template <bool FWD, bool SCALE, int PROBE = 0, bool R2C = false, bool C2R = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(224))) CLFA_RES16_TARGET void k_fft_res16(const cpx *data, cpx *out, cpx *__restrict__ slots,
                                                   const cpx *__restrict__ tabs_g, long batch,
                                                   unsigned long long *__restrict__ dbg = nullptr,
                                                   const cpx *__restrict__ w2_g = nullptr) {
#ifndef CLFA_RES16_PROBE
  static_assert(PROBE == 0, "the timing experiments exist only in tools/res16_probe.hip (CLFA_RES16_PROBE)");
#endif
  static_assert(!R2C || (FWD && SCALE && PROBE == 0 && !C2R), "the fused forward pair map");
  static_assert(!C2R || (!FWD && !SCALE && PROBE == 0), "the fused inverse pair map");
  __shared__ __attribute__((aligned(16))) cpx s_tab[(R2C || C2R) ? kTabSizeR : kTabSize];
  __shared__ __attribute__((aligned(16))) cpx s_x[kXSize];
  __shared__ __attribute__((aligned(16))) char s_spill[256 * kSpillStride];
  __shared__ __attribute__((aligned(16))) cpx s_c0[(R2C || C2R) ? 256 : 1];
  __shared__ __attribute__((aligned(16))) cpx s_col0[C2R ? 2 * kXA : 1];   // first-pass results of the B blocks' column 0
  acc_claim_all();
  const int tid = threadIdx.x;
  for (int i = tid; i < kTabSize; i += 256) {
    cpx w = tabs_g[i];
    // exact: powers of two (R2C: the pair map's 1/2 as well)
    if (SCALE && i >= kTabLo && i < kTabHi) w = cscale(w, R2C ? 0.5f / (float)kN : 1.0f / (float)kN);
    if (C2R && i >= kTabLo && i < kTabHi) w = cscale(w, 0.5f);   // the inverse map's two factors 1/2
    s_tab[i] = w;
  }
  if constexpr (R2C || C2R) {
    s_tab[kTabPair + tid] = w2_g[tid];
    if (tid < 16) s_tab[kTabPair + 256 + tid] = w2_g[256 * tid];
  }
  // The lane's addresses are recomputed from an opaque copy of the lane index wherever a block
  // starts: as loop invariants they would cost ~13 VGPRs for the whole kernel, which hipcc then
  // parks in AGPRs (this kernel's own)
  auto lane = [&]() {
    int l = tid;
    asm volatile("" : "+v"(l));
    ResLane L;
    L.c = l & 15;
    L.t = l >> 4;
    L.voff = L.t * 2048 + L.c * 8;
    L.xa_w = s_x + L.c * kXA + 16 * L.t;
    L.xa_r = s_x + L.c * kXA + L.t;
    L.xb_w = s_x + L.t * kXB + 18 * L.c;
    L.xb_r = s_x + L.c * kXB + L.t;
    L.tw_row = s_tab + kTabTw + 16 * L.t;
    L.spill = s_spill + l * kSpillStride;
    L.slot_off = l * 8;
    return L;
  };
  // R2C, the B block of a pair: row (16 - c) mod 16 of the exchange, first-pass output 15 - t (and the stores follow)
  [[maybe_unused]] auto lane_m = [&]() {
    ResLane L = lane();
    const int cm = (16 - L.c) & 15, tm = 15 - L.t;
    L.voff = tm * 2048 + cm * 8;
    L.xb_r = s_x + cm * kXB + tm;
    L.tw_row = s_tab + kTabTw + 16 * tm;
    return L;
  };
  // C2R, the B block of a pair: first-pass results go to column slot 16 - c at position 16 (15 - t); the lanes c = 0
  // hold the NEXT pair's column 0 (-> copy buffer `save_w`) and read this block's from the previous pair's (`save_r`)
  [[maybe_unused]] auto lane_b = [&](cpx *save_w, const cpx *save_r) {
    ResLane L = lane();
    const int cm = 16 - L.c, tm = 15 - L.t;
    L.xa_w = (L.c ? s_x + cm * kXA : save_w) + 16 * tm;
    if (L.c == 0) L.xa_r = save_r + L.t;
    return L;
  };
  const int rot = (PROBE & kProbeRotate) ? (int)(blockIdx.x & 15) : 0;   // probe only: 0 in the library
  int so[16];   // row offsets e * 32 KiB of the asm loads, pinned in SGPRs (never rematerialised next to a load)
#pragma unroll
  for (int e = 0; e < 16; e++) {
    so[e] = e * 32768;
    asm volatile("" : "+s"(so[e]));
  }
  const __amdgpu_buffer_rsrc_t slot = res_rsrc(slots + (long)blockIdx.x * 4096);
  __syncthreads();

  f32x32 K[kVgprBlk];
#pragma unroll
  for (int j = 0; j < kVgprBlk; j++) K[j] = 0.f;
  cpx v[16];
  long b = xcd_first(blockIdx.x, gridDim.x);   // XCD-compact assignment (fft_device.hpp)
#ifdef CLFA_RES16_PROBE
  // probe only: other transform -> workgroup assignments (dbg[3000]; needs batch % (16 * gridDim.x) == 0): 0 the library's,
  // 5 workgroup i takes i, i + G, ... (rounds 1-3), 1 chunks (i * per + k), 2 / 3 mixtures, 4 a bit permutation
  const int mapmode = dbg ? (int)dbg[3000] : 0;
  const long per = batch / gridDim.x;
  long kk = 0;
  auto bmap = [&](long k) -> long {
    const long i = blockIdx.x;
    switch (mapmode) {
      case 1: return i * per + k;
      case 2: return (i & 15) + 16 * k + 16 * per * (i >> 4);
      case 3: return (i >> 4) + (gridDim.x >> 4) * k + (gridDim.x >> 4) * per * (i & 15);
      case 4: {   // bit j of b = bit dbg[3001 + j] of (i | k << 8), 12 bits (256 workgroups x 16 transforms)
        const long v = i | (k << 8);
        long r = 0;
        for (int j = 0; j < 12; j++) r |= ((v >> dbg[3001 + j]) & 1) << j;
        return __builtin_amdgcn_readfirstlane((int)r);
      }
      case 5: return i + k * gridDim.x;
      default: return xcd_first(blockIdx.x, gridDim.x) + k * gridDim.x;   // the library's
    }
  };
  b = bmap(0);
#endif
#ifdef CLFA_RES16_PROBE
  unsigned long long clk1 = 0, clk2 = 0;
  unsigned epoch = 0;   // probe only (the host zeroes the counter before the launch)
  unsigned long long slot_next = 0, slot_p1 = 0, slot_p2 = 0;   // probe only
  if constexpr (PROBE & kProbeSlots) {
    slot_next = __builtin_amdgcn_s_memrealtime();
    slot_p1 = dbg[2048];
    slot_p2 = dbg[2049];
  }
#endif
  // blocks 0 and 1 of the first transform
  if constexpr (C2R) {
    // the first pair: block 7 -> Z0, block 8 mirrored -> landing registers, column 128 (lanes c = 0, mirrored rows) -> Z1
    const ResLane L0 = lane();
    const cpx *x0 = data + b * (long)kN;
    res_issue_all(HookAcc<kZone0, kC2rKeepA>{res_rsrc(x0 + 7 * 16), L0.voff, so});
    res_issue_all(HookLandT<true>{res_rsrc(x0 + 8 * 16), (15 - L0.t) * 2048 + (16 - L0.c) * 8, so});
    res_issue_all(HookAcc<kZone1, true>{res_rsrc(x0 + 8 * 16), (15 - L0.t) * 2048 + L0.c * 8, so});
  } else if constexpr (!(PROBE & kProbeNoLoad)) {
    const ResLane L0 = lane();
    res_load_acc<kZone0>(data + b * (long)kN + (rot & 15) * 16, L0.voff);
    res_load_land(data + b * (long)kN + ((1 + rot) & 15) * 16, L0.voff);
  }
#pragma unroll 1
#ifdef CLFA_RES16_PROBE
  for (; kk < per; kk++, b = bmap(kk)) {
#else
  for (; b < batch; b += gridDim.x) {
#endif
    const cpx *x = data + b * (long)kN;
    cpx *y = out + b * (long)kN;   // out == data: in place
#ifdef CLFA_RES16_PROBE
    unsigned long long t0 = 0;
    if constexpr (PROBE & kProbeStamps) t0 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (C2R) {
      // ---- phase 1 of the packed real inverse: pairs of column blocks, the reference's iconv map first (see above)
      res_wait_vm<0>();
      {   // column 128 (lanes c = 0, out of Z1): pairs within itself; its first-pass results -> copy buffer 0
        const ResLane L = lane();
        acc_fetch_raw<kZone1>(v, std::make_integer_sequence<int, 8>());
        const cpx bm = cmul(s_tab[kTabPair + 128], s_tab[kTabPair + 256 + 15 - L.t]);
        res_unpair_self_col<false>(v, L.c, L.t, bm, s_c0);
        if (L.c == 0) {
          const HookNone none;
          CLFA_DFT16_H(false, v, none, kMapColA);
          f4 *pw = reinterpret_cast<f4 *>(s_col0 + 16 * (15 - L.t));
#pragma unroll
          for (int i = 0; i < 8; i++) pw[i] = pack2(v[2 * i], v[2 * i + 1]);
        }
      }
      // MODE 0: pairs 0..5, 1: pair 6 (the last pair's B block comes through the landing registers), 2: pair 7
      auto pair_step = [&](auto MODE, int p) {
        constexpr int mode = decltype(MODE)::value;
        const int q = 7 - p;
        {
          const ResLane L = lane();
          // loads of this pair: issued along the previous pair's A block (pair 7's B block: along pair 6's B block);
          // younger than they are only the slot stores of the blocks since (pair 0: waited for above)
          if (mode == 2) res_wait_vm<kC2rWait ? 1 : 0>();
          else if (p > 0) res_wait_vm<kC2rWait ? 2 : 0>();
          acc_fetch_raw<kZone0>(v, std::make_integer_sequence<int, 8>());
          if (mode == 1 || (mode == 0 && p > 0)) res_zone1_to_land();
          const cpx base = cmul(s_tab[kTabPair + 16 * q + L.c], s_tab[kTabPair + 256 + L.t]);
          if constexpr (mode == 2) {
            res_unpair_self_col<true>(v, L.c, L.t, base, s_c0);   // column 0
            if (L.c != 0) res_unpair_map(v, base);
          } else {
            res_unpair_map(v, base);
          }
          const int voff_m = (15 - L.t) * 2048 + (16 - L.c) * 8;
          if constexpr (mode == 0) {
            res_col_block<false, 0>(v, L, q, s_tab, s_x,
                                    Hook2<HookAcc<kZone0, kC2rKeepA>, HookAcc<kZone1, true>>{HookAcc<kZone0, kC2rKeepA>{res_rsrc(x + (q - 1) * 16), L.voff, so},
                                                                                 HookAcc<kZone1, true>{res_rsrc(x + (16 - q) * 16), voff_m, so}});
          } else if constexpr (mode == 1) {
            res_col_block<false, 0>(v, L, q, s_tab, s_x, HookAcc<kZone0, kC2rKeepA>{res_rsrc(x + (q - 1) * 16), L.voff, so});
          } else {
            res_col_block<false, 0>(v, L, q, s_tab, s_x);
          }
          res_deposit<0, false, true>(v, L, q, K, slot);
        }
        {
          const ResLane L = lane_b(s_col0 + ((p + 1) & 1) * kXA, s_col0 + (p & 1) * kXA);
          res_land_fetch(v);
          if constexpr (mode == 0) {
            res_col_block<false, 0>(v, L, 15 - q, s_tab, s_x);
          } else if constexpr (mode == 1) {
            // block 15 mirrored (its lanes c = 0 have no partner column to fetch: out of the buffer's range)
            const int voff_m = L.c ? (15 - L.t) * 2048 + (16 - L.c) * 8 : (int)0x80000000;
            res_col_block<false, 0>(v, L, 15 - q, s_tab, s_x, HookLandT<true>{res_rsrc(x + 15 * 16), voff_m, so});
          } else {
            res_col_block<false, 0>(v, L, 15 - q, s_tab, s_x, HookSlot{slot, L.slot_off, so});
          }
          res_deposit<0, mode == 2, true>(v, L, 15 - q, K, slot);
        }
      };
#pragma unroll 1
      for (int p = 0; p < 6; p++) pair_step(ic<0>(), p);
      pair_step(ic<1>(), 6);
      pair_step(ic<2>(), 7);
    } else {
    // ---- phase 1: four column blocks per round (landing zones Z0, v[224:255], Z1, v[224:255]); on entry
    // block 0 is in (or on its way to) Z0 and block 1 on its way to the landing registers
#pragma unroll 1
    for (int cb = 0; cb < 12; cb += 4) {
      res_phase1_block<FWD, PROBE, kZone0, kZone1, true>(v, lane(), x, cb, rot, so, K, slot, s_tab, s_x);
      res_phase1_block<FWD, PROBE, -1, -1, true>(v, lane(), x, cb + 1, rot, so, K, slot, s_tab, s_x);
      res_phase1_block<FWD, PROBE, kZone1, kZone0, true>(v, lane(), x, cb + 2, rot, so, K, slot, s_tab, s_x);
      res_phase1_block<FWD, PROBE, -1, -1, true>(v, lane(), x, cb + 3, rot, so, K, slot, s_tab, s_x);
    }
    res_phase1_block<FWD, PROBE, kZone0, kZone1, true>(v, lane(), x, 12, rot, so, K, slot, s_tab, s_x);
    res_phase1_block<FWD, PROBE, -1, -1, true>(v, lane(), x, 13, rot, so, K, slot, s_tab, s_x);
    res_phase1_block<FWD, PROBE, kZone1, -1, false>(v, lane(), x, 14, rot, so, K, slot, s_tab, s_x);
    if constexpr (R2C) res_phase1_block<FWD, PROBE, -1, -1, false, 0, false>(v, lane(), x, 15, rot, so, K, slot, s_tab, s_x);
    else res_phase1_block<FWD, PROBE, -1, -1, false, 0, true>(v, lane(), x, 15, rot, so, K, slot, s_tab, s_x);
    }   // !C2R
#ifdef CLFA_RES16_PROBE
    if constexpr (PROBE & kProbeStamps) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      clk1 += t1 - t0;
      t0 = t1;
    }
    if constexpr (PROBE & kProbeGridSync) {
      res_probe_grid_sync(dbg, epoch);
      t0 = __builtin_amdgcn_s_memtime();
    }
    if constexpr (PROBE & kProbeSlots) {
      slot_next += slot_p1;
      while ((long long)(__builtin_amdgcn_s_memrealtime() - slot_next) < 0) __builtin_amdgcn_s_sleep(8);
      if constexpr (PROBE & kProbeStamps) t0 = __builtin_amdgcn_s_memtime();
    }
#endif
    // ---- phase 2: row blocks in the order slot (its data are in the landing registers by now), AGPR
    // (the accumulation file is then free for the next transform's block 0), VGPR, LDS.  A block's results
    // are parked in the landing registers and stored while the next block is computed.
    long bn = b + gridDim.x;   // next transform (clamped: its first loads are issued unconditionally)
    bn = bn < batch ? bn : batch - 1;
#ifdef CLFA_RES16_PROBE
    bn = bmap(kk + 1 < per ? kk + 1 : kk);
#endif
    const cpx *xn = data + bn * (long)kN;
    if constexpr (R2C) {
      // ---- phase 2 of the packed real transform: pairs of row blocks (A = q natural, B = 15 - q mirrored), see above
#pragma unroll 1
      for (int q = 0; q < 8; q++) {
        // stores into a zero-length buffer are dropped: pair 0 has nothing to store yet
        const unsigned live = q ? 0x7fffffffu : 0u;
        cpx base;
        {
          const ResLane L = lane();
          switch (q) {
            case 0: res_fetch_static<0>(v, L, K); break;
            case 1: res_fetch_static<1>(v, L, K); break;
            case 2: res_fetch_static<2>(v, L, K); break;
            case 3:   // the slot's block: loaded along block A_2; the 16 stores of block B_2 are younger
              res_wait_vm<kC2rWait ? 16 : 0>();
              acc_fetch_flat<kSlotAcc>(v);
              break;
            case 4: res_fetch_static<4>(v, L, K); break;
            case 5: res_fetch_static<5>(v, L, K); break;
            case 6: res_fetch_static<6>(v, L, K); break;
            default: res_fetch_static<7>(v, L, K); break;
          }
          // ... with the stores of A'_(q-1) (landing registers) riding along
          const __amdgpu_buffer_rsrc_t ra =
              __builtin_amdgcn_make_buffer_rsrc(y + (q - 1) * 16, 0, live, 0x00020000);
          if (q == 2) res_row_block<FWD, PROBE, false, false>(v, L, Hook2<HookStore, HookSlotAcc>{HookStore{ra, L.voff, so}, HookSlotAcc{slot, L.slot_off, so}});
          else res_row_block<FWD, PROBE, false, false>(v, L, HookStore{ra, L.voff, so});
          base = cmul(s_tab[kTabPair + 16 * q + L.c], s_tab[kTabPair + 256 + L.t]);
          if (q == 0) {
            res_pair_self_row<true>(v, L.c, L.t, base, s_c0);
          } else if (L.c == 0) {
            res_pair_patch_c0(v, base);
          }
          res_stage(v);   // A_q: raw in the lanes c != 0, finished in the lanes c = 0
        }
        {
          const ResLane L = lane_m();
          switch (q) {
            case 0: res_fetch_static<15>(v, L, K); break;
            case 1: res_fetch_static<14>(v, L, K); break;
            case 2: res_fetch_static<13>(v, L, K); break;
            case 3: res_fetch_static<12>(v, L, K); break;
            case 4: res_fetch_static<11>(v, L, K); break;
            case 5: res_fetch_static<10>(v, L, K); break;
            case 6: res_fetch_static<9>(v, L, K); break;
            default: res_fetch_static<8>(v, L, K); break;
          }
          // ... with the stores of B'_(q-1) (parked in the accumulation registers, finished by the patch above)
          const __amdgpu_buffer_rsrc_t rb =
              __builtin_amdgcn_make_buffer_rsrc(y + (16 - q) * 16, 0, live, 0x00020000);
          res_row_block<FWD, PROBE, false, false>(v, L, HookStoreAcc{rb, L.voff, so});
          if (L.c != 0) res_pair_map(v, base);
          if (q == 7) {
            const cpx bm = cmul(s_tab[kTabPair + 128], s_tab[kTabPair + 256 + 15 - L.t]);
            res_pair_self_row<false>(v, L.c, L.t, bm, s_c0);
          } else {
            acc_park_flat<kParkAcc>(v);   // B'_q (its lanes c = 0 still raw)
          }
        }
      }
      {
        const ResLane L = lane(), Lm = lane_m();
        res_store_land(res_rsrc(y + 7 * 16), L.voff, so);
        res_store_so(v, res_rsrc(y + 8 * 16), Lm.voff, so);
        // the next transform's blocks 0 and 1
        res_load_acc<kZone0>(xn, L.voff);
        res_load_land(xn + 16, L.voff);
      }
    } else {
    constexpr bool kLand = PROBE == 0;   // row blocks leave their results in the landing registers themselves (dft16_hp_land)
    {
      const ResLane L = lane();
      if constexpr (!(PROBE & kProbeNoSlot)) res_wait_vm<0>();
      res_land_fetch(v);
      res_row_block<FWD, PROBE, kLand>(v, L);
      if constexpr (!kLand) res_stage(v);
    }
    int rb_prev = kLdsBlk;
#pragma unroll 1
    for (int it = (PROBE & kProbePhase1Only) ? 14 : 1; it < 15; it++) {
      const ResLane L = lane();
      int rb;
      switch (it) {
#define CLFA_C(k, r) case k: res_fetch_static<r, C2R>(v, L, K); rb = r; break;
        CLFA_C(1, 8) CLFA_C(2, 9) CLFA_C(3, 10) CLFA_C(4, 11) CLFA_C(5, 12) CLFA_C(6, 13) CLFA_C(7, 14) CLFA_C(8, 15)
        CLFA_C(9, 4) CLFA_C(10, 5) CLFA_C(11, 6) CLFA_C(12, 7) CLFA_C(13, 0)
#undef CLFA_C
        default: res_fetch_static<1>(v, L, K); rb = 1; break;
      }
      if constexpr (!(PROBE & kProbeNoLoad)) {
        if (it == 13) {
          if constexpr (C2R) res_issue_all(HookAcc<kZone0, kC2rKeepA>{res_rsrc(xn + 7 * 16), L.voff, so});   // the next transform's block 7
          else res_load_acc<kZone0>(xn + (rot & 15) * 16, L.voff);
        }
      }
      if constexpr (!(PROBE & kProbeNoStore)) {
        res_row_block<FWD, PROBE, kLand>(v, L, HookStore{res_rsrc(y + ((rb_prev + rot) & 15) * 16), L.voff, so});
      } else {
        res_row_block<FWD, PROBE>(v, L);
      }
      if constexpr (!kLand) res_stage(v);
      rb_prev = rb;
    }
    {
      const ResLane L = lane();
      res_fetch_static<2>(v, L, K);
      if constexpr (!(PROBE & kProbeNoStore)) {
        res_row_block<FWD, PROBE>(v, L, HookStore{res_rsrc(y + ((rb_prev + rot) & 15) * 16), L.voff, so});
      } else {
        res_row_block<FWD, PROBE>(v, L);
      }
      res_store<PROBE>(v, res_rsrc(y + ((2 + rot) & 15) * 16), L.voff);
      // the next transform's block 1 -> landing registers (after this block's parked stores have been issued)
      if constexpr (C2R) {   // the next transform's block 8 (mirrored) and column 128
        res_issue_all(HookLandT<true>{res_rsrc(xn + 8 * 16), (15 - L.t) * 2048 + (16 - L.c) * 8, so});
        res_issue_all(HookAcc<kZone1, true>{res_rsrc(xn + 8 * 16), (15 - L.t) * 2048 + L.c * 8, so});
      } else if constexpr (!(PROBE & kProbeNoLoad)) res_load_land(xn + ((1 + rot) & 15) * 16, L.voff);
    }
    }   // !R2C
#ifdef CLFA_RES16_PROBE
    if constexpr (PROBE & kProbePack) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const __amdgpu_buffer_rsrc_t ry = res_rsrc(y);
      int lt = tid;
      asm volatile("" : "+v"(lt));
#pragma unroll 1
      for (int k = 0; k < 128; k += 8) {
        u32x2 a[8], bq[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int i = lt + 256 * (k + u);
          a[u] = __builtin_amdgcn_raw_buffer_load_b64(ry, i * 8, 0, 2);
          bq[u] = __builtin_amdgcn_raw_buffer_load_b64(ry, (kN - 1 - i) * 8, 0, 2);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int i = lt + 256 * (k + u);
          const cpx ci = __builtin_bit_cast(cpx, a[u]), cj = __builtin_bit_cast(cpx, bq[u]);
          cpx oi, oj;
          r2c_pair(ci, cj, s_tab[(i >> 7) & 255], oi, oj);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, oi), ry, i * 8, 0, 2);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, oj), ry, (kN - 1 - i) * 8, 0, 2);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (PROBE & kProbeStamps) clk2 += __builtin_amdgcn_s_memtime() - t0;
    if constexpr (PROBE & kProbeGridSync) res_probe_grid_sync(dbg, epoch);
    if constexpr (PROBE & kProbeSlots) {
      slot_next += slot_p2;
      while ((long long)(__builtin_amdgcn_s_memrealtime() - slot_next) < 0) __builtin_amdgcn_s_sleep(8);
    }
#endif
  }
#ifdef CLFA_RES16_PROBE
  if constexpr (PROBE & kProbeStamps) {
    if (tid == 0) {
      dbg[2 * blockIdx.x] = clk1;
      dbg[2 * blockIdx.x + 1] = clk2;
    }
  }
#endif
}

hipError_t launch_fft_res16(bool fwd, bool scale, const cpx *data, cpx *out, cpx *slots, const cpx *tabs, long batch,
                            const DeviceInfo &di, hipStream_t s) {
  if (batch <= 0) return hipSuccess;
  const int grid = (int)(batch < di.num_cus ? batch : di.num_cus);
  if (fwd && scale) hipLaunchKernelGGL((k_fft_res16<true, true>), dim3(grid), dim3(256), 0, s, data, out, slots, tabs, batch);
  else if (fwd) hipLaunchKernelGGL((k_fft_res16<true, false>), dim3(grid), dim3(256), 0, s, data, out, slots, tabs, batch);
  else if (!scale) hipLaunchKernelGGL((k_fft_res16<false, false>), dim3(grid), dim3(256), 0, s, data, out, slots, tabs, batch);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_crfft_res16(const cpx *data, cpx *out, cpx *slots, const cpx *tabs, const cpx *w2, long batch,
                              const DeviceInfo &di, hipStream_t s) {
  if (batch <= 0) return hipSuccess;
  const int grid = (int)(batch < di.num_cus ? batch : di.num_cus);
  hipLaunchKernelGGL((k_fft_res16<false, false, 0, false, true>), dim3(grid), dim3(256), 0, s, data, out, slots, tabs, batch,
                     (unsigned long long *)nullptr, w2);
  return hipGetLastError();
}

hipError_t launch_rfft_res16(const cpx *data, cpx *out, cpx *slots, const cpx *tabs, const cpx *w2, long batch,
                             const DeviceInfo &di, hipStream_t s) {
  if (batch <= 0) return hipSuccess;
  const int grid = (int)(batch < di.num_cus ? batch : di.num_cus);
  hipLaunchKernelGGL((k_fft_res16<true, true, 0, true>), dim3(grid), dim3(256), 0, s, data, out, slots, tabs, batch,
                     (unsigned long long *)nullptr, w2);
  return hipGetLastError();
}

}  // namespace clfa
