// internal.hpp — launcher prototypes shared by the kernel files and the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_device.hpp"

namespace clfa {

enum FftMode { MODE_C2C = 0, MODE_R2C = 1, MODE_C2R = 2 };

// Largest complex length the single-workgroup LDS kernel handles; above it the four-step kernel
// (two phases, the intermediate in LDS + registers + a small global scratch) takes over.
constexpr int kLdsMaxLog = 13;
// n = 8192 uses lane-addressed twiddle tables instead of the half table (fft_device.hpp, LaneTab13):
// [W_256^(j t), j, t < 16 | W_4096^(2^k j mod 4096), k < 4, j < 256 | W_8192^t, t < 512]; the first
// kLane13Lds entries live in LDS
constexpr bool kLdsTwoLevel(int logn) { return logn >= 13; }
constexpr int kLane13Lds = 1280, kLane13Size = 1792;
// ... in LDS the rows of the 16 x 16 part are kRow16Stride entries apart: the sixteen lanes of a ds_read_b128 group read
// sixteen DIFFERENT rows (row = tid & 15); 16 entries = 32 dwords apart they would fall on two groups of four banks (an
// 8-way conflict, 32 LDS cycles per read — rocprofv3 round 4: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.41-0.46 for
// config 3's kernel); 18 entries = 36 dwords apart they cover the 64 banks exactly once (MI355X_MICROARCH.md, LDS table)
constexpr int kRow16Stride = kRow16StrideDev;   // (fft_device.hpp: CLFA_ROW16_STRIDE)
static_assert(kRow16Stride >= 16 && kRow16Stride % 2 == 0, "rows stay 16-byte aligned");
constexpr int kRow16Lds = 16 * kRow16Stride;            // the s256 part follows
constexpr int kLaneLds = kRow16Lds + (kLane13Lds - 256);   // entries of the LDS copy
// LDS index of entry i < kLane13Lds of the global table
__host__ __device__ constexpr int lane_lds_index(int i) { return i < 256 ? (i >> 4) * kRow16Stride + (i & 15) : i + (kRow16Lds - 256); }
// the 16384-point chains of k_rfft_2x<14> (packed real size 65536: 1024 lanes, passes 16 x 16 x 16 x 4, fft_device.hpp
// LaneTab14): the same LDS part, then [W_16384^t | W_16384^(2 t) | W_16384^(3 t)], t < 1024
constexpr int kLds14Log = 14, kLane14Size = kLane13Lds + 3 * 1024;
constexpr int kMaxLog = 16;  // reference int32 index bound, cl_fft.cpp:32

struct FftTables {      // all device pointers, owned by the plan
  const cpx *half = nullptr;   // W_n^k, k < n/2, forward sign (LDS path; n = complex length)
  const cpx *w2 = nullptr;     // r2c table (cl_fft.cpp:233-238), sign of the plan's direction, m entries
  const cpx *four = nullptr;   // four-step tables: [half N1 | half N2 | lo | hi]
  const cpx *res16 = nullptr;  // n = 65536 only: tables of the resident kernel (kRes16TabSize entries)
};

struct DeviceInfo {
  int device = 0;
  int num_cus = 256;
};

// single-workgroup LDS FFT, n = 2^logn <= 2^kLdsMaxLog.  mode selects the fused
// r2c epilogue / c2r prologue.  scale: multiply by 1/n (forward plans).
// out_off (here and below): the results go to data + out_off complex elements — 0 = in place, otherwise a destination
// that does not overlap the source (clfa_fft_exec_dev_oop); the source is only read
hipError_t launch_fft_lds(int logn, bool fwd, int mode, bool scale, cpx *data, const FftTables &t,
                          long batch, const DeviceInfo &di, hipStream_t s, long out_off = 0);
const char *name_fft_lds(int logn, bool fwd, int mode);
// packed real size 65536 (n = 32768): two runs of the 16384-point machinery per transform, radix-2 step and pair
// map in registers; t.half = the n = 16384 lane tables (kLane14Size), t.w2 = the plan's r2c table (n entries)
hipError_t launch_rfft_lds15(bool fwd, cpx *data, const FftTables &t, long batch, const DeviceInfo &di, hipStream_t s,
                             long out_off = 0);
// packed real size 32768 the same way on two 8192-point runs (two 512-lane workgroups per CU); t.half = the n = 8192
// lane tables (kLane13Size), t.w2 = the plan's r2c table
hipError_t launch_rfft_2x13(bool fwd, cpx *data, const FftTables &t, long batch, const DeviceInfo &di, hipStream_t s,
                            long out_off = 0);
// complex n = 16384 as two 8192-point runs + a radix-2 step in registers; t.half = the n = 8192 lane tables
// (kLane13Size) followed by W_16384^t, t < 512
hipError_t launch_cfft_2x13(bool fwd, bool scale, cpx *data, const FftTables &t, long batch, const DeviceInfo &di,
                            hipStream_t s, long out_off = 0);

// four-step FFT, n = 2^logn in (2^kLdsMaxLog, 2^kMaxLog]; scratch = fourstep_grid() * n complex
// (n = 65536 with more than a few transforms runs the resident kernel below and uses the first
// kRes16SlotBytes * grid bytes of the scratch as its slots)
int fourstep_grid(const DeviceInfo &di);
hipError_t launch_fft_4step(int logn, bool fwd, bool scale, cpx *data, cpx *scratch, const FftTables &t, long batch,
                            const DeviceInfo &di, hipStream_t s, long out_off = 0);
const char *name_fft_4step(int logn);
int fourstep_split(int logn, int *logn1, int *logn2, int *loglo);
// n = 65536, the whole intermediate resident on the CU (fft_resident.hip): one HBM pass, no scratch.
// tabs: kRes16TabSize entries, forward sign: [W_256^(t j), t, j < 16 | W_n^k, k < 256 | W_256^k, k < 256 |
// W_4096^(m k mod 4096), m = 1, 2, 4, 8, k < 256]
constexpr int kRes16TabSize = 1792;
// slots: 32 KiB per workgroup, grid = min(batch, CUs) workgroups (kRes16SlotBytes each)
constexpr size_t kRes16SlotBytes = 32768;
// out == data: in place (what every plan does); out != data: out of place (measured in tools/res16_probe.hip)
hipError_t launch_fft_res16(bool fwd, bool scale, const cpx *data, cpx *out, cpx *slots, const cpx *tabs, long batch,
                            const DeviceInfo &di, hipStream_t s);
// packed real transforms of size 131072, forward: the same kernel with the reference's pair map inside its second
// phase (w2 = the plan's pair twiddles, 65536 entries) — one HBM pass instead of the transform + k_r2c_pack
hipError_t launch_rfft_res16(const cpx *data, cpx *out, cpx *slots, const cpx *tabs, const cpx *w2, long batch,
                             const DeviceInfo &di, hipStream_t s);
// ... and the inverse: the reference's iconv map inside the first phase (w2 = the inverse plan's pair twiddles)
hipError_t launch_crfft_res16(const cpx *data, cpx *out, cpx *slots, const cpx *tabs, const cpx *w2, long batch,
                              const DeviceInfo &di, hipStream_t s);
// n = 2^17 .. 2^kBigMaxLog (extension: the reference overflows above 65536): columns + rows + transpose
constexpr int kBigMaxLog = 24;
struct BigGeom {
  int logn, logn1, logn2;   // n = 2^logn1 x 2^logn2
  bool two_run;                    // two-pass sizes: 1024-point columns / rows as two 512-point runs (two workgroups per CU)
};
int big_split(int logn, BigGeom *g);
// bigtabs: [half N1 | W_n^k, k < 128 | W_n^(128 k), k < 128 | W_n^(16384 k), k < n / 16384]; sub: tables of the 2^logn2 row transform; scratch holds `batch`
// transforms (batch <= 65535), scratch2 the row transform's own workspace (logn2 > kLdsMaxLog)
// (out: where the last pass writes; data itself is only read)
hipError_t launch_fft_big(const BigGeom &g, bool fwd, bool scale, cpx *data, cpx *out, cpx *scratch, cpx *scratch2,
                          const cpx *bigtabs, const FftTables &sub, long batch, const DeviceInfo &di, hipStream_t s);

// stand-alone pack / unpack (reference kernels conv / iconv) for M above the LDS path
hipError_t launch_r2c_pack(cpx *data, const cpx *w2, int m, long batch, hipStream_t s, long out_off = 0);
hipError_t launch_c2r_unpack(cpx *data, const cpx *w2, int m, long batch, hipStream_t s, long out_off = 0);

// arbitrary (non power-of-two) complex lengths, an extension: Bluestein's algorithm around two m-point
// power-of-two transforms, m >= 2 n - 1 (fft_kernels.hip); n up to kBlueMaxN
constexpr int kBlueMaxN = 1 << 22;
bool blue_lds_ok(int m);
hipError_t launch_blue_lds(int m, const cpx *x, cpx *y, const cpx *w, const cpx *bt, const cpx *tab, int n, float scale,
                           long batch, const DeviceInfo &di, hipStream_t s);
hipError_t launch_blue_pre(const cpx *x, const cpx *w, cpx *a, int n, int m, long batch, hipStream_t s);
hipError_t launch_blue_mul(cpx *a, const cpx *bt, int m, long batch, hipStream_t s);
hipError_t launch_blue_post(const cpx *a, const cpx *w, cpx *x, int n, int m, float scale, long batch, hipStream_t s);

// reference `reorder` kernel as an op
hipError_t launch_reorder(cpx *out, const cpx *in, int logn, long batch, hipStream_t s);

// ---- partitioned convolution --------------------------------------------------
struct PconvGeom {
  int logb;      // log2(bins), bins = pts
  int bins;
  int nparts;
  int channels;
};
// input block (channels x pts floats) -> zero-padded 2*pts real FFT -> packed
// spectrum frame `frame` of ring (channels x nparts x bins complex).  Unscaled,
// reference pack (cl_conv_kernels.h:46-85).
// in_b / ring_b / frame_b: optional second input of a time-varying block, transformed by the same launch
hipError_t launch_pconv_forward(const PconvGeom &g, const float *in, long in_stride, cpx *ring, int frame,
                                const cpx *half, const cpx *w2f, hipStream_t s, const float *in_b = nullptr,
                                cpx *ring_b = nullptr, int frame_b = 0);
// acc = sum_p A[(wp+p)%nparts] (.) B[p]  (cl_conv_kernels.h:102-118), acc: channels x bins complex
int pconv_mac_split(const PconvGeom &g);   // partial accumulators the MAC writes (acc must hold that many)
// reduce = false leaves the pconv_mac_split() partial sums for launch_pconv_inverse(nsplit) to add up
hipError_t launch_pconv_mac(const PconvGeom &g, const cpx *ringA, const cpx *ringB, int wp, cpx *acc,
                            hipStream_t s, bool reduce = true);
// acc -> c2r -> inverse FFT -> overlap-add (cl_conv_kernels.h:87-100,120-124); out channels x pts,
// tail channels x pts (unscaled second half kept for the next block)
hipError_t launch_pconv_inverse(const PconvGeom &g, const cpx *acc, float *tail, float *out,
                                const cpx *half, const cpx *w2i, hipStream_t s, int nsplit = 1);
// one launch per block (forward + MAC + inverse in one workgroup per channel); used when
// one launch per block for a FEW channels (conv_kernels.hip, k_pconv_coop): 2^logs bin slices x sparts segments of
// the partition axis per channel (logs = -1: the kernel does not apply); xacc: channels x sparts x bins complex
// (hand-over of the accumulator slices), counters: one zero-initialised unsigned per channel (returned to zero by
// every launch)
struct PconvCoop {
  int logs, sparts;
};
PconvCoop pconv_coop_plan(const PconvGeom &g, const DeviceInfo &di);
hipError_t launch_pconv_coop(const PconvGeom &g, PconvCoop c, const float *in1, const float *in2, cpx *ringA, cpx *ringB,
                             float *tail, float *out, int frame1, int frame2, int wp, const cpx *half, const cpx *w2f,
                             const cpx *w2i, cpx *xacc, unsigned *counters, int num_cus, hipStream_t s);
// pconv_fused_ok(): bins 512..4096 and enough channels to fill the chip
bool pconv_fused_ok(const PconvGeom &g, const DeviceInfo &di);
hipError_t launch_pconv_fused(const PconvGeom &g, const float *in1, const float *in2, cpx *ringA, cpx *ringB,
                              float *tail, float *out, int frame1, int frame2, int wp, const cpx *half,
                              const cpx *w2f, const cpx *w2i, hipStream_t s, bool deep = false);   // deep: fewer channels than CUs
constexpr int kPconvMaxLogBins = 15;   // pts up to 32768 (the reference harness' largest, csound/tests.py:13)
// ends of the composed chain used when bins exceed the LDS FFT sizes
hipError_t launch_pconv_pad(const float *in, long in_stride, cpx *work, int bins, int channels, hipStream_t s);
hipError_t launch_pconv_olap(const float *work, float *tail, float *out, int bins, int channels, hipStream_t s);

// ---- direct convolution ----------------------------------------------------------
struct DconvPlan {
  int C;    // taps per workgroup
  int G;    // chunks of the tap axis (grid x)
  int VB;   // output blocks (grid y): block y takes the tiles of 64 outputs y, y + VB, ...
};
DconvPlan dconv_plan(int irsize, int vsize);
// one block: out[0..vsize) from the rings as they stand with in1 (and in2) written at wp; files the block in the rings.
// part: G x vsize floats, counters: VB zeroed words (both only touched when G > 1).  out must not overlap in1 / in2.
hipError_t launch_dconv_block(const DconvPlan &pl, float *out, const float *in1, const float *in2, float *del, float *coefs,
                              float *part, unsigned *counter, int irsize, int vsize, int wp, int num_cus, hipStream_t s);

}  // namespace clfa
