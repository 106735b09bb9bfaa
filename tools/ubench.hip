// dev tool (not product): the yardsticks the FFT kernels are judged against, measured on the box.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench [mem|seg|issue|all]
//
// mem   : read / write / copy bandwidth with single-instruction 16-byte accesses (plain and
//         non-temporal), UNR loads in flight per lane, >= 100 timed launches after a warm-up that is
//         long enough to get past the clock ramp.  This replaces the round-1 membench copy numbers
//         (one float4 in flight, nt split into four dword accesses).
// seg   : the four-step FFT's global access shape without any arithmetic: 256 x 256 matrices of
//         8-byte elements, a workgroup reads column blocks (SEG bytes of every row, rows 2 KiB apart)
//         and writes column blocks of another matrix; SEG = 128 / 256 / 512 / 2048 bytes, 8- or
//         16-byte accesses per lane.  Says what segment width costs at the HBM.
// issue : VALU issue cost per wave-instruction (v_fma_f32, v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32,
//         v_accvgpr_write/read) at 1, 2 and 4 waves per SIMD, and an LDS exchange loop
//         (16 x ds_write_b64 + barrier + 16 x ds_read_b64) at 4 and 8 waves per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <vector>

#define CK(x)                                                         \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                        \
    }                                                                 \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ f4 ld16(const f4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st16(f4 *p, f4 v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <bool NT> __device__ __forceinline__ f2 ld8(const f2 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st8(f2 *p, f2 v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ---------------------------------------------------------------- mem
// a workgroup walks tiles of 256 * UNR float4; all UNR loads are issued before the first store
template <int UNR, bool NT> __global__ __launch_bounds__(256) void k_copy(f4 *__restrict__ dst, const f4 *__restrict__ src, long tiles) {
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const f4 *s = src + t * (256 * UNR) + threadIdx.x;
    f4 *d = dst + t * (256 * UNR) + threadIdx.x;
    f4 r[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) r[u] = ld16<NT>(s + u * 256);
#pragma unroll
    for (int u = 0; u < UNR; u++) st16<NT>(d + u * 256, r[u]);
  }
}
template <int UNR, bool NT> __global__ __launch_bounds__(256) void k_read(float *__restrict__ sink, const f4 *__restrict__ src, long tiles) {
  f4 acc = 0;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const f4 *s = src + t * (256 * UNR) + threadIdx.x;
    f4 r[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) r[u] = ld16<NT>(s + u * 256);
#pragma unroll
    for (int u = 0; u < UNR; u++) acc += r[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = acc.x;
}
template <int UNR, bool NT> __global__ __launch_bounds__(256) void k_write(f4 *__restrict__ dst, long tiles) {
  f4 v = {1, 2, 3, 4};
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    f4 *d = dst + t * (256 * UNR) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNR; u++) st16<NT>(d + u * 256, v);
  }
}

// ---------------------------------------------------------------- seg
// matrices of 256 rows x 2048 bytes; workgroup b walks matrices b, b + grid, ...; per matrix it reads
// every column block (SEG bytes wide) of `src` and writes the same block of `dst`.  B16: 16 bytes per
// lane (SEG/16 lanes per row segment) else 8 bytes per lane.  256 lanes; per step a lane moves 16
// accesses (8-byte) or 8 (16-byte): 32 KiB per workgroup step like one FFT column block.
template <int SEG, bool B16, bool NT, int MODE>   // MODE 0 copy, 1 read only, 2 write only
__global__ __launch_bounds__(256) void k_seg(char *__restrict__ dst, const char *__restrict__ src, long mats, float *sink) {
  constexpr int AB = B16 ? 16 : 8;            // bytes per access
  constexpr int LPS = SEG / AB;               // lanes per segment
  constexpr int RPI = 256 / LPS;              // rows covered per access instruction
  constexpr int NACC = 32768 / (256 * AB);    // accesses per lane per step (32 KiB per step)
  constexpr int ROWS_STEP = RPI * NACC;       // rows per step
  constexpr int STEPS_ROW = 256 / ROWS_STEP;  // steps to cover the 256 rows of a column block (>= 1)
  static_assert(ROWS_STEP <= 256, "");
  const int l = threadIdx.x, c = l % LPS, r0 = l / LPS;
  float acc = 0;
  for (long m = blockIdx.x; m < mats; m += gridDim.x) {
    const char *s = src + m * 524288;
    char *d = dst + m * 524288;
    for (int cb = 0; cb < 2048 / SEG; cb++) {
      for (int st = 0; st < STEPS_ROW; st++) {
        const long off = (long)(st * ROWS_STEP + r0) * 2048 + cb * SEG + c * AB;
        if constexpr (B16) {
          f4 r[NACC];
#pragma unroll
          for (int e = 0; e < NACC; e++) r[e] = MODE == 2 ? f4{1, 2, 3, 4} : ld16<NT>((const f4 *)(s + off + (long)e * RPI * 2048));
#pragma unroll
          for (int e = 0; e < NACC; e++) {
            if (MODE == 1) acc += r[e].x + r[e].w;
            else st16<NT>((f4 *)(d + off + (long)e * RPI * 2048), r[e]);
          }
        } else {
          f2 r[NACC];
#pragma unroll
          for (int e = 0; e < NACC; e++) r[e] = MODE == 2 ? f2{1, 2} : ld8<NT>((const f2 *)(s + off + (long)e * RPI * 2048));
#pragma unroll
          for (int e = 0; e < NACC; e++) {
            if (MODE == 1) acc += r[e].x + r[e].y;
            else st8<NT>((f2 *)(d + off + (long)e * RPI * 2048), r[e]);
          }
        }
      }
    }
  }
  if (acc == 123.456f) *sink = acc;
}


// GRP adjacent 128-byte column blocks per step, 8 bytes per lane: the accesses of the GRP blocks are
// interleaved instruction by instruction ((cb, e), (cb + 1, e), ...), so adjacent 128-byte segments are
// requested back to back by the same wave — does the memory system treat that like one GRP x 128-byte segment?
template <int GRP, bool NT, int MODE, bool INPLACE, bool FAR = false>
__global__ __launch_bounds__(256) void k_seg_grp(char *dst, const char *src, long mats, float *sink) {
  const int l = threadIdx.x, c = l & 15, r0 = l >> 4;
  float acc = 0;
  for (long m = blockIdx.x; m < mats; m += gridDim.x) {
    const char *s = src + m * 524288;
    char *d = dst + m * 524288;
    constexpr int GS = FAR ? 2048 / GRP : 128;   // FAR: the GRP blocks of a step are 2048 / GRP bytes apart (not adjacent)
    for (int cb = 0; cb < 16 / GRP; cb++) {
      const long off = (long)r0 * 2048 + cb * (FAR ? 128 : 128 * GRP) + c * 8;
      f2 r[16 * GRP];
#pragma unroll
      for (int e = 0; e < 16; e++)
#pragma unroll
        for (int g = 0; g < GRP; g++) r[e * GRP + g] = MODE == 2 ? f2{1, 2} : ld8<NT>((const f2 *)(s + off + g * GS + (long)e * 32768));
#pragma unroll
      for (int e = 0; e < 16; e++)
#pragma unroll
        for (int g = 0; g < GRP; g++) {
          if (MODE == 1) acc += r[e * GRP + g].x + r[e * GRP + g].y;
          else st8<NT>((f2 *)(d + off + g * GS + (long)e * 32768), r[e * GRP + g]);
        }
    }
  }
  if (acc == 123.456f) *sink = acc;
}


// The resident FFT kernel's memory behaviour with its arithmetic as dummy work: one 256-lane workgroup per CU
// (one wave per SIMD), a matrix (256 x 2048 B) at a time, in place.  Phase 1 reads the 16 column blocks, GRP adjacent
// blocks per step with the step's loads issued before WORK dependent v_pk_fma per block (the loads of the next step
// fly behind them: two steps in flight); phase 2 writes 16 column blocks the same way.  GRP = 1: the kernel as it
// is; GRP = 4: four adjacent column blocks per burst (512 contiguous bytes requested together).
template <int GRP> __global__ __launch_bounds__(256) void k_resident_model(char *data, long mats, int work, float *sink) {
  const int l = threadIdx.x, c = l & 15, r0 = l >> 4;
  f2 acc = {1.0f, 0.5f};
  const f2 mm = {0.999f, 1.001f}, cc = {1e-3f, -1e-3f};
  auto busy = [&](int n) {
    for (int i = 0; i < n; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(mm), "v"(cc));
  };
  for (long m = blockIdx.x; m < mats; m += gridDim.x) {
    char *d = data + m * 524288;
    f2 cur[16 * GRP], nxt[16 * GRP];
    auto load = [&](f2 (&r)[16 * GRP], int cb) {
      const long off = (long)r0 * 2048 + cb * 128 + c * 8;
#pragma unroll
      for (int e = 0; e < 16; e++)
#pragma unroll
        for (int g = 0; g < GRP; g++) r[e * GRP + g] = ld8<true>((const f2 *)(d + off + g * 128 + (long)e * 32768));
    };
    load(cur, 0);
#pragma unroll 1
    for (int cb = 0; cb < 16; cb += GRP) {
      load(nxt, (cb + GRP) & 15);   // (the last step re-reads block 0: one extra step of 16 / GRP + 1, same for every GRP ratio-wise)
      busy(work * GRP);
#pragma unroll
      for (int i = 0; i < 16 * GRP; i++) {
        acc += cur[i];
        cur[i] = nxt[i];
      }
    }
#pragma unroll 1
    for (int cb = 0; cb < 16; cb += GRP) {
      busy(work * GRP);
      const long off = (long)r0 * 2048 + cb * 128 + c * 8;
#pragma unroll
      for (int e = 0; e < 16; e++)
#pragma unroll
        for (int g = 0; g < GRP; g++) st8<true>((f2 *)(d + off + g * 128 + (long)e * 32768), acc + f2{(float)e, (float)g});
    }
  }
  if (acc.x == 123.456f) *sink = acc.y;
}


// the same traffic with the phases interleaved inside the CU: step k writes column block k of matrix m and reads
// column block k of matrix m + grid (what a pipeline that drains transform b row block by row block while it fills
// transform b + 1 column block by column block would do; the register file could just hold it: c <= r cells)
__global__ __launch_bounds__(256) void k_resident_model_il(char *data, long mats, int work, float *sink) {
  const int l = threadIdx.x, c = l & 15, r0 = l >> 4;
  f2 acc = {1.0f, 0.5f};
  const f2 mm = {0.999f, 1.001f}, cc = {1e-3f, -1e-3f};
  auto busy = [&](int n) {
    for (int i = 0; i < n; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(mm), "v"(cc));
  };
  f2 cur[16], nxt[16];
  auto load = [&](f2 (&r)[16], long m, int cb) {
    const char *d = data + m * 524288 + (long)r0 * 2048 + cb * 128 + c * 8;
#pragma unroll
    for (int e = 0; e < 16; e++) r[e] = ld8<true>((const f2 *)(d + (long)e * 32768));
  };
  long m = blockIdx.x;
  if (m >= mats) return;
  // prologue: read the first matrix (phase 1 alone)
  load(cur, m, 0);
  for (int cb = 0; cb < 16; cb++) {
    load(nxt, m, (cb + 1) & 15);
    busy(work);
#pragma unroll
    for (int i = 0; i < 16; i++) { acc += cur[i]; cur[i] = nxt[i]; }
  }
  for (; m < mats; m += gridDim.x) {
    const long mn = m + gridDim.x < mats ? m + gridDim.x : m;   // the last round re-reads its own matrix
    load(cur, mn, 0);
#pragma unroll 1
    for (int cb = 0; cb < 16; cb++) {
      load(nxt, mn, (cb + 1) & 15);
      busy(work);          // row block cb of matrix m
      char *d = data + m * 524288 + (long)r0 * 2048 + cb * 128 + c * 8;
#pragma unroll
      for (int e = 0; e < 16; e++) st8<true>((f2 *)(d + (long)e * 32768), acc + f2{(float)e, 1.f});
      busy(work);          // column block cb of matrix m + grid
#pragma unroll
      for (int i = 0; i < 16; i++) { acc += cur[i]; cur[i] = nxt[i]; }
    }
  }
  if (acc.x == 123.456f) *sink = acc.y;
}

// ---------------------------------------------------------------- issue
// OP: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_add_f32, 3 v_pk_mul_f32, 4 v_accvgpr_write+read pair, 5 v_add_f32
template <int OP> __global__ void k_issue(float *out, int iters) {
  f2 a[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = f2{(float)threadIdx.x * 1e-3f + i, 1.0f + i};
  const f2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 8; rep++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
        if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if constexpr (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if constexpr (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if constexpr (OP == 4) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a1" : "+v"(a[i].x) : : "a0", "a1");
        if constexpr (OP == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
      }
    }
  }
  f2 s = a[0];
#pragma unroll
  for (int i = 1; i < 8; i++) s += a[i];
  if (s.x == 123.456f) out[0] = s.y;
}

// LDS exchange: every lane writes 16 x 8 bytes (stride-16 scatter, padded), barrier, reads 16 x 8
// bytes (contiguous, padded), NB barriers per exchange (1 = double-buffered, 2 = single buffer);
// FMAS v_pk_fma per value between the read and the next write.
template <int THREADS, int NB, int FMAS> __global__ __launch_bounds__(THREADS) void k_xchg(float *out, int iters) {
  __shared__ f2 buf[2][4096 + 256 + 16];
  const int l = threadIdx.x % 256, g = threadIdx.x / 256;   // two independent 256-lane groups when THREADS = 512
  __shared__ f2 buf2[THREADS > 256 ? 2 : 1][THREADS > 256 ? 4096 + 256 + 16 : 1];
  f2 v[16];
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = f2{(float)l + e, (float)e};
  const f2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
  for (int it = 0; it < iters; it++) {
    f2 *b = (g == 0 ? buf[NB == 1 ? (it & 1) : 0] : buf2[NB == 1 ? (it & 1) : 0]);
    const int base = l * 16;
    f2 *pw = b + base + (base >> 4);
#pragma unroll
    for (int e = 0; e < 16; e++) pw[e] = v[e];
    __syncthreads();
    const f2 *pr = b + l + (l >> 4);
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = pr[e * (256 + 16)];
    if (NB == 2) __syncthreads();
#pragma unroll
    for (int f = 0; f < FMAS; f++)
#pragma unroll
      for (int e = 0; e < 16; e++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[e]) : "v"(m), "v"(c));
  }
  f2 s = v[0];
#pragma unroll
  for (int e = 1; e < 16; e++) s += v[e];
  if (s.x == 123.456f) out[0] = s.y;
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

  if (all || !strcmp(what, "mem")) {
    const size_t bytes = 2ul << 30;   // 2 GiB each: far beyond the 256 MiB Infinity Cache
    f4 *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    printf("\n[mem] 2 GiB buffers, 16-byte accesses, 30 warm-up + 100 timed launches; TB/s = bytes moved / time\n");
    printf("%-28s %8s %8s %8s\n", "kernel (grid = CUs x wg/CU)", "read", "write", "copy");
#define MEM_ROW(UNR, NT, WPC)                                                                                  \
  {                                                                                                            \
    const long tiles = bytes / (256ul * UNR * 16);                                                             \
    const int grid = cus * WPC;                                                                                \
    float tr = time_launches(30, 100, [&] { hipLaunchKernelGGL((k_read<UNR, NT>), dim3(grid), dim3(256), 0, 0, sink, a, tiles); }); \
    float tw = time_launches(30, 100, [&] { hipLaunchKernelGGL((k_write<UNR, NT>), dim3(grid), dim3(256), 0, 0, b, tiles); });      \
    float tc = time_launches(30, 100, [&] { hipLaunchKernelGGL((k_copy<UNR, NT>), dim3(grid), dim3(256), 0, 0, b, a, tiles); });    \
    printf("unr %d %-5s wg/CU %-2d          %8.2f %8.2f %8.2f\n", UNR, NT ? "nt" : "plain", WPC, bytes / tr * 1e-9,   \
           bytes / tw * 1e-9, 2.0 * bytes / tc * 1e-9);                                                        \
  }
    MEM_ROW(4, false, 8) MEM_ROW(4, true, 8) MEM_ROW(8, false, 4) MEM_ROW(8, true, 4) MEM_ROW(8, true, 8)
    MEM_ROW(4, true, 2) MEM_ROW(8, true, 1) MEM_ROW(8, true, 2)
    CK(hipFree(a));
    CK(hipFree(b));
  }

  if (all || !strcmp(what, "seg")) {
    const long mats = 4096;   // 2 GiB in, 2 GiB out
    char *a, *b;
    CK(hipMalloc(&a, mats * 524288));
    CK(hipMalloc(&b, mats * 524288));
    CK(hipMemset(a, 1, mats * 524288));
    CK(hipMemset(b, 2, mats * 524288));
    printf("\n[seg] 4096 matrices of 256 x 2048 B, column blocks of SEG bytes, nt accesses, 1 wg (256 lanes) per CU x WPC;\n"
           "      TB/s = bytes moved / time (copy counts read + write)\n");
    printf("%-34s %8s %8s %8s\n", "shape", "read", "write", "copy");
#define SEG_ROW(SEG, B16, WPC)                                                                                     \
  {                                                                                                                \
    const int grid = cus * WPC;                                                                                    \
    const double by = (double)mats * 524288;                                                                       \
    float tr = time_launches(10, 50, [&] { hipLaunchKernelGGL((k_seg<SEG, B16, true, 1>), dim3(grid), dim3(256), 0, 0, b, a, mats, sink); }); \
    float tw = time_launches(10, 50, [&] { hipLaunchKernelGGL((k_seg<SEG, B16, true, 2>), dim3(grid), dim3(256), 0, 0, b, a, mats, sink); }); \
    float tc = time_launches(10, 50, [&] { hipLaunchKernelGGL((k_seg<SEG, B16, true, 0>), dim3(grid), dim3(256), 0, 0, b, a, mats, sink); }); \
    printf("seg %4d B, %2d B/lane, wg/CU %d       %8.2f %8.2f %8.2f\n", SEG, B16 ? 16 : 8, WPC, by / tr * 1e-9,       \
           by / tw * 1e-9, 2 * by / tc * 1e-9);                                                                    \
  }
    SEG_ROW(128, false, 1) SEG_ROW(128, false, 2) SEG_ROW(128, false, 4)
    SEG_ROW(256, false, 1) SEG_ROW(256, false, 2) SEG_ROW(256, true, 1) SEG_ROW(256, true, 2)
    SEG_ROW(512, false, 2) SEG_ROW(512, true, 1) SEG_ROW(512, true, 2)
    SEG_ROW(2048, true, 1) SEG_ROW(2048, true, 2) SEG_ROW(2048, true, 4)

    printf("\n[grp] as seg 128 B / 8 B per lane, but GRP adjacent column blocks per step with their accesses interleaved\n"
           "      (adjacent 128-byte segments requested back to back); ip = in place (dst = src).\n"
           "      copy TB/s (read + write counted): 6 rounds of 40 launches per shape, shapes interleaved; min / median / max over rounds\n");
    {
      struct Shape { const char *name; std::function<void()> launch; std::vector<float> ms; };
      std::vector<Shape> shapes;
#define GRP_SHAPE(GRP, WPC, IP) shapes.push_back({"grp " #GRP " x 128 B, wg/CU " #WPC, [&] { hipLaunchKernelGGL((k_seg_grp<GRP, true, 0, IP>), dim3(cus * WPC), dim3(256), 0, 0, IP ? a : b, a, mats, sink); }, {}});
      GRP_SHAPE(1, 1, false) GRP_SHAPE(2, 1, false) GRP_SHAPE(4, 1, false) GRP_SHAPE(1, 2, false) GRP_SHAPE(2, 2, false)
      GRP_SHAPE(1, 1, true) GRP_SHAPE(2, 1, true) GRP_SHAPE(4, 1, true) GRP_SHAPE(1, 2, true) GRP_SHAPE(2, 2, true)
#define FAR_SHAPE(GRP, WPC, IP) shapes.push_back({"far " #GRP " x 128 B, wg/CU " #WPC " ip=" #IP, [&] { hipLaunchKernelGGL((k_seg_grp<GRP, true, 0, IP, true>), dim3(cus * WPC), dim3(256), 0, 0, IP ? a : b, a, mats, sink); }, {}});
      FAR_SHAPE(2, 1, true) FAR_SHAPE(4, 1, true) FAR_SHAPE(2, 1, false) FAR_SHAPE(4, 1, false) FAR_SHAPE(8, 1, true)
      shapes.push_back({"grp 8 x 128 B, wg/CU 1 ip", [&] { hipLaunchKernelGGL((k_seg_grp<8, true, 0, true>), dim3(cus), dim3(256), 0, 0, a, a, mats, sink); }, {}});
      shapes.push_back({"seg 256 B, 8 B/lane, wg/CU 1", [&] { hipLaunchKernelGGL((k_seg<256, false, true, 0>), dim3(cus), dim3(256), 0, 0, b, a, mats, sink); }, {}});
      shapes.push_back({"seg 512 B, 16 B/lane, wg/CU 1", [&] { hipLaunchKernelGGL((k_seg<512, true, true, 0>), dim3(cus), dim3(256), 0, 0, b, a, mats, sink); }, {}});
      for (int round = 0; round < 6; round++)
        for (auto &sh : shapes) sh.ms.push_back(time_launches(8, 40, sh.launch));
      const double by = 2.0 * (double)mats * 524288;
      int idx = 0;
      for (auto &sh : shapes) {
        std::vector<float> v = sh.ms;
        std::sort(v.begin(), v.end());
        printf("%-32s %s  %6.2f / %6.2f / %6.2f\n", sh.name, (idx >= 5 && idx < 10) ? "ip" : "  ", by / v.back() * 1e-9, by / v[v.size() / 2] * 1e-9, by / v.front() * 1e-9);
        idx++;
      }
    }

    printf("\n[model] the resident kernel's memory behaviour with dummy arithmetic (WORK v_pk_fma per block and phase, one wave\n"
           "        per SIMD): ms per pass over 4096 matrices, in place; TB/s = 2 x 2 GiB / time\n");
    {
      struct Shape { const char *name; std::function<void()> launch; std::vector<float> ms; };
      std::vector<Shape> shapes;
      for (int work : {0, 50, 100}) {
        static char names[24][64];
        static int ni = 0;
        snprintf(names[ni], 64, "work %3d  1 block / step ", work);
        shapes.push_back({names[ni++], [&, work] { hipLaunchKernelGGL((k_resident_model<1>), dim3(cus), dim3(256), 0, 0, a, mats, work, sink); }, {}});
        snprintf(names[ni], 64, "work %3d  2 blocks / step", work);
        shapes.push_back({names[ni++], [&, work] { hipLaunchKernelGGL((k_resident_model<2>), dim3(cus), dim3(256), 0, 0, a, mats, work, sink); }, {}});
        snprintf(names[ni], 64, "work %3d  4 blocks / step", work);
        shapes.push_back({names[ni++], [&, work] { hipLaunchKernelGGL((k_resident_model<4>), dim3(cus), dim3(256), 0, 0, a, mats, work, sink); }, {}});
        snprintf(names[ni], 64, "work %3d  phases interleaved", work);
        shapes.push_back({names[ni++], [&, work] { hipLaunchKernelGGL(k_resident_model_il, dim3(cus), dim3(256), 0, 0, a, mats, work, sink); }, {}});
      }
      for (int round = 0; round < 4; round++)
        for (auto &sh : shapes) sh.ms.push_back(time_launches(6, 30, sh.launch));
      const double by = 2.0 * (double)mats * 524288;
      for (auto &sh : shapes) {
        std::vector<float> v = sh.ms;
        std::sort(v.begin(), v.end());
        printf("%-28s %7.3f ms (min %.3f)  %6.2f TB/s\n", sh.name, v[v.size() / 2], v.front(), by / v[v.size() / 2] * 1e-9);
      }
    }
    CK(hipFree(a));
    CK(hipFree(b));
  }

  if (all || !strcmp(what, "issue")) {
    printf("\n[issue] cycles per wave-instruction per SIMD (at %d MHz nominal; the clock under load may be lower)\n", prop.clockRate / 1000);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "accvgpr w+r pair", "v_add_f32"};
    const int iters = 20000;
    printf("%-18s %10s %10s %10s\n", "op", "1 wave/SIMD", "2 waves", "4 waves");
#define ISSUE_ROW(OP)                                                                                             \
  {                                                                                                               \
    double cyc[3];                                                                                                \
    int wi = 0;                                                                                                   \
    for (int waves : {4, 8, 16}) {                                                                                \
      float ms = time_launches(3, 10, [&] { hipLaunchKernelGGL((k_issue<OP>), dim3(cus), dim3(64 * waves), 0, 0, sink, iters); }); \
      const double inst = (double)iters * 64 * (OP == 4 ? 2 : 1) * (waves / 4);                                   \
      cyc[wi++] = ms * 1e-3 * prop.clockRate * 1e3 / inst;                                                        \
    }                                                                                                             \
    printf("%-18s %10.2f %10.2f %10.2f\n", names[OP], cyc[0], cyc[1], cyc[2]);                                    \
  }
    ISSUE_ROW(0) ISSUE_ROW(5) ISSUE_ROW(1) ISSUE_ROW(2) ISSUE_ROW(3) ISSUE_ROW(4)
    printf("\n[xchg] LDS exchange of 16 x 8 B per lane (32 KiB per 256 lanes) + barrier(s) + F v_pk_fma per value;\n"
           "       us per exchange per workgroup, one workgroup per CU\n");
    printf("%-44s %8s %8s %8s\n", "shape", "F=0", "F=4", "F=12");
#define X_ROW(T, NB)                                                                                              \
  {                                                                                                               \
    const int it = 4000;                                                                                          \
    float t0 = time_launches(2, 5, [&] { hipLaunchKernelGGL((k_xchg<T, NB, 0>), dim3(cus), dim3(T), 0, 0, sink, it); });  \
    float t4 = time_launches(2, 5, [&] { hipLaunchKernelGGL((k_xchg<T, NB, 4>), dim3(cus), dim3(T), 0, 0, sink, it); });  \
    float t12 = time_launches(2, 5, [&] { hipLaunchKernelGGL((k_xchg<T, NB, 12>), dim3(cus), dim3(T), 0, 0, sink, it); }); \
    printf("%d lanes (%d x 32 KiB per step), %d barrier(s)    %8.3f %8.3f %8.3f\n", T, T / 256, NB, t0 / it * 1e3,   \
           t4 / it * 1e3, t12 / it * 1e3);                                                                        \
  }
    X_ROW(256, 1) X_ROW(256, 2) X_ROW(512, 1) X_ROW(512, 2)
  }
  return 0;
}
