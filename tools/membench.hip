// dev tool (not product): memory-system microbenchmarks that bound the large-N FFT design.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o gpurun_out/membench && gpurun_out/membench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);          \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef float4 v4;

template <bool NT> __device__ __forceinline__ v4 ld(const v4 *p) {
  if constexpr (NT) {
    v4 r;
    r.x = __builtin_nontemporal_load(&p->x);
    r.y = __builtin_nontemporal_load(&p->y);
    r.z = __builtin_nontemporal_load(&p->z);
    r.w = __builtin_nontemporal_load(&p->w);
    return r;
  } else
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(v4 *p, v4 v) {
  if constexpr (NT) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z);
    __builtin_nontemporal_store(v.w, &p->w);
  } else
    *p = v;
}

template <bool NT> __global__ __launch_bounds__(256) void k_copy(v4 *dst, const v4 *src, size_t n, int reps) {
  for (int r = 0; r < reps; r++)
    for (size_t i = blockIdx.x * 256ul + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) st<NT>(dst + i, ld<NT>(src + i));
}
__global__ __launch_bounds__(256) void k_read(float *sink, const v4 *src, size_t n, int reps) {
  float acc = 0;
  for (int r = 0; r < reps; r++)
    for (size_t i = blockIdx.x * 256ul + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      v4 v = src[i];
      acc += v.x + v.y + v.z + v.w;
    }
  if (acc == 123.456f) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write(v4 *dst, size_t n, int reps) {
  v4 v = make_float4(1, 2, 3, 4);
  for (int r = 0; r < reps; r++)
    for (size_t i = blockIdx.x * 256ul + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}

// "in -> private scratch -> out" emulation of the two-phase FFT, no compute.
// Each workgroup handles chunks of CH bytes: phase A copies chunk from `in` to its scratch
// slot, barrier, phase B copies the slot to `out`.  UNR float4 per lane in flight.
template <int THREADS, int UNR, bool NT, bool SEG>
__global__ __launch_bounds__(THREADS) void k_twophase(v4 *out, const v4 *in, v4 *scratch, int chunks, int chunk_v4) {
  v4 *slot = scratch + (size_t)blockIdx.x * chunk_v4;
  for (int c = blockIdx.x; c < chunks; c += gridDim.x) {
    const v4 *src = in + (size_t)c * chunk_v4;
    v4 *dst = out + (size_t)c * chunk_v4;
    for (int base = 0; base < chunk_v4; base += THREADS * UNR) {
      v4 r[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        int i = base + u * THREADS + threadIdx.x;
        // SEG: 128-byte segments at 2 KiB stride (the column-block access of the real kernel)
        int j = SEG ? ((i & 7) | ((i >> 3 & 255) << 7) | ((i >> 11 & 15) << 3) | (i & ~32767)) : i;
        r[u] = ld<NT>(src + j);
      }
#pragma unroll
      for (int u = 0; u < UNR; u++) slot[base + u * THREADS + threadIdx.x] = r[u];
    }
    __syncthreads();
    for (int base = 0; base < chunk_v4; base += THREADS * UNR) {
      v4 r[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) r[u] = slot[base + u * THREADS + threadIdx.x];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        int i = base + u * THREADS + threadIdx.x;
        int j = SEG ? ((i & 7) | ((i >> 3 & 255) << 7) | ((i >> 11 & 15) << 3) | (i & ~32767)) : i;
        st<NT>(dst + j, r[u]);
      }
    }
    __syncthreads();
  }
}

// the FFT kernel's exact global access shape: 8 bytes per lane, 16 lanes per 128-byte row segment,
// rows 2 KiB apart (column block in -> row-major scratch; scratch rows -> column block out)
template <bool NT>
__global__ __launch_bounds__(256) void k_twophase8(unsigned long long *out, const unsigned long long *in,
                                                   unsigned long long *scratch, int chunks) {
  const int W = 65536;   // 8-byte words per 512 KiB chunk
  unsigned long long *slot = scratch + (size_t)blockIdx.x * W;
  const int col = threadIdx.x & 15, tf = threadIdx.x >> 4;
  for (int c = blockIdx.x; c < chunks; c += gridDim.x) {
    const unsigned long long *src = in + (size_t)c * W;
    unsigned long long *dst = out + (size_t)c * W;
    for (int cb = 0; cb < 16; cb++) {      // phase 1: column block cb (rows tf + 16 e) -> scratch, same places
      unsigned long long r[16];
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const unsigned long long *p = src + (tf + 16 * e) * 256 + cb * 16 + col;
        r[e] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int e = 0; e < 16; e++) slot[(tf + 16 * e) * 256 + cb * 16 + col] = r[e];
    }
    __syncthreads();
    for (int rb = 0; rb < 16; rb++) {      // phase 2: row block rb contiguous -> column block rb of out
      unsigned long long r[16];
      const int t2 = threadIdx.x & 15, row = threadIdx.x >> 4;
#pragma unroll
      for (int e = 0; e < 16; e++) r[e] = slot[(rb * 16 + row) * 256 + t2 + 16 * e];
#pragma unroll
      for (int e = 0; e < 16; e++) {
        unsigned long long *p = dst + (tf + 16 * e) * 256 + rb * 16 + col;
        if (NT) __builtin_nontemporal_store(r[e], p); else *p = r[e];
      }
    }
    __syncthreads();
  }
}

template <class F> static float timeit(F f, int iters = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main() {
  const size_t big = (size_t)2 << 30;  // 2 GiB
  v4 *a, *b, *scratch;
  float *sink;
  CK(hipMalloc(&a, big));
  CK(hipMalloc(&b, big));
  CK(hipMalloc(&scratch, (size_t)1 << 30));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, big));
  CK(hipMemset(b, 0, big));
  const int grid = 256 * 8;
  printf("== streaming 2 GiB (HBM)\n");
  {
    size_t n = big / 16;
    float ms = timeit([&] { k_copy<false><<<grid, 256>>>(b, a, n, 1); });
    printf("copy plain      : %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0 * big / ms / 1e9);
    ms = timeit([&] { k_copy<true><<<grid, 256>>>(b, a, n, 1); });
    printf("copy nt         : %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0 * big / ms / 1e9);
    ms = timeit([&] { k_read<<<grid, 256>>>(sink, a, n, 1); });
    printf("read only       : %.3f ms  %.2f TB/s\n", ms, 1.0 * big / ms / 1e9);
    ms = timeit([&] { k_write<<<grid, 256>>>(b, n, 1); });
    printf("write only      : %.3f ms  %.2f TB/s\n", ms, 1.0 * big / ms / 1e9);
  }
  for (size_t mb : {16, 64, 128}) {
    size_t bytes = mb << 20, n = bytes / 16;
    int reps = (int)(big / bytes);
    printf("== %zu MiB buffers, %d passes in one launch (cache-resident)\n", mb, reps);
    float ms = timeit([&] { k_copy<false><<<grid, 256>>>(b, a, n, reps); });
    printf("copy plain      : %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0 * bytes * reps / ms / 1e9);
    ms = timeit([&] { k_read<<<grid, 256>>>(sink, a, n, reps); });
    printf("read only       : %.3f ms  %.2f TB/s\n", ms, 1.0 * bytes * reps / ms / 1e9);
    ms = timeit([&] { k_write<<<grid, 256>>>(b, n, reps); });
    printf("write only      : %.3f ms  %.2f TB/s\n", ms, 1.0 * bytes * reps / ms / 1e9);
  }
  printf("== two-phase emulation: 4096 chunks of 512 KiB, in -> scratch -> out (alg = 2 x 2 GiB)\n");
  const int chunks = 4096, chunk_v4 = (512 << 10) / 16;
#define TP(THREADS, UNR, NT, SEG, G)                                                                        \
  {                                                                                                          \
    float ms = timeit([&] { k_twophase<THREADS, UNR, NT, SEG><<<G, THREADS>>>(b, a, scratch, chunks, chunk_v4); }); \
    printf("threads %4d unr %d nt %d seg %d grid %4d : %.3f ms  alg %.2f TB/s\n", THREADS, UNR, NT, SEG, G, ms,     \
           2.0 * big / ms / 1e9);                                                                            \
  }
  TP(1024, 4, false, false, 256)
  TP(1024, 4, true, false, 256)
  TP(1024, 8, true, false, 256)
  TP(512, 8, true, false, 512)
  TP(512, 8, true, false, 256)
  TP(256, 8, true, false, 1024)
  TP(256, 8, true, false, 512)
  TP(256, 8, true, false, 256)
  TP(256, 8, true, false, 128)
  TP(1024, 8, true, true, 256)
  TP(512, 8, true, true, 512)
  TP(256, 8, true, true, 1024)
  printf("== two-phase emulation with the FFT kernel's own access shape (8 B per lane, 128-B segments)\n");
  for (int G : {512, 1024}) {
    float ms = timeit([&] { k_twophase8<true><<<G, 256>>>((unsigned long long *)b, (const unsigned long long *)a, (unsigned long long *)scratch, 4096); });
    printf("8B lanes nt 1 grid %4d : %.3f ms  alg %.2f TB/s\n", G, ms, 2.0 * big / ms / 1e9);
    ms = timeit([&] { k_twophase8<false><<<G, 256>>>((unsigned long long *)b, (const unsigned long long *)a, (unsigned long long *)scratch, 4096); });
    printf("8B lanes nt 0 grid %4d : %.3f ms  alg %.2f TB/s\n", G, ms, 2.0 * big / ms / 1e9);
  }
  printf("== two-phase emulation with L2-resident scratch: chunks of 32 KiB / 64 KiB / 128 KiB\n");
#define TPC(THREADS, UNR, NT, G, CHKB)                                                                     \
  {                                                                                                          \
    int cv4 = (CHKB << 10) / 16, nch = (int)(big / ((size_t)CHKB << 10));                                    \
    float ms = timeit([&] { k_twophase<THREADS, UNR, NT, false><<<G, THREADS>>>(b, a, scratch, nch, cv4); }); \
    printf("chunk %4d KiB threads %4d unr %d nt %d grid %4d (scratch %.1f MiB/XCD): %.3f ms  alg %.2f TB/s\n", CHKB, THREADS, UNR, NT, G, \
           G * (double)CHKB / 1024 / 8, ms, 2.0 * big / ms / 1e9);                                          \
  }
  TPC(256, 8, true, 512, 32)
  TPC(256, 8, true, 1024, 32)
  TPC(256, 8, false, 1024, 32)
  TPC(256, 8, true, 2048, 32)
  TPC(256, 4, true, 1024, 16)
  TPC(256, 8, true, 512, 64)
  TPC(256, 8, true, 1024, 64)
  TPC(512, 8, true, 512, 64)
  TPC(512, 8, true, 256, 128)
  TPC(512, 8, true, 512, 128)
  return 0;
}
