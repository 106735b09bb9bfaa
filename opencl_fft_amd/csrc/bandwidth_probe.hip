// bandwidth_probe.hip — the yardsticks bench.py prints beside the FFT's roofline fraction
// ("the bench must also print a measured device-copy bandwidth").
//
// Not part of the reference's surface: a measurement aid behind one C entry point.  Kernels are the
// ones of tools/ubench.hip: single-instruction 16-byte non-temporal accesses, several in flight per
// lane; plus the FFT's own global access shape — 256 x 256 matrices of 8-byte elements moved in
// column blocks of 16 columns (128-byte row segments 2 KiB apart) — which is what bounds a
// four-step transform that touches HBM once.
#include <hip/hip_runtime.h>

#include "../../include/clfft_amd.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int UNR> __global__ __launch_bounds__(256) void k_bw_copy(f4 *__restrict__ dst, const f4 *__restrict__ src, long tiles) {
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const f4 *s = src + t * (256 * UNR) + threadIdx.x;
    f4 *d = dst + t * (256 * UNR) + threadIdx.x;
    f4 r[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) r[u] = __builtin_nontemporal_load(s + u * 256);
#pragma unroll
    for (int u = 0; u < UNR; u++) __builtin_nontemporal_store(r[u], d + u * 256);
  }
}
template <int UNR> __global__ __launch_bounds__(256) void k_bw_read(float *__restrict__ sink, const f4 *__restrict__ src, long tiles) {
  f4 acc = 0;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const f4 *s = src + t * (256 * UNR) + threadIdx.x;
    f4 r[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) r[u] = __builtin_nontemporal_load(s + u * 256);
#pragma unroll
    for (int u = 0; u < UNR; u++) acc += r[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = acc.x;
}
template <int UNR> __global__ __launch_bounds__(256) void k_bw_write(f4 *__restrict__ dst, long tiles) {
  const f4 v = {1, 2, 3, 4};
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    f4 *d = dst + t * (256 * UNR) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNR; u++) __builtin_nontemporal_store(v, d + u * 256);
  }
}
// column blocks of 16 columns of 512 KiB matrices: lane (c, t) moves rows t + 16 e of column c, like
// one column / row block of the resident FFT kernel (8 bytes per lane, 16 in flight)
__global__ __launch_bounds__(256) void k_bw_colblock(f2 *__restrict__ dst, const f2 *__restrict__ src, long mats) {
  const int c = threadIdx.x & 15, t = threadIdx.x >> 4;
  for (long m = blockIdx.x; m < mats; m += gridDim.x) {
    const f2 *s = src + m * 65536 + t * 256 + c;
    f2 *d = dst + m * 65536 + t * 256 + c;
    for (int cb = 0; cb < 16; cb++) {
      f2 r[16];
#pragma unroll
      for (int e = 0; e < 16; e++) r[e] = __builtin_nontemporal_load(s + cb * 16 + e * 4096);
#pragma unroll
      for (int e = 0; e < 16; e++) __builtin_nontemporal_store(r[e], d + cb * 16 + e * 4096);
    }
  }
}

}  // namespace

extern "C" int clfa_bandwidth_probe(int device, int what, size_t bytes, int launches, double *tb_per_s) {
  if (!tb_per_s || launches < 1 || bytes < (size_t)(1 << 20) || what < 0 || what > 3) return CLFA_INVALID_VALUE;
  int prev = -1, count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return CLFA_DEVICE_NOT_FOUND;
  }
  if (device < 0 || device >= count) return CLFA_INVALID_DEVICE;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(device);
  hipDeviceProp_t prop;
  void *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = CLFA_SUCCESS;
  bytes &= ~((size_t)524288 - 1);   // whole 512 KiB matrices
  auto fail = [&](hipError_t e) {
    if (e != hipSuccess) {
      (void)hipGetLastError();
      rc = e == hipErrorOutOfMemory ? CLFA_MEM_OBJECT_ALLOCATION_FAILURE : CLFA_OUT_OF_RESOURCES;
    }
    return e != hipSuccess;
  };
  do {
    if (fail(hipGetDeviceProperties(&prop, device))) break;
    if (fail(hipMalloc(&a, bytes)) || fail(hipMalloc(&b, bytes))) break;
    if (fail(hipMemset(a, 1, bytes)) || fail(hipMemset(b, 2, bytes))) break;
    if (fail(hipEventCreate(&e0)) || fail(hipEventCreate(&e1))) break;
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    auto launch = [&]() {
      switch (what) {
        case 0: hipLaunchKernelGGL((k_bw_read<8>), dim3(cus * 2), dim3(256), 0, 0, (float *)b, (const f4 *)a, (long)(bytes / (256 * 8 * 16))); break;
        case 1: hipLaunchKernelGGL((k_bw_write<8>), dim3(cus * 4), dim3(256), 0, 0, (f4 *)b, (long)(bytes / (256 * 8 * 16))); break;
        case 2: hipLaunchKernelGGL((k_bw_copy<4>), dim3(cus * 2), dim3(256), 0, 0, (f4 *)b, (const f4 *)a, (long)(bytes / (256 * 4 * 16))); break;
        default: hipLaunchKernelGGL((k_bw_colblock), dim3(cus * 2), dim3(256), 0, 0, (f2 *)b, (const f2 *)a, (long)(bytes / 524288)); break;
      }
    };
    for (int i = 0; i < launches / 4 + 2; i++) launch();   // warm-up
    if (fail(hipEventRecord(e0, 0))) break;
    for (int i = 0; i < launches; i++) launch();
    if (fail(hipEventRecord(e1, 0)) || fail(hipEventSynchronize(e1)) || fail(hipGetLastError())) break;
    float ms = 0;
    if (fail(hipEventElapsedTime(&ms, e0, e1))) break;
    const double moved = (what == 2 || what == 3 ? 2.0 : 1.0) * (double)bytes * launches;
    *tb_per_s = moved / (ms * 1e-3) * 1e-12;
  } while (0);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  return rc;
}
