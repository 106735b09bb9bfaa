// dev tool: same-XCD cross-CU hand-off through L2: do the reader's loads hit L2?
// pairs of workgroups (b, b+8) share an XCD (verified with HW_REG_XCC_ID); writer copies
// 32 KiB blocks in -> scratch and bumps a counter; reader waits, copies scratch -> out.
// MODE 0: reader uses sc1 loads; MODE 1: plain loads after an agent-scope acquire fence;
// MODE 2: plain loads, no fence (may be stale; traffic only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
template <int MODE>
__global__ __launch_bounds__(256) void k_pair(const u64 *in, u64 *out, u64 *scratch, unsigned *flags, int iters, unsigned *bad) {
  const int pair = (blockIdx.x / 16) * 8 + (blockIdx.x % 8);   // blocks b and b+8 form a pair
  const bool writer = ((blockIdx.x / 8) & 1) == 0;
  const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 7u;
  const int W = 32768 / 8;                                      // u64 per block
  u64 *slot = scratch + (size_t)pair * 2 * W;                   // 2 slots per pair
  unsigned *fw = flags + 64 * pair, *fr = flags + 64 * pair + 32, *fx = flags + 64 * pair + 16;
  if (threadIdx.x == 0) { if (writer) __hip_atomic_store(fx, xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  for (int it = 0; it < iters; it++) {
    const u64 *src = in + ((size_t)pair * iters + it) * W;
    u64 *dst = out + ((size_t)pair * iters + it) * W;
    u64 *s = slot + (it & 1) * W;
    if (writer) {
      if (threadIdx.x == 0 && it >= 2) { unsigned n = 0; while (__hip_atomic_load(fr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it - 1)) { __builtin_amdgcn_s_sleep(1); if (++n > (1u << 22)) break; } }
      __syncthreads();
      u64 r[16];
      for (int u = 0; u < 16; u++) r[u] = __builtin_nontemporal_load(src + u * 256 + threadIdx.x);
      for (int u = 0; u < 16; u++) s[u * 256 + threadIdx.x] = r[u];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(fw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      if (threadIdx.x == 0) {
        unsigned n = 0; while (__hip_atomic_load(fw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1)) { __builtin_amdgcn_s_sleep(1); if (++n > (1u << 22)) break; }
        if (it == 0 && __hip_atomic_load(fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc + 1) atomicAdd(bad, 1);
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      u64 r[16];
      for (int u = 0; u < 16; u++) {
        const u64 *p = s + u * 256 + threadIdx.x;
        r[u] = MODE == 0 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *(volatile const u64 *)p;
      }
      for (int u = 0; u < 16; u++) __builtin_nontemporal_store(r[u], dst + u * 256 + threadIdx.x);
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(fr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}
int main(int argc, char **argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int grid = 1024, pairs = grid / 2, iters = 128;          // 512 pairs x 128 x 32 KiB = 2 GiB
  size_t bytes = (size_t)pairs * iters * 32768;
  u64 *in, *out, *scratch; unsigned *flags, *bad;
  hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMalloc(&scratch, (size_t)pairs * 65536); hipMalloc(&flags, pairs * 256); hipMalloc(&bad, 4);
  hipMemset(in, 1, bytes); hipMemset(bad, 0, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; rep++) {
    hipMemset(flags, 0, pairs * 256);
    hipEventRecord(a);
    if (mode == 0) k_pair<0><<<grid, 256>>>(in, out, scratch, flags, iters, bad);
    else if (mode == 1) k_pair<1><<<grid, 256>>>(in, out, scratch, flags, iters, bad);
    else k_pair<2><<<grid, 256>>>(in, out, scratch, flags, iters, bad);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("mode %d: %.3f ms  alg %.2f TB/s  (pairs on different XCC: %u)\n", mode, ms, 2.0 * bytes / ms / 1e9, hb);
  }
  return 0;
}
