// dev tool: what does HW_REG_XCC_ID return per workgroup, and does a same-XCC_ID reader hit L2?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_ids(unsigned *out) {
  unsigned xcc = __builtin_amdgcn_s_getreg(6164);          // hwreg(HW_REG_XCC_ID, 0, 4)
  unsigned full = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // all 32 bits
  unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
  if (threadIdx.x == 0) { out[3 * blockIdx.x] = xcc; out[3 * blockIdx.x + 1] = full; out[3 * blockIdx.x + 2] = hwid; }
}
int main() {
  const int G = 64;
  unsigned *d; hipMalloc(&d, G * 12);
  k_ids<<<G, 64>>>(d);
  std::vector<unsigned> h(3 * G);
  hipMemcpy(h.data(), d, G * 12, hipMemcpyDeviceToHost);
  for (int b = 0; b < G; b++) printf("block %2d xcc(4b)=%u xcc_id_reg=0x%08x hw_id=0x%08x\n", b, h[3*b], h[3*b+1], h[3*b+2]);
  return 0;
}
