// dev tool (not product): where the resident n = 65536 kernel spends its time.
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -fno-slp-vectorize -I opencl_fft_amd/csrc tools/res16_probe.hip -o /tmp/res16_probe
// Times k_fft_res16 on 4096 transforms with parts left out (PROBE bits, fft_resident.hip) and reads the
// per-phase clock stamps.  Results are garbage by construction for every mode but "full".
#define CLFA_RES16_PROBE 1   // the kernel's timing experiments (stamps, grid barrier, time slots) exist only in this tool
#include "../opencl_fft_amd/csrc/fft_resident.hip"

#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace clfa;

#define CK(x)                                                         \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                        \
    }                                                                 \
  } while (0)

template <int PROBE> static void run(const char *name, cpx *data, cpx *slots, cpx *tabs, unsigned long long *dbg, long batch, int cus) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int warm = 10, reps = 40;
  auto launch = [&] {
    if (PROBE & 512) CK(hipMemsetAsync(dbg + 1024, 0, 8, 0));
    hipLaunchKernelGGL((k_fft_res16<true, false, PROBE>), dim3(cus), dim3(256), 0, 0, data, slots, tabs, batch, dbg);
  };
  for (int i = 0; i < warm; i++) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  double p1 = 0, p2 = 0;
  if (PROBE & kProbeStamps) {
    std::vector<unsigned long long> h(2 * cus);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < cus; i++) {
      p1 += h[2 * i];
      p2 += h[2 * i + 1];
    }
    p1 /= cus;
    p2 /= cus;
    if (PROBE == kProbeStamps) {   // spread over the workgroups (one per CU; workgroup b usually sits on XCD b % 8)
      double mn = 1e30, mx = 0, xs[8] = {0};
      for (int i = 0; i < cus; i++) {
        const double t = (double)h[2 * i] + (double)h[2 * i + 1];
        mn = t < mn ? t : mn;
        mx = t > mx ? t : mx;
        xs[i & 7] += t / (cus / 8);
      }
      printf("  per-workgroup busy cycles: min %.3f  avg %.3f  max %.3f M  (max/avg %.3f);  by blockIdx %% 8:", mn * 1e-6, (p1 + p2) * 1e-6,
             mx * 1e-6, mx / (p1 + p2));
      for (int x = 0; x < 8; x++) printf(" %.3f", xs[x] * 1e-6);
      printf("\n");
    }
  }
  printf("%-34s %8.3f ms  %6.2f TB/s alg", name, ms, batch * 65536.0 * 16 / ms * 1e-9);
  if (PROBE & kProbeStamps) printf("   phase1 %.1f  phase2 %.1f  kcycles per transform (s_memtime, 100 MHz ticks x?)", p1 / (batch / cus) * 1e-3, p2 / (batch / cus) * 1e-3);
  printf("\n");
}

// time slots (PROBE 1024): phase k of every workgroup starts no earlier than (its start) + S[k]
static void run_slots(cpx *data, cpx *slots, cpx *tabs, unsigned long long *dbg, long batch, int cus, unsigned p1, unsigned p2) {
  unsigned long long h[2] = {p1, p2};
  CK(hipMemcpy(dbg + 2048, h, 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto launch = [&] { hipLaunchKernelGGL((k_fft_res16<true, false, 1024 | 16>), dim3(cus), dim3(256), 0, 0, data, slots, tabs, batch, dbg); };
  for (int i = 0; i < 10; i++) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < 40; i++) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 40;
  std::vector<unsigned long long> st(2 * cus);
  CK(hipMemcpy(st.data(), dbg, st.size() * 8, hipMemcpyDeviceToHost));
  double a1 = 0, a2 = 0;
  for (int i = 0; i < cus; i++) {
    a1 += st[2 * i];
    a2 += st[2 * i + 1];
  }
  printf("slots %5.2f + %5.2f us                 %8.3f ms  %6.2f TB/s alg   busy phase1 %.1f phase2 %.1f kcycles per transform\n", p1 * 0.01, p2 * 0.01,
         ms, batch * 65536.0 * 16 / ms * 1e-9, a1 / cus / (batch / cus) * 1e-3, a2 / cus / (batch / cus) * 1e-3);
}

// the same work as `parts` launches of batch / parts transforms each: kernel boundaries keep the
// workgroups' read and write phases aligned chip-wide
static void run_split(cpx *data, cpx *slots, cpx *tabs, long batch, int cus, int parts) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int warm = 5, reps = 20;
  const long per = batch / parts;
  auto once = [&] {
    for (int p = 0; p < parts; p++)
      hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(cus), dim3(256), 0, 0, data + p * per * 65536, slots, tabs, per, (unsigned long long *)nullptr);
  };
  for (int i = 0; i < warm; i++) once();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) once();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%2d launches of %4ld transforms          %8.3f ms  %6.2f TB/s alg\n", parts, per, ms, batch * 65536.0 * 16 / ms * 1e-9);
}

// per-launch times of the first launches after an idle period (optionally after `pre` ms of another
// kernel): how long does the chip take to reach its steady state for this kernel?
__global__ void k_spin(float *out, int iters) {
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; i++) a = a * 0.999f + 1e-3f;
  if (a == 123.456f) out[0] = a;
}
static void run_ramp(const char *name, cpx *data, cpx *slots, cpx *tabs, long batch, int cus, int pre_kind, float *sink) {
  usleep(300000);
  const int n = 30;
  std::vector<hipEvent_t> ev(n + 1);
  for (auto &evt : ev) CK(hipEventCreate(&evt));
  if (pre_kind == 1) {        // ALU-only activity, ~50 ms
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_spin, dim3(cus * 8), dim3(256), 0, 0, sink, 300000);
  } else if (pre_kind == 2) { // memory activity: 50 launches of the FFT kernel itself
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(cus), dim3(256), 0, 0, data, slots, tabs, batch, (unsigned long long *)nullptr);
  }
  CK(hipEventRecord(ev[0]));
  for (int i = 0; i < n; i++) {
    hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(cus), dim3(256), 0, 0, data, slots, tabs, batch, (unsigned long long *)nullptr);
    CK(hipEventRecord(ev[i + 1]));
  }
  CK(hipEventSynchronize(ev[n]));
  printf("%-44s", name);
  for (int i = 0; i < n; i++) {
    float ms;
    CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    printf(" %.3f", ms);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const long batch = 4096;
  cpx *data, *slots, *tabs;
  unsigned long long *dbg;
  CK(hipMalloc(&data, batch * 65536 * 8));
  CK(hipMalloc(&slots, (size_t)cus * 32768));
  CK(hipMalloc(&tabs, 1792 * 8));
  CK(hipMalloc(&dbg, 32768));
  CK(hipMemset(dbg, 0, 32768));
  CK(hipMemset(data, 0, batch * 65536 * 8));
  std::vector<cpx> t(1792);
  for (int i = 0; i < 1792; i++) t[i] = mk((float)cos(i * 0.001), (float)sin(i * 0.001));   // unit-modulus stand-ins: timing only
  CK(hipMemcpy(tabs, t.data(), 1792 * 8, hipMemcpyHostToDevice));
  printf("k_fft_res16 probe: %ld transforms, %d workgroups\n", batch, cus);
  run<0>("full", data, slots, tabs, dbg, batch, cus);
  run<16>("full + stamps", data, slots, tabs, dbg, batch, cus);
  for (unsigned p1 : {2300u, 2400u, 2500u, 2600u, 2700u})
    for (unsigned p2 : {2300u, 2400u, 2500u, 2600u}) run_slots(data, slots, tabs, dbg, batch, cus, p1, p2);
  run<0>("full (again)", data, slots, tabs, dbg, batch, cus);
  if (getenv("PROBE_SLOTS_ONLY")) return 0;
  run<1 | 16>("no loads", data, slots, tabs, dbg, batch, cus);
  run<2 | 16>("no stores", data, slots, tabs, dbg, batch, cus);
  run<1 | 2 | 8 | 16>("no global traffic at all", data, slots, tabs, dbg, batch, cus);
  run<32 | 16>("loads never waited for", data, slots, tabs, dbg, batch, cus);
  run<64 | 16>("no arithmetic", data, slots, tabs, dbg, batch, cus);
  run<64 | 4 | 16>("no arithmetic, no barriers", data, slots, tabs, dbg, batch, cus);
  run<128 | 2 | 16>("phase 1 only (+2 row blocks), no stores", data, slots, tabs, dbg, batch, cus);
  run<128 | 2 | 64 | 16>("phase 1 only, no stores, no arithmetic", data, slots, tabs, dbg, batch, cus);
  run<256 | 16>("rotated column-block order", data, slots, tabs, dbg, batch, cus);
  run<256 | 2 | 16>("rotated, no stores", data, slots, tabs, dbg, batch, cus);
  run<256 | 1 | 16>("rotated, no loads", data, slots, tabs, dbg, batch, cus);
  run<512 | 16>("grid barrier at phase boundaries", data, slots, tabs, dbg, batch, cus);
  run<0>("full (again)", data, slots, tabs, dbg, batch, cus);
  for (int parts : {1, 2, 4}) run_split(data, slots, tabs, batch, cus, parts);
  float *sink;
  CK(hipMalloc(&sink, 64));
  run_ramp("30 launches after 0.3 s idle", data, slots, tabs, batch, cus, 0, sink);
  run_ramp("... after idle + 50 ms of ALU-only kernels", data, slots, tabs, batch, cus, 1, sink);
  run_ramp("... after idle + 50 launches of itself", data, slots, tabs, batch, cus, 2, sink);
  run_ramp("30 launches after 0.3 s idle (again)", data, slots, tabs, batch, cus, 0, sink);
  {
    // the same with random data of O(1) magnitude (as bench.py), forward (scaled 1/N) / inverse alternating
    std::vector<cpx> h(1 << 20);
    unsigned sd = 1;
    for (auto &c : h) {
      sd = sd * 1664525u + 1013904223u;
      float re = (sd >> 8) / 8388608.f - 1.f;
      sd = sd * 1664525u + 1013904223u;
      c = mk(re, (sd >> 8) / 8388608.f - 1.f);
    }
    for (long off = 0; off < batch * 65536; off += (1 << 20)) CK(hipMemcpy(data + off, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<cpx> tt(1792);
    // real tables this time (unit-modulus stand-ins would let the values drift)
    for (int t = 0; t < 16; t++) for (int j = 0; j < 16; j++) tt[16 * t + j] = mk((float)cos(t * j * 2 * M_PI / 256), -(float)sin(t * j * 2 * M_PI / 256));
    for (int k = 0; k < 256; k++) tt[256 + k] = mk((float)cos(k * 2 * M_PI / 65536), -(float)sin(k * 2 * M_PI / 65536));
    for (int k = 0; k < 256; k++) tt[512 + k] = mk((float)cos(k * 2 * M_PI / 256), -(float)sin(k * 2 * M_PI / 256));
    for (int m = 0; m < 4; m++) for (int k = 0; k < 256; k++) { int idx = ((1 << m) * k) & 4095; tt[768 + 256 * m + k] = mk((float)cos(idx * 2 * M_PI / 4096), -(float)sin(idx * 2 * M_PI / 4096)); }
    CK(hipMemcpy(tabs, tt.data(), 1792 * 8, hipMemcpyHostToDevice));
    usleep(300000);
    const int n = 30;
    std::vector<hipEvent_t> ev(n + 1);
    for (auto &evt : ev) CK(hipEventCreate(&evt));
    CK(hipEventRecord(ev[0]));
    for (int i = 0; i < n; i++) {
      if (i & 1) hipLaunchKernelGGL((k_fft_res16<false, false, 0>), dim3(cus), dim3(256), 0, 0, data, slots, tabs, batch, (unsigned long long *)nullptr);
      else hipLaunchKernelGGL((k_fft_res16<true, true, 0>), dim3(cus), dim3(256), 0, 0, data, slots, tabs, batch, (unsigned long long *)nullptr);
      CK(hipEventRecord(ev[i + 1]));
    }
    CK(hipEventSynchronize(ev[n]));
    printf("%-44s", "random data, fwd/inv alternating, after idle");
    for (int i = 0; i < n; i++) {
      float ms;
      CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
      printf(" %.3f", ms);
    }
    printf("\n");
  }
  return 0;
}
