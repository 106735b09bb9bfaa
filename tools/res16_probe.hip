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
#include <algorithm>
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

static cpx *g_out = nullptr;   // != nullptr: the launches write there (out of place)

template <int PROBE> static void run(const char *name, cpx *data, cpx *slots, cpx *tabs, unsigned long long *dbg, long batch, int cus) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int warm = 10, reps = 40;
  auto launch = [&] {
    if (PROBE & 512) CK(hipMemsetAsync(dbg + 1024, 0, 8, 0));
    hipLaunchKernelGGL((k_fft_res16<true, false, PROBE>), dim3(cus), dim3(256), 0, 0, data, g_out ? g_out : data, slots, tabs, batch, dbg);
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
  auto launch = [&] { hipLaunchKernelGGL((k_fft_res16<true, false, 1024 | 16>), dim3(cus), dim3(256), 0, 0, data, g_out ? g_out : data, slots, tabs, batch, dbg); };
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
      hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(cus), dim3(256), 0, 0, data + p * per * 65536, data + p * per * 65536, slots, tabs, per, (unsigned long long *)nullptr);
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
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(cus), dim3(256), 0, 0, data, g_out ? g_out : data, slots, tabs, batch, (unsigned long long *)nullptr);
  }
  CK(hipEventRecord(ev[0]));
  for (int i = 0; i < n; i++) {
    hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(cus), dim3(256), 0, 0, data, g_out ? g_out : data, slots, tabs, batch, (unsigned long long *)nullptr);
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
  if (getenv("PROBE_NOXCHG")) {   // what do the LDS exchanges (and their barriers) cost the full kernel?
    for (int round = 0; round < 3; round++) {
      run<16>("full + stamps", data, slots, tabs, dbg, batch, cus);
      run<4096 | 16>("no LDS exchange (garbage) + stamps", data, slots, tabs, dbg, batch, cus);
      run<4 | 16>("no barriers (garbage) + stamps", data, slots, tabs, dbg, batch, cus);
      run<4096 | 1 | 2 | 8 | 16>("no exchange, no global traffic", data, slots, tabs, dbg, batch, cus);
      run<1 | 2 | 8 | 16>("no global traffic", data, slots, tabs, dbg, batch, cus);
    }
    return 0;
  }
  if (getenv("PROBE_PACK")) {   // a pair-map phase behind phase 2, inside the launch
    for (int round = 0; round < 3; round++) {
      run<16>("full + stamps", data, slots, tabs, dbg, batch, cus);
      run<2048 | 16>("full + in-launch pair pass + stamps", data, slots, tabs, dbg, batch, cus);
    }
    return 0;
  }
  if (getenv("PROBE_GRID")) {   // fewer workgroups than CUs: does the chip need all 256 to move these bytes?
    for (int round = 0; round < 3; round++)
      for (int g : {256, 248, 240, 224, 208, 192, 160, 128}) {
        const long b2 = (batch / g) * g;   // whole rounds only
        char nm[64];
        snprintf(nm, sizeof nm, "grid %3d, %4ld transforms", g, b2);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        auto launch = [&] { hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(g), dim3(256), 0, 0, data, data, slots, tabs, b2, (unsigned long long *)nullptr); };
        for (int i = 0; i < 6; i++) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 30; i++) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-30s %8.3f ms  %6.2f TB/s alg\n", nm, ms / 30, b2 * 65536.0 * 16 / (ms / 30) * 1e-9);
      }
    return 0;
  }
  if (getenv("PROBE_MAP")) {   // transform -> workgroup assignments, in place
    for (int round = 0; round < 3; round++)
      for (unsigned long long mode = 0; mode < 4; mode++) {
        CK(hipMemcpy(dbg + 3000, &mode, 8, hipMemcpyHostToDevice));
        char nm[64];
        snprintf(nm, sizeof nm, "assignment %llu, in place", mode);
        run<16>(nm, data, slots, tabs, dbg, batch, cus);
      }
    return 0;
  }
  if (getenv("PROBE_PERM")) {   // bit permutations of (workgroup, iteration) -> transform, in place; needs 256 CUs
    struct P { int p[12]; float best; };
    std::vector<P> perms;
    auto add = [&](std::initializer_list<int> l) { P q; int j = 0; for (int x : l) q.p[j++] = x; q.best = 1e9f; perms.push_back(q); };
    add({0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11});     // the library's
    for (int pos = 0; pos <= 8; pos++) {             // the 4 iteration bits as a group at bit `pos` of b
      P q; int src = 0;
      for (int j = 0; j < 12; j++) q.p[j] = (j >= pos && j < pos + 4) ? 8 + (j - pos) : src++;
      q.best = 1e9f; perms.push_back(q);
    }
    for (int pos = 0; pos <= 8; pos++) {             // ... with the workgroup bits reversed (XCD bits on top)
      P q; int src = 7;
      for (int j = 0; j < 12; j++) q.p[j] = (j >= pos && j < pos + 4) ? 8 + (j - pos) : src--;
      q.best = 1e9f; perms.push_back(q);
    }
    if (getenv("PROBE_PERM_XCD")) {                  // XCD-compact candidates only (w0..w2 = XCD, w3..w7 = workgroup in the XCD)
      perms.resize(1);
      add({3, 4, 5, 6, 7, 0, 1, 2, 8, 9, 10, 11});   // A: window of G transforms, XCD x takes its x-th eighth
      add({7, 6, 5, 4, 3, 2, 1, 0, 8, 9, 10, 11});   // B: the same, bits reversed
      add({3, 4, 5, 6, 7, 8, 9, 10, 11, 0, 1, 2});   // C: every XCD streams through its own eighth of the batch
      add({3, 4, 5, 6, 7, 0, 8, 9, 10, 11, 1, 2});   // D
      add({3, 4, 5, 6, 7, 0, 1, 8, 9, 10, 11, 2});   // E
      add({8, 9, 10, 11, 3, 4, 5, 6, 7, 0, 1, 2});   // F: every workgroup takes 16 adjacent transforms, XCDs an eighth each
      add({8, 9, 3, 4, 5, 6, 7, 10, 11, 0, 1, 2});   // G
      add({3, 4, 5, 6, 7, 1, 2, 8, 9, 10, 11, 0});   // H: as E with the XCD's low bit on top
      add({4, 6, 5, 7, 3, 2, 1, 8, 9, 10, 11, 0});   // the local search's best
      add({7, 3, 6, 8, 11, 2, 5, 1, 9, 10, 0, 4});
    }
    unsigned sd = 12345;
    for (int r = 0; r < (getenv("PROBE_PERM_XCD") ? 0 : 60); r++) {                   // random ones
      P q; for (int j = 0; j < 12; j++) q.p[j] = j;
      for (int j = 11; j > 0; j--) { sd = sd * 1664525u + 1013904223u; int o = (sd >> 8) % (j + 1); std::swap(q.p[j], q.p[o]); }
      q.best = 1e9f; perms.push_back(q);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int round = 0; round < (getenv("PROBE_PERM_XCD") ? 6 : 3); round++)
      for (auto &q : perms) {
        unsigned long long h[13] = {4};
        for (int j = 0; j < 12; j++) h[1 + j] = q.p[j];
        CK(hipMemcpy(dbg + 3000, h, sizeof h, hipMemcpyHostToDevice));
        auto launch = [&] { hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(cus), dim3(256), 0, 0, data, data, slots, tabs, batch, dbg); };
        for (int i = 0; i < 5; i++) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; i++) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        q.best = std::min(q.best, ms / 20);
      }
    CK(hipGetLastError());
    for (auto &q : perms) {
      printf("b bits 0..11 <- ");
      for (int j = 0; j < 12; j++) printf(q.p[j] < 8 ? "w%d " : "k%d ", q.p[j] < 8 ? q.p[j] : q.p[j] - 8);
      printf("  %8.3f ms\n", q.best);
    }
    return 0;
  }
  if (getenv("PROBE_SEARCH")) {   // local search over the bit permutations (pairwise swaps from the best so far), in place
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto eval = [&](const int *p, int reps) {
      unsigned long long h[13] = {4};
      for (int j = 0; j < 12; j++) h[1 + j] = p[j];
      CK(hipMemcpy(dbg + 3000, h, sizeof h, hipMemcpyHostToDevice));
      auto launch = [&] { hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(cus), dim3(256), 0, 0, data, data, slots, tabs, batch, dbg); };
      for (int i = 0; i < 4; i++) launch();
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; i++) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      return ms / reps;
    };
    auto show = [&](const int *p, float ms, const char *tag) {
      printf("%s b bits 0..11 <- ", tag);
      for (int j = 0; j < 12; j++) printf(p[j] < 8 ? "w%d " : "k%d ", p[j] < 8 ? p[j] : p[j] - 8);
      printf("  %8.3f ms\n", ms);
      fflush(stdout);
    };
    const int starts[3][12] = {{7, 6, 5, 4, 3, 2, 1, 8, 9, 10, 11, 0}, {7, 3, 6, 8, 11, 2, 0, 1, 9, 10, 5, 4}, {10, 3, 5, 0, 8, 9, 2, 7, 11, 1, 4, 6}};
    for (int st = 0; st < 3; st++) {
      int cur[12];
      for (int j = 0; j < 12; j++) cur[j] = starts[st][j];
      float best = eval(cur, 30);
      show(cur, best, "start ");
      for (int sweep = 0; sweep < 3; sweep++) {
        bool improved = false;
        for (int a = 0; a < 12; a++)
          for (int c = a + 1; c < 12; c++) {
            int t[12];
            for (int j = 0; j < 12; j++) t[j] = cur[j];
            std::swap(t[a], t[c]);
            float ms = eval(t, 12);
            if (ms < best - 0.004f) {
              ms = eval(t, 30);   // confirm
              const float again = eval(cur, 30);
              if (ms < again - 0.003f) {
                for (int j = 0; j < 12; j++) cur[j] = t[j];
                best = ms;
                improved = true;
                show(cur, best, "better");
              }
            }
          }
        if (!improved) break;
      }
      show(cur, eval(cur, 40), "final ");
    }
    const int lib[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
    show(lib, eval(lib, 40), "lib   ");
    return 0;
  }
  if (getenv("PROBE_DELTA")) {   // dst = src + delta inside ONE allocation: which address relation of the two streams matters?
    char *big;
    const size_t two_g = (size_t)batch * 65536 * 8;
    CK(hipMalloc(&big, 2 * two_g + (256u << 20)));
    CK(hipMemset(big, 0, 2 * two_g + (256u << 20)));
    const long deltas[] = {0, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 2 << 20, 4 << 20, 8 << 20,
                           16 << 20, 32 << 20, 64 << 20, 128 << 20, (128 << 20) + 524288, (128 << 20) + 4096};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; round++)
      for (int far = 0; far < 2; far++)
        for (long d : deltas) {
          const cpx *src = (const cpx *)big;
          cpx *dst = (cpx *)(big + (far ? two_g : 0) + d);
          auto launch = [&] { hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(cus), dim3(256), 0, 0, src, dst, slots, tabs, batch, (unsigned long long *)nullptr); };
          for (int i = 0; i < 6; i++) launch();
          CK(hipEventRecord(e0));
          for (int i = 0; i < 20; i++) launch();
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          printf("dst = src + %s%10ld   %8.3f ms\n", far ? "2 GiB + " : "        ", d, ms / 20);
        }
    return 0;
  }
  if (getenv("PROBE_OOP")) {   // in place against out of place, interleaved
    cpx *out2;
    CK(hipMalloc(&out2, batch * 65536 * 8));
    CK(hipMemset(out2, 0, batch * 65536 * 8));
    for (int round = 0; round < (getenv("PROBE_PP") ? 0 : 4); round++) {
      g_out = nullptr;
      run<0>("full, in place", data, slots, tabs, dbg, batch, cus);
      run<16>("full + stamps, in place", data, slots, tabs, dbg, batch, cus);
      g_out = out2;
      run<0>("full, OUT OF PLACE", data, slots, tabs, dbg, batch, cus);
      run<16>("full + stamps, OUT OF PLACE", data, slots, tabs, dbg, batch, cus);
    }
    g_out = nullptr;
    // ping-pong: A -> B, B -> A (every buffer is read and written in turn, as a caller alternating directions would)
    for (int round = 0; round < 4; round++) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int mode = 0; mode < 2; mode++) {
        auto launch = [&](int i) {
          const cpx *src = mode == 0 ? data : ((i & 1) ? out2 : data);
          cpx *dst = mode == 0 ? data : ((i & 1) ? data : out2);
          hipLaunchKernelGGL((k_fft_res16<true, false, 0>), dim3(cus), dim3(256), 0, 0, src, dst, slots, tabs, batch, (unsigned long long *)nullptr);
        };
        for (int i = 0; i < 10; i++) launch(i);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 40; i++) launch(i);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8.3f ms  %6.2f TB/s alg\n", mode ? "ping-pong A -> B, B -> A" : "in place", ms / 40, batch * 65536.0 * 16 / (ms / 40) * 1e-9);
      }
    }
    return 0;
  }
#ifndef PROBE_OOP_ONLY   // -DPROBE_OOP_ONLY: compile the two instantiations above only
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
      if (i & 1) hipLaunchKernelGGL((k_fft_res16<false, false, 0>), dim3(cus), dim3(256), 0, 0, data, g_out ? g_out : data, slots, tabs, batch, (unsigned long long *)nullptr);
      else hipLaunchKernelGGL((k_fft_res16<true, true, 0>), dim3(cus), dim3(256), 0, 0, data, g_out ? g_out : data, slots, tabs, batch, (unsigned long long *)nullptr);
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
#endif
  return 0;
}
