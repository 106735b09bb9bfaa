// Golden-vector generator that drives the UNMODIFIED reference classes.
//
// TEST INFRASTRUCTURE, not product code.  This file is ours; it is compiled
// (oracle/Makefile, target `ref`) together with the reference sources where
// they lie under /root/reference (cl_fft.cpp, cl_conv.cpp, cl_dconv.cpp), the
// products go only to oracle/_ref/ (git-ignored).  It needs an OpenCL device
// at RUN time: the authoring container has none, the MI355X box has the AMD
// OpenCL runtime, so `oracle/_ref/ref_driver <outdir>` is run there and the
// numeric outputs are committed under tests/golden/ref/.
//
// It only uses the reference's public/protected class surface:
//   cl_fft::Clcfft / Clrfft     (cl_fft.h:29-111)
//   cl_conv::Clpconv            (cl_conv.h:124-188)
//   cl_conv::Cldconv            (cl_dconv.h:17-66)
// Inputs follow the fixture PRNG of SURVEY.md §8c (LCG 1664525/1013904223).
#include <cl_conv.h>
#include <cl_dconv.h>
#include <cl_fft.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

typedef std::complex<float> cf;

static std::string g_dir;
static FILE *g_manifest = nullptr;
static bool g_first = true;

static void put(const std::string &name, const void *p, size_t count,
                const char *dtype, const std::string &shape,
                const std::string &note) {
  size_t esz = (!strcmp(dtype, "f64")) ? 8 : 4;
  std::string path = g_dir + "/" + name + ".bin";
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) {
    perror(path.c_str());
    exit(2);
  }
  fwrite(p, esz, count, f);
  fclose(f);
  fprintf(g_manifest, "%s\n  \"%s\": {\"dtype\": \"%s\", \"shape\": [%s], \"note\": \"%s\"}",
          g_first ? "" : ",", name.c_str(), dtype, shape.c_str(), note.c_str());
  g_first = false;
}
static void put_f32(const std::string &name, const float *p, size_t n,
                    const std::string &note) {
  put(name, p, n, "f32", std::to_string(n), note);
}
static void put_c64(const std::string &name, const cf *p, size_t n,
                    const std::string &note) {
  put(name, p, 2 * n, "f32", std::to_string(n) + ", 2", note);
}

// fixture PRNG
struct Lcg {
  uint32_t s;
  explicit Lcg(uint32_t seed) : s(seed) {}
  uint32_t next() { return s = s * 1664525u + 1013904223u; }
  float sym() { return (float)(next() >> 8) / 8388608.0f - 1.0f; }        // [-1,1)
  float half() { return (float)(next() >> 8) / 16777216.0f - 0.5f; }      // [-.5,.5)
};

// decimation used for large vectors: first 64, last 64, every 16th element
static std::vector<cf> decimate(const std::vector<cf> &v) {
  std::vector<cf> d;
  size_t n = v.size();
  for (size_t i = 0; i < 64; i++) d.push_back(v[i]);
  for (size_t i = n - 64; i < n; i++) d.push_back(v[i]);
  for (size_t i = 0; i < n; i += 16) d.push_back(v[i]);
  return d;
}
static void checks(const std::vector<cf> &v, double out[3]) {
  double sr = 0, si = 0, e = 0;
  for (auto &c : v) {
    sr += c.real();
    si += c.imag();
    e += (double)c.real() * c.real() + (double)c.imag() * c.imag();
  }
  out[0] = sr;
  out[1] = si;
  out[2] = e;
}

// exposes the protected bit-reversal table of the reference (cl_fft.cpp:96-104)
struct PeekCfft : cl_fft::Clcfft {
  PeekCfft(cl_device_id d, int n) : Clcfft(d, n, true) {}
  std::vector<int> bitrev() {
    std::vector<int> t(N);
    clEnqueueReadBuffer(commands, b, CL_TRUE, 0, sizeof(cl_int) * N, t.data(),
                        0, NULL, NULL);
    return t;
  }
  std::vector<cf> twiddle() {
    std::vector<cf> t(N);
    clEnqueueReadBuffer(commands, w, CL_TRUE, 0, sizeof(cl_float2) * N,
                        t.data(), 0, NULL, NULL);
    return t;
  }
};

// ---- G11: direct convolution over several ring cycles -------------------------
// With irsize % vsize == 0 the reference's write point takes the values 0, vsize, ..., irsize only,
// so its defective `wp > irsize` branch (cl_dconv.cpp:112-119) never runs; the delay ring (and, in
// the two-input form, the coefficient ring) is uninitialised device memory only until every slot
// has been written once, i.e. from block irsize / vsize + 1 on the output is fully defined.
static void run_g11(cl_device_id dev) {
  auto run_dconv = [&](const std::string &tag, int irsize, int vsize, int blocks, bool tv) {
    cl_conv::Cldconv c(dev, irsize, vsize);
    if (c.get_cl_err() != CL_SUCCESS) {
      fprintf(stderr, "Cldconv setup error %d\n", c.get_cl_err());
      exit(6);
    }
    Lcg r(tv ? 13 : 11);
    std::vector<float> ir(irsize), in(blocks * vsize), in2(blocks * vsize), out(blocks * vsize);
    for (auto &v : ir) v = r.half();
    for (auto &v : in) v = r.half();
    for (auto &v : in2) v = r.half();
    c.push_ir(ir.data());
    for (int b = 0; b < blocks; b++) {
      if (tv) c.convolution(&out[b * vsize], &in[b * vsize], &in2[b * vsize]);
      else c.convolution(&out[b * vsize], &in[b * vsize]);
    }
    const std::string from = std::to_string(irsize / vsize + 1);
    put_f32(tag + "_ir", ir.data(), ir.size(), "impulse response (push_ir)");
    put_f32(tag + "_in", in.data(), in.size(), "input blocks");
    if (tv) put_f32(tag + "_in2", in2.data(), in2.size(), "second input blocks");
    put_f32(tag + "_out", out.data(), out.size(),
            std::string(tv ? "Cldconv::convolution(out,in1,in2)" : "Cldconv::convolution(out,in)") +
                " per block; independent of uninitialised device memory from block " + from + " on");
  };
  run_dconv("g11_dconv_i16_v8", 16, 8, 12, false);      // 3 ring slots, 4 cycles
  run_dconv("g11_dconv_i1024_v64", 1024, 64, 56, false);  // 17 ring slots, > 3 cycles
  run_dconv("g11_dconv_i64_v64", 64, 64, 8, false);       // 2 ring slots
  run_dconv("g11_tvdconv_i16_v8", 16, 8, 12, true);
  run_dconv("g11_tvdconv_i256_v32", 256, 32, 30, true);
}

// PCI bus number of an OpenCL device (AMD's cl_amd_device_attribute_query: CL_DEVICE_TOPOLOGY_AMD), -1 if unknown
static int pci_bus_of(cl_device_id d) {
  struct {
    cl_uint type;
    cl_char unused[17];
    cl_char bus, device, function;
  } topo;
  memset(&topo, 0, sizeof(topo));
  if (clGetDeviceInfo(d, 0x4037 /* CL_DEVICE_TOPOLOGY_AMD */, sizeof(topo), &topo, NULL) != CL_SUCCESS || topo.type != 1) return -1;
  return (int)(unsigned char)topo.bus;
}

// ---- timing of the reference's own calls (bench.py / tools/rt_sweep.py run these live; nothing is written) ----------
// Clrfft::transform(c, r), forward / inverse alternating (cl_fft.cpp:267-296): "<us per call> <G real samples/s> <calls>"
static int time_rfft(cl_device_id dev, int size, const char *name) {
  cl_fft::Clrfft f(dev, size, true), g(dev, size, false);
  if (f.get_error() != CL_SUCCESS || g.get_error() != CL_SUCCESS) return 4;
  std::vector<float> r(size);
  std::vector<cf> c(size / 2);
  Lcg l(12345);
  for (auto &v : r) v = l.sym();
  for (int k = 0; k < 4; k++) (k & 1) ? g.transform(c.data(), r.data()) : f.transform(c.data(), r.data());
  const int reps = 200;
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < reps; k++) (k & 1) ? g.transform(c.data(), r.data()) : f.transform(c.data(), r.data());
  auto t1 = std::chrono::steady_clock::now();
  const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
  printf("%.3f %.6f %d %s\n", us, size / us * 1e-3, reps, name);
  return 0;
}
// Clpconv::convolution per block, one instance (cl_conv.cpp:393-458 static / 460-548 time-varying): microseconds per block
static double time_pconv_block(cl_device_id dev, int pts, int cvs, bool tv, int blocks) {
  cl_conv::Clpconv c(dev, cvs, pts);
  if (c.get_cl_err() != CL_SUCCESS) return -1.0;
  Lcg l(7);
  std::vector<float> ir((size_t)(cvs / pts) * pts), a(pts), b(pts), out(pts);
  const float g = 1.0f / std::sqrt((float)cvs);
  for (auto &v : ir) v = l.half() * g;
  for (auto &v : a) v = l.sym();
  for (auto &v : b) v = l.half() * g;
  if (!tv && c.push_ir(ir.data()) != CL_SUCCESS) return -1.0;
  for (int k = 0; k < 3; k++) tv ? c.convolution(out.data(), a.data(), b.data()) : c.convolution(out.data(), a.data());
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < blocks; k++) tv ? c.convolution(out.data(), a.data(), b.data()) : c.convolution(out.data(), a.data());
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / blocks;
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: ref_driver <outdir> [device-index | pci:<bus>] [g11 | time | time-rfft <size> | time-pconv <pts> <cvs> <blocks> [tv] | rt-sweep <blocks>]\n");
    return 2;
  }
  g_dir = argv[1];
  cl_device_id ids[32];
  cl_uint num = 0;
  int err = clGetDeviceIDs(NULL, CL_DEVICE_TYPE_ALL, 32, ids, &num);
  if (err != CL_SUCCESS || num == 0) {
    fprintf(stderr, "no OpenCL device: %s\n", cl_fft::cl_error_string(err));
    return 3;
  }
  // the device: an index into clGetDeviceIDs' list (as the reference's callers pick it), or pci:<bus> — the OpenCL device on
  // that PCI bus, which is how bench.py names the GPU its own rank runs on (OpenCL's enumeration need not follow HIP's)
  int devidx = 0;
  if (argc > 2 && !strncmp(argv[2], "pci:", 4)) {
    const int want = (int)strtol(argv[2] + 4, NULL, 16);
    devidx = -1;
    for (cl_uint i = 0; i < num; i++)
      if (pci_bus_of(ids[i]) == want) devidx = (int)i;
    if (devidx < 0) {
      fprintf(stderr, "no OpenCL device on PCI bus %02x\n", want);
      return 3;
    }
  } else if (argc > 2) {
    devidx = atoi(argv[2]);
  }
  if (devidx < 0 || devidx >= (int)num) {
    fprintf(stderr, "device index %d out of range (%u devices)\n", devidx, num);
    return 3;
  }
  char name[160] = {0};
  clGetDeviceInfo(ids[devidx], CL_DEVICE_NAME, 128, name, NULL);
  {
    const int bus = pci_bus_of(ids[devidx]);
    if (bus >= 0) snprintf(name + strlen(name), sizeof(name) - strlen(name), " [pci bus %02x]", bus);
  }
  fprintf(stderr, "ref_driver: %u device(s), using %d: %s\n", num, devidx, name);
  cl_device_id dev = ids[devidx];

  if (argc > 3 && !strcmp(argv[3], "time-rfft")) return time_rfft(dev, argc > 4 ? atoi(argv[4]) : 16384, name);
  if (argc > 6 && !strcmp(argv[3], "time-pconv")) {
    // "<us per block> <G samples/s of one instance> <blocks> <device>"
    const int pts = atoi(argv[4]), cvs = atoi(argv[5]), blocks = atoi(argv[6]);
    const double us = time_pconv_block(dev, pts, cvs, argc > 7 && !strcmp(argv[7], "tv"), blocks);
    if (us < 0) return 4;
    printf("%.3f %.6f %d %s\n", us, pts / us * 1e-3, blocks, name);
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "rt-sweep")) {
    // the grid of the reference's own benchmark (csound/tests.py:10-36: cltvconv, partition M = 2^{9,11,13,15} x filter
    // length L = 2^16..2^22): one line per cell, "<M> <log2 L> <us per block>", time-varying convolution, one instance
    const int blocks = argc > 4 ? atoi(argv[4]) : 50;
    printf("# %s\n", name);
    for (int m : {9, 11, 13, 15})
      for (int l = 16; l <= 22; l++) {
        const double us = time_pconv_block(dev, 1 << m, 1 << l, true, blocks);
        printf("%d %d %.3f\n", 1 << m, l, us);
        fflush(stdout);
      }
    return 0;
  }

  if (argc > 3 && !strcmp(argv[3], "time")) {
    // only the timing of the reference's own path (bench.py runs this live where an OpenCL device exists): one line on stdout,
    // "<us per transform> <Gsamples/s> <transforms timed> <device name>"; nothing is written to <outdir>
    const int N = 65536;
    cl_fft::Clcfft f(dev, N, true);
    if (f.get_error() != CL_SUCCESS) {
      fprintf(stderr, "Clcfft setup: %s\n", cl_fft::cl_error_string(f.get_error()));
      return 4;
    }
    std::vector<cf> x(N);
    unsigned sd = 12345;
    for (auto &c : x) {
      sd = sd * 1664525u + 1013904223u;
      const float re = (sd >> 8) / 8388608.f - 1.f;
      sd = sd * 1664525u + 1013904223u;
      c = cf(re, (sd >> 8) / 8388608.f - 1.f);
    }
    cl_fft::Clcfft g(dev, N, false);
    for (int k = 0; k < 4; k++) (k & 1 ? g : f).transform(x.data());   // warm-up; forward / inverse keep the data O(1)
    const int reps = 200;
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; k++) (k & 1 ? g : f).transform(x.data());
    auto t1 = std::chrono::steady_clock::now();
    const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
    printf("%.3f %.6f %d %s\n", us, N / us * 1e-3, reps, name);
    return 0;
  }
  const bool only11 = argc > 3 && !strcmp(argv[3], "g11");   // only the G11 vectors (manifest_g11.json)
  g_manifest = fopen((g_dir + (only11 ? "/manifest_g11.json" : "/manifest.json")).c_str(), "w");
  fprintf(g_manifest, "{\n  \"_device\": \"%s\"", name);
  g_first = false;
  if (only11) {
    run_g11(dev);
    fprintf(g_manifest, "\n}\n");
    fclose(g_manifest);
    fprintf(stderr, "ref_driver: G11 done\n");
    return 0;
  }

  const double PI = cl_fft::PI;

  // ---- G1: test_cfft.cpp input, N=16 ------------------------------------
  {
    const int N = 16;
    cl_fft::Clcfft f(dev, N, true), i(dev, N, false);
    if (f.get_error() || i.get_error()) {
      fprintf(stderr, "Clcfft setup error %d %d (%s)\n", f.get_error(),
              i.get_error(), f.get_log());
      return 4;
    }
    std::vector<cf> s(N);
    for (int k = 0; k < N; k++) s[k] = cf((float)sin(k * 2 * PI / N), 0.f);
    put_c64("g1_cfft16_in", s.data(), N, "test_cfft.cpp:54-56 input");
    f.transform(s.data());
    put_c64("g1_cfft16_fwd", s.data(), N, "forward (scaled 1/N)");
    i.transform(s.data());
    put_c64("g1_cfft16_inv", s.data(), N, "inverse of forward");
  }
  // ---- G2: test_rfft.cpp input, N=16 ------------------------------------
  {
    const int N = 16;
    cl_fft::Clrfft f(dev, N, true), i(dev, N, false);
    std::vector<float> sig(N);
    std::vector<cf> spec(N / 2);
    for (int k = 0; k < N; k++)
      sig[k] = 0.5 + sin(k * 2 * PI / N) + 0.5 * cos(k * PI);
    put_f32("g2_rfft16_in", sig.data(), N, "test_rfft.cpp:54-57 input");
    f.transform(spec.data(), sig.data());
    put_c64("g2_rfft16_fwd", spec.data(), N / 2, "packed spectrum");
    std::vector<float> back(N);
    i.transform(spec.data(), back.data());
    put_f32("g2_rfft16_inv", back.data(), N, "inverse of forward");
  }
  // ---- G3/G4: c2c all powers of two 2..65536, LCG seed 12345 -------------
  for (int lg = 1; lg <= 16; lg++) {
    int N = 1 << lg;
    cl_fft::Clcfft f(dev, N, true), i(dev, N, false);
    Lcg r(12345);
    std::vector<cf> x(N);
    for (int k = 0; k < N; k++) {
      float re = r.sym();
      float im = r.sym();
      x[k] = cf(re, im);
    }
    std::vector<cf> y = x, z = x;
    f.transform(y.data());
    i.transform(z.data());
    std::vector<cf> rt = y;
    i.transform(rt.data());
    double c[3];
    std::string tag = "cfft" + std::to_string(N);
    if (N <= 4096) {
      put_c64("g3_" + tag + "_fwd", y.data(), N, "forward of LCG(12345)");
      put_c64("g3_" + tag + "_inv", z.data(), N, "inverse (unscaled) of LCG(12345)");
      put_c64("g3_" + tag + "_rt", rt.data(), N, "inverse(forward(x))");
    } else {
      auto d = decimate(y);
      put_c64("g4_" + tag + "_fwd_dec", d.data(), d.size(), "first64,last64,every16th");
      d = decimate(z);
      put_c64("g4_" + tag + "_inv_dec", d.data(), d.size(), "first64,last64,every16th");
      d = decimate(rt);
      put_c64("g4_" + tag + "_rt_dec", d.data(), d.size(), "first64,last64,every16th");
    }
    checks(y, c);
    put("g4_" + tag + "_fwd_chk", c, 3, "f64", "3", "sum re, sum im, energy");
    checks(z, c);
    put("g4_" + tag + "_inv_chk", c, 3, "f64", "3", "sum re, sum im, energy");
  }
  // ---- G5: r2c/c2r, sizes 4..16384 (real points), LCG seed 12345 --------
  for (int lg = 2; lg <= 17; lg++) {
    int S = 1 << lg, M = S / 2;
    cl_fft::Clrfft f(dev, S, true), i(dev, S, false);
    Lcg r(12345);
    std::vector<float> x(S);
    for (int k = 0; k < S; k++) x[k] = r.sym();
    std::vector<cf> spec(M);
    f.transform(spec.data(), x.data());
    std::vector<float> back(S);
    std::vector<cf> tmp = spec;
    i.transform(tmp.data(), back.data());
    // inverse applied to an arbitrary (non-hermitian-derived) packed spectrum
    Lcg r2(777);
    std::vector<cf> arb(M);
    for (int k = 0; k < M; k++) {
      float re = r2.sym();
      float im = r2.sym();
      arb[k] = cf(re, im);
    }
    std::vector<float> arbout(S);
    std::vector<cf> arbc = arb;
    i.transform(arbc.data(), arbout.data());
    std::string tag = "rfft" + std::to_string(S);
    if (S <= 4096) {
      put_c64("g5_" + tag + "_fwd", spec.data(), M, "packed spectrum of LCG(12345)");
      put_f32("g5_" + tag + "_rt", back.data(), S, "inverse(forward(x))");
      put_f32("g5_" + tag + "_invarb", arbout.data(), S, "inverse of LCG(777) packed spectrum");
    } else {
      auto d = decimate(spec);
      d.push_back(spec[M / 2]);  // the self-paired bin (quirk, cl_fft.cpp:278)
      put_c64("g5_" + tag + "_fwd_dec", d.data(), d.size(), "first64,last64,every16th,+bin M/2");
      std::vector<cf> bc(S / 2), ac(S / 2);
      memcpy(bc.data(), back.data(), S * 4);
      memcpy(ac.data(), arbout.data(), S * 4);
      d = decimate(bc);
      put_c64("g5_" + tag + "_rt_dec", d.data(), d.size(), "real pairs; first64,last64,every16th");
      d = decimate(ac);
      put_c64("g5_" + tag + "_invarb_dec", d.data(), d.size(), "real pairs; first64,last64,every16th");
    }
    double c[3];
    checks(spec, c);
    put("g5_" + tag + "_fwd_chk", c, 3, "f64", "3", "sum re, sum im, energy");
  }
  // ---- G6: bit-reversal + twiddle tables ---------------------------------
  for (int N : {16, 1024, 65536}) {
    PeekCfft p(dev, N);
    auto t = p.bitrev();
    put("g6_bitrev" + std::to_string(N), t.data(), t.size(), "i32",
        std::to_string(N), "cl_fft.cpp:96-104");
    if (N <= 1024) {
      auto w = p.twiddle();
      put_c64("g6_twiddle" + std::to_string(N), w.data(), w.size(), "cl_fft.cpp:86-91 forward");
    }
  }
  // ---- G7/G8: partitioned convolution -----------------------------------
  auto run_pconv = [&](const std::string &tag, int pts, int nparts, int blocks,
                       bool ones) {
    int cvs = pts * nparts;
    cl_conv::Clpconv c(dev, cvs, pts);
    if (c.get_cl_err() != CL_SUCCESS) {
      fprintf(stderr, "Clpconv setup error %d\n", c.get_cl_err());
      exit(5);
    }
    Lcg r(7);
    std::vector<float> ir(cvs), in(blocks * pts), out(blocks * pts);
    for (auto &v : ir) v = ones ? 1.0f / cvs : r.half();
    for (auto &v : in) v = ones ? 1.0f : r.half();
    c.push_ir(ir.data());
    for (int b = 0; b < blocks; b++)
      c.convolution(&out[b * pts], &in[b * pts]);
    put_f32(tag + "_ir", ir.data(), ir.size(), "impulse response");
    put_f32(tag + "_in", in.data(), in.size(), "input blocks");
    put_f32(tag + "_out", out.data(), out.size(), "Clpconv::convolution(out,in) per block");
  };
  run_pconv("g7_pconv_p8_n4", 8, 4, 12, false);
  run_pconv("g7_pconv_p8_n4_ones", 8, 4, 12, true);
  run_pconv("g7_pconv_p2_n3", 2, 3, 9, false);
  run_pconv("g7_pconv_p64_n1", 64, 1, 4, false);
  run_pconv("g8_pconv_p1024_n8", 1024, 8, 24, false);
  // ---- G9: time-varying partitioned convolution --------------------------
  {
    int pts = 8, nparts = 4, blocks = 12, cvs = pts * nparts;
    cl_conv::Clpconv c(dev, cvs, pts);
    Lcg r(7);
    std::vector<float> in1(blocks * pts), in2(blocks * pts), out(blocks * pts);
    for (auto &v : in1) v = r.half();
    for (auto &v : in2) v = r.half();
    for (int b = 0; b < blocks; b++)
      c.convolution(&out[b * pts], &in1[b * pts], &in2[b * pts]);
    put_f32("g9_tvconv_p8_n4_in1", in1.data(), in1.size(), "input 1");
    put_f32("g9_tvconv_p8_n4_in2", in2.data(), in2.size(), "input 2");
    put_f32("g9_tvconv_p8_n4_out", out.data(), out.size(), "Clpconv::convolution(out,in1,in2)");
  }
  {
    int pts = 256, nparts = 5, blocks = 14, cvs = pts * nparts;
    cl_conv::Clpconv c(dev, cvs, pts);
    Lcg r(7);
    std::vector<float> in1(blocks * pts), in2(blocks * pts), out(blocks * pts);
    for (auto &v : in1) v = r.half();
    for (auto &v : in2) v = r.half();
    for (int b = 0; b < blocks; b++)
      c.convolution(&out[b * pts], &in1[b * pts], &in2[b * pts]);
    put_f32("g9_tvconv_p256_n5_in1", in1.data(), in1.size(), "input 1");
    put_f32("g9_tvconv_p256_n5_in2", in2.data(), in2.size(), "input 2");
    put_f32("g9_tvconv_p256_n5_out", out.data(), out.size(), "Clpconv::convolution(out,in1,in2)");
  }
  // ---- G10: direct convolution (first non-wrapping blocks only: the
  //           reference's wrap branch and uninitialised `del` are defects,
  //           SURVEY.md §8a) ------------------------------------------------
  {
    int irsize = 16, vsize = 8, blocks = 2;
    cl_conv::Cldconv c(dev, irsize, vsize);
    Lcg r(7);
    std::vector<float> ir(irsize), in(blocks * vsize), out(blocks * vsize);
    for (auto &v : ir) v = r.half();
    for (auto &v : in) v = r.half();
    c.push_ir(ir.data());
    for (int b = 0; b < blocks; b++)
      c.convolution(&out[b * vsize], &in[b * vsize]);
    put_f32("g10_dconv_ir", ir.data(), ir.size(), "impulse response");
    put_f32("g10_dconv_in", in.data(), in.size(), "input");
    put_f32("g10_dconv_out", out.data(), out.size(),
            "Cldconv::convolution; depends on uninitialised device memory in the reference");
  }
  run_g11(dev);
  // ---- timing of the reference's own path on this device ----------------
  {
    const int N = 65536, reps = 20;
    cl_fft::Clcfft f(dev, N, true);
    std::vector<cf> x(N, cf(0.25f, -0.5f));
    f.transform(x.data());
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; k++) f.transform(x.data());
    auto t1 = std::chrono::steady_clock::now();
    double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
    double v[2] = {us, N / us * 1e-3};
    put("timing_ref_cfft65536", v, 2, "f64", "2",
        "reference Clcfft::transform on this OpenCL device: us per transform, Gsamples/s (PCIe copies included)");
    fprintf(stderr, "reference Clcfft N=65536: %.1f us/transform = %.4f Gsamples/s\n", us, v[1]);
  }
  fprintf(g_manifest, "\n}\n");
  fclose(g_manifest);
  fprintf(stderr, "ref_driver: done\n");
  return 0;
}
