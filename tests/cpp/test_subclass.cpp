// A SUBCLASS written against the reference's header: it reaches the protected members the reference
// declares (cl_fft.h:31-44: N, forward, w, b, data1, data2, commands, fft()) with the calls the reference
// itself makes on them (cl_fft.cpp:155-158: clEnqueueWriteBuffer / fft() / clEnqueueReadBuffer).  The same
// source compiles against the reference's cl_fft.h (oracle/ref_driver.cpp's PeekCfft does exactly this
// there); here it must compile against include/cl_fft.h, and on a GPU the tables must be the reference's
// formulas bit for bit and fft() must give what transform() gives.
#include <cl_fft.h>

#include "golden.h"

#include <cmath>
#include <cstdio>
#include <iostream>
#include <vector>

typedef std::complex<float> cf;

struct PeekCfft : cl_fft::Clcfft {
  PeekCfft(cl_device_id d, int n, bool fwd) : Clcfft(d, n, fwd) {}
  std::vector<int> bitrev() {
    std::vector<int> t(N);
    clEnqueueReadBuffer(commands, b, CL_TRUE, 0, sizeof(cl_int) * N, t.data(), 0, NULL, NULL);
    return t;
  }
  std::vector<cf> twiddle() {
    std::vector<cf> t(N);
    clEnqueueReadBuffer(commands, w, CL_TRUE, 0, sizeof(cl_float2) * N, t.data(), 0, NULL, NULL);
    return t;
  }
  // Clcfft::transform as the reference writes it (cl_fft.cpp:153-161)
  int own_transform(cf *c) {
    clEnqueueWriteBuffer(commands, data1, CL_TRUE, 0, sizeof(cl_float2) * N, c, 0, NULL, NULL);
    int err = fft();
    clEnqueueReadBuffer(commands, data2, CL_TRUE, 0, sizeof(cl_float2) * N, c, 0, NULL, NULL);
    return err;
  }
  bool is_forward() const { return forward; }
};

// ... and a subclass of Clrfft: there the inherited N is size / 2 and fft() is the complex N-point transform of
// data1 -> data2 ALONE (cl_fft.cpp:210, 138-151); the pack / unpack kernels belong to Clrfft::transform (cl_fft.cpp:267-296)
struct PeekRfft : cl_fft::Clrfft {
  PeekRfft(cl_device_id d, int size, bool fwd) : Clrfft(d, size, fwd) {}
  int complex_points() const { return N; }
  int own_fft(cf *c) {
    clEnqueueWriteBuffer(commands, data1, CL_TRUE, 0, sizeof(cl_float2) * N, c, 0, NULL, NULL);
    int err = fft();
    clEnqueueReadBuffer(commands, data2, CL_TRUE, 0, sizeof(cl_float2) * N, c, 0, NULL, NULL);
    return err;
  }
  std::vector<cf> twiddle() {
    std::vector<cf> t(N);
    clEnqueueReadBuffer(commands, w, CL_TRUE, 0, sizeof(cl_float2) * N, t.data(), 0, NULL, NULL);
    return t;
  }
};

int main() {
  cl_device_id ids[32];
  cl_uint num = 0;
  if (clGetDeviceIDs(NULL, CL_DEVICE_TYPE_ALL, 32, ids, &num) != CL_SUCCESS) {
    std::cout << "failed to find a device!" << std::endl;
    return 2;
  }
  int bad = 0;
  for (int n : {16, 1024, 65536}) {
    for (int fwd = 1; fwd >= 0; fwd--) {
      PeekCfft p(ids[0], n, fwd != 0);
      if (p.get_error() != 0 || p.is_forward() != (fwd != 0)) return 1;
      // tables: the reference's expressions (cl_fft.cpp:86-101), bit for bit
      const std::vector<int> br = p.bitrev();
      const std::vector<cf> tw = p.twiddle();
      std::vector<int> want(n);
      want[0] = 0;
      for (int i = 1, m = n / 2; i < n; i <<= 1, m >>= 1)
        for (int j = 0; j < i; j++) want[i + j] = want[j] + m;
      for (int i = 0; i < n; i++) {
        if (br[i] != want[i]) bad++;
        const float re = (float)cos(i * 2 * cl_fft::PI / n), im = (fwd ? -1.f : 1.f) * (float)sin(i * 2 * cl_fft::PI / n);
        if (tw[i].real() != re || tw[i].imag() != im) bad++;
      }
      // fft() on data1 -> data2 against transform()
      golden::Lcg r(12345);
      std::vector<cf> x(n);
      for (auto &c : x) {
        float re = r.sym();
        float im = r.sym();
        c = cf(re, im);
      }
      std::vector<cf> y = x, z = x;
      if (p.own_transform(y.data()) != 0 || p.transform(z.data()) != 0) return 1;
      for (int i = 0; i < n; i++)
        if (y[i] != z[i]) bad++;
      if (n == 1024) {
        const std::vector<float> g = golden::load_f32(fwd ? "g3_cfft1024_fwd" : "g3_cfft1024_inv");
        bad += g.size() != 2048 || !golden::parity(reinterpret_cast<float *>(y.data()), g.data(), 2048, 1e-6, "subclass fft() vs reference");
      }
    }
  }
  // Clrfft(2048): N = 1024 and fft() must be the reference's 1024-point complex transform (G3), packed nothing
  for (int fwd = 1; fwd >= 0; fwd--) {
    PeekRfft p(ids[0], 2048, fwd != 0);
    if (p.get_error() != 0 || p.complex_points() != 1024) return 1;
    const std::vector<cf> tw = p.twiddle();   // the base class's table on N = 1024 points (cl_fft.cpp:86-91)
    for (int i = 0; i < 1024; i++) {
      const float re = (float)cos(i * 2 * cl_fft::PI / 1024), im = (fwd ? -1.f : 1.f) * (float)sin(i * 2 * cl_fft::PI / 1024);
      if (tw[i].real() != re || tw[i].imag() != im) bad++;
    }
    golden::Lcg r(12345);
    std::vector<cf> x(1024);
    for (auto &c : x) {
      float re = r.sym();
      float im = r.sym();
      c = cf(re, im);
    }
    if (p.own_fft(x.data()) != 0) return 1;
    const std::vector<float> g = golden::load_f32(fwd ? "g3_cfft1024_fwd" : "g3_cfft1024_inv");
    bad += g.size() != 2048 || !golden::parity(reinterpret_cast<float *>(x.data()), g.data(), 2048, 1e-6, "Clrfft subclass fft() vs reference cfft");
    // the object still transforms real data afterwards (the sub-plan shares nothing with the packed real plan)
    std::vector<float> re(2048);
    for (auto &v : re) v = r.sym();
    std::vector<cf> spec(1024);
    if (fwd && p.transform(spec.data(), re.data()) != 0) return 1;
  }
  std::cout << (bad ? "FAIL" : "OK") << std::endl;
  return bad ? 1 : 0;
}
