// C++ surface check, counterpart of the reference's print-only test_cfft.cpp (N = 16 sine),
// but asserting: forward spectrum = (0,-0.5) at bin 1, (0,+0.5) at bin 15, inverse returns
// the input.  Uses only what a caller of the reference uses: clGetDeviceIDs, clGetDeviceInfo,
// cl_fft::Clcfft, get_error, transform, cl_error_string.
#include <cl_fft.h>

#include "golden.h"

#include <cmath>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <vector>

#define DEVID 0
#define N 16
using namespace cl_fft;

int main() {
  cl_device_id device_ids[32];
  cl_uint num = 0;
  char name[128];
  int err = clGetDeviceIDs(NULL, CL_DEVICE_TYPE_ALL, 32, device_ids, &num);
  if (err != CL_SUCCESS) {
    std::cout << "failed to find a device! " << cl_error_string(err) << std::endl;
    return 2;
  }
  clGetDeviceInfo(device_ids[DEVID], CL_DEVICE_NAME, 128, name, NULL);
  std::cout << "using device " << DEVID << ":" << name << std::endl;

  Clcfft dft(device_ids[DEVID], N, true), idft(device_ids[DEVID], N, false);
  if ((err = dft.get_error()) != 0 || (err = idft.get_error()) != 0) {
    std::cout << cl_error_string(err) << std::endl;
    return 1;
  }
  std::vector<std::complex<float>> sig(N), in(N);
  for (int i = 0; i < N; i++) in[i] = sig[i] = std::complex<float>((float)sin(i * 2 * PI / N), 0.f);
  if ((err = dft.transform(sig.data())) != 0) return 1;
  std::cout << std::fixed << std::setprecision(3) << "spec =[";
  for (int i = 0; i < N; i++) std::cout << sig[i] << (i < N - 1 ? "," : "]\n");
  int bad = 0;
  for (int i = 0; i < N; i++) {
    std::complex<float> want(0.f, i == 1 ? -0.5f : (i == N - 1 ? 0.5f : 0.f));
    if (std::abs(sig[i] - want) > 1e-6f) bad++;
  }
  if ((err = idft.transform(sig.data())) != 0) return 1;
  for (int i = 0; i < N; i++)
    if (std::abs(sig[i] - in[i]) > 1e-6f) bad++;
  // batched extension: 5 transforms at once equal 5 single calls
  Clcfft big(device_ids[DEVID], 1024, true);
  std::vector<std::complex<float>> a(5 * 1024), b;
  unsigned s = 1;
  for (auto &c : a) { s = s * 1664525u + 1013904223u; c = std::complex<float>((s >> 8) / 8388608.f - 1.f, 0.25f); }
  b = a;
  if (big.transform(a.data(), 5) != 0) return 1;
  for (int k = 0; k < 5; k++) big.transform(b.data() + 1024 * k);
  for (size_t i = 0; i < a.size(); i++)
    if (a[i] != b[i]) bad++;
  // config 1 of BASELINE.json: N = 1024 through the class surface against the reference's own outputs
  // (G3: input LCG(12345), forward / inverse / round trip of the unmodified reference)
  {
    Clcfft f1k(device_ids[DEVID], 1024, true), i1k(device_ids[DEVID], 1024, false);
    golden::Lcg r(12345);
    std::vector<std::complex<float>> x(1024);
    for (auto &c : x) {
      float re = r.sym();
      float im = r.sym();
      c = std::complex<float>(re, im);
    }
    std::vector<std::complex<float>> y = x, z = x;
    if (f1k.transform(y.data()) != 0 || i1k.transform(z.data()) != 0) return 1;
    std::vector<std::complex<float>> rt = y;
    if (i1k.transform(rt.data()) != 0) return 1;
    const std::vector<float> gf = golden::load_f32("g3_cfft1024_fwd"), gi = golden::load_f32("g3_cfft1024_inv"),
                             gr = golden::load_f32("g3_cfft1024_rt");
    if (gf.size() != 2048 || gi.size() != 2048 || gr.size() != 2048) bad++;
    else {
      bad += !golden::parity(reinterpret_cast<float *>(y.data()), gf.data(), 2048, 1e-6, "Clcfft 1024 forward vs reference");
      bad += !golden::parity(reinterpret_cast<float *>(z.data()), gi.data(), 2048, 1e-6, "Clcfft 1024 inverse vs reference");
      bad += !golden::parity(reinterpret_cast<float *>(rt.data()), gr.data(), 2048, 1e-6, "Clcfft 1024 round trip vs reference");
    }
  }
  // error convention: a bad size is reported through get_error(), nothing throws
  Clcfft wrong(device_ids[DEVID], 1, true);
  if (wrong.get_error() != CL_INVALID_VALUE || std::string(cl_error_string(wrong.get_error())) != "Invalid value") bad++;
  std::cout << (bad ? "FAIL" : "OK") << std::endl;
  return bad ? 1 : 0;
}
