// counterpart of the reference's print-only test_rfft.cpp: dc + fundamental + nyquist, N = 16;
// expected packed spectrum (0.5,0.5) (0,-1) 0 0 0 0 0 0, inverse returns the input.
#include <cl_fft.h>

#include "golden.h"

#include <cmath>
#include <iomanip>
#include <iostream>
#include <vector>

#define DEVID 0
#define N 16
using namespace cl_fft;

int main() {
  cl_device_id device_ids[32];
  cl_uint num = 0;
  int err = clGetDeviceIDs(NULL, CL_DEVICE_TYPE_ALL, 32, device_ids, &num);
  if (err != CL_SUCCESS) {
    std::cout << "failed to find a device! " << cl_error_string(err) << std::endl;
    return 2;
  }
  Clrfft dft(device_ids[DEVID], N, true), idft(device_ids[DEVID], N, false);
  if ((err = dft.get_error()) != 0 || (err = idft.get_error()) != 0) {
    std::cout << cl_error_string(err) << std::endl;
    return 1;
  }
  std::vector<std::complex<float>> spec(N / 2);
  std::vector<float> sig(N), in(N);
  for (int i = 0; i < N; i++) in[i] = sig[i] = 0.5 + sin(i * 2 * PI / N) + 0.5 * cos(i * PI);
  if (dft.transform(spec.data(), sig.data()) != 0) return 1;
  std::cout << std::fixed << std::setprecision(3) << "spec =[";
  for (int i = 0; i < N / 2; i++) std::cout << spec[i] << (i < N / 2 - 1 ? "," : "]\n");
  int bad = 0;
  for (int i = 0; i < N / 2; i++) {
    std::complex<float> want = i == 0 ? std::complex<float>(0.5f, 0.5f) : (i == 1 ? std::complex<float>(0.f, -1.f) : 0.f);
    if (std::abs(spec[i] - want) > 1e-6f) bad++;
  }
  std::vector<float> back(N, 0.f);
  if (idft.transform(spec.data(), back.data()) != 0) return 1;
  for (int i = 0; i < N; i++)
    if (std::fabs(back[i] - in[i]) > 1e-6f) bad++;
  // in-place form through the virtual transform(c) (cl_fft.h:104-109)
  std::vector<std::complex<float>> buf(N / 2);
  for (int i = 0; i < N; i++) reinterpret_cast<float *>(buf.data())[i] = in[i];
  Clcfft *base = &dft;
  if (base->transform(buf.data()) != 0) return 1;
  for (int i = 0; i < N / 2; i++)
    if (std::abs(buf[i] - spec[i]) > 1e-6f) bad++;
  // size 2048 against the reference's own outputs (G5: LCG(12345) forward, round trip; inverse of the
  // arbitrary packed spectrum LCG(777)) — includes the un-conjugated self-paired bin M/2
  {
    const int S = 2048, M = S / 2;
    Clrfft f(device_ids[DEVID], S, true), inv(device_ids[DEVID], S, false);
    golden::Lcg r(12345);
    std::vector<float> x(S), back(S), arbout(S);
    for (auto &v : x) v = r.sym();
    std::vector<std::complex<float>> sp(M);
    if (f.transform(sp.data(), x.data()) != 0) return 1;
    std::vector<std::complex<float>> tmp = sp;
    if (inv.transform(tmp.data(), back.data()) != 0) return 1;
    golden::Lcg r2(777);
    std::vector<std::complex<float>> arb(M);
    for (auto &c : arb) {
      float re = r2.sym();
      float im = r2.sym();
      c = std::complex<float>(re, im);
    }
    if (inv.transform(arb.data(), arbout.data()) != 0) return 1;
    const std::vector<float> gf = golden::load_f32("g5_rfft2048_fwd"), gr = golden::load_f32("g5_rfft2048_rt"),
                             ga = golden::load_f32("g5_rfft2048_invarb");
    if (gf.size() != (size_t)S || gr.size() != (size_t)S || ga.size() != (size_t)S) bad++;
    else {
      bad += !golden::parity(reinterpret_cast<float *>(sp.data()), gf.data(), S, 1e-6, "Clrfft 2048 forward vs reference");
      bad += !golden::parity(back.data(), gr.data(), S, 1e-6, "Clrfft 2048 round trip vs reference");
      bad += !golden::parity(arbout.data(), ga.data(), S, 1e-6, "Clrfft 2048 inverse (arbitrary) vs reference");
    }
  }
  std::cout << (bad ? "FAIL" : "OK") << std::endl;
  return bad ? 1 : 0;
}
