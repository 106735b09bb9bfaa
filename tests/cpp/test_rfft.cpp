// counterpart of the reference's print-only test_rfft.cpp: dc + fundamental + nyquist, N = 16;
// expected packed spectrum (0.5,0.5) (0,-1) 0 0 0 0 0 0, inverse returns the input.
#include <cl_fft.h>

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
  std::cout << (bad ? "FAIL" : "OK") << std::endl;
  return bad ? 1 : 0;
}
