// C++ surface check for cl_conv::Clpconv / Cldconv: the reference has no convolution test at all
// (SURVEY.md §4).  Checks the partitioned output against a direct float64 linear convolution with
// the reference's DC/Nyquist half-gain removed by using a zero-mean, Nyquist-free construction:
// we instead compare against the per-block closed form (Y[0]*=0.5, Y[pts]*=0.5) evaluated by DFT.
#include <cl_dconv.h>

#include <cmath>
#include <complex>
#include <iostream>
#include <vector>

typedef std::complex<double> cd;
static std::vector<cd> dft(const std::vector<cd> &x, int sign) {
  size_t n = x.size();
  std::vector<cd> y(n);
  for (size_t k = 0; k < n; k++) {
    cd a = 0;
    for (size_t j = 0; j < n; j++) a += x[j] * std::polar(1.0, sign * 2 * M_PI * (double)(k * j % n) / n);
    y[k] = a;
  }
  return y;
}
static int g_msgs = 0;
static void on_err(std::string s, void *d) { g_msgs++; (void)s; (void)d; }

int main() {
  cl_device_id ids[32];
  cl_uint num = 0;
  if (clGetDeviceIDs(NULL, CL_DEVICE_TYPE_ALL, 32, ids, &num) != CL_SUCCESS) return 2;
  const int pts = 16, nparts = 3, blocks = 7, cvs = pts * nparts;
  cl_conv::Clpconv conv(ids[0], cvs, pts, on_err, (void *)&g_msgs);
  if (conv.get_cl_err() != CL_SUCCESS) return 1;
  std::vector<float> ir(cvs), in(pts * blocks), out(pts * blocks);
  unsigned s = 7;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.f - 0.5f; };
  for (auto &v : ir) v = rnd();
  for (auto &v : in) v = rnd();
  if (conv.push_ir(ir.data()) != CL_SUCCESS) return 1;
  for (int b = 0; b < blocks; b++)
    if (conv.convolution(&out[b * pts], &in[b * pts]) != CL_SUCCESS) return 1;
  // closed-form model
  const int L = 2 * pts;
  std::vector<std::vector<cd>> H(nparts), X(blocks);
  for (int p = 0; p < nparts; p++) {
    std::vector<cd> h(L, 0.0);
    for (int i = 0; i < pts; i++) h[i] = ir[p * pts + i];
    H[p] = dft(h, -1);
  }
  std::vector<double> tail(pts, 0.0);
  int bad = 0;
  double worst = 0;
  for (int t = 0; t < blocks; t++) {
    std::vector<cd> x(L, 0.0);
    for (int i = 0; i < pts; i++) x[i] = in[t * pts + i];
    X[t] = dft(x, -1);
    std::vector<cd> Y(L, 0.0);
    for (int a = 0; a < nparts && t - a >= 0; a++)
      for (int k = 0; k < L; k++) Y[k] += X[t - a][k] * H[a][k];
    Y[0] *= 0.5;
    Y[pts] *= 0.5;
    std::vector<cd> y = dft(Y, +1);
    for (int i = 0; i < pts; i++) {
      double want = y[i].real() / L + tail[i];
      worst = std::fmax(worst, std::fabs(want - out[t * pts + i]));
      tail[i] = y[pts + i].real() / L;
    }
  }
  if (worst > 2e-6) bad++;
  // direct convolution: y[n] = sum_k h[k] x[n-1-k] (cl_dconv.cpp:32-43)
  const int irsize = 24, vsize = 8, dblocks = 9;
  cl_conv::Cldconv dc(ids[0], irsize, vsize);
  std::vector<float> h(irsize), x(vsize * dblocks), y(vsize * dblocks);
  for (auto &v : h) v = rnd();
  for (auto &v : x) v = rnd();
  if (dc.push_ir(h.data()) != CL_SUCCESS) return 1;
  for (int b = 0; b < dblocks; b++)
    if (dc.convolution(&y[b * vsize], &x[b * vsize]) != CL_SUCCESS) return 1;
  for (int n = 0; n < vsize * dblocks; n++) {
    double want = 0;
    for (int k = 0; k < irsize; k++)
      if (n - 1 - k >= 0) want += (double)h[k] * x[n - 1 - k];
    if (std::fabs(want - y[n]) > 2e-6) bad++;
  }
  // error callback by value on a bad geometry
  cl_conv::Clpconv wrong(ids[0], 100, 24, on_err, (void *)&g_msgs);
  if (wrong.get_cl_err() != CL_INVALID_VALUE || g_msgs != 1) bad++;
  std::cout << "pconv max abs err " << worst << (bad ? "  FAIL" : "  OK") << std::endl;
  return bad ? 1 : 0;
}
