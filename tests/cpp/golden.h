// Loader for the reference's golden vectors (tests/golden/ref/*.bin, written by oracle/ref_driver.cpp from
// the unmodified reference on the MI355X's OpenCL device) and the 1e-6 parity criterion of SURVEY.md
// section 8d, for the C++ surface programs.
#pragma once
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace golden {

inline std::string dir() {
  char buf[4096];
  ssize_t n = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
  std::string p = n > 0 ? std::string(buf, (size_t)n) : std::string(".");
  for (int up = 0; up < 3; up++) p = p.substr(0, p.find_last_of('/'));   // tests/cpp/build/prog -> tests
  return p + "/golden/ref/";
}

inline std::vector<float> load_f32(const std::string &name) {
  std::vector<float> v;
  FILE *f = fopen((dir() + name + ".bin").c_str(), "rb");
  if (!f) {
    printf("golden vector %s not found under %s\n", name.c_str(), dir().c_str());
    return v;
  }
  fseek(f, 0, SEEK_END);
  long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize((size_t)bytes / 4);
  if (fread(v.data(), 4, v.size(), f) != v.size()) v.clear();
  fclose(f);
  return v;
}

// fixture PRNG of SURVEY.md section 8c
struct Lcg {
  uint32_t s;
  explicit Lcg(uint32_t seed) : s(seed) {}
  uint32_t next() { return s = s * 1664525u + 1013904223u; }
  float sym() { return (float)(next() >> 8) / 8388608.0f - 1.0f; }
  float half() { return (float)(next() >> 8) / 16777216.0f - 0.5f; }
};

// ||y - ref||2 / ||ref||2 <= tol and max|y - ref| / max|ref| <= tol over n floats
inline bool parity(const float *y, const float *ref, size_t n, double tol, const char *what) {
  double num = 0, den = 0, mx = 0, mr = 0;
  for (size_t i = 0; i < n; i++) {
    double d = (double)y[i] - ref[i];
    num += d * d;
    den += (double)ref[i] * ref[i];
    mx = std::fmax(mx, std::fabs(d));
    mr = std::fmax(mr, std::fabs((double)ref[i]));
  }
  const double l2 = std::sqrt(num / (den > 0 ? den : 1)), mxr = mx / (mr > 0 ? mr : 1);
  printf("%-34s relL2 %.2e  max %.2e  (bar %.0e) %s\n", what, l2, mxr, tol, (l2 <= tol && mxr <= tol) ? "ok" : "MISMATCH");
  return l2 <= tol && mxr <= tol;
}

}  // namespace golden
