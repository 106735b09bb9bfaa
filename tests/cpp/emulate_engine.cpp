// CPU emulation of the lane-level FFT engine (opencl_fft_amd/csrc/fft_device.hpp):
// runs the very same pass_compute / pass_scatter / pass_gather code lane by lane,
// with a std::vector standing in for LDS, and checks it against a float64 DFT.
// Built and run by tests/test_engine_emulation.py (no GPU needed).
#include "../../opencl_fft_amd/csrc/fft_device.hpp"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

using namespace clfa;

template <int LOGN, int LOGE, int LOGNS, bool FWD>
static void run(std::vector<cpx> &regs, const std::vector<cpx> &tab, std::vector<cpx> &lds) {
  constexpr int E = 1 << LOGE, T = 1 << (LOGN - LOGE);
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  for (int tid = 0; tid < T; tid++)
    pass_compute<LOGN, LOGE, LOGNS, FWD>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, tab);
  if constexpr (LOGNS + LOGR < LOGN) {
    for (int tid = 0; tid < T; tid++)
      pass_scatter<LOGN, LOGE, LOGNS>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid,
                                      [&](int p, cpx v) { lds[lds_pad(p)] = v; });
    for (int tid = 0; tid < T; tid++)
      pass_gather<LOGN, LOGE>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid,
                              [&](int p) { return lds[lds_pad(p)]; });
    run<LOGN, LOGE, LOGNS + LOGR, FWD>(regs, tab, lds);
  }
}

template <int LOGN, int LOGE, bool FWD> static double check() {
  constexpr int n = 1 << LOGN, E = 1 << LOGE, T = n / E;
  std::vector<cpx> x(n), tab(n / 2 > 0 ? n / 2 : 1), regs(n), lds(lds_padded_size(n));
  unsigned s = 12345u + LOGN;
  for (auto &c : x) {
    s = s * 1664525u + 1013904223u; c.x = (float)(s >> 8) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; c.y = (float)(s >> 8) / 8388608.0f - 1.0f;
  }
  const double PI = 3.141592653589793;
  for (int i = 0; i < n / 2; i++) tab[i] = mk((float)cos(i * 2 * PI / n), -(float)sin(i * 2 * PI / n));
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) regs[tid * E + e] = x[tid + T * e];
  run<LOGN, LOGE, 0, FWD>(regs, tab, lds);
  // float64 DFT (O(n^2) for small n, else recursive radix-2)
  std::vector<std::complex<double>> ref(n);
  {
    std::vector<std::complex<double>> a(n);
    for (int i = 0; i < n; i++) a[i] = {x[i].x, x[i].y};
    // iterative radix-2 in double
    for (int i = 0, j = 0; i < n; i++) {
      if (i < j) std::swap(a[i], a[j]);
      int m = n >> 1;
      while (m >= 1 && (j & m)) { j ^= m; m >>= 1; }
      j |= m;
    }
    for (int len = 2; len <= n; len <<= 1)
      for (int i = 0; i < n; i += len)
        for (int k = 0; k < len / 2; k++) {
          double ang = (FWD ? -2 : 2) * PI * k / len;
          std::complex<double> w(cos(ang), sin(ang)), u = a[i + k], t = w * a[i + k + len / 2];
          a[i + k] = u + t; a[i + k + len / 2] = u - t;
        }
    ref = a;
  }
  double num = 0, den = 0;
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) {
      std::complex<double> y(regs[tid * E + e].x, regs[tid * E + e].y);
      num += std::norm(y - ref[tid + T * e]);
      den += std::norm(ref[tid + T * e]);
    }
  return sqrt(num / den);
}

static int g_fail = 0;
// same engine with the two-level twiddle table (W_n^k = hi[k >> LOGLO] * lo[k & mask])
template <int LOGN, int LOGE, int LOGNS, bool FWD, int LOGLO>
static void run2(std::vector<cpx> &regs, const TwoLevelTab<LOGLO> &tab, std::vector<cpx> &lds) {
  constexpr int E = 1 << LOGE, T = 1 << (LOGN - LOGE);
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  for (int tid = 0; tid < T; tid++)
    pass_compute<LOGN, LOGE, LOGNS, FWD>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, tab);
  if constexpr (LOGNS + LOGR < LOGN) {
    for (int tid = 0; tid < T; tid++)
      pass_scatter<LOGN, LOGE, LOGNS>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid,
                                      [&](int p, cpx v) { lds[lds_pad(p)] = v; });
    for (int tid = 0; tid < T; tid++)
      pass_gather<LOGN, LOGE>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, [&](int p) { return lds[lds_pad(p)]; });
    run2<LOGN, LOGE, LOGNS + LOGR, FWD, LOGLO>(regs, tab, lds);
  }
}
template <int LOGN, int LOGE, int LOGLO, bool FWD> static int two_level() {
  constexpr int n = 1 << LOGN, E = 1 << LOGE, T = n / E, LO = 1 << LOGLO, HI = n >> LOGLO;
  std::vector<cpx> x(n), hi(HI), lo(LO), regs(n), regs1(n), lds(lds_padded_size(n)), half(n / 2);
  unsigned s = 777u;
  for (auto &c : x) {
    s = s * 1664525u + 1013904223u; c.x = (float)(s >> 8) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; c.y = (float)(s >> 8) / 8388608.0f - 1.0f;
  }
  const double PI = 3.141592653589793;
  for (int i = 0; i < HI; i++) hi[i] = mk((float)cos((double)i * LO * 2 * PI / n), -(float)sin((double)i * LO * 2 * PI / n));
  for (int i = 0; i < LO; i++) lo[i] = mk((float)cos(i * 2 * PI / n), -(float)sin(i * 2 * PI / n));
  for (int i = 0; i < n / 2; i++) half[i] = mk((float)cos(i * 2 * PI / n), -(float)sin(i * 2 * PI / n));
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) regs1[tid * E + e] = regs[tid * E + e] = x[tid + T * e];
  TwoLevelTab<LOGLO> tab{hi.data(), lo.data()};
  run2<LOGN, LOGE, 0, FWD, LOGLO>(regs, tab, lds);
  run<LOGN, LOGE, 0, FWD>(regs1, half, lds);      // exact half table as the yardstick
  double num = 0, den = 0;
  for (int i = 0; i < n; i++) {
    double dx = regs[i].x - regs1[i].x, dy = regs[i].y - regs1[i].y;
    num += dx * dx + dy * dy;
    den += (double)regs1[i].x * regs1[i].x + (double)regs1[i].y * regs1[i].y;
  }
  double e = sqrt(num / den);
  printf("n=2^%-2d E=%-2d two-level table vs exact table: relL2 %.3g\n", LOGN, E, e);
  return !(e < 3e-7);
}
// ---- packed real transforms with the pair maps done in registers ---------------------------
// forward: pass chain whose last (remainder) pass is pass_last_paired, pairs picked up by pairs_visit;
// inverse: pass_first_paired + the transposed (decimation-in-frequency) chain.  Yardstick: the plain chain plus the
// reference's pair loops (cl_fft.cpp:178-205) over natural-order arrays.
template <int LOGN, int LOGE, int LOGNS>
static void run_pairlast(std::vector<cpx> &regs, const std::vector<cpx> &tab, std::vector<cpx> &lds) {
  constexpr int E = 1 << LOGE, T = 1 << (LOGN - LOGE);
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS), NEXT = LOGNS + LOGR;
  for (int tid = 0; tid < T; tid++)
    pass_compute<LOGN, LOGE, LOGNS, true>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, tab);
  for (int tid = 0; tid < T; tid++)
    pass_scatter_padded<LOGN, LOGE, LOGNS>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
  if constexpr (NEXT + pass_logr(LOGN, LOGE, NEXT) == LOGN) {
    for (int tid = 0; tid < T; tid++)
      pass_last_paired<LOGN, LOGE, true>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, tab, lds.data());
  } else {
    for (int tid = 0; tid < T; tid++)
      pass_gather_padded<LOGN, LOGE>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
    run_pairlast<LOGN, LOGE, NEXT>(regs, tab, lds);
  }
}
template <int LOGN, int LOGE, int LOGNS>
static void run_dif_inv(std::vector<cpx> &regs, const std::vector<cpx> &tab, std::vector<cpx> &lds) {
  constexpr int E = 1 << LOGE, T = 1 << (LOGN - LOGE);
  for (int tid = 0; tid < T; tid++) {
    auto &v = *reinterpret_cast<cpx(*)[E]>(&regs[tid * E]);
    dif_gather_padded<LOGN, LOGE, LOGNS>(v, tid, lds.data());
    dif_compute<LOGN, LOGE, LOGNS, false>(v, tid, tab);
  }
  if constexpr (LOGNS > 0) {
    for (int tid = 0; tid < T; tid++)
      dif_scatter_padded<LOGN, LOGE>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
    run_dif_inv<LOGN, LOGE, LOGNS - LOGE>(regs, tab, lds);
  }
}
template <int LOGN, int LOGE> static int paired() {
  constexpr int m = 1 << LOGN, E = 1 << LOGE, T = m / E, R = 1 << pass_rem_logr(LOGN, LOGE), U = E / R;
  static_assert(pair_ok(LOGN, LOGE), "");
  std::vector<cpx> x(m), tab(m / 2), w2f(m), w2i(m), regs(m), lds(lds_padded_size(m));
  unsigned s = 4242u + LOGN;
  for (auto &c : x) {
    s = s * 1664525u + 1013904223u; c.x = (float)(s >> 8) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; c.y = (float)(s >> 8) / 8388608.0f - 1.0f;
  }
  const double PI = 3.141592653589793;
  for (int i = 0; i < m / 2; i++) tab[i] = mk((float)cos(i * 2 * PI / m), -(float)sin(i * 2 * PI / m));
  for (int i = 0; i < m; i++) {
    w2f[i] = mk((float)cos(i * PI / m), -(float)sin(i * PI / m));
    w2i[i] = mk((float)cos(i * PI / m), (float)sin(i * PI / m));
  }
  int bad = 0;
  // ---- forward ----
  std::vector<cpx> z(m), want(m), got(m);
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) regs[tid * E + e] = x[tid + T * e];
  run<LOGN, LOGE, 0, true>(regs, tab, lds);
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) z[tid + T * e] = regs[tid * E + e];
  want = z;
  want[0] = mk((z[0].x + z[0].y) * .5f, (z[0].x - z[0].y) * .5f);
  for (int i = 1; i < m / 2; i++) r2c_pair(z[i], z[m - i], w2f[i], want[i], want[m - i]);
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) regs[tid * E + e] = x[tid + T * e];
  run_pairlast<LOGN, LOGE, 0>(regs, tab, lds);
  std::vector<int> seen(m, 0);
  for (int tid = 0; tid < T; tid++)
    pairs_visit<LOGN, LOGE>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid, [&](int k, int i, cpx ci, cpx cj) {
      (void)k;
      const int j = i == 0 ? m / 2 : m - i;
      cpx oi, oj;
      r2c_pair(ci, cj, w2f[i], oi, oj);
      if (i == 0) {
        oi = mk((ci.x + ci.y) * .5f, (ci.x - ci.y) * .5f);
        oj = cj;
      }
      if (i < 0 || i >= m / 2 + (i == 0)) bad |= 1;
      got[i] = oi; got[j] = oj; seen[i]++; seen[j]++;
    });
  for (int i = 0; i < m; i++)
    if (seen[i] != 1 || got[i].x != want[i].x || got[i].y != want[i].y) bad |= 2;
  // ---- inverse ----
  std::vector<cpx> y = x;
  y[0] = mk(x[0].x + x[0].y, x[0].x - x[0].y);
  for (int i = 1; i < m / 2; i++) c2r_pair(x[i], x[m - i], w2i[i], y[i], y[m - i]);
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) regs[tid * E + e] = y[tid + T * e];
  run<LOGN, LOGE, 0, false>(regs, tab, lds);
  std::vector<cpx> wanti(m);
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) wanti[tid + T * e] = regs[tid * E + e];
  std::fill(seen.begin(), seen.end(), 0);
  for (int tid = 0; tid < T; tid++) {
    cpx oi[E / 2], oj[E / 2];
    for (int u = 0; u < U / 2; u++)
      for (int q = 0; q < R; q++) {
        const int k = u * R + q, i = pair_index<LOGN, LOGE>(tid, u, q), j = i == 0 ? m / 2 : m - i;
        c2r_pair(x[i], x[j], w2i[i], oi[k], oj[k]);
        if (i == 0) {
          oi[k] = mk(x[0].x + x[0].y, x[0].x - x[0].y);
          oj[k] = x[j];
        }
        seen[i]++; seen[j]++;
      }
    pass_first_paired<LOGN, LOGE, false>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, oi, oj, tab);
  }
  for (int i = 0; i < m; i++) if (seen[i] != 1) bad |= 4;
  for (int tid = 0; tid < T; tid++)
    pass_first_paired_scatter<LOGN, LOGE>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
  run_dif_inv<LOGN, LOGE, pass_last_logns(LOGN, LOGE) - LOGE>(regs, tab, lds);
  double num = 0, den = 0;
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) {
      cpx a = regs[tid * E + e], b = wanti[tid + T * e];
      num += (double)(a.x - b.x) * (a.x - b.x) + (double)(a.y - b.y) * (a.y - b.y);
      den += (double)b.x * b.x + (double)b.y * b.y;
    }
  double e = sqrt(num / den);
  if (!(e < 5e-7)) bad |= 8;
  printf("m=2^%-2d E=%-2d paired r2c %s, paired c2r relL2 %.3g%s\n", LOGN, E, (bad & 3) ? "MISMATCH" : "bit-exact", e,
         bad ? "  FAIL" : "");
  return bad != 0;
}

// ---- lane-addressed twiddle tables (n = 8192: LaneTab13, n = 16384: LaneTab14) -----------------------
// The chain with every pass's twiddles taken from the lane tables, plain and with the paired remainder
// pass (forward: pass_last_paired, inverse: pass_first_paired + transposed chain), against the chain on the
// exact half table.  The table blob is laid out like the host's (clfft_amd.cpp, fft_setup).
template <int LOGN> struct LaneOf { using type = LaneTab13; };
template <> struct LaneOf<14> { using type = LaneTab14; };
template <int LOGN> static std::vector<typename LaneOf<LOGN>::type> make_lane_tabs(std::vector<cpx> &blob) {
  constexpr int T = (1 << LOGN) / 16;
  const double PI = 3.141592653589793;
  blob.clear();
  auto w = [&](long k, long n) { blob.push_back(mk((float)cos(k * 2 * PI / n), -(float)sin(k * 2 * PI / n))); };
  for (int j = 0; j < 16; j++)
    for (int t = 0; t < 16; t++) w(j * t, 256);
  for (int k = 0; k < 4; k++)
    for (int j = 0; j < 256; j++) w(((1 << k) * j) & 4095, 4096);
  if (LOGN == 14) {
    for (int m = 1; m <= 3; m++)
      for (int t = 0; t < 1024; t++) w(m * t, 16384);
  } else {
    for (int t = 0; t < 512; t++) w(t, 8192);
  }
  std::vector<typename LaneOf<LOGN>::type> tabs(T);
  for (int t = 0; t < T; t++) {
    if constexpr (LOGN == 14) tabs[t] = LaneTab14{&blob[16 * (t & 15)], &blob[256 + (t & 255)], blob[1280 + t], blob[1280 + 1024 + t], blob[1280 + 2048 + t]};
    else tabs[t] = LaneTab13{&blob[16 * (t & 15)], &blob[256 + (t & 255)], blob[1280 + t]};
  }
  return tabs;
}
template <int LOGN, int LOGNS, bool FWD, bool PAIRLAST, class LT>
static void run_lane(std::vector<cpx> &regs, const std::vector<LT> &tabs, std::vector<cpx> &lds) {
  constexpr int LOGE = 4, E = 16, T = 1 << (LOGN - LOGE);
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS), NEXT = LOGNS + LOGR;
  for (int tid = 0; tid < T; tid++)
    pass_compute<LOGN, LOGE, LOGNS, FWD>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, tabs[tid]);
  if constexpr (NEXT < LOGN) {
    for (int tid = 0; tid < T; tid++)
      pass_scatter_padded<LOGN, LOGE, LOGNS>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
    if constexpr (PAIRLAST && NEXT + pass_logr(LOGN, LOGE, NEXT) == LOGN) {
      for (int tid = 0; tid < T; tid++)
        pass_last_paired<LOGN, LOGE, FWD>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, tabs[tid], lds.data());
    } else {
      for (int tid = 0; tid < T; tid++)
        pass_gather_padded<LOGN, LOGE>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
      run_lane<LOGN, NEXT, FWD, PAIRLAST>(regs, tabs, lds);
    }
  }
}
// fft_wg.hpp's wg_passes_sigma, lane by lane: registers slot `tid` is PHYSICAL lane tid; in the middle passes it works
// butterfly lane_sigma(tid) with that lane's tables.  Also counts what the permutation is for: the bank conflicts of the
// gathers' ds_read_b64 (two groups of 32 lanes over 64 banks of 4 bytes, MI355X_MICROARCH.md).
static long g_conf_plain = 0, g_conf_sigma = 0;
template <int LOGN> static long gather_conflicts(bool sigma) {
  constexpr int T = (1 << LOGN) / 16;
  long extra = 0;
  for (int g = 0; g < T; g += 32) {
    int cnt[64] = {0};
    for (int l = g; l < g + 32; l++) {
      const int L = sigma ? lane_sigma(l) : l, dw = 2 * lds_pad(L);
      cnt[dw & 63]++;
    }
    int worst = 0;
    for (int b = 0; b < 64; b++) worst = cnt[b] > worst ? cnt[b] : worst;
    extra += worst - 1;
  }
  return extra;
}
template <int LOGN, int LOGNS, bool FWD, bool PAIRLAST, class LT>
static void run_lane_sigma(std::vector<cpx> &regs, const std::vector<LT> &tabs, std::vector<cpx> &lds) {
  constexpr int LOGE = 4, E = 16, T = 1 << (LOGN - LOGE);
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS), NEXT = LOGNS + LOGR;
  constexpr bool FIRST = LOGNS == 0, LAST = NEXT == LOGN;
  auto own = [&](int tid) { return (FIRST || LAST) ? tid : lane_sigma(tid); };
  for (int tid = 0; tid < T; tid++)
    pass_compute<LOGN, LOGE, LOGNS, FWD>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), own(tid), tabs[own(tid)]);
  if constexpr (!LAST) {
    constexpr bool NEXT_LAST = NEXT + pass_logr(LOGN, LOGE, NEXT) == LOGN;
    for (int tid = 0; tid < T; tid++)
      pass_scatter_padded<LOGN, LOGE, LOGNS>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), FIRST ? tid : lane_sigma(tid), lds.data());
    if constexpr (PAIRLAST && NEXT_LAST) {
      for (int tid = 0; tid < T; tid++)
        pass_last_paired<LOGN, LOGE, FWD>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, tabs[tid], lds.data());
    } else {
      for (int tid = 0; tid < T; tid++)
        pass_gather_padded<LOGN, LOGE>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), NEXT_LAST ? tid : lane_sigma(tid), lds.data());
      run_lane_sigma<LOGN, NEXT, FWD, PAIRLAST>(regs, tabs, lds);
    }
  }
}
template <int LOGN, int LOGNS, class LT>
static void run_lane_dif_inv(std::vector<cpx> &regs, const std::vector<LT> &tabs, std::vector<cpx> &lds) {
  constexpr int LOGE = 4, E = 16, T = 1 << (LOGN - LOGE);
  for (int tid = 0; tid < T; tid++) {
    auto &v = *reinterpret_cast<cpx(*)[E]>(&regs[tid * E]);
    dif_gather_padded<LOGN, LOGE, LOGNS>(v, tid, lds.data());
    dif_compute<LOGN, LOGE, LOGNS, false>(v, tid, tabs[tid]);
  }
  if constexpr (LOGNS > 0) {
    for (int tid = 0; tid < T; tid++)
      dif_scatter_padded<LOGN, LOGE>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
    run_lane_dif_inv<LOGN, LOGNS - LOGE>(regs, tabs, lds);
  }
}
static double rel_l2(const std::vector<cpx> &a, const std::vector<cpx> &b) {
  double num = 0, den = 0;
  for (size_t i = 0; i < a.size(); i++) {
    num += (double)(a[i].x - b[i].x) * (a[i].x - b[i].x) + (double)(a[i].y - b[i].y) * (a[i].y - b[i].y);
    den += (double)b[i].x * b[i].x + (double)b[i].y * b[i].y;
  }
  return sqrt(num / den);
}
template <int LOGN> static int lane_tables() {
  constexpr int LOGE = 4, n = 1 << LOGN, E = 16, T = n / E, R = 1 << pass_rem_logr(LOGN, LOGE), U = E / R;
  std::vector<cpx> blob, x(n), half(n / 2), regs(n), ref(n), lds(lds_padded_size(n));
  auto tabs = make_lane_tabs<LOGN>(blob);
  unsigned s = 9001u + LOGN;
  for (auto &c : x) {
    s = s * 1664525u + 1013904223u; c.x = (float)(s >> 8) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; c.y = (float)(s >> 8) / 8388608.0f - 1.0f;
  }
  const double PI = 3.141592653589793;
  for (int i = 0; i < n / 2; i++) half[i] = mk((float)cos(i * 2 * PI / n), -(float)sin(i * 2 * PI / n));
  int bad = 0;
  auto load = [&](std::vector<cpx> &r, const std::vector<cpx> &src) {
    for (int tid = 0; tid < T; tid++)
      for (int e = 0; e < E; e++) r[tid * E + e] = src[tid + T * e];
  };
  // plain chain, both directions
  double e_fwd, e_inv;
  load(regs, x); load(ref, x);
  run_lane<LOGN, 0, true, false>(regs, tabs, lds);
  run<LOGN, LOGE, 0, true>(ref, half, lds);
  e_fwd = rel_l2(regs, ref);
  load(regs, x); load(ref, x);
  run_lane<LOGN, 0, false, false>(regs, tabs, lds);
  run<LOGN, LOGE, 0, false>(ref, half, lds);
  e_inv = rel_l2(regs, ref);
  if (!(e_fwd < 3e-7) || !(e_inv < 3e-7)) bad |= 1;
  // forward with the paired remainder pass: the pairs it hands out against the natural-order spectrum
  std::vector<cpx> z(n), got(n);
  load(ref, x);
  run<LOGN, LOGE, 0, true>(ref, half, lds);
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) z[tid + T * e] = ref[tid * E + e];
  load(regs, x);
  run_lane<LOGN, 0, true, true>(regs, tabs, lds);
  std::vector<int> seen(n, 0);
  for (int tid = 0; tid < T; tid++)
    pairs_visit<LOGN, LOGE>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid, [&](int k, int i, cpx ci, cpx cj) {
      (void)k;
      const int j = i == 0 ? n / 2 : n - i;
      got[i] = ci; got[j] = cj; seen[i]++; seen[j]++;
    });
  for (int i = 0; i < n; i++) if (seen[i] != 1) bad |= 2;
  const double e_pair = rel_l2(got, z);
  if (!(e_pair < 3e-7)) bad |= 4;
  // the chain with its middle passes on permuted lanes: the same values in the same registers, and conflict-free gathers
  {
    std::vector<cpx> a(n), b(n);
    for (int pl = 0; pl < 2; pl++)
      for (int dir = 0; dir < 2; dir++) {
        load(a, x); load(b, x);
        if (pl && dir) { run_lane<LOGN, 0, true, true>(a, tabs, lds); run_lane_sigma<LOGN, 0, true, true>(b, tabs, lds); }
        else if (pl) continue;   // (the paired last pass exists for forward transforms)
        else if (dir) { run_lane<LOGN, 0, true, false>(a, tabs, lds); run_lane_sigma<LOGN, 0, true, false>(b, tabs, lds); }
        else { run_lane<LOGN, 0, false, false>(a, tabs, lds); run_lane_sigma<LOGN, 0, false, false>(b, tabs, lds); }
        for (int i = 0; i < n; i++) if (!(a[i].x == b[i].x && a[i].y == b[i].y)) bad |= 16;
      }
    const long cp = gather_conflicts<LOGN>(false), cs = gather_conflicts<LOGN>(true);
    printf("n=2^%-2d gather bank conflicts (extra LDS cycles per gather instruction, all lane groups): lane = tid %ld, lane = sigma(tid) %ld\n", LOGN, cp, cs);
    if (cs != 0 || cp == 0) bad |= 32;
  }
  // inverse: pass_first_paired fed with natural-order data in pair order + transposed chain
  load(ref, x);
  run<LOGN, LOGE, 0, false>(ref, half, lds);
  std::vector<cpx> wanti(n), goti(n);
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) wanti[tid + T * e] = ref[tid * E + e];
  for (int tid = 0; tid < T; tid++) {
    cpx oi[E / 2], oj[E / 2];
    for (int u = 0; u < U / 2; u++)
      for (int q = 0; q < R; q++) {
        const int k = u * R + q, i = pair_index<LOGN, LOGE>(tid, u, q), j = i == 0 ? n / 2 : n - i;
        oi[k] = x[i]; oj[k] = x[j];
      }
    pass_first_paired<LOGN, LOGE, false>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, oi, oj, tabs[tid]);
  }
  for (int tid = 0; tid < T; tid++)
    pass_first_paired_scatter<LOGN, LOGE>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
  run_lane_dif_inv<LOGN, pass_last_logns(LOGN, LOGE) - LOGE>(regs, tabs, lds);
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) goti[tid + T * e] = regs[tid * E + e];
  const double e_pinv = rel_l2(goti, wanti);
  if (!(e_pinv < 3e-7)) bad |= 8;
  printf("n=2^%-2d lane tables vs exact half table: fwd %.3g inv %.3g, paired fwd %.3g, paired inv (transposed chain) %.3g%s\n",
         LOGN, e_fwd, e_inv, e_pair, e_pinv, bad ? "  FAIL" : "");
  return bad != 0;
}

// ---- packed real sizes 32768 / 65536 on two 8192- / 16384-point sub-transforms (rfft2x_fwd_slot / rfft2x_inv_slot) -----
// (LOGN = 13, 14: lane tables, 16 points per lane; LOGN = 11 with 8 points per lane: the half table)
template <int LOGN, int LOGE = 4> static int rfft2x() {
  constexpr int M = 1 << LOGN, n = 2 * M, E = 1 << LOGE, T = M / E, R = 1 << pass_rem_logr(LOGN, LOGE), U = E / R, NP = E / 2;
  constexpr bool LANE = LOGN >= 13;
  std::vector<cpx> blob, z(n), half(n / 2), w2f(n), w2i(n), lds(lds_padded_size(n));
  auto tabs = make_lane_tabs<LANE ? LOGN : 13>(blob);
  std::vector<cpx> halfc(M / 2);   // the sub-transforms' half table (LOGN < 13)
  unsigned s = 31337u;
  for (auto &c : z) {
    s = s * 1664525u + 1013904223u; c.x = (float)(s >> 8) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; c.y = (float)(s >> 8) / 8388608.0f - 1.0f;
  }
  const double PI = 3.141592653589793;
  for (int i = 0; i < n / 2; i++) half[i] = mk((float)cos(i * 2 * PI / n), -(float)sin(i * 2 * PI / n));
  for (int i = 0; i < M / 2; i++) halfc[i] = mk((float)cos(i * 2 * PI / M), -(float)sin(i * 2 * PI / M));
  for (int i = 0; i < n; i++) {
    w2f[i] = mk((float)cos(i * PI / n), -(float)sin(i * PI / n));
    w2i[i] = mk((float)cos(i * PI / n), (float)sin(i * PI / n));
  }
  int bad = 0;
  // ---- forward: yardstick = 32768-point chain on the exact half table, scaled 1/n, + the reference pair loop
  std::vector<cpx> regs(n), Z(n), want(n), got(n);
  {
    constexpr int T15 = n / E;
    for (int tid = 0; tid < T15; tid++)
      for (int e = 0; e < E; e++) regs[tid * E + e] = z[tid + T15 * e];
    run<LOGN + 1, LOGE, 0, true>(regs, half, lds);
    for (int tid = 0; tid < T15; tid++)
      for (int e = 0; e < E; e++) Z[tid + T15 * e] = cscale(regs[tid * E + e], 1.0f / (float)n);
  }
  want = Z;
  want[0] = mk((Z[0].x + Z[0].y) * .5f, (Z[0].x - Z[0].y) * .5f);
  for (int i = 1; i < n / 2; i++) r2c_pair(Z[i], Z[n - i], w2f[i], want[i], want[n - i]);
  std::vector<cpx> ra(M), rb(M);
  for (int tid = 0; tid < T; tid++)
    for (int e = 0; e < E; e++) {
      ra[tid * E + e] = z[2 * (tid + T * e)];
      rb[tid * E + e] = z[2 * (tid + T * e) + 1];
    }
  if constexpr (LANE) {
    run_lane<LOGN, 0, true, true>(ra, tabs, lds);
    run_lane<LOGN, 0, true, true>(rb, tabs, lds);
  } else {
    run_pairlast<LOGN, LOGE, 0>(ra, halfc, lds);
    run_pairlast<LOGN, LOGE, 0>(rb, halfc, lds);
  }
  std::vector<int> seen(n, 0);
  for (int tid = 0; tid < T; tid++) {
    cpx ai[NP], aj[NP], bi[NP], bj[NP];
    int ii[NP];
    pairs_visit<LOGN, LOGE>(*reinterpret_cast<const cpx(*)[E]>(&ra[tid * E]), tid, [&](int k, int i, cpx ci, cpx cj) {
      ai[k] = cscale(ci, 1.0f / (float)n); aj[k] = cscale(cj, 1.0f / (float)n); ii[k] = i;
    });
    pairs_visit<LOGN, LOGE>(*reinterpret_cast<const cpx(*)[E]>(&rb[tid * E]), tid, [&](int k, int i, cpx ci, cpx cj) {
      (void)i; bi[k] = cscale(ci, 1.0f / (float)n); bj[k] = cscale(cj, 1.0f / (float)n);
    });
    for (int k = 0; k < NP; k++)
      rfft2x_fwd_slot<LOGN>(tid, k / R, k % R, ii[k], ai[k], aj[k], bi[k], bj[k], w2f[2 * tid], w2f[tid],
                      [&](int pos, cpx v) { got[pos] = v; seen[pos]++; });
  }
  for (int i = 0; i < n; i++) if (seen[i] != 1) bad |= 1;
  const double e_f = rel_l2(got, want);
  double mx = 0, mxref = 0;
  for (int i = 0; i < n; i++) {
    mx = std::max(mx, (double)std::max(fabsf(got[i].x - want[i].x), fabsf(got[i].y - want[i].y)));
    mxref = std::max(mxref, (double)std::max(fabsf(want[i].x), fabsf(want[i].y)));
  }
  if (!(e_f < 3e-7) || !(mx / mxref < 1e-6)) bad |= 2;
  // ---- inverse: yardstick = reference pair loop, then the 32768-point inverse chain (unscaled)
  std::vector<cpx> y = z, wanti(n), goti(n);
  y[0] = mk(z[0].x + z[0].y, z[0].x - z[0].y);
  for (int i = 1; i < n / 2; i++) c2r_pair(z[i], z[n - i], w2i[i], y[i], y[n - i]);
  {
    constexpr int T15 = n / E;
    for (int tid = 0; tid < T15; tid++)
      for (int e = 0; e < E; e++) regs[tid * E + e] = y[tid + T15 * e];
    run<LOGN + 1, LOGE, 0, false>(regs, half, lds);
    for (int tid = 0; tid < T15; tid++)
      for (int e = 0; e < E; e++) wanti[tid + T15 * e] = regs[tid * E + e];
  }
  std::fill(seen.begin(), seen.end(), 0);
  std::vector<cpx> oa(M / 2 * 1), dummy;
  std::vector<cpx> OA(T * NP), PA(T * NP), OB(T * NP), PB(T * NP);
  for (int tid = 0; tid < T; tid++)
    for (int u = 0; u < U / 2; u++)
      for (int q = 0; q < R; q++) {
        const int k = u * R + q, i = pair_index<LOGN, LOGE>(tid, u, q);
        cpx in[4];
        for (int w = 0; w < 4; w++) {
          const int pos = rfft2x_pos<LOGN>(i, w);
          seen[pos]++;
          in[w] = z[pos];
        }
        rfft2x_inv_slot<LOGN>(tid, u, q, i, w2i[2 * tid], w2i[tid], in[0], in[1], in[2], in[3], OA[tid * NP + k], PA[tid * NP + k],
                        OB[tid * NP + k], PB[tid * NP + k]);
      }
  for (int i = 0; i < n; i++) if (seen[i] != 1) bad |= 4;
  for (int half_ = 0; half_ < 2; half_++) {
    std::vector<cpx> &O = half_ ? OB : OA, &P = half_ ? PB : PA;
    std::vector<cpx> r(M);
    for (int tid = 0; tid < T; tid++) {
      auto &rv = *reinterpret_cast<cpx(*)[E]>(&r[tid * E]);
      const auto &ov = *reinterpret_cast<const cpx(*)[NP]>(&O[tid * NP]);
      const auto &pv = *reinterpret_cast<const cpx(*)[NP]>(&P[tid * NP]);
      if constexpr (LANE) pass_first_paired<LOGN, LOGE, false>(rv, tid, ov, pv, tabs[tid]);
      else pass_first_paired<LOGN, LOGE, false>(rv, tid, ov, pv, halfc);
    }
    for (int tid = 0; tid < T; tid++)
      pass_first_paired_scatter<LOGN, LOGE>(*reinterpret_cast<const cpx(*)[E]>(&r[tid * E]), tid, lds.data());
    if constexpr (LANE) run_lane_dif_inv<LOGN, pass_last_logns(LOGN, LOGE) - LOGE>(r, tabs, lds);
    else run_dif_inv<LOGN, LOGE, pass_last_logns(LOGN, LOGE) - LOGE>(r, halfc, lds);
    for (int tid = 0; tid < T; tid++)
      for (int e = 0; e < E; e++) goti[2 * (tid + T * e) + half_] = r[tid * E + e];
  }
  const double e_i = rel_l2(goti, wanti);
  if (!(e_i < 3e-7)) bad |= 8;
  printf("real size %d on two %d-point sub-transforms: forward relL2 %.3g (max %.3g), inverse relL2 %.3g%s\n", 2 * n, M, e_f,
         mx / mxref, e_i, bad ? "  FAIL" : "");
  if (bad) printf("  flags %d\n", bad);
  return bad != 0;
}
template <int LOGN, int LOGE> static void both() {
  double a = check<LOGN, LOGE, true>(), b = check<LOGN, LOGE, false>();
  printf("n=2^%-2d E=%-2d relL2 fwd %.3g inv %.3g\n", LOGN, 1 << LOGE, a, b);
  if (!(a < 5e-7) || !(b < 5e-7)) g_fail = 1;
}

// xcd_first(): for every grid the persistent kernels launch, the workgroups' first transforms are a permutation of
// 0 .. grid - 1 (then every workgroup steps by the grid size: each transform is taken exactly once), workgroups i and
// i + 8 (one XCD under round-robin dispatch) take ADJACENT transforms, and grids that are not multiples of 8 fall back to
// "workgroup i takes i"
static int assignment() {
  int bad = 0;
  for (unsigned grid : {1u, 2u, 7u, 8u, 16u, 70u, 72u, 128u, 248u, 256u, 304u, 512u, 1024u}) {
    std::vector<int> seen(grid, 0);
    for (unsigned i = 0; i < grid; i++) {
      const long f = clfa::xcd_first(i, grid);
      if (f < 0 || f >= (long)grid) { bad++; continue; }
      seen[f]++;
      if (grid % 8 == 0) {
        if (i + 8 < grid && clfa::xcd_first(i + 8, grid) != f + 1) bad++;
      } else if (f != (long)i) bad++;
    }
    for (unsigned i = 0; i < grid; i++) bad += seen[i] != 1;
  }
  printf("xcd_first: %s\n", bad ? "FAIL" : "a permutation for every grid, XCD-compact for multiples of 8");
  return bad != 0;
}

// ---- the 16 x 16 twiddle table of the pass that starts at 16 points (HalfRowTab: n = 256 .. 4096, 16 points per lane)
// against the half table they are filled from: the same transform, value for value
template <int LOGN, int LOGE, int LOGNS, bool FWD, class Mk>
static void run_tab(std::vector<cpx> &regs, const Mk &mk_tab, std::vector<cpx> &lds) {
  constexpr int E = 1 << LOGE, T = 1 << (LOGN - LOGE);
  constexpr int LOGR = pass_logr(LOGN, LOGE, LOGNS);
  for (int tid = 0; tid < T; tid++)
    pass_compute<LOGN, LOGE, LOGNS, FWD>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, mk_tab(tid));
  if constexpr (LOGNS + LOGR < LOGN) {
    for (int tid = 0; tid < T; tid++)
      pass_scatter_padded<LOGN, LOGE, LOGNS>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
    for (int tid = 0; tid < T; tid++)
      pass_gather_padded<LOGN, LOGE>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
    run_tab<LOGN, LOGE, LOGNS + LOGR, FWD>(regs, mk_tab, lds);
  }
}
// ... and the transposed chain (dif_compute), whose twiddles sit on the outputs
template <int LOGN, int LOGE, int LOGNS, bool FWD, class Mk>
static void run_tab_dif(std::vector<cpx> &regs, const Mk &mk_tab, std::vector<cpx> &lds) {
  constexpr int E = 1 << LOGE, T = 1 << (LOGN - LOGE);
  for (int tid = 0; tid < T; tid++)
    dif_gather_padded<LOGN, LOGE, LOGNS>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
  for (int tid = 0; tid < T; tid++)
    dif_compute<LOGN, LOGE, LOGNS, FWD>(*reinterpret_cast<cpx(*)[E]>(&regs[tid * E]), tid, mk_tab(tid));
  if constexpr (LOGNS > 0) {
    for (int tid = 0; tid < T; tid++)
      dif_scatter_padded<LOGN, LOGE>(*reinterpret_cast<const cpx(*)[E]>(&regs[tid * E]), tid, lds.data());
    run_tab_dif<LOGN, LOGE, LOGNS - LOGE, FWD>(regs, mk_tab, lds);
  }
}
template <int LOGN, int LOGE> static int small_tables() {
  constexpr int n = 1 << LOGN, E = 1 << LOGE, T = n / E;
  std::vector<cpx> x(n), half(n / 2), a(n), b(n), lds(lds_padded_size(n));
  unsigned s = 4242u + LOGN;
  for (auto &c : x) {
    s = s * 1664525u + 1013904223u; c.x = (float)(s >> 8) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; c.y = (float)(s >> 8) / 8388608.0f - 1.0f;
  }
  const double PI = 3.141592653589793;
  for (int i = 0; i < n / 2; i++) half[i] = mk((float)cos(i * 2 * PI / n), -(float)sin(i * 2 * PI / n));
  static_assert(LOGE == 4, "16 points per lane");
  std::vector<cpx> row(16 * kRow16StrideDev);
  lds_fill_row16<LOGN>(row.data(), half.data(), 0, 1);
  auto plain = [&](int) { return static_cast<const cpx *>(half.data()); };
  auto small = [&](int tid) { return HalfRowTab{half.data(), row.data() + kRow16StrideDev * (tid & 15)}; };
  int bad = 0;
  for (int dir = 0; dir < 2; dir++) {
    for (int tid = 0; tid < T; tid++)
      for (int e = 0; e < E; e++) a[tid * E + e] = b[tid * E + e] = x[tid + T * e];
    if (dir) {
      run_tab<LOGN, LOGE, 0, true>(a, plain, lds);
      run_tab<LOGN, LOGE, 0, true>(b, small, lds);
    } else {
      run_tab<LOGN, LOGE, 0, false>(a, plain, lds);
      run_tab<LOGN, LOGE, 0, false>(b, small, lds);
    }
    for (int i = 0; i < n; i++) bad += !(a[i].x == b[i].x && a[i].y == b[i].y);
  }
  if constexpr (LOGN % LOGE == 0) {   // the transposed chain from its last full-radix pass downwards (any data: both sides see the same)
    constexpr int L0 = LOGN - LOGE;
    for (int i = 0; i < n; i++) lds[lds_pad(i)] = x[i];
    std::vector<cpx> l2 = lds;
    run_tab_dif<LOGN, LOGE, L0, false>(a, plain, lds);
    run_tab_dif<LOGN, LOGE, L0, false>(b, small, l2);
    for (int i = 0; i < n; i++) bad += !(a[i].x == b[i].x && a[i].y == b[i].y);
  }
  printf("n=2^%-2d E=%-2d 16 x 16 twiddle table vs the half table: %d values differ\n", LOGN, E, bad);
  return bad != 0;
}

int main() {
  g_fail |= assignment();
  g_fail |= small_tables<8, 4>() | small_tables<10, 4>() | small_tables<12, 4>();
  both<1, 1>(); both<2, 2>(); both<3, 3>(); both<4, 4>(); both<5, 4>(); both<6, 4>(); both<7, 4>();
  both<8, 4>(); both<9, 4>(); both<10, 4>(); both<11, 4>(); both<12, 4>(); both<13, 4>(); both<14, 4>();
  both<6, 2>(); both<6, 3>(); both<10, 3>(); both<9, 2>(); both<7, 3>(); both<8, 3>();
  both<5, 5>(); both<10, 5>(); both<13, 5>(); both<12, 5>(); both<8, 5>();   // 32 points per lane (radix-32 passes)
  g_fail |= two_level<13, 5, 6, true>() | two_level<13, 5, 6, false>() | two_level<12, 4, 6, true>();
  g_fail |= paired<5, 4>() | paired<6, 4>() | paired<7, 4>() | paired<9, 4>() | paired<10, 4>() | paired<11, 4>() |
            paired<13, 4>() | paired<4, 3>() | paired<5, 3>() | paired<7, 3>();
  g_fail |= paired<14, 4>();
  g_fail |= lane_tables<13>() | lane_tables<14>();
  g_fail |= rfft2x<14>() | rfft2x<13>() | rfft2x<11, 3>();
  puts(g_fail ? "FAIL" : "OK");
  return g_fail;
}
