// fft_kernels.hip — batched 1-D FFT kernels for gfx950 (MI355X).
//
// Replaces the reference's launch chain reorder + log2(N) x fft (+ conv/iconv)
// (cl_fft.cpp:24-41, 138-151, 178-205) by
//   k_fft_lds    one HBM pass: a transform (n <= 8192) lives in VGPRs + LDS of one
//                workgroup; r2c pack / c2r unpack fused as LDS epilogue / prologue;
//   k_fft_4step  n = 2^14..2^16: N1 x N2 decomposition, both phases in ONE
//                persistent kernel; the intermediate goes through a per-workgroup
//                scratch slot that is small enough (grid x 512 KiB) to live in the
//                256 MiB Infinity Cache, so HBM sees one read + one write per sample;
//   k_r2c_pack / k_c2r_unpack, k_reorder  stand-alone forms of the reference's
//                conv / iconv / reorder kernels.
#include <cstdlib>

#include "fft_wg.hpp"

namespace clfa {

// ---------------------------------------------------------------------------------
// single-workgroup LDS FFT
// ---------------------------------------------------------------------------------

template <int LOGN, bool FWD, int MODE, bool SCALE>
__global__ __launch_bounds__(LdsGeom<LOGN>::WG) void k_fft_lds(cpx *__restrict__ data,
                                                              const cpx *__restrict__ tab_g,
                                                              const cpx *__restrict__ w2_g, long batch) {
  using G = LdsGeom<LOGN>;
  constexpr int N = G::N, E = G::E, T = G::T, WG = G::WG, FPW = G::FPW;
  __shared__ cpx s_tab[G::HALF];
  __shared__ cpx s_x[FPW * G::PADN];

  const int tid = threadIdx.x;
  const int f = tid / T, t = tid % T;
  for (int i = tid; i < N / 2; i += WG) s_tab[i] = tab_g[i];
  __syncthreads();
  cpx *xb = s_x + f * G::PADN;

  const long groups = (batch + FPW - 1) / FPW;
#pragma unroll 1
  for (long g = blockIdx.x; g < groups; g += gridDim.x) {
    const long b = g * FPW + f;
    const bool active = b < batch;
    cpx *x = data + (active ? b : 0) * (long)N;
    cpx v[E];
    if constexpr (MODE == MODE_C2R) {
      // fused reference `iconv` on the way in
      __syncthreads();
      if (active) {
        for (int i = t; i < N / 2; i += T) {
          if (i == 0) {
            cpx c0 = x[0];
            xb[0] = mk(c0.x + c0.y, c0.x - c0.y);
            xb[lds_pad(N / 2)] = x[N / 2];
          } else {
            cpx oi, oj;
            c2r_pair(x[i], x[N - i], w2_g[i], oi, oj);
            xb[lds_pad(i)] = oi;
            xb[lds_pad(N - i)] = oj;
          }
        }
      }
      __syncthreads();
      pass_gather<LOGN, G::LOGE>(v, t, [&](int p) { return xb[lds_pad(p)]; });
    } else {
#pragma unroll
      for (int e = 0; e < E; e++) v[e] = active ? x[t + T * e] : mk(0.f, 0.f);
    }

    wg_passes<LOGN, G::LOGE, 0, FWD>(v, t, s_tab, xb);

    if constexpr (SCALE) {
      constexpr float inv = 1.0f / (float)N;
#pragma unroll
      for (int e = 0; e < E; e++) v[e] = cscale(v[e], inv);
    }

    if constexpr (MODE == MODE_R2C) {
      // fused reference `conv` on the way out
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; e++) xb[lds_pad(t + T * e)] = v[e];
      __syncthreads();
      if (active) {
        for (int i = t; i < N / 2; i += T) {
          if (i == 0) {
            cpx z = xb[0];
            x[0] = mk((z.x + z.y) * .5f, (z.x - z.y) * .5f);
            x[N / 2] = xb[lds_pad(N / 2)];
          } else {
            cpx oi, oj;
            r2c_pair(xb[lds_pad(i)], xb[lds_pad(N - i)], w2_g[i], oi, oj);
            x[i] = oi;
            x[N - i] = oj;
          }
        }
      }
    } else {
      if (active) {
#pragma unroll
        for (int e = 0; e < E; e++) x[t + T * e] = v[e];
      }
    }
  }
}

template <int LOGN, bool FWD, int MODE, bool SCALE>
static hipError_t launch_lds_one(cpx *data, const FftTables &t, long batch, const DeviceInfo &di,
                                 hipStream_t s) {
  using G = LdsGeom<LOGN>;
  long groups = (batch + G::FPW - 1) / G::FPW;
  // persistent-ish grid: enough workgroups to fill the chip several times over,
  // grid-stride over the rest so the LDS twiddle table is loaded once per workgroup
  long cap = (long)di.num_cus * 16;
  int grid = (int)(groups < cap ? groups : cap);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_fft_lds<LOGN, FWD, MODE, SCALE>), dim3(grid), dim3(G::WG), 0, s, data, t.half, t.w2,
                     batch);
  return hipGetLastError();
}

template <int LOGN>
static hipError_t launch_lds_n(bool fwd, int mode, bool scale, cpx *data, const FftTables &t, long batch,
                               const DeviceInfo &di, hipStream_t s) {
#define CLFA_CASE(F, M, S) \
  if (fwd == F && mode == M && scale == S) return launch_lds_one<LOGN, F, M, S>(data, t, batch, di, s);
  CLFA_CASE(true, MODE_C2C, true)
  CLFA_CASE(true, MODE_C2C, false)
  CLFA_CASE(false, MODE_C2C, false)
  CLFA_CASE(true, MODE_R2C, true)
  CLFA_CASE(false, MODE_C2R, false)
#undef CLFA_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_fft_lds(int logn, bool fwd, int mode, bool scale, cpx *data, const FftTables &t,
                          long batch, const DeviceInfo &di, hipStream_t s) {
  if (batch <= 0) return hipSuccess;
  switch (logn) {
#define CLFA_N(L) \
  case L:         \
    return launch_lds_n<L>(fwd, mode, scale, data, t, batch, di, s);
    CLFA_N(1) CLFA_N(2) CLFA_N(3) CLFA_N(4) CLFA_N(5) CLFA_N(6) CLFA_N(7) CLFA_N(8) CLFA_N(9) CLFA_N(10)
    CLFA_N(11) CLFA_N(12) CLFA_N(13)
#undef CLFA_N
    default:
      return hipErrorInvalidValue;
  }
}

const char *name_fft_lds(int, bool, int) { return "k_fft_lds"; }

// ---------------------------------------------------------------------------------
// four-step FFT for n = 2^14 .. 2^16
// ---------------------------------------------------------------------------------

int fourstep_split(int logn, int *l1, int *l2, int *loglo) {
  if (logn < 14 || logn > 16) return -1;
  *l1 = logn / 2;
  *l2 = logn - *l1;
  *loglo = logn / 2;
  return 0;
}

template <int LOGN> struct FourGeom {
  static constexpr int LOGN1 = LOGN / 2, LOGN2 = LOGN - LOGN1;
  static constexpr int N = 1 << LOGN, N1 = 1 << LOGN1, N2 = 1 << LOGN2;
  static constexpr int LOGLO = LOGN / 2, LO = 1 << LOGLO, HI = 1 << (LOGN - LOGLO);
  static constexpr int SLICE = 256;                 // lanes per slice
  static constexpr int T1 = N1 / 16, C1 = SLICE / T1;   // lanes per column FFT, columns per slice
  static constexpr int T2 = N2 / 16, R2 = SLICE / T2;   // lanes per row FFT, rows per slice
  static constexpr int S2 = lds_padded_size(N2) | 1;    // odd row stride in LDS
  static constexpr int SL = (N1 * C1 > R2 * S2) ? N1 * C1 : R2 * S2;  // exchange elements per slice
  static constexpr int NCB = N2 / C1, NRB = N1 / R2;    // column blocks, row blocks per transform
  static constexpr int TABS = N1 / 2 + N2 / 2 + LO + HI;
};

// phase 1 of one slice: column block cb of `src` (N1 x N2, row-major) ->
// N1-point FFT down the columns, times W_N^(n2*k1), stored to dst[k1][n2].
template <int LOGN, bool FWD, bool NT>
__device__ __forceinline__ void four_phase1(const cpx *__restrict__ src, cpx *__restrict__ dst, int cb, int l,
                                            const cpx *tab1, const cpx *tlo, const cpx *thi, cpx *sx) {
  using G = FourGeom<LOGN>;
  const int col = l % G::C1, tf = l / G::C1;
  const int n2 = cb * G::C1 + col;
  cpx v[16];
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const cpx *p = src + (long)(tf + G::T1 * e) * G::N2 + n2;
    if constexpr (NT) {
      const unsigned long long raw = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(p));
      v[e] = *reinterpret_cast<const cpx *>(&raw);
    } else {
      v[e] = *p;
    }
  }
  pass_compute<G::LOGN1, 4, 0, FWD>(v, tf, tab1);
  __syncthreads();
  pass_scatter<G::LOGN1, 4, 0>(v, tf, [&](int p, cpx val) { sx[p * G::C1 + col] = val; });
  __syncthreads();
  pass_gather<G::LOGN1, 4>(v, tf, [&](int p) { return sx[p * G::C1 + col]; });
  pass_compute<G::LOGN1, 4, 4, FWD>(v, tf, tab1);
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k1 = tf + G::T1 * e;
    const int ex = n2 * k1;  // < N
    cpx w = cmul(tlo[ex & (G::LO - 1)], thi[ex >> G::LOGLO]);
    if (!FWD) w.y = -w.y;
    dst[(long)k1 * G::N2 + n2] = cmul(v[e], w);
  }
}

// phase 2 of one slice: row block rb of `src` (rows k1, contiguous n2) ->
// N2-point FFT along each row -> dst[k1 + N1*k2] (natural order of the result)
template <int LOGN, bool FWD, bool SCALE, bool NT>
__device__ __forceinline__ void four_phase2(const cpx *__restrict__ src, cpx *__restrict__ dst, int rb, int l,
                                            const cpx *tab2, cpx *sx) {
  using G = FourGeom<LOGN>;
  cpx v[16];
  {
    const int tf = l % G::T2, row = l / G::T2;
    const cpx *p = src + (long)(rb * G::R2 + row) * G::N2 + tf;
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = p[G::T2 * e];
    pass_compute<G::LOGN2, 4, 0, FWD>(v, tf, tab2);
    __syncthreads();
    cpx *xr = sx + row * G::S2;
    pass_scatter<G::LOGN2, 4, 0>(v, tf, [&](int q, cpx val) { xr[lds_pad(q)] = val; });
    __syncthreads();
  }
  // the last pass runs with rows on the fast lane index so that the transposed
  // store below is contiguous across lanes
  const int row = l % G::R2, tf = l / G::R2;
  const cpx *xr = sx + row * G::S2;
  pass_gather<G::LOGN2, 4>(v, tf, [&](int q) { return xr[lds_pad(q)]; });
  pass_compute<G::LOGN2, 4, 4, FWD>(v, tf, tab2);
  const int k1 = rb * G::R2 + row;
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k2 = tf + G::T2 * e;
    cpx o = v[e];
    if constexpr (SCALE) o = cscale(o, 1.0f / (float)G::N);
    cpx *p = dst + (long)k2 * G::N1 + k1;
    if constexpr (NT) {
      __builtin_nontemporal_store(*reinterpret_cast<unsigned long long *>(&o),
                                  reinterpret_cast<unsigned long long *>(p));
    } else {
      *p = o;
    }
  }
}

template <int LOGN, bool FWD, bool SCALE, int NSLICE, bool NT>
__global__ __launch_bounds__(256 * NSLICE) void k_fft_4step(cpx *__restrict__ data, cpx *__restrict__ scratch,
                                                           const cpx *__restrict__ tabs_g, long batch) {
  using G = FourGeom<LOGN>;
  __shared__ cpx s_tabs[G::TABS];
  __shared__ cpx s_x[NSLICE * G::SL];
  const int tid = threadIdx.x;
  for (int i = tid; i < G::TABS; i += 256 * NSLICE) s_tabs[i] = tabs_g[i];
  const cpx *tab1 = s_tabs, *tab2 = s_tabs + G::N1 / 2, *tlo = tab2 + G::N2 / 2, *thi = tlo + G::LO;
  const int slice = tid / G::SLICE, l = tid % G::SLICE;
  cpx *sx = s_x + slice * G::SL;
  cpx *mid = scratch + (long)blockIdx.x * G::N;
  __syncthreads();

#pragma unroll 1
  for (long b = blockIdx.x; b < batch; b += gridDim.x) {
    cpx *x = data + b * (long)G::N;
#pragma unroll 1
    for (int cb = slice; cb < G::NCB; cb += NSLICE) four_phase1<LOGN, FWD, NT>(x, mid, cb, l, tab1, tlo, thi, sx);
    // the workgroup re-reads what it has just stored: workgroup-scope release/acquire
    __syncthreads();
#pragma unroll 1
    for (int rb = slice; rb < G::NRB; rb += NSLICE) four_phase2<LOGN, FWD, SCALE, NT>(mid, x, rb, l, tab2, sx);
    __syncthreads();
  }
}

struct FourVariant {
  int nslice;
  bool nt;
  int wg_per_cu;
};
static FourVariant four_variant(int variant) {
  switch (variant) {
    case 1: return {4, false, 1};
    case 2: return {2, false, 2};
    case 3: return {2, true, 2};
    case 4: return {1, false, 4};
    case 5: return {1, false, 2};
    case 6: return {2, false, 1};
    default: return {4, true, 1};  // 0
  }
}

int fourstep_grid(int logn, int variant, const DeviceInfo &di) {
  (void)logn;
  // tuning knob for experiments: CLFA_4STEP_GRID=<workgroups>
  if (const char *e = getenv("CLFA_4STEP_GRID")) {
    int g = atoi(e);
    if (g > 0) return g;
  }
  FourVariant v = four_variant(variant);
  return di.num_cus * v.wg_per_cu;
}

template <int LOGN, bool FWD, bool SCALE>
static hipError_t launch_4step_v(int variant, cpx *data, cpx *scratch, const FftTables &t, long batch,
                                 const DeviceInfo &di, hipStream_t s) {
  FourVariant v = four_variant(variant);
  int grid = fourstep_grid(LOGN, variant, di);
  if (batch < grid) grid = (int)batch;
#define CLFA_V(NS, NT)                                                                                    \
  if (v.nslice == NS && v.nt == NT) {                                                                     \
    hipLaunchKernelGGL((k_fft_4step<LOGN, FWD, SCALE, NS, NT>), dim3(grid), dim3(256 * NS), 0, s, data,   \
                       scratch, t.four, batch);                                                           \
    return hipGetLastError();                                                                             \
  }
  CLFA_V(4, true) CLFA_V(4, false) CLFA_V(2, false) CLFA_V(2, true) CLFA_V(1, false)
#undef CLFA_V
  return hipErrorInvalidValue;
}

template <int LOGN>
static hipError_t launch_4step_n(bool fwd, bool scale, int variant, cpx *data, cpx *scratch, const FftTables &t,
                                 long batch, const DeviceInfo &di, hipStream_t s) {
  if (fwd && scale) return launch_4step_v<LOGN, true, true>(variant, data, scratch, t, batch, di, s);
  if (fwd && !scale) return launch_4step_v<LOGN, true, false>(variant, data, scratch, t, batch, di, s);
  if (!fwd && !scale) return launch_4step_v<LOGN, false, false>(variant, data, scratch, t, batch, di, s);
  return hipErrorInvalidValue;
}

hipError_t launch_fft_4step(int logn, bool fwd, bool scale, int variant, cpx *data, cpx *scratch,
                            const FftTables &t, long batch, const DeviceInfo &di, hipStream_t s) {
  if (batch <= 0) return hipSuccess;
  switch (logn) {
    case 14: return launch_4step_n<14>(fwd, scale, variant, data, scratch, t, batch, di, s);
    case 15: return launch_4step_n<15>(fwd, scale, variant, data, scratch, t, batch, di, s);
    case 16: return launch_4step_n<16>(fwd, scale, variant, data, scratch, t, batch, di, s);
    default: return hipErrorInvalidValue;
  }
}

const char *name_fft_4step(int, bool, int) { return "k_fft_4step"; }

// ---------------------------------------------------------------------------------
// stand-alone pack / unpack / reorder
// ---------------------------------------------------------------------------------

// reference conv (cl_fft.cpp:178-191) over a batch; thread per pair
__global__ __launch_bounds__(256) void k_r2c_pack(cpx *__restrict__ data, const cpx *__restrict__ w2, int m,
                                                  long total_pairs) {
  const int hp = m / 2;
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total_pairs; g += (long)gridDim.x * 256) {
    long b = g / hp;
    int i = (int)(g % hp);
    cpx *c = data + b * m;
    if (i == 0) {
      cpx z = c[0];
      c[0] = mk((z.x + z.y) * .5f, (z.x - z.y) * .5f);
    } else {
      cpx oi, oj;
      r2c_pair(c[i], c[m - i], w2[i], oi, oj);
      c[i] = oi;
      c[m - i] = oj;
    }
  }
}
// reference iconv (cl_fft.cpp:192-205)
__global__ __launch_bounds__(256) void k_c2r_unpack(cpx *__restrict__ data, const cpx *__restrict__ w2, int m,
                                                    long total_pairs) {
  const int hp = m / 2;
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total_pairs; g += (long)gridDim.x * 256) {
    long b = g / hp;
    int i = (int)(g % hp);
    cpx *c = data + b * m;
    if (i == 0) {
      cpx z = c[0];
      c[0] = mk(z.x + z.y, z.x - z.y);
    } else {
      cpx oi, oj;
      c2r_pair(c[i], c[m - i], w2[i], oi, oj);
      c[i] = oi;
      c[m - i] = oj;
    }
  }
}

static int grid_for(long items) {
  long g = (items + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  if (g < 1) g = 1;
  return (int)g;
}

hipError_t launch_r2c_pack(cpx *data, const cpx *w2, int m, long batch, hipStream_t s) {
  long pairs = batch * (m / 2);
  if (pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_r2c_pack, dim3(grid_for(pairs)), dim3(256), 0, s, data, w2, m, pairs);
  return hipGetLastError();
}
hipError_t launch_c2r_unpack(cpx *data, const cpx *w2, int m, long batch, hipStream_t s) {
  long pairs = batch * (m / 2);
  if (pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_c2r_unpack, dim3(grid_for(pairs)), dim3(256), 0, s, data, w2, m, pairs);
  return hipGetLastError();
}

// reference reorder (cl_fft.cpp:24-27): out[k] = in[bitrev(k)].  The table of
// cl_fft.cpp:96-101 is exactly the log2(n)-bit reversal, computed here with
// v_bfrev_b32 instead of a table read.
__global__ __launch_bounds__(256) void k_reorder(cpx *__restrict__ out, const cpx *__restrict__ in, int logn,
                                                 long total) {
  const unsigned mask = (1u << logn) - 1u;
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    unsigned k = (unsigned)g & mask;
    unsigned r = __brev(k) >> (32 - logn);
    out[g] = in[(g - k) + r];
  }
}

hipError_t launch_reorder(cpx *out, const cpx *in, int logn, long batch, hipStream_t s) {
  long total = batch << logn;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_reorder, dim3(grid_for(total)), dim3(256), 0, s, out, in, logn, total);
  return hipGetLastError();
}

}  // namespace clfa
