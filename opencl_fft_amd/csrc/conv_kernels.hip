// conv_kernels.hip — uniformly partitioned overlap-add convolution and direct
// convolution for gfx950 (MI355X), batched over independent channels.
//
// The reference runs 26 launches per block and channel (cl_conv.cpp:393-458:
// reorder, 10 x fft, r2c, convol with float CAS atomics, c2r, reorder, 10 x fft,
// olap).  Here a block is three launches for ALL channels:
//   k_pconv_fwd  real block -> zero-padded real FFT -> packed frame in the ring
//                (reorder + fft + r2c fused; the transform lives in VGPRs + LDS)
//   k_pconv_mac  acc[n] = sum_p A[(wp+p) % nparts][n] (.) B[p][n]; one lane owns
//                two bins and walks the partitions in registers: no atomics,
//                deterministic order, 16-byte coalesced streaming of both rings
//   k_pconv_inv  c2r + inverse FFT + overlap-add + 1/bins scaling fused
// The rings are channels x nparts x bins complex64, resident in HBM.
#include <cstdlib>

#include "fft_wg.hpp"

namespace clfa {

// ---------------------------------------------------------------------------------
// forward: in (channels x pts floats) -> ring frame
// ---------------------------------------------------------------------------------
template <int LOGB>
__global__ __launch_bounds__(LdsGeom<LOGB>::WG) void k_pconv_fwd(const float *__restrict__ in, long in_stride,
                                                                cpx *__restrict__ ring, int frame, int nparts,
                                                                int channels, const cpx *__restrict__ tab_g,
                                                                const cpx *__restrict__ w2_g,
                                                                const float *__restrict__ in_b, cpx *__restrict__ ring_b,
                                                                int frame_b) {
  // blockIdx.y = 1: the second input of a time-varying block (its own ring and frame), same launch
  if (blockIdx.y == 1) {
    in = in_b;
    ring = ring_b;
    frame = frame_b;
  }
  using G = LdsGeom<LOGB>;
  constexpr int N = G::N, E = G::E, T = G::T, WG = G::WG, FPW = G::FPW;
  __shared__ cpx s_tab[G::HALF];
  __shared__ cpx s_x[FPW * G::PADN];
  const int tid = threadIdx.x;
  const int f = tid / T, t = tid % T;
  for (int i = tid; i < N / 2; i += WG) s_tab[i] = tab_g[i];
  __syncthreads();
  cpx *xb = s_x + f * G::PADN;
  const int groups = (channels + FPW - 1) / FPW;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int ch = g * FPW + f;
    const bool active = ch < channels;
    // the real block reinterpreted as N/2 complex values, upper half zero
    // (cl_conv.cpp:399: only bytes>>1 of in1 are written; the rest is zero)
    const cpx *src = reinterpret_cast<const cpx *>(in + (long)(active ? ch : 0) * in_stride);
    cpx v[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int p = t + T * e;
      v[e] = (active && p < N / 2) ? src[p] : mk(0.f, 0.f);
    }
    wg_passes<LOGB, G::LOGE, 0, true>(v, t, s_tab, xb);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; e++) xb[lds_pad(t + T * e)] = v[e];
    __syncthreads();
    if (active) {
      cpx *x = ring + ((long)ch * nparts + frame) * N;
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
  }
}

template <int LOGB>
static hipError_t launch_fwd_one(const PconvGeom &g, const float *in, long in_stride, cpx *ring, int frame,
                                 const cpx *half, const cpx *w2f, hipStream_t s, const float *in_b, cpx *ring_b,
                                 int frame_b) {
  using G = LdsGeom<LOGB>;
  int groups = (g.channels + G::FPW - 1) / G::FPW;
  int grid = groups < 4096 ? groups : 4096;
  hipLaunchKernelGGL((k_pconv_fwd<LOGB>), dim3(grid, in_b ? 2 : 1), dim3(G::WG), 0, s, in, in_stride, ring, frame,
                     g.nparts, g.channels, half, w2f, in_b, ring_b, frame_b);
  return hipGetLastError();
}

hipError_t launch_pconv_forward(const PconvGeom &g, const float *in, long in_stride, cpx *ring, int frame,
                                const cpx *half, const cpx *w2f, hipStream_t s, const float *in_b, cpx *ring_b,
                                int frame_b) {
  switch (g.logb) {
#define CLFA_B(L) \
  case L:         \
    return launch_fwd_one<L>(g, in, in_stride, ring, frame, half, w2f, s, in_b, ring_b, frame_b);
    CLFA_B(1) CLFA_B(2) CLFA_B(3) CLFA_B(4) CLFA_B(5) CLFA_B(6) CLFA_B(7) CLFA_B(8) CLFA_B(9) CLFA_B(10)
    CLFA_B(11) CLFA_B(12) CLFA_B(13)
#undef CLFA_B
    default:
      return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------
// multiply-accumulate over partitions (reference convol, cl_conv_kernels.h:102-118)
// ---------------------------------------------------------------------------------
struct alignas(16) cpx2 {
  cpx a, b;
};

// both rings are read exactly once per block and exceed the Infinity Cache at config 4:
// non-temporal 16-byte loads
__device__ __forceinline__ cpx2 ld_stream(const cpx2 *p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f r = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
  cpx2 o;
  o.a = mk(r.x, r.y);
  o.b = mk(r.z, r.w);
  return o;
}

// one lane = two adjacent bins (16 B) of one channel; loops the partitions of its segment.
// blockIdx.y = segment of the partition axis (1 segment when there are enough channels to fill
// the chip; few channels with long filters are split and summed by k_pconv_reduce in fixed order)
template <int UNROLL>
__global__ __launch_bounds__(256) void k_pconv_mac(const cpx *__restrict__ A, const cpx *__restrict__ B,
                                                   cpx *__restrict__ acc, int wp, int bins, int nparts,
                                                   long total /* channels * bins/2 */, int chunk) {
  const int hb = bins >> 1;
  const int p_begin = blockIdx.y * chunk;
  const int p_end = p_begin + chunk < nparts ? p_begin + chunk : nparts;
  cpx *dst = acc + (long)blockIdx.y * total * 2;
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const long ch = g / hb;
    const int i2 = (int)(g % hb);
    const cpx2 *a = reinterpret_cast<const cpx2 *>(A + ch * (long)nparts * bins) + i2;
    const cpx2 *b = reinterpret_cast<const cpx2 *>(B + ch * (long)nparts * bins) + i2;
    cpx s0 = mk(0.f, 0.f), s1 = mk(0.f, 0.f);
    int fr = wp + p_begin;  // ring frame of partition p_begin (wp = frame of the oldest input block)
    fr = fr < nparts ? fr : fr - nparts;
    int p = p_begin;
    for (; p + UNROLL <= p_end; p += UNROLL) {
      cpx2 av[UNROLL], bv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        int f = fr + u;
        f = f < nparts ? f : f - nparts;
        av[u] = ld_stream(a + (long)f * hb);
        bv[u] = ld_stream(b + (long)(p + u) * hb);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        cpx pr = cmul_plain(av[u].a, bv[u].a);
        const bool dc = i2 == 0;  // packed DC / Nyquist bin: (re*re, im*im); a select keeps the loop body one block
        pr = mk(dc ? av[u].a.x * bv[u].a.x : pr.x, dc ? av[u].a.y * bv[u].a.y : pr.y);
        s0 = cadd(s0, pr);
        s1 = cadd(s1, cmul_plain(av[u].b, bv[u].b));
      }
      fr += UNROLL;
      fr = fr < nparts ? fr : fr - nparts;
    }
    for (; p < p_end; p++) {
      cpx2 av = a[(long)fr * hb], bv = b[(long)p * hb];
      if (i2 == 0) {
        s0.x += av.a.x * bv.a.x;
        s0.y += av.a.y * bv.a.y;
      } else {
        s0 = cadd(s0, cmul_plain(av.a, bv.a));
      }
      s1 = cadd(s1, cmul_plain(av.b, bv.b));
      fr = fr + 1 < nparts ? fr + 1 : 0;
    }
    cpx2 o;
    o.a = s0;
    o.b = s1;
    reinterpret_cast<cpx2 *>(dst)[g] = o;
  }
}

// partial accumulators k = base, base + stride, ..., (count of them, base = blockIdx.y * count * stride, as far as
// nsplit goes) summed in ascending order into accumulator `base`: deterministic.  One launch with count = nsplit sums
// everything; many segments (a single channel with a long filter) are summed as a two-level tree so that the sum is
// not one workgroup's serial walk over hundreds of strided loads.  MAXC > 0: count <= MAXC, all loads are issued
// before the first add (one memory latency instead of `count`).
template <int MAXC>
__global__ __launch_bounds__(256) void k_pconv_reduce(cpx *__restrict__ acc, long total2, int nsplit, int count, int stride) {
  const int base = blockIdx.y * count * stride;
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total2; g += (long)gridDim.x * 256) {
    if constexpr (MAXC > 0) {
      cpx v[MAXC];
#pragma unroll
      for (int k = 0; k < MAXC; k++) {
        const bool ok = k < count && base + k * stride < nsplit;
        v[k] = acc[(long)(ok ? base + k * stride : base) * total2 + g];   // clamped: straight-line loads
        if (!ok) v[k] = mk(0.f, 0.f);
      }
      cpx s = v[0];
#pragma unroll
      for (int k = 1; k < MAXC; k++) s = cadd(s, v[k]);
      acc[(long)base * total2 + g] = s;
    } else {
      cpx s = acc[(long)base * total2 + g];
      for (int k = 1; k < count && base + k * stride < nsplit; k++) s = cadd(s, acc[(long)(base + k * stride) * total2 + g]);
      acc[(long)base * total2 + g] = s;
    }
  }
}

int pconv_mac_split(const PconvGeom &g) {
  // split only when channels x bins/2 gives fewer than ~64K lanes (config 4 has 131072: no split).  A single
  // channel with a long filter — the reference harness' own case, csound/tests.py — has to put the whole chip on
  // the partition axis to stream its rings at HBM speed: up to 512 segments of at least 4 partitions
  // (CLFA_PCONV_SPLIT_MAX: tuning switch, read once).
  static const int cap = [] {
    const char *e = getenv("CLFA_PCONV_SPLIT_MAX");
    const int v = e ? atoi(e) : 512;
    return v < 1 ? 1 : (v > 2048 ? 2048 : v);   // the two-level sum holds 64 groups of 32 partial accumulators
  }();
  long lanes = (long)g.channels * (g.bins / 2);
  const long target = lanes <= 1024 ? 128L * 1024 : 64L * 1024;   // measured: the finer split pays up to pts = 2048
  long want = (target + lanes - 1) / lanes;
  if (lanes >= 64L * 1024) want = 1;
  if (want > cap) want = cap;
  if (want > g.nparts / 4) want = g.nparts / 4;   // (segments of 1 or 2 partitions: measured, mixed — not kept)
  return want < 1 ? 1 : (int)want;
}

hipError_t launch_pconv_mac(const PconvGeom &g, const cpx *ringA, const cpx *ringB, int wp, cpx *acc,
                            hipStream_t s, bool reduce) {
  long total = (long)g.channels * (g.bins / 2);
  long grid = (total + 255) / 256;
  if (grid > 256 * 64) grid = 256 * 64;
  const int nsplit = pconv_mac_split(g);
  const int chunk = (g.nparts + nsplit - 1) / nsplit;
  hipLaunchKernelGGL((k_pconv_mac<4>), dim3((int)grid, nsplit), dim3(256), 0, s, ringA, ringB, acc, wp, g.bins,
                     g.nparts, total, chunk);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || nsplit == 1 || !reduce) return e;
  long total2 = total * 2, rgrid = (total2 + 255) / 256;
  if (rgrid > 4096) rgrid = 4096;
  auto sum = [&](int groups, int count, int stride) -> hipError_t {   // the smallest unrolled form that holds `count`
    const dim3 grid((int)rgrid, groups), block(256);
    if (count > 64) return hipErrorInvalidValue;   // (cannot happen below the 2048 cap: a dropped partial sum must not pass silently)
    if (count <= 2) hipLaunchKernelGGL(k_pconv_reduce<2>, grid, block, 0, s, acc, total2, nsplit, count, stride);
    else if (count <= 4) hipLaunchKernelGGL(k_pconv_reduce<4>, grid, block, 0, s, acc, total2, nsplit, count, stride);
    else if (count <= 8) hipLaunchKernelGGL(k_pconv_reduce<8>, grid, block, 0, s, acc, total2, nsplit, count, stride);
    else if (count <= 16) hipLaunchKernelGGL(k_pconv_reduce<16>, grid, block, 0, s, acc, total2, nsplit, count, stride);
    else if (count <= 32) hipLaunchKernelGGL(k_pconv_reduce<32>, grid, block, 0, s, acc, total2, nsplit, count, stride);
    else hipLaunchKernelGGL(k_pconv_reduce<64>, grid, block, 0, s, acc, total2, nsplit, count, stride);
    return hipGetLastError();
  };
  if (nsplit > 64) {   // groups of 32, then the group sums
    const int groups = (nsplit + 31) / 32;
    if ((e = sum(groups, 32, 1)) != hipSuccess) return e;
    return sum(1, groups, 32);
  }
  return sum(1, nsplit, 1);
}

// ---------------------------------------------------------------------------------
// inverse: acc -> c2r -> inverse FFT -> overlap-add (reference c2r + reorder +
// fft + olap, cl_conv_kernels.h:87-100, 120-124)
// ---------------------------------------------------------------------------------
template <int LOGB>
__global__ __launch_bounds__(LdsGeom<LOGB>::WG) void k_pconv_inv(const cpx *__restrict__ acc,
                                                                float *__restrict__ tail,
                                                                float *__restrict__ out, int channels,
                                                                const cpx *__restrict__ tab_g,
                                                                const cpx *__restrict__ w2_g, int nsplit) {
  using G = LdsGeom<LOGB>;
  constexpr int N = G::N, E = G::E, T = G::T, WG = G::WG, FPW = G::FPW;
  __shared__ cpx s_tab[G::HALF];
  __shared__ cpx s_x[FPW * G::PADN];
  const int tid = threadIdx.x;
  const int f = tid / T, t = tid % T;
  for (int i = tid; i < N / 2; i += WG) s_tab[i] = tab_g[i];
  __syncthreads();
  cpx *xb = s_x + f * G::PADN;
  const int groups = (channels + FPW - 1) / FPW;
  const long part = (long)channels * N;   // one partial accumulator (MAC split over the partition axis)
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int ch = g * FPW + f;
    const bool active = ch < channels;
    const cpx *x0 = acc + (long)(active ? ch : 0) * N;
    // sum of the partial accumulators, ascending (the order k_pconv_reduce uses): no separate launch
    auto x = [&](int i) {
      cpx sum = x0[i];
      for (int k = 1; k < nsplit; k++) sum = cadd(sum, x0[k * part + i]);
      return sum;
    };
    __syncthreads();
    if (active) {
      for (int i = t; i < N / 2; i += T) {
        if (i == 0) {
          cpx c0 = x(0);
          xb[0] = mk(c0.x + c0.y, c0.x - c0.y);
          xb[lds_pad(N / 2)] = x(N / 2);
        } else {
          cpx oi, oj;
          c2r_pair(x(i), x(N - i), w2_g[i], oi, oj);
          xb[lds_pad(i)] = oi;
          xb[lds_pad(N - i)] = oj;
        }
      }
    }
    __syncthreads();
    cpx v[E];
    pass_gather<LOGB, G::LOGE>(v, t, [&](int p) { return xb[lds_pad(p)]; });
    wg_passes<LOGB, G::LOGE, 0, false>(v, t, s_tab, xb);
    if (active) {
      // v[e] holds real samples 2p, 2p+1 of the 2*bins-point block, p = t + T*e.
      // p < N/2: output half (+ old tail, / bins); p >= N/2: the new tail, unscaled.
      constexpr float inv = 1.0f / (float)N;
      cpx *o = reinterpret_cast<cpx *>(out + (long)ch * N);
      cpx *tl = reinterpret_cast<cpx *>(tail + (long)ch * N);
      if constexpr (E >= 2) {
#pragma unroll
        for (int e = 0; e < E / 2; e++) {
          const int p = t + T * e;
          cpx old = tl[p];
          o[p] = mk((v[e].x + old.x) * inv, (v[e].y + old.y) * inv);
          tl[p] = v[e + E / 2];
        }
      }
    }
  }
}

template <int LOGB>
static hipError_t launch_inv_one(const PconvGeom &g, const cpx *acc, float *tail, float *out, const cpx *half,
                                 const cpx *w2i, hipStream_t s, int nsplit) {
  using G = LdsGeom<LOGB>;
  int groups = (g.channels + G::FPW - 1) / G::FPW;
  int grid = groups < 4096 ? groups : 4096;
  hipLaunchKernelGGL((k_pconv_inv<LOGB>), dim3(grid), dim3(G::WG), 0, s, acc, tail, out, g.channels, half, w2i, nsplit);
  return hipGetLastError();
}

hipError_t launch_pconv_inverse(const PconvGeom &g, const cpx *acc, float *tail, float *out, const cpx *half,
                                const cpx *w2i, hipStream_t s, int nsplit) {
  switch (g.logb) {
#define CLFA_B(L) \
  case L:         \
    return launch_inv_one<L>(g, acc, tail, out, half, w2i, s, nsplit);
    CLFA_B(1) CLFA_B(2) CLFA_B(3) CLFA_B(4) CLFA_B(5) CLFA_B(6) CLFA_B(7) CLFA_B(8) CLFA_B(9) CLFA_B(10)
    CLFA_B(11) CLFA_B(12) CLFA_B(13)
#undef CLFA_B
    default:
      return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------
// fused block: forward FFT -> MAC over all partitions -> inverse FFT + overlap-add, one
// workgroup per channel, ONE launch per block (used when there are enough channels to fill the
// chip).  Everything a channel needs stays inside its workgroup, so the only synchronisation is
// __syncthreads(): the new spectrum frame is stored to the ring and re-read by the same
// workgroup (workgroup-scope visibility), the accumulator lives in LDS.
// ---------------------------------------------------------------------------------
template <int LOGB, bool TV, bool DEEP = false>
__global__ __launch_bounds__(256) void k_pconv_fused(const float *__restrict__ in1, const float *__restrict__ in2,
                                                     cpx *__restrict__ ringA, cpx *__restrict__ ringB,
                                                     float *__restrict__ tail, float *__restrict__ out, int frame1,
                                                     int frame2, int wp, int nparts, const cpx *__restrict__ tab_g,
                                                     const cpx *__restrict__ w2f_g, const cpx *__restrict__ w2i_g) {
  using G = LdsGeom<LOGB>;
  constexpr int N = G::N, E = G::E, T = G::T;   // N = bins; T = N/16 lanes run the FFTs
  static_assert(T <= 256 && N / 2 >= 256, "fused block kernel covers bins 512..4096");
  constexpr int HB = N / 2, IPT = HB / 256;     // 16-byte items (two bins) per lane in the MAC
  __shared__ cpx s_tab[G::HALF];
  __shared__ cpx s_x[G::PADN];
  __shared__ cpx s_acc[N];
  const int tid = threadIdx.x;
  const int ch = blockIdx.x;
  for (int i = tid; i < N / 2; i += 256) s_tab[i] = tab_g[i];
  __syncthreads();

  // ---- forward chain(s): reference reorder + fft + r2c (cl_conv.cpp:399-419 / 465-513) ----------
  auto forward_load = [&](const float *in, cpx (&v)[E]) {
    if (tid < T) {
      const cpx *src = reinterpret_cast<const cpx *>(in + (long)ch * N);
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int p = tid + T * e;
        v[e] = p < N / 2 ? src[p] : mk(0.f, 0.f);
      }
    }
  };
  auto forward_rest = [&](cpx (&v)[E], cpx *ring, int frame) {
    // all 256 lanes walk the barriers; lanes >= T carry dummies and touch no LDS slot of the transform
    if (tid < T) pass_compute<LOGB, G::LOGE, 0, true>(v, tid, s_tab);
    constexpr int LOGR0 = pass_logr(LOGB, G::LOGE, 0);
    static_assert(LOGR0 == 4, "16 points per lane");
    // unrolled pass chain with workgroup-wide barriers
    __syncthreads();
    if (tid < T) pass_scatter<LOGB, G::LOGE, 0>(v, tid, [&](int p, cpx val) { s_x[lds_pad(p)] = val; });
    __syncthreads();
    if (tid < T) {
      pass_gather<LOGB, G::LOGE>(v, tid, [&](int p) { return s_x[lds_pad(p)]; });
      pass_compute<LOGB, G::LOGE, 4, true>(v, tid, s_tab);
    }
    if constexpr (LOGB > 8) {
      __syncthreads();
      if (tid < T) pass_scatter<LOGB, G::LOGE, 4>(v, tid, [&](int p, cpx val) { s_x[lds_pad(p)] = val; });
      __syncthreads();
      if (tid < T) {
        pass_gather<LOGB, G::LOGE>(v, tid, [&](int p) { return s_x[lds_pad(p)]; });
        pass_compute<LOGB, G::LOGE, 8, true>(v, tid, s_tab);
      }
    }
    __syncthreads();
    if (tid < T) {
#pragma unroll
      for (int e = 0; e < E; e++) s_x[lds_pad(tid + T * e)] = v[e];
    }
    __syncthreads();
    cpx *x = ring + ((long)ch * nparts + frame) * N;
    for (int i = tid; i < N / 2; i += 256) {
      const int j = i == 0 ? N / 2 : N - i;
      const cpx ci = s_x[lds_pad(i)], cj = s_x[lds_pad(j)];
      cpx oi, oj;
      r2c_pair(ci, cj, w2f_g[i], oi, oj);
      if (i == 0) {
        oi = mk((ci.x + ci.y) * .5f, (ci.x - ci.y) * .5f);
        oj = cj;
      }
      x[i] = oi;
      x[j] = oj;
    }
  };
  // Fewer channels than CUs (DEEP), static response: the MAC needs the frame stored below only for its LAST partition (the ring
  // position frame1 = wp - 1 pairs with partition nparts - 1, clfa_pconv_process_dev), so the first kPre partitions are
  // requested in front of the forward chain — behind the block's own samples: the counter of outstanding loads is in order —
  // and land while it runs: a workgroup that is alone with its latency starts its MAC with a full queue (160 channels:
  // 51.2-52.1 -> 46.0-46.5 us per block).  Same products, same order of the sums.  With a workgroup on every CU it buys
  // nothing (256 channels 67.3-68.2 -> 68.7-69.5), profiles/pconv_prefetch_r05.txt.
  constexpr int kPre = (TV || !DEEP || IPT > 2) ? 0 : 8;
  const cpx2 *const mac_a = reinterpret_cast<const cpx2 *>(ringA + (long)ch * nparts * N);
  const cpx2 *const mac_b = reinterpret_cast<const cpx2 *>(ringB + (long)ch * nparts * N);
  [[maybe_unused]] cpx2 pa[kPre ? kPre : 1][IPT], pb[kPre ? kPre : 1][IPT];
  [[maybe_unused]] const bool pre = kPre > 0 && nparts >= 4 * kPre;   // uniform (9 .. 12 partitions: +3 %, 40 and more: 0 .. -12 %)
  cpx vin[E];
  forward_load(in1, vin);
  if constexpr (kPre > 0) {
    if (pre) {
#pragma unroll
      for (int q = 0; q < kPre; q++) {
        const int fq = wp + q < nparts ? wp + q : wp + q - nparts;
#pragma unroll
        for (int k = 0; k < IPT; k++) {
          pa[q][k] = ld_stream(mac_a + (long)fq * HB + tid + 256 * k);
          pb[q][k] = ld_stream(mac_b + (long)q * HB + tid + 256 * k);
        }
      }
    }
  }
  forward_rest(vin, ringA, frame1);
  if constexpr (TV) {
    forward_load(in2, vin);
    forward_rest(vin, ringB, frame2);
  }
  __syncthreads();   // the frames just stored are re-read below by this workgroup

  // ---- MAC over all partitions (reference convol, cl_conv_kernels.h:102-118) -----------------------
  {
    const cpx2 *a = mac_a;
    const cpx2 *b = mac_b;
    cpx s0[IPT], s1[IPT];
#pragma unroll
    for (int k = 0; k < IPT; k++) s0[k] = s1[k] = mk(0.f, 0.f);
    int fr = wp;
    auto mac = [&](const cpx2 (&av)[IPT], const cpx2 (&bv)[IPT]) {
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        cpx pr = cmul_plain(av[k].a, bv[k].a);
        if (k == 0) {  // lane 0: packed DC / Nyquist bin, (re*re, im*im) — a select, not a branch (a branch
          const bool dc = tid == 0;   // splits the loop body and the streaming loads stop overlapping)
          pr = mk(dc ? av[k].a.x * bv[k].a.x : pr.x, dc ? av[k].a.y * bv[k].a.y : pr.y);
        }
        s0[k] = cadd(s0[k], pr);
        s1[k] = cadd(s1[k], cmul_plain(av[k].b, bv[k].b));
      }
    };
    auto step = [&](int p) {
      cpx2 av[IPT], bv[IPT];
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        av[k] = ld_stream(a + (long)fr * HB + tid + 256 * k);
        bv[k] = ld_stream(b + (long)p * HB + tid + 256 * k);
      }
      mac(av, bv);
      fr = fr + 1 < nparts ? fr + 1 : 0;
    };
    int p0 = 0;
    if constexpr (kPre > 0) {
      if (pre) {
#pragma unroll
        for (int q = 0; q < kPre; q++) mac(pa[q], pb[q]);
        p0 = kPre;
        fr = wp + kPre < nparts ? wp + kPre : wp + kPre - nparts;
      }
    }
    // loads of 4 partitions in flight per lane fill the memory system when every CU has a workgroup; with fewer
    // channels than CUs (DEEP) a workgroup is alone with its latency and 8 pay (160 channels: 55.5 -> 51.8 us)
    if constexpr (DEEP) {
#pragma unroll 8
      for (int p = p0; p < nparts; p++) step(p);
    } else {
#pragma unroll 4
      for (int p = p0; p < nparts; p++) step(p);
    }
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      s_acc[2 * (tid + 256 * k)] = s0[k];
      s_acc[2 * (tid + 256 * k) + 1] = s1[k];
    }
  }
  __syncthreads();

  // ---- inverse chain: c2r + inverse FFT + overlap-add (cl_conv_kernels.h:87-100, 120-124) -------------
  for (int i = tid; i < N / 2; i += 256) {
    if (i == 0) {
      const cpx c0 = s_acc[0];
      s_x[0] = mk(c0.x + c0.y, c0.x - c0.y);
      s_x[lds_pad(N / 2)] = s_acc[N / 2];
    } else {
      cpx oi, oj;
      c2r_pair(s_acc[i], s_acc[N - i], w2i_g[i], oi, oj);
      s_x[lds_pad(i)] = oi;
      s_x[lds_pad(N - i)] = oj;
    }
  }
  __syncthreads();
  {
    cpx v[E];
    if (tid < T) {
      pass_gather<LOGB, G::LOGE>(v, tid, [&](int p) { return s_x[lds_pad(p)]; });
      pass_compute<LOGB, G::LOGE, 0, false>(v, tid, s_tab);
    }
    __syncthreads();
    if (tid < T) pass_scatter<LOGB, G::LOGE, 0>(v, tid, [&](int p, cpx val) { s_x[lds_pad(p)] = val; });
    __syncthreads();
    if (tid < T) {
      pass_gather<LOGB, G::LOGE>(v, tid, [&](int p) { return s_x[lds_pad(p)]; });
      pass_compute<LOGB, G::LOGE, 4, false>(v, tid, s_tab);
    }
    if constexpr (LOGB > 8) {
      __syncthreads();
      if (tid < T) pass_scatter<LOGB, G::LOGE, 4>(v, tid, [&](int p, cpx val) { s_x[lds_pad(p)] = val; });
      __syncthreads();
      if (tid < T) {
        pass_gather<LOGB, G::LOGE>(v, tid, [&](int p) { return s_x[lds_pad(p)]; });
        pass_compute<LOGB, G::LOGE, 8, false>(v, tid, s_tab);
      }
    }
    if (tid < T) {
      constexpr float inv = 1.0f / (float)N;
      cpx *o = reinterpret_cast<cpx *>(out + (long)ch * N);
      cpx *tl = reinterpret_cast<cpx *>(tail + (long)ch * N);
#pragma unroll
      for (int e = 0; e < E / 2; e++) {
        const int p = tid + T * e;
        const cpx old = tl[p];
        o[p] = mk((v[e].x + old.x) * inv, (v[e].y + old.y) * inv);
        tl[p] = v[e + E / 2];
      }
    }
  }
}

bool pconv_fused_ok(const PconvGeom &g, const DeviceInfo &di) {
  // one workgroup per channel: below ~8/15 of the CUs the chip is too empty for it (measured at pts 1024, 94
  // partitions: 128 channels 48 us fused against 46 us on the three-kernel chain, 144 channels 50 against 56)
  return g.logb >= 9 && g.logb <= 12 && g.channels * 15 >= di.num_cus * 8;
}

template <int LOGB>
static hipError_t launch_fused_one(const PconvGeom &g, const float *in1, const float *in2, cpx *ringA, cpx *ringB,
                                   float *tail, float *out, int frame1, int frame2, int wp, const cpx *half,
                                   const cpx *w2f, const cpx *w2i, hipStream_t s, bool deep) {
#define CLFA_FUSED(TVF, DP)                                                                                          \
  hipLaunchKernelGGL((k_pconv_fused<LOGB, TVF, DP>), dim3(g.channels), dim3(256), 0, s, in1, in2, ringA, ringB, tail, out, \
                     frame1, frame2, wp, g.nparts, half, w2f, w2i)
  if (in2 && deep) CLFA_FUSED(true, true);
  else if (in2) CLFA_FUSED(true, false);
  else if (deep) CLFA_FUSED(false, true);
  else CLFA_FUSED(false, false);
#undef CLFA_FUSED
  return hipGetLastError();
}

hipError_t launch_pconv_fused(const PconvGeom &g, const float *in1, const float *in2, cpx *ringA, cpx *ringB,
                              float *tail, float *out, int frame1, int frame2, int wp, const cpx *half,
                              const cpx *w2f, const cpx *w2i, hipStream_t s, bool deep) {
  switch (g.logb) {
    case 9: return launch_fused_one<9>(g, in1, in2, ringA, ringB, tail, out, frame1, frame2, wp, half, w2f, w2i, s, deep);
    case 10: return launch_fused_one<10>(g, in1, in2, ringA, ringB, tail, out, frame1, frame2, wp, half, w2f, w2i, s, deep);
    case 11: return launch_fused_one<11>(g, in1, in2, ringA, ringB, tail, out, frame1, frame2, wp, half, w2f, w2i, s, deep);
    case 12: return launch_fused_one<12>(g, in1, in2, ringA, ringB, tail, out, frame1, frame2, wp, half, w2f, w2i, s, deep);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------
// cooperative block: ONE launch per block for a FEW channels — the single-instance call that the reference's
// opcodes and its own harness make (cl_conv.cpp:393-458 / 460-548 once per ksmps block; csound/tests.py:22-29).
// With one channel the chain above is four or five dependent launches of 5 us each.  Here S workgroups per
// channel split the BIN axis of the multiply-accumulate (no partial sums to add up: a bin's whole sum over the
// partitions is formed inside one workgroup, rows of lanes walking the partitions in parallel and meeting in
// LDS in fixed order), every workgroup transforms the new input block itself (a few microseconds of redundant
// arithmetic instead of a grid-wide hand-over of the new frame; workgroup 0 also files it in the ring for the
// blocks to come), and the only inter-workgroup step is the hand-over of the finished accumulator slices —
// bins x 8 bytes per channel in all — to whichever workgroup arrives LAST at the channel's counter: it runs the
// inverse chain.  No workgroup ever waits for another one: nothing spins.
// Hand-over protocol (MI355X_MICROARCH.md, "Valid forms"): slices stored with agent-scope (sc1) stores, every
// storing wave s_waitcnt vmcnt(0), workgroup barrier, ONE lane's agent-scope atomic add on the channel's counter;
// the workgroup whose add returns S - 1 reads every slice with agent-scope (sc1) loads after its own barrier.
// ---------------------------------------------------------------------------------
// The hand-overs below are written for gfx950's memory pipeline (sc1 stores write through, vmcnt counts stores, sc1 loads
// are served past the CU's L1): another target needs the C++ release / acquire forms instead.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "conv_kernels.hip: the inter-workgroup hand-overs are gfx950-specific (see MI355X_MICROARCH.md, 'Valid forms')"
#endif
// The form without an acquire is the one MI355X_MICROARCH.md measured ("Valid forms": sc1 stores, every storing wave's
// vmcnt(0), barrier, one lane's agent-scope add; the workgroup whose add came last loads with sc1 loads) — for launches of
// at most ONE workgroup per CU.  A launch with more workgroups than CUs (acquire != 0, set by the launcher) is outside
// that table: there the arriving lane of the last workgroup runs the documented consumer form as well — one agent-scope
// acquire (buffer_inv sc1) and its wait, in front of the barrier that releases the other waves' loads.
__device__ __forceinline__ void handover_acquire(int acquire) {
  if (acquire) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
// The arriving lane's add carries the C++ model's release as well (CLFA_HANDOVER_RELEASE): the hand-written form above
// orders the slice stores in hardware, but nothing in it tells hipcc that they must stay above the add — with the
// release a future compiler cannot sink a slice store below the counter.  On gfx950 it costs a buffer_wbl2 sc1 and a wait
// in ONE lane after the barrier (profiles/handover_release_r05.txt).
#ifndef CLFA_HANDOVER_RELEASE
#define CLFA_HANDOVER_RELEASE 1
#endif
__device__ __forceinline__ unsigned handover_arrive(unsigned *counter) {
  return __hip_atomic_fetch_add(counter, 1u, CLFA_HANDOVER_RELEASE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(cpx *p, cpx v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ cpx ld_agent(const cpx *p) {
  return __builtin_bit_cast(cpx, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT));
}

template <int LOGB, bool TV>
__global__ __launch_bounds__(LdsGeom<LOGB>::WG) void k_pconv_coop(const float *__restrict__ in1, const float *__restrict__ in2,
                                                    cpx *__restrict__ ringA, cpx *__restrict__ ringB,
                                                    float *__restrict__ tail, float *__restrict__ out, int frame1,
                                                    int frame2, int wp, int nparts, const cpx *__restrict__ tab_g,
                                                    const cpx *__restrict__ w2f_g, const cpx *__restrict__ w2i_g,
                                                    cpx *__restrict__ xacc, unsigned *__restrict__ counters, int logs,
                                                    int sparts, int acquire) {
  using G = LdsGeom<LOGB>;
  constexpr int N = G::N, E = G::E, T = G::T, HB = N / 2;   // N = bins; T = N/16 lanes run the FFTs
  constexpr int WG = G::WG;                                  // 256 lanes; 512 for partitions of 8192 samples
  constexpr int kSliceMax = 512;                             // bins per workgroup (host: logs >= LOGB - 9)
  static_assert(LOGB >= 5 && LOGB <= 13 && T <= WG, "bins 32..8192: slices of 32 bins, 16 points per lane in the transforms");
  __shared__ cpx s_tab[G::HALF];
  __shared__ cpx s_x[G::PADN];
  __shared__ cpx s_fa[kSliceMax];            // this workgroup's slice of the new input block's packed spectrum (frame1 of ring A)
  __shared__ cpx s_fb[TV ? kSliceMax : 1];   // ... of the second input's (frame2 of ring B)
  __shared__ cpx2 s_red[WG];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  // workgroup = (slice sl of the bins, segment ps of the partition axis)
  const int ch = blockIdx.y, S = 1 << logs, sl = blockIdx.x & (S - 1), ps = blockIdx.x >> logs;
  for (int i = tid; i < N / 2; i += WG) s_tab[i] = tab_g[i];
  __syncthreads();

  // ---- this workgroup's part of the multiply-accumulate: slice = N >> logs bins = IW 16-byte items; lane = item li of
  // partition row pr; rows walk p = p_begin + pr, + NR, ...  The first loads of the walk (and the ring operands of the
  // new frames' terms) are issued HERE, before the forward transforms: they need nothing from them, and their memory
  // latency then runs under 1-2 us of butterflies instead of after them.
  const int iw = HB >> logs, nr = WG / iw;
  const int li = tid % iw, pr = tid / iw;
  const int item = sl * iw + li;                       // 16-byte item (bins 2 item, 2 item + 1) of the frame
  const cpx2 *ra = reinterpret_cast<const cpx2 *>(ringA + (long)ch * nparts * N) + item;
  const cpx2 *rb = reinterpret_cast<const cpx2 *>(ringB + (long)ch * nparts * N) + item;
  const int p1 = nparts - 1;                           // (wp + p1) % nparts == frame1: wp = frame1 + 1
  const int chunk = (nparts + sparts - 1) / sparts;    // this workgroup's partitions [p_begin, p_end)
  const int p_begin = ps * chunk, p_end = p_begin + chunk < nparts ? p_begin + chunk : nparts;
  constexpr int UNR = 4;
  cpx2 av0[UNR], bv0[UNR];
  bool live0[UNR];
#pragma unroll
  for (int u = 0; u < UNR; u++) {   // (clamped, not predicated: straight-line loads)
    const int pp = p_begin + pr + u * nr;
    const bool ok = pp < p_end;
    const int pc = ok ? pp : p_end - 1;
    int fr = wp + pc;
    fr = fr < nparts ? fr : fr - nparts;
    av0[u] = ld_stream(ra + (long)fr * HB);
    bv0[u] = ld_stream(rb + (long)pc * HB);
    live0[u] = ok && pc != p1 && !(TV && pc == frame2);
  }
  const cpx2 b_p1 = rb[(long)p1 * HB];                 // ring operand of the new A frame's term
  int fr2 = wp + (TV ? frame2 : 0);
  fr2 = fr2 < nparts ? fr2 : fr2 - nparts;
  const cpx2 a_f2 = ra[(long)fr2 * HB];                // ... of the new B frame's term (time-varying blocks)
  // ... and so are the pack / unpack twiddles of the lane's bins and the overlap-add tail: every global load that does
  // not depend on this launch's results leaves at the top of the kernel — a block is a chain of dependent steps of a
  // microsecond each, and every load left in the middle of it is one more
  constexpr int NI = (N / 2 + WG - 1) / WG;
  cpx w2f_r[NI], w2i_r[NI];
#pragma unroll
  for (int q = 0; q < NI; q++) {
    const int i = tid + q * WG;
    w2f_r[q] = w2f_g[i < N / 2 ? i : 0];
    w2i_r[q] = w2i_g[i < N / 2 ? i : 0];
  }
  cpx tail_r[E / 2];
  {
    const cpx *tl = reinterpret_cast<const cpx *>(tail + (long)ch * N);
#pragma unroll
    for (int e = 0; e < E / 2; e++) tail_r[e] = tl[(tid < T ? tid : 0) + T * e];
  }

  // ---- forward chain(s) in every workgroup: reference reorder + fft + r2c (cl_conv.cpp:399-419 / 465-513) ----
  // Time-varying blocks transform both inputs AT ONCE where the lanes allow it (2 T <= 256): lanes [0, T) take in1,
  // lanes [T, 2 T) in2, each group with its own exchange buffer — one pass chain's worth of barriers, not two.
  constexpr bool DUAL = TV && 2 * T <= WG;
  __shared__ cpx s_x2[DUAL ? G::PADN : 1];
  auto forward = [&](const float *inA, const float *inB, bool dual) {
    // inB / ring B only when dual; otherwise one input (inA) by lanes [0, T)
    const int grp = dual ? tid / T : 0, tt = dual ? tid % T : tid;
    const bool work = dual ? tid < 2 * T : tid < T;
    cpx *sx = (DUAL && grp == 1) ? s_x2 : s_x;
    cpx v[E];
    if (work) {
      const cpx *src = reinterpret_cast<const cpx *>((grp == 1 ? inB : inA) + (long)ch * N);
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int p = tt + T * e;
        v[e] = p < N / 2 ? src[p] : mk(0.f, 0.f);
      }
      pass_compute<LOGB, G::LOGE, 0, true>(v, tt, s_tab);
    }
    // (LOGB > 4 always: a second pass of radix 2^min(4, LOGB - 4), a third one above 256 bins)
    __syncthreads();
    if (work) pass_scatter<LOGB, G::LOGE, 0>(v, tt, [&](int p, cpx val) { sx[lds_pad(p)] = val; });
    __syncthreads();
    if (work) {
      pass_gather<LOGB, G::LOGE>(v, tt, [&](int p) { return sx[lds_pad(p)]; });
      pass_compute<LOGB, G::LOGE, 4, true>(v, tt, s_tab);
    }
    if constexpr (LOGB > 8) {
      __syncthreads();
      if (work) pass_scatter<LOGB, G::LOGE, 4>(v, tt, [&](int p, cpx val) { sx[lds_pad(p)] = val; });
      __syncthreads();
      if (work) {
        pass_gather<LOGB, G::LOGE>(v, tt, [&](int p) { return sx[lds_pad(p)]; });
        pass_compute<LOGB, G::LOGE, 8, true>(v, tt, s_tab);
      }
    }
    if constexpr (LOGB > 12) {
      __syncthreads();
      if (work) pass_scatter<LOGB, G::LOGE, 8>(v, tt, [&](int p, cpx val) { sx[lds_pad(p)] = val; });
      __syncthreads();
      if (work) {
        pass_gather<LOGB, G::LOGE>(v, tt, [&](int p) { return sx[lds_pad(p)]; });
        pass_compute<LOGB, G::LOGE, 12, true>(v, tt, s_tab);
      }
    }
    __syncthreads();
    if (work) {
#pragma unroll
      for (int e = 0; e < E; e++) sx[lds_pad(tt + T * e)] = v[e];
    }
    __syncthreads();
  };
  // packed spectrum (reference r2c) of the transform left in `sx` -> sf (LDS) and, by workgroup 0, the ring frame
  const int bw = N >> logs, b0 = sl * bw;   // this workgroup's bins [b0, b0 + bw)
  auto pack = [&](const cpx *sx, cpx *ring, int frame, cpx *sf) {
    cpx *x = ring + ((long)ch * nparts + frame) * N;
#pragma unroll
    for (int q = 0; q < NI; q++) {
      const int i = tid + q * WG;
      if (i >= N / 2) break;
      const int j = i == 0 ? N / 2 : N - i;
      const cpx ci = sx[lds_pad(i)], cj = sx[lds_pad(j)];
      cpx oi, oj;
      r2c_pair(ci, cj, w2f_r[q], oi, oj);
      if (i == 0) {
        oi = mk((ci.x + ci.y) * .5f, (ci.x - ci.y) * .5f);
        oj = cj;
      }
      if (i >= b0 && i < b0 + bw) sf[i - b0] = oi;
      if (j >= b0 && j < b0 + bw) sf[j - b0] = oj;
      if (blockIdx.x == 0) {   // filed in the ring for the blocks to come; nobody reads it from there in this launch
        x[i] = oi;
        x[j] = oj;
      }
    }
  };
  if constexpr (DUAL) {
    forward(in1, in2, true);
    pack(s_x, ringA, frame1, s_fa);
    pack(s_x2, ringB, frame2, s_fb);
    __syncthreads();
  } else {
    forward(in1, nullptr, false);
    pack(s_x, ringA, frame1, s_fa);
    __syncthreads();
    if constexpr (TV) {
      forward(in2, nullptr, false);
      pack(s_x, ringB, frame2, s_fb);
      __syncthreads();
    }
  }

  // ---- MAC over all partitions for this workgroup's slice of the bins (reference convol, cl_conv_kernels.h:102-118)
  {
    const cpx2 *a = ra, *b = rb;
    const bool dc = item == 0;                           // packed DC / Nyquist bin: (re*re, im*im)
    cpx s0 = mk(0.f, 0.f), s1 = mk(0.f, 0.f);
    auto term = [&](const cpx2 &av, const cpx2 &bv, bool live) {
      cpx pa = cmul_plain(av.a, bv.a);
      pa = mk(dc ? av.a.x * bv.a.x : pa.x, dc ? av.a.y * bv.a.y : pa.y);
      const cpx pb = cmul_plain(av.b, bv.b);
      s0 = cadd(s0, mk(live ? pa.x : 0.f, live ? pa.y : 0.f));
      s1 = cadd(s1, mk(live ? pb.x : 0.f, live ? pb.y : 0.f));
    };
    // the frames written by THIS launch (frame1 of A; frame2 of B) are taken from LDS below: in the loop their
    // (stale) ring contents are read like any other frame and dropped by a select — no branch in the stream
#pragma unroll
    for (int u = 0; u < UNR; u++) term(av0[u], bv0[u], live0[u]);   // the batch fetched before the transforms
    int p = p_begin + pr + UNR * nr;
    for (; p + (UNR - 1) * nr < p_end; p += UNR * nr) {
      cpx2 av[UNR], bv[UNR];
      bool live[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const int pp = p + u * nr;
        int fr = wp + pp;
        fr = fr < nparts ? fr : fr - nparts;
        av[u] = ld_stream(a + (long)fr * HB);
        bv[u] = ld_stream(b + (long)pp * HB);
        live[u] = pp != p1 && !(TV && pp == frame2);
      }
#pragma unroll
      for (int u = 0; u < UNR; u++) term(av[u], bv[u], live[u]);
    }
    for (; p < p_end; p += nr) {
      int fr = wp + p;
      fr = fr < nparts ? fr : fr - nparts;
      term(ld_stream(a + (long)fr * HB), ld_stream(b + (long)p * HB), p != p1 && !(TV && p == frame2));
    }
    // the terms of the new frames, by the row that owns their partition
    const cpx2 *fa = reinterpret_cast<const cpx2 *>(s_fa) + li;
    if (p1 >= p_begin && p1 < p_end && pr == (p1 - p_begin) % nr) {
      cpx2 bv;
      if (TV && p1 == frame2) bv = reinterpret_cast<const cpx2 *>(s_fb)[li];
      else bv = b_p1;
      term(*fa, bv, true);
    }
    if constexpr (TV) {
      if (frame2 != p1 && frame2 >= p_begin && frame2 < p_end && pr == (frame2 - p_begin) % nr) {
        term(a_f2, reinterpret_cast<const cpx2 *>(s_fb)[li], true);
      }
    }
    cpx2 mine;
    mine.a = s0;
    mine.b = s1;
    s_red[tid] = mine;
    __syncthreads();
    if (pr == 0) {   // rows summed in ascending order: deterministic
      cpx t0 = s_red[li].a, t1 = s_red[li].b;
      for (int r = 1; r < nr; r++) {
        t0 = cadd(t0, s_red[r * iw + li].a);
        t1 = cadd(t1, s_red[r * iw + li].b);
      }
      cpx *dst = xacc + ((long)ch * sparts + ps) * N + 2 * item;
      st_agent(dst, t0);
      st_agent(dst + 1, t1);
    }
  }
  // ---- hand-over: the last workgroup to arrive at the channel's counter owns the inverse chain
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned old = handover_arrive(counters + ch);
    s_last = old == (unsigned)(S * sparts - 1);
    if (s_last) {
      __hip_atomic_store(counters + ch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next block's launch
      handover_acquire(acquire);
    }
  }
  __syncthreads();
  if (!s_last) return;

  // ---- inverse chain: c2r + inverse FFT + overlap-add (cl_conv_kernels.h:87-100, 120-124) -------------
  const cpx *xa = xacc + (long)ch * sparts * N;
  auto xsum = [&](int i) {   // the segments' partial sums in ascending order
    cpx sum = ld_agent(xa + i);
    for (int k = 1; k < sparts; k++) sum = cadd(sum, ld_agent(xa + (long)k * N + i));
    return sum;
  };
#pragma unroll
  for (int q = 0; q < NI; q++) {
    const int i = tid + q * WG;
    if (i >= N / 2) break;
    if (i == 0) {
      const cpx c0 = xsum(0);
      s_x[0] = mk(c0.x + c0.y, c0.x - c0.y);
      s_x[lds_pad(N / 2)] = xsum(N / 2);
    } else {
      cpx oi, oj;
      c2r_pair(xsum(i), xsum(N - i), w2i_r[q], oi, oj);
      s_x[lds_pad(i)] = oi;
      s_x[lds_pad(N - i)] = oj;
    }
  }
  __syncthreads();
  {
    cpx v[E];
    if (tid < T) {
      pass_gather<LOGB, G::LOGE>(v, tid, [&](int p) { return s_x[lds_pad(p)]; });
      pass_compute<LOGB, G::LOGE, 0, false>(v, tid, s_tab);
    }
    __syncthreads();
    if (tid < T) pass_scatter<LOGB, G::LOGE, 0>(v, tid, [&](int p, cpx val) { s_x[lds_pad(p)] = val; });
    __syncthreads();
    if (tid < T) {
      pass_gather<LOGB, G::LOGE>(v, tid, [&](int p) { return s_x[lds_pad(p)]; });
      pass_compute<LOGB, G::LOGE, 4, false>(v, tid, s_tab);
    }
    if constexpr (LOGB > 8) {
      __syncthreads();
      if (tid < T) pass_scatter<LOGB, G::LOGE, 4>(v, tid, [&](int p, cpx val) { s_x[lds_pad(p)] = val; });
      __syncthreads();
      if (tid < T) {
        pass_gather<LOGB, G::LOGE>(v, tid, [&](int p) { return s_x[lds_pad(p)]; });
        pass_compute<LOGB, G::LOGE, 8, false>(v, tid, s_tab);
      }
    }
    if constexpr (LOGB > 12) {
      __syncthreads();
      if (tid < T) pass_scatter<LOGB, G::LOGE, 8>(v, tid, [&](int p, cpx val) { s_x[lds_pad(p)] = val; });
      __syncthreads();
      if (tid < T) {
        pass_gather<LOGB, G::LOGE>(v, tid, [&](int p) { return s_x[lds_pad(p)]; });
        pass_compute<LOGB, G::LOGE, 12, false>(v, tid, s_tab);
      }
    }
    if (tid < T) {
      constexpr float inv = 1.0f / (float)N;
      cpx *o = reinterpret_cast<cpx *>(out + (long)ch * N);
      cpx *tl = reinterpret_cast<cpx *>(tail + (long)ch * N);
#pragma unroll
      for (int e = 0; e < E / 2; e++) {
        const int p = tid + T * e;
        const cpx old = tail_r[e];
        o[p] = mk((v[e].x + old.x) * inv, (v[e].y + old.y) * inv);
        tl[p] = v[e + E / 2];
      }
    }
  }
}

// Shape of the cooperative block, or logs = -1 when it does not apply: bins 32..4096; slices of 32 bins (256-byte
// segments of a frame) unless the channels alone would overfill the chip; the partition axis cut into segments until
// a workgroup's share of the two rings is at most CLFA_PCONV_COOP_MAX_KB (tuning switch, read at plan creation; 0 switches the
// kernel off); filters that would need more than half the CUs that way stay with the launch chain above (the split
// MAC + tree sum), which puts the whole chip on the partition axis.
PconvCoop pconv_coop_plan(const PconvGeom &g, const DeviceInfo &di) {
  const char *cap_env = getenv("CLFA_PCONV_COOP_MAX_KB");   // read per plan, like every other tuning switch
  const long cap_kb = cap_env ? atol(cap_env) : 128L;
  PconvCoop c{-1, 1};
  // (partitions of 8192 samples were measured on this kernel with 512-lane workgroups: 27 us static, 36 us time-varying
  // against 29 us on the chain — a workgroup's own 8192-point transforms take 10 us each — so they stay on the chain)
  if (g.logb < 5 || g.logb > 12 || cap_kb <= 0) return c;
  int logs = g.logb - 5;                                  // 32 bins per workgroup
  const int logs_min = g.logb > 9 ? g.logb - 9 : 0;       // at most 256 16-byte items per workgroup (one per lane)
  while (logs > logs_min && ((long)g.channels << logs) > di.num_cus) logs--;
  if (((long)g.channels << logs) > 2L * di.num_cus) return c;
  const long share = 2L * g.nparts * (g.bins >> logs) * 8;   // bytes of the rings one bin slice streams
  long sparts = (share + cap_kb * 1024 - 1) / (cap_kb * 1024);
  // ... as far as HALF the CUs go: a block that needs the whole chip to stream its rings is faster on the chain
  // (measured, real-time ratio of one time-varying channel: M = 512, L = 2^21: 448 here against 482 on the chain,
  // L = 2^22: 384 / 401; M = 2048, L = 2^21: 1761 / 1804 — every workgroup repeats the forward transform, and
  // the last one adds up all the segments)
  const long wgs = (long)g.channels << logs;
  long room = (di.num_cus / 2) / wgs;
  if (room < 1) room = 1;
  if (sparts > g.nparts / 4) sparts = g.nparts / 4;          // segments of at least 4 partitions
  if (sparts < 1) sparts = 1;
  if (sparts > room) {
    // Many channels: their bin slices alone occupy half the chip or more, every workgroup streams its whole share
    // (up to 1 MiB) and nothing is cut or added up.  Measured at pts 1024 x 94 partitions, per block: 24 channels
    // 24.0 (chain) -> 15.5 us, 32: 24.1 -> 16.4, 64: 33.0 -> 24.0, 100: 41.4 -> 32.6, 128: 45.4 -> 40.4, 136: 55.9
    // -> 50.1; from 137 channels on k_pconv_fused takes over (144: 49.7 against 49.7 here, 256: 68.7 against 73.3).
    if (2 * wgs < di.num_cus || share > 1024L * 1024) return c;
    sparts = 1;
  }
  c.logs = logs;
  c.sparts = (int)sparts;
  return c;
}

template <int LOGB>
static hipError_t launch_coop_one(const PconvGeom &g, PconvCoop c, const float *in1, const float *in2, cpx *ringA, cpx *ringB,
                                  float *tail, float *out, int frame1, int frame2, int wp, const cpx *half, const cpx *w2f,
                                  const cpx *w2i, cpx *xacc, unsigned *counters, int num_cus, hipStream_t s) {
  const dim3 grid(c.sparts << c.logs, g.channels);
  const int acquire = (long)grid.x * grid.y > num_cus;   // more workgroups than CUs: see handover_acquire()
  if (in2)
    hipLaunchKernelGGL((k_pconv_coop<LOGB, true>), grid, dim3(LdsGeom<LOGB>::WG), 0, s, in1, in2, ringA, ringB, tail, out, frame1, frame2,
                       wp, g.nparts, half, w2f, w2i, xacc, counters, c.logs, c.sparts, acquire);
  else
    hipLaunchKernelGGL((k_pconv_coop<LOGB, false>), grid, dim3(LdsGeom<LOGB>::WG), 0, s, in1, in2, ringA, ringB, tail, out, frame1, frame2,
                       wp, g.nparts, half, w2f, w2i, xacc, counters, c.logs, c.sparts, acquire);
  return hipGetLastError();
}

hipError_t launch_pconv_coop(const PconvGeom &g, PconvCoop c, const float *in1, const float *in2, cpx *ringA, cpx *ringB,
                             float *tail, float *out, int frame1, int frame2, int wp, const cpx *half, const cpx *w2f,
                             const cpx *w2i, cpx *xacc, unsigned *counters, int num_cus, hipStream_t s) {
  switch (g.logb) {
#define CLFA_B(L) \
  case L:         \
    return launch_coop_one<L>(g, c, in1, in2, ringA, ringB, tail, out, frame1, frame2, wp, half, w2f, w2i, xacc, counters, num_cus, s);
    CLFA_B(5) CLFA_B(6) CLFA_B(7) CLFA_B(8) CLFA_B(9) CLFA_B(10) CLFA_B(11) CLFA_B(12)
#undef CLFA_B
    default:
      return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------
// partitions above the LDS sizes (pts = 16384, 32768): the same chain composed from the
// large-N FFT kernel; these two kernels are its zero-padding and overlap-add ends
// ---------------------------------------------------------------------------------
// work[ch][p] = p < bins/2 ? (in[ch][2p], in[ch][2p+1]) : 0   (cl_conv.cpp:399: half of in1 is written)
__global__ __launch_bounds__(256) void k_pconv_pad(const float *__restrict__ in, long in_stride,
                                                   cpx *__restrict__ work, int bins, long total) {
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const long ch = g / bins;
    const int p = (int)(g % bins);
    work[g] = p < bins / 2 ? reinterpret_cast<const cpx *>(in + ch * in_stride)[p] : mk(0.f, 0.f);
  }
}
// reference olap (cl_conv_kernels.h:120-124) on work viewed as 2*bins floats per channel
__global__ __launch_bounds__(256) void k_pconv_olap(const float *__restrict__ work, float *__restrict__ tail,
                                                    float *__restrict__ out, int bins, long total) {
  const float inv = 1.0f / (float)bins;
  for (long g = blockIdx.x * 256L + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const long ch = g / bins;
    const int n = (int)(g % bins);
    const float *t = work + ch * 2L * bins;
    out[g] = (t[n] + tail[g]) * inv;
    tail[g] = t[bins + n];
  }
}
hipError_t launch_pconv_pad(const float *in, long in_stride, cpx *work, int bins, int channels, hipStream_t s) {
  long total = (long)channels * bins, grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_pconv_pad, dim3((int)grid), dim3(256), 0, s, in, in_stride, work, bins, total);
  return hipGetLastError();
}
hipError_t launch_pconv_olap(const float *work, float *tail, float *out, int bins, int channels, hipStream_t s) {
  long total = (long)channels * bins, grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_pconv_olap, dim3((int)grid), dim3(256), 0, s, work, tail, out, bins, total);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// direct convolution (reference convol, cl_dconv.cpp:32-43; host side cl_dconv.cpp:109-148): ONE launch per block.
// The reference runs irsize x vsize work items that each add one product to out[n] with CAS atomics.  Here the work is
// a grid of (tap chunks) x (output blocks): workgroup (g, y) forms the partial sums of its tiles of 64 outputs over
// its chunk of C taps from LDS copies of the chunk's coefficients and of the delay-ring window they meet (every ring
// sample is read from memory once per workgroup and tile, not once per output), four lane rows walking a quarter of
// the chunk each.  The block's new samples are taken straight from the input (workgroup (0, 0) also files them in the
// ring — and, for the two-input form, in the coefficient ring — for the blocks to come), so no launch has to precede
// this one.  With more than one chunk the partial sums of an output block go to whichever of its workgroups arrives
// LAST at the block's counter (same hand-over as k_pconv_coop: agent-scope stores, s_waitcnt vmcnt(0), barrier, one
// atomic add; nothing spins), which adds them in a fixed order: the result does not depend on the arrival order.
// ---------------------------------------------------------------------------------
constexpr int kDconvNT = 64;        // outputs per tile
constexpr int kDconvMaxC = 4096;    // taps per workgroup at most
constexpr int kDconvMaxVB = 512;    // output blocks (= counters) at most

DconvPlan dconv_plan(int irsize, int vsize) {
  // about 64 K products per workgroup (a microsecond), at most ~512 workgroups; outputs first, then taps
  const long work = (long)irsize * vsize;
  long want = work >> 16;
  want = want < 1 ? 1 : want > 512 ? 512 : want;
  const int tiles = (vsize + kDconvNT - 1) / kDconvNT;
  DconvPlan pl;
  pl.VB = (int)(tiles < want ? tiles : want);
  if (pl.VB > kDconvMaxVB) pl.VB = kDconvMaxVB;
  const long gwant = (want + pl.VB - 1) / pl.VB;
  int c = (int)((irsize + gwant - 1) / gwant);
  c = (c + 63) / 64 * 64;
  pl.C = c < 64 ? 64 : c > kDconvMaxC ? kDconvMaxC : c;
  pl.G = (irsize + pl.C - 1) / pl.C;
  return pl;
}

__device__ __forceinline__ void st_agent_f(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent_f(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_dconv_block(float *__restrict__ out, const float *__restrict__ in1,
                                                     const float *__restrict__ in2, float *__restrict__ del,
                                                     float *__restrict__ coefs, float *__restrict__ part,
                                                     unsigned *__restrict__ counters, int irsize, int vsize, int wp, int C,
                                                     int acquire) {
  __shared__ float s_k[kDconvMaxC];
  __shared__ float s_d[kDconvMaxC + kDconvNT];
  __shared__ float s_red[256];
  __shared__ bool s_last;
  const int tid = threadIdx.x, g = blockIdx.x, G = gridDim.x, y = blockIdx.y, VB = gridDim.y;
  const int end = irsize + vsize;
  const int rp = (wp + vsize) % end;   // the read point is the write point AFTER this block (cl_dconv.cpp:124)
  const int h0 = g * C;
  const int taps = irsize - h0 < C ? irsize - h0 : C;
  // the rings as they stand once this block's samples are in (cl_dconv.cpp:112-122, 134-147)
  auto fresh = [&](int r) {   // position of ring index r inside the block being written, or >= vsize
    const int off = r - wp;
    return off < 0 ? off + end : off;
  };
  // (staging loops: eight independent loads per lane in flight, indices clamped instead of branched around — one at a
  // time they cost a memory round trip each)
  for (int jb = tid; jb < taps; jb += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int j = jb + 256 * u < taps ? jb + 256 * u : taps - 1;
      const int q = irsize - 1 - (h0 + j);
      const int off = fresh(q);
      const float *src = (in2 != nullptr && off < vsize) ? in2 + off : coefs + q;
      v[u] = *src;
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (jb + 256 * u < taps) s_k[jb + 256 * u] = v[u];
  }
  const int n = tid & (kDconvNT - 1), row = tid / kDconvNT;   // (row is wave-uniform: its coefficient reads broadcast)
  const int per = (taps + 3) / 4;
  const int j0 = row * per, j1 = j0 + per < taps ? j0 + per : taps;
  for (int n0 = y * kDconvNT; n0 < vsize; n0 += VB * kDconvNT) {
    // (no barrier here: the previous tile's readers of s_d passed the barrier behind their sums, and the first tile's
    // window loads fly together with the coefficient loads above — the barrier below covers both)
    const int span = taps + kDconvNT - 1;
    const int base = (int)(((long)rp + n0 + h0) % end);
    for (int jb = tid; jb < span; jb += 8 * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = jb + 256 * u < span ? jb + 256 * u : span - 1;
        int r = base + j;
        if (r >= end) {
          r -= end;
          if (r >= end) r %= end;   // (a ring shorter than the tile's window)
        }
        const int off = fresh(r);
        const float *src = off < vsize ? in1 + off : del + r;
        v[u] = *src;
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (jb + 256 * u < span) s_d[jb + 256 * u] = v[u];
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll 16   // (sixteen LDS reads in flight; the sum stays in ascending tap order)
    for (int j = j0; j < j1; j++) acc += s_d[n + j] * s_k[j];
    s_red[tid] = acc;
    __syncthreads();
    if (row == 0 && n0 + n < vsize) {
      const float sum = (s_red[n] + s_red[kDconvNT + n]) + (s_red[2 * kDconvNT + n] + s_red[3 * kDconvNT + n]);
      if (G == 1) out[n0 + n] = sum;
      else st_agent_f(part + (long)g * vsize + n0 + n, sum);
    }
  }
  if (g == 0 && y == 0) {   // file the block: nobody reads these ring positions in this launch (everybody takes them from the input)
    for (int i = tid; i < vsize; i += 256) {
      const int r = (wp + i) % end;
      del[r] = in1[i];
      if (in2 != nullptr) coefs[r] = in2[i];
    }
  }
  if (G == 1) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned old = handover_arrive(counters + y);
    s_last = old == (unsigned)(G - 1);
    if (s_last) {
      __hip_atomic_store(counters + y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next block's launch
      handover_acquire(acquire);
    }
  }
  __syncthreads();
  if (!s_last) return;
  // the output block's partial sums: lane row r adds the chunks of its quarter in ascending order (eight loads in
  // flight at a time: one after the other they cost a cache round trip each), the quarters meet in LDS in fixed order
  const int gq = (G + 3) / 4;
  const int k0 = row * gq, k1 = k0 + gq < G ? k0 + gq : G;
  for (int n0 = y * kDconvNT; n0 < vsize; n0 += VB * kDconvNT) {
    float sum = 0.f;
    if (n0 + n < vsize) {
      const float *pp = part + n0 + n;
      int k = k0;
      for (; k + 8 <= k1; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = ld_agent_f(pp + (long)(k + u) * vsize);
#pragma unroll
        for (int u = 0; u < 8; u++) sum += v[u];
      }
      for (; k < k1; k++) sum += ld_agent_f(pp + (long)k * vsize);
    }
    s_red[tid] = sum;
    __syncthreads();
    if (row == 0 && n0 + n < vsize)
      out[n0 + n] = (s_red[n] + s_red[kDconvNT + n]) + (s_red[2 * kDconvNT + n] + s_red[3 * kDconvNT + n]);
    __syncthreads();
  }
}

hipError_t launch_dconv_block(const DconvPlan &pl, float *out, const float *in1, const float *in2, float *del, float *coefs,
                              float *part, unsigned *counters, int irsize, int vsize, int wp, int num_cus, hipStream_t s) {
  if (pl.C < 1 || pl.C > kDconvMaxC || pl.G < 1 || (long)pl.C * pl.G < irsize || pl.VB < 1 || pl.VB > kDconvMaxVB)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_dconv_block, dim3(pl.G, pl.VB), dim3(256), 0, s, out, in1, in2, del, coefs, part, counters, irsize,
                     vsize, wp, pl.C, (long)pl.G * pl.VB > num_cus ? 1 : 0);
  return hipGetLastError();
}

}  // namespace clfa
