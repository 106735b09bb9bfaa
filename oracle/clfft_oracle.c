/* oracle/clfft_oracle.c — CPU restatement of the reference algorithm (plain C).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call into this file.
 *
 * What it is: the reference's device kernels and host dispatch loops restated
 * as serial C loops ("for gid in 0..G" per NDRange launch), float32 arithmetic
 * in the reference's operation order, tables computed in double and rounded to
 * float exactly as the reference's constructors do.  Compiled with
 * -ffp-contract=off so no multiply-add is fused.
 *
 * Parity pin (tests/test_oracle_golden.py):
 *   1. tests/golden/ref/ — outputs of the UNMODIFIED reference classes
 *      (oracle/_ref, built in place from /root/reference by oracle/Makefile)
 *      run on the MI355X through the AMD OpenCL runtime by oracle/ref_driver.cpp.
 *   2. the two known-answer programs of the reference (test_cfft.cpp:54-56,
 *      test_rfft.cpp:54-57), whose expected spectra are analytic.
 *   3. a float64 DFT (numpy) as ground truth.
 *
 * Citations are file:line relative to /root/reference.
 */
#include "clfft_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const double ORC_PI = 3.141592653589793; /* cl_fft.h:24 */

typedef struct {
  float x, y;
} cpx;

/* cl_fft.cpp:20-22 (also cl_conv_kernels.h:17-19): complex product */
static inline cpx c_prod(cpx a, cpx b) {
  cpx r;
  r.x = a.x * b.x - a.y * b.y;
  r.y = a.x * b.y + a.y * b.x;
  return r;
}
static inline cpx c_add(cpx a, cpx b) { cpx r = {a.x + b.x, a.y + b.y}; return r; }
static inline cpx c_sub(cpx a, cpx b) { cpx r = {a.x - b.x, a.y - b.y}; return r; }
static inline cpx c_scale(float s, cpx a) { cpx r = {s * a.x, s * a.y}; return r; }
/* cl_fft.cpp:170-172 */
static inline cpx c_conj(cpx a) { cpx r = {a.x, -a.y}; return r; }
/* cl_fft.cpp:174-176: multiplication by +i (the reference calls it "rotation by pi") */
static inline cpx c_rot(cpx a) { cpx r = {-a.y, a.x}; return r; }

static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

/* ---- tables ------------------------------------------------------------ */

/* cl_fft.cpp:96-101 (twin: cl_conv.cpp:290-295): doubling construction of
 * the bit-reversal permutation */
void orc_bitrev_table(int n, int *bp) {
  for (int i = 0; i < n; i++) bp[i] = i;
  for (int i = 1, h = n / 2; i < n; i <<= 1, h >>= 1)
    for (int j = 0; j < i; j++) bp[i + j] = bp[j] + h;
}

/* cl_fft.cpp:86-91: w[i] = (cos(2 pi i/N), -/+ sin(2 pi i/N)), double -> float.
 * The expression order `i * 2 * PI / N` is kept (it fixes the double rounding). */
void orc_twiddle_table(int n, int forward, float *w) {
  float sign = forward ? -1.f : 1.f;
  for (int i = 0; i < n; i++) {
    w[2 * i] = (float)cos(i * 2 * ORC_PI / n);
    w[2 * i + 1] = sign * (float)sin(i * 2 * ORC_PI / n);
  }
}

/* cl_fft.cpp:233-238: w2[i] = (cos(pi i/M), -/+ sin(pi i/M)) */
void orc_r2c_twiddle_table(int m, int forward, float *w2) {
  float sign = forward ? -1.f : 1.f;
  for (int i = 0; i < m; i++) {
    w2[2 * i] = (float)cos(i * ORC_PI / m);
    w2[2 * i + 1] = sign * (float)sin(i * ORC_PI / m);
  }
}

/* ---- kernels ------------------------------------------------------------ */

/* cl_fft.cpp:24-27, G = n */
void orc_reorder(float *out, const float *in, const int *b, int n) {
  cpx *o = (cpx *)out;
  const cpx *s = (const cpx *)in;
  for (int k = 0; k < n; k++) o[k] = s[b[k]];
}

/* cl_fft.cpp:29-41, G = n/2: one radix-2 DIT stage, butterflies of span n2/2.
 * scale_fwd: the reference divides by N in the last stage of a forward plan. */
void orc_fft_stage(float *sf, const float *wf, int n, int n2, int scale_fwd) {
  cpx *s = (cpx *)sf;
  const cpx *w = (const cpx *)wf;
  int half = n2 >> 1;
  int last = (n2 == n) && scale_fwd;
  for (int g = 0; g < n / 2; g++) {
    int k = g * n2;
    int m = k / n;
    k = k % n + m;
    int i = k + half;
    cpx e = s[k];
    cpx o = c_prod(s[i], w[m * n / n2]);
    cpx a = c_add(e, o), d = c_sub(e, o);
    if (last) {
      a.x = a.x / (float)n; a.y = a.y / (float)n;
      d.x = d.x / (float)n; d.y = d.y / (float)n;
    }
    s[k] = a;
    s[i] = d;
  }
}

/* cl_fft.cpp:178-191, G = m/2: forward real-FFT post-process.  Bin m/2 is not
 * visited (thread count m/2), reproducing the reference. */
void orc_r2c_conv(float *cf, const float *wf, int m) {
  cpx *c = (cpx *)cf;
  const cpx *w = (const cpx *)wf;
  for (int i = 0; i < m / 2; i++) {
    if (i == 0) {
      cpx z = c[0];
      c[0].x = (z.x + z.y) * .5f;
      c[0].y = (z.x - z.y) * .5f;
      continue;
    }
    int j = m - i;
    cpx cj = c_conj(c[j]);
    cpx e = c_scale(.5f, c_add(c[i], cj));
    cpx o = c_scale(.5f, c_rot(c_sub(cj, c[i])));
    cpx p = c_prod(w[i], o);
    c[i] = c_add(e, p);
    c[j] = c_conj(c_sub(e, p));
  }
}

/* cl_fft.cpp:192-205, G = m/2: inverse real-FFT pre-process */
void orc_c2r_iconv(float *cf, const float *wf, int m) {
  cpx *c = (cpx *)cf;
  const cpx *w = (const cpx *)wf;
  for (int i = 0; i < m / 2; i++) {
    if (i == 0) {
      cpx z = c[0];
      c[0].x = z.x + z.y;
      c[0].y = z.x - z.y;
      continue;
    }
    int j = m - i;
    cpx cj = c_conj(c[j]);
    cpx e = c_scale(.5f, c_add(c[i], cj));
    cpx o = c_scale(.5f, c_rot(c_sub(c[i], cj)));
    cpx p = c_prod(w[i], o);
    c[i] = c_add(e, p);
    c[j] = c_conj(c_sub(e, p));
  }
}

/* ---- plans (tables cached per call site) -------------------------------- */

typedef struct {
  int n;
  int *b;
  float *w;   /* n complex */
  float *tmp; /* n complex: the reference's data1 (input side of reorder) */
} cplan;

static int cplan_init(cplan *p, int n, int forward) {
  p->n = n;
  p->b = (int *)malloc(sizeof(int) * n);
  p->w = (float *)malloc(sizeof(float) * 2 * n);
  p->tmp = (float *)malloc(sizeof(float) * 2 * n);
  if (!p->b || !p->w || !p->tmp) return -6; /* CL_OUT_OF_HOST_MEMORY */
  orc_bitrev_table(n, p->b);
  orc_twiddle_table(n, forward, p->w);
  return 0;
}
static void cplan_free(cplan *p) {
  free(p->b);
  free(p->w);
  free(p->tmp);
}

/* Clcfft::fft(), cl_fft.cpp:138-151: reorder data1 -> data2, then log2(N)
 * stage launches on data2.  `data` plays data2 on return. */
static void cplan_exec(cplan *p, float *data, int forward) {
  int n = p->n;
  memcpy(p->tmp, data, sizeof(float) * 2 * n);   /* cl_fft.cpp:155 write data1 */
  orc_reorder(data, p->tmp, p->b, n);            /* cl_fft.cpp:141 */
  for (int h = 1; h < n; h *= 2)                 /* cl_fft.cpp:143-149 */
    orc_fft_stage(data, p->w, n, h << 1, forward);
}

/* Clcfft::transform, cl_fft.cpp:153-161 */
int orc_cfft(float *data, int n, int forward) {
  if (!is_pow2(n) || n < 2) return -30; /* CL_INVALID_VALUE */
  cplan p;
  int e = cplan_init(&p, n, forward);
  if (!e) cplan_exec(&p, data, forward);
  cplan_free(&p);
  return e;
}

/* Clrfft::transform (in place), cl_fft.cpp:267-296.  size real points,
 * M = size/2 complex (cl_fft.cpp:210). */
static void rplan_exec(cplan *p, const float *w2, float *data, int forward) {
  int m = p->n;
  if (forward) {
    cplan_exec(p, data, 1);                      /* cl_fft.cpp:275-277 */
    orc_r2c_conv(data, w2, m);                   /* cl_fft.cpp:278-280 */
  } else {
    orc_c2r_iconv(data, w2, m);                  /* cl_fft.cpp:286-288 (on data1) */
    cplan_exec(p, data, 0);                      /* cl_fft.cpp:289 */
  }
}

int orc_rfft(float *data, int size, int forward) {
  if (!is_pow2(size) || size < 4) return -30;
  int m = size / 2;
  cplan p;
  int e = cplan_init(&p, m, forward);
  float *w2 = (float *)malloc(sizeof(float) * 2 * m);
  if (!e && w2) {
    orc_r2c_twiddle_table(m, forward, w2);
    rplan_exec(&p, w2, data, forward);
  }
  free(w2);
  cplan_free(&p);
  return e;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* B sequential transform() calls of the reference == one batched call here
 * (batch-major contiguous); batches spread over host threads. */
int orc_cfft_batched(float *data, int n, long batch, int forward, int nthreads) {
  if (!is_pow2(n) || n < 2) return -30;
  int err = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    cplan p;
    int e = cplan_init(&p, n, forward);
    if (e) err = e;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long bi = 0; bi < batch; bi++)
      if (!e) cplan_exec(&p, data + 2 * (size_t)n * bi, forward);
    cplan_free(&p);
  }
  (void)nthreads;
  return err;
}

int orc_rfft_batched(float *data, int size, long batch, int forward, int nthreads) {
  if (!is_pow2(size) || size < 4) return -30;
  int m = size / 2, err = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    cplan p;
    int e = cplan_init(&p, m, forward);
    float *w2 = (float *)malloc(sizeof(float) * 2 * m);
    if (e || !w2) err = -6;
    else orc_r2c_twiddle_table(m, forward, w2);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long bi = 0; bi < batch; bi++)
      if (!e && w2) rplan_exec(&p, w2, data + (size_t)size * bi, forward);
    free(w2);
    cplan_free(&p);
  }
  (void)nthreads;
  return err;
}

/* ---- partitioned convolution (Clpconv) ---------------------------------- */

struct orc_pconv {
  int bins, nparts, bsize, wp, wp2;         /* cl_conv.cpp:143-144 */
  int *b;
  float *w[2], *w2[2];                       /* cl_conv.cpp:263-287 */
  float *in1, *in2, *out, *olap;             /* bins complex each, cl_conv.cpp:232-240 */
  float *spec1, *spec2;                      /* bsize complex each, cl_conv.cpp:243-246 */
};

/* cl_conv_kernels.h:46-52: gather AND zero the source */
static void pc_reorder(float *outf, float *inf, const int *b, int offs, int bins) {
  cpx *out = (cpx *)outf + offs;
  cpx *in = (cpx *)inf;
  for (int k = 0; k < bins; k++) {
    out[k] = in[b[k]];
    in[b[k]].x = 0.f;
    in[b[k]].y = 0.f;
  }
}
/* cl_conv.cpp:53-67 + cl_conv_kernels.h:54-68: log2(bins) unscaled stages on a frame */
static void pc_fft(float *data, const float *w, int bins, int offs) {
  for (int h = 1; h < bins; h *= 2)
    orc_fft_stage(data + 2 * (size_t)offs, w, bins, h << 1, 0);
}

orc_pconv *orc_pconv_create(int cvs, int pts) {
  if (!is_pow2(pts) || pts < 2 || cvs < pts) return NULL;
  orc_pconv *p = (orc_pconv *)calloc(1, sizeof(*p));
  p->bins = pts;
  p->nparts = cvs / pts;                     /* floor: remainder dropped, cl_conv.cpp:143 */
  p->bsize = p->nparts * p->bins;
  p->wp = 0;
  p->wp2 = p->nparts - 1;
  int bins = p->bins;
  p->b = (int *)malloc(sizeof(int) * bins);
  orc_bitrev_table(bins, p->b);              /* cl_conv.cpp:290-295 */
  for (int d = 0; d < 2; d++) {
    p->w[d] = (float *)malloc(sizeof(float) * 2 * bins);
    p->w2[d] = (float *)malloc(sizeof(float) * 2 * bins);
    orc_twiddle_table(bins, d == 0, p->w[d]);        /* cl_conv.cpp:264-275 */
    orc_r2c_twiddle_table(bins, d == 0, p->w2[d]);   /* cl_conv.cpp:276-287 */
  }
  p->in1 = (float *)calloc(2 * bins, sizeof(float)); /* zeroed: cl_conv.cpp:303-313 */
  p->in2 = (float *)calloc(2 * bins, sizeof(float));
  p->out = (float *)calloc(2 * bins, sizeof(float));
  p->olap = (float *)calloc(2 * bins, sizeof(float));
  p->spec1 = (float *)calloc(2 * (size_t)p->bsize, sizeof(float));
  p->spec2 = (float *)calloc(2 * (size_t)p->bsize, sizeof(float));
  return p;
}

void orc_pconv_destroy(orc_pconv *p) {
  if (!p) return;
  free(p->b);
  for (int d = 0; d < 2; d++) { free(p->w[d]); free(p->w2[d]); }
  free(p->in1); free(p->in2); free(p->out); free(p->olap);
  free(p->spec1); free(p->spec2);
  free(p);
}
int orc_pconv_wp(const orc_pconv *p) { return p->wp; }
int orc_pconv_wp2(const orc_pconv *p) { return p->wp2; }
int orc_pconv_nparts(const orc_pconv *p) { return p->nparts; }

/* forward chain shared by push_ir and both convolution(): write pts floats to
 * the first half of the staging buffer, reorder into ring frame `fr`, fft, r2c
 * (cl_conv.cpp:361-380 / 399-419) */
static void pc_forward(orc_pconv *p, float *stage, float *ring, int fr, const float *src) {
  int bins = p->bins;
  memcpy(stage, src, sizeof(float) * bins);          /* bytes>>1 = bins floats */
  pc_reorder(ring, stage, p->b, fr * bins, bins);
  pc_fft(ring, p->w[0], bins, fr * bins);
  orc_r2c_conv(ring + 2 * (size_t)fr * bins, p->w2[0], bins); /* cl_conv_kernels.h:70-85 */
}

/* Clpconv::push_ir, cl_conv.cpp:353-388: IR partition i -> spec2 frame wp2,
 * wp2 decrementing from nparts-1 */
int orc_pconv_push_ir(orc_pconv *p, const float *ir) {
  for (int i = 0; i < p->nparts; i++) {
    pc_forward(p, p->in2, p->spec2, p->wp2, ir + (size_t)i * p->bins);
    p->wp2 = p->wp2 == 0 ? p->nparts - 1 : p->wp2 - 1;
  }
  return 0;
}

/* cl_conv_kernels.h:102-118, G = bsize.  The reference accumulates with
 * float CAS atomics (order unspecified); here partitions are summed in
 * ascending thread order. */
static void pc_convol(orc_pconv *p) {
  int bins = p->bins, nparts = p->nparts;
  const cpx *in = (const cpx *)p->spec1;
  const cpx *coef = (const cpx *)p->spec2;
  float *out = p->in1;
  for (int k = 0; k < p->bsize; k++) {
    int n = k % bins;
    int rp = p->wp + k / bins;
    const cpx *fr = in + (size_t)(rp < nparts ? rp : rp - nparts) * bins;
    cpx s;
    if (n) s = c_prod(fr[n], coef[k]);
    else { s.x = fr[0].x * coef[k].x; s.y = fr[0].y * coef[k].y; }
    out[2 * n] += s.x;
    out[2 * n + 1] += s.y;
  }
}

/* inverse chain, cl_conv.cpp:428-455 */
static void pc_inverse(orc_pconv *p, float *output) {
  int bins = p->bins;
  pc_convol(p);                                       /* :428 */
  orc_c2r_iconv(p->in1, p->w2[1], bins);              /* :434, cl_conv_kernels.h:87-100 */
  pc_reorder(p->out, p->in1, p->b, 0, bins);          /* :439 (zeroes in1 again) */
  pc_fft(p->out, p->w[1], bins, 0);                   /* :444 */
  /* olap kernel, cl_conv_kernels.h:120-124, G = bins */
  for (int n = 0; n < bins; n++) {
    p->olap[n] = (p->out[n] + p->olap[bins + n]) / (float)bins;
    p->olap[bins + n] = p->out[bins + n];
  }
  memcpy(output, p->olap, sizeof(float) * bins);      /* :455 */
}

/* Clpconv::convolution(out,in), cl_conv.cpp:393-458 */
int orc_pconv_convolution(orc_pconv *p, float *output, const float *input) {
  pc_forward(p, p->in1, p->spec1, p->wp, input);
  p->wp = p->wp != p->nparts - 1 ? p->wp + 1 : 0;     /* :424 */
  pc_inverse(p, output);
  return 0;
}

/* Clpconv::convolution(out,in1,in2), cl_conv.cpp:460-548 */
int orc_pconv_convolution_tv(orc_pconv *p, float *output, const float *in1, const float *in2) {
  pc_forward(p, p->in1, p->spec1, p->wp, in1);
  pc_forward(p, p->in2, p->spec2, p->wp2, in2);
  p->wp = p->wp != p->nparts - 1 ? p->wp + 1 : 0;     /* :516 */
  p->wp2 = p->wp2 == 0 ? p->nparts - 1 : p->wp2 - 1;  /* :519 */
  pc_inverse(p, output);
  return 0;
}

/* ---- direct convolution (Cldconv) ---------------------------------------- */

struct orc_dconv {
  int irsize, vsize, wp;
  float *del, *coefs;  /* irsize+vsize floats each, cl_dconv.cpp:88-91 */
};

/* Buffers are zero-initialised here; the reference leaves them uninitialised
 * (cl_dconv.cpp:87-91), a defect SURVEY.md §8a says not to reproduce. */
orc_dconv *orc_dconv_create(int irsize, int vsize) {
  if (irsize < 1 || vsize < 1) return NULL;
  orc_dconv *d = (orc_dconv *)calloc(1, sizeof(*d));
  d->irsize = irsize;
  d->vsize = vsize;
  d->wp = 0;
  d->del = (float *)calloc((size_t)irsize + vsize, sizeof(float));
  d->coefs = (float *)calloc((size_t)irsize + vsize, sizeof(float));
  return d;
}
void orc_dconv_destroy(orc_dconv *d) {
  if (!d) return;
  free(d->del);
  free(d->coefs);
  free(d);
}
/* cl_dconv.cpp:150-153 */
int orc_dconv_push_ir(orc_dconv *d, const float *ir) {
  memcpy(d->coefs, ir, sizeof(float) * d->irsize);
  return 0;
}
/* ring write with wrap, the intent of cl_dconv.cpp:112-122 */
static void dc_ring_write(float *ring, int end, int wp, const float *src, int n) {
  for (int i = 0; i < n; i++) ring[(wp + i) % end] = src[i];
}
/* Cldconv::convolution(out,in), cl_dconv.cpp:109-132 + kernel :32-43.
 * Full vsize outputs are produced every call (the reference's wrap branch
 * clobbers `bytes` and returns only `front` samples — a defect, not copied). */
int orc_dconv_convolution(orc_dconv *d, float *out, const float *in) {
  int irsize = d->irsize, vsize = d->vsize, end = irsize + vsize;
  dc_ring_write(d->del, end, d->wp, in, vsize);
  d->wp = (d->wp + vsize) % end;                      /* :124 */
  for (int n = 0; n < vsize; n++) out[n] = 0.f;       /* :123 */
  for (int t = 0; t < irsize * vsize; t++) {          /* kernel, G = irsize*vsize */
    int n = t % vsize, h = t / vsize;
    int rp = d->wp + n + h;
    float tap = d->del[rp < end ? rp : rp % end] * d->coefs[irsize - 1 - h];
    out[n] += tap;
  }
  return 0;
}
/* Cldconv::convolution(out,in1,in2), cl_dconv.cpp:134-148: in2 is written
 * into the coefficient ring at the same write point */
int orc_dconv_convolution_tv(orc_dconv *d, float *out, const float *in1, const float *in2) {
  dc_ring_write(d->coefs, d->irsize + d->vsize, d->wp, in2, d->vsize);
  return orc_dconv_convolution(d, out, in1);
}
