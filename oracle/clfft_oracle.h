/* oracle/clfft_oracle.h — CPU restatement of the reference algorithm.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use this.  The product (opencl_fft_amd/) never links,
 * imports or calls anything in oracle/.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * it restates.  Parity pin: see the header of clfft_oracle.c.
 */
#ifndef CLFFT_ORACLE_H
#define CLFFT_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* tables */
void orc_bitrev_table(int n, int *bp);                       /* cl_fft.cpp:96-101  */
void orc_twiddle_table(int n, int forward, float *w);        /* cl_fft.cpp:86-91   */
void orc_r2c_twiddle_table(int m, int forward, float *w2);   /* cl_fft.cpp:233-238 */

/* kernels, one call == one NDRange launch of the reference */
void orc_reorder(float *out, const float *in, const int *b, int n);            /* cl_fft.cpp:24-27 */
void orc_fft_stage(float *s, const float *w, int n, int n2, int scale_fwd);    /* cl_fft.cpp:29-41 */
void orc_r2c_conv(float *c, const float *w2, int m);                           /* cl_fft.cpp:178-191 */
void orc_c2r_iconv(float *c, const float *w2, int m);                          /* cl_fft.cpp:192-205 */

/* class-level operations (in place on interleaved complex64) */
int orc_cfft(float *data, int n, int forward);               /* Clcfft::transform cl_fft.cpp:138-161 */
int orc_rfft(float *data, int size, int forward);            /* Clrfft::transform cl_fft.cpp:267-296 */
/* batched helpers (batch-major contiguous); nthreads<=0 -> all cores (OpenMP) */
int orc_cfft_batched(float *data, int n, long batch, int forward, int nthreads);
int orc_rfft_batched(float *data, int size, long batch, int forward, int nthreads);
int orc_num_threads(void);

/* partitioned convolution: Clpconv, cl_conv.cpp:140-548 + cl_conv_kernels.h:46-124 */
typedef struct orc_pconv orc_pconv;
orc_pconv *orc_pconv_create(int cvs, int pts);
void orc_pconv_destroy(orc_pconv *p);
int orc_pconv_push_ir(orc_pconv *p, const float *ir);
int orc_pconv_convolution(orc_pconv *p, float *out, const float *in);
int orc_pconv_convolution_tv(orc_pconv *p, float *out, const float *in1, const float *in2);
int orc_pconv_wp(const orc_pconv *p);
int orc_pconv_wp2(const orc_pconv *p);
int orc_pconv_nparts(const orc_pconv *p);

/* direct convolution: Cldconv, cl_dconv.cpp:32-153 */
typedef struct orc_dconv orc_dconv;
orc_dconv *orc_dconv_create(int irsize, int vsize);
void orc_dconv_destroy(orc_dconv *d);
int orc_dconv_push_ir(orc_dconv *d, const float *ir);
int orc_dconv_convolution(orc_dconv *d, float *out, const float *in);
int orc_dconv_convolution_tv(orc_dconv *d, float *out, const float *in1, const float *in2);

#ifdef __cplusplus
}
#endif
#endif
