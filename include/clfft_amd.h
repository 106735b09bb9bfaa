/* clfft_amd.h — C ABI of libclfft_amd.so (MI355X / gfx950 native FFT and
 * partitioned-convolution engine).
 *
 * This is the drop-in boundary for the hot path of vlazzarini/opencl_fft:
 * the reference has no FFI of its own — its public surface is the C++ classes
 * cl_fft::Clcfft / Clrfft (cl_fft.h:29-111), cl_conv::Clpconv (cl_conv.h:124-188)
 * and cl_conv::Cldconv (cl_dconv.h:17-66).  Every entry point below names the
 * reference member it replaces (file:line relative to the reference tree).
 * include/cl_fft.h, cl_conv.h and cl_dconv.h rebuild those classes as
 * header-only wrappers over this ABI; opencl_fft_amd/ binds it with ctypes.
 *
 * Conventions kept from the reference:
 *   - every call returns an OpenCL-numbered status: 0 = CL_SUCCESS, negative =
 *     error (cl_fft.cpp:298-395); nothing throws across this boundary;
 *   - complex data are interleaved float32 (re, im), batch-major contiguous;
 *   - "host" entry points copy in/out and block, like the reference's
 *     blocking clEnqueueWrite/ReadBuffer (cl_fft.cpp:155-159);
 *   - forward c2c is scaled by 1/N, inverse is unscaled (cl_fft.cpp:39-40);
 *     r2c uses the reference's packed amplitude layout incl. the untouched
 *     self-paired bin M/2 (cl_fft.cpp:178-205).
 * Extensions (the reference does one transform per call): a batch count, and
 * "_dev" entry points that work in place on device-resident buffers on a
 * caller-supplied hipStream_t (passed as void*; NULL is the HIP default stream, as
 * everywhere in HIP).  Work is ordered by that stream only; nothing blocks.
 * A plan / convolution object owns ONE device workspace, so it may have work in flight on one
 * stream at a time: when a call names another stream than the object's previous call, the library
 * first waits (on the host) for that previous stream (not while the new stream is being captured into
 * a hipGraph: a captured launch is ordered by its graph, and the object's earlier work must be complete
 * when the graph is replayed).  Every entry point leaves the calling
 * thread's current HIP device as it found it.
 */
#ifndef CLFFT_AMD_H
#define CLFFT_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLFA_API __attribute__((visibility("default")))

/* status codes: the OpenCL numbers the reference returns (CL/cl.h) */
#define CLFA_SUCCESS 0
#define CLFA_DEVICE_NOT_FOUND (-1)
#define CLFA_DEVICE_NOT_AVAILABLE (-2)
#define CLFA_MEM_OBJECT_ALLOCATION_FAILURE (-4)
#define CLFA_OUT_OF_RESOURCES (-5)
#define CLFA_OUT_OF_HOST_MEMORY (-6)
#define CLFA_INVALID_VALUE (-30)
#define CLFA_INVALID_DEVICE (-33)
#define CLFA_INVALID_COMMAND_QUEUE (-36)
#define CLFA_INVALID_MEM_OBJECT (-38)
#define CLFA_INVALID_KERNEL_ARGS (-52)
#define CLFA_INVALID_OPERATION (-59)
#define CLFA_INVALID_BUFFER_SIZE (-61)

typedef struct clfa_fft clfa_fft;     /* c2c or r2c/c2r plan: Clcfft / Clrfft object */
typedef struct clfa_pconv clfa_pconv; /* Clpconv object, `channels` independent instances */
typedef struct clfa_dconv clfa_dconv; /* Cldconv object */

/* ---- library / devices ---------------------------------------------------- */
/* replaces clGetDeviceIDs(NULL, CL_DEVICE_TYPE_ALL, ...) at test_cfft.cpp:31,
 * opencl.cpp call sites opcode.cpp:57,113,172,269: HIP device ordinals */
CLFA_API int clfa_device_count(int *count);
/* replaces clGetDeviceInfo(id, CL_DEVICE_NAME, ...) at test_cfft.cpp:37 */
CLFA_API int clfa_device_name(int device, char *buf, size_t len);
/* cl_fft::cl_error_string (cl_fft.cpp:298-395) / cl_conv::cl_string (cl_conv.h:25-122) */
CLFA_API const char *clfa_error_string(int err);
CLFA_API const char *clfa_version(void);

/* ---- tables (host, exact reference formulas) ------------------------------- */
/* bit-reversal index table, cl_fft.cpp:96-101 (twin cl_conv.cpp:290-295) */
CLFA_API int clfa_bitrev_table(int n, int *out);
/* w[i] = (cos(2 pi i/n), -/+ sin(2 pi i/n)), i in [0,n): cl_fft.cpp:86-91 */
CLFA_API int clfa_twiddle_table(int n, int forward, float *out);
/* w2[i] = (cos(pi i/m), -/+ sin(pi i/m)), i in [0,m): cl_fft.cpp:233-238 */
CLFA_API int clfa_r2c_twiddle_table(int m, int forward, float *out);

/* ---- complex FFT: cl_fft::Clcfft ------------------------------------------- */
/* Clcfft::Clcfft(device_id, size, fwd), cl_fft.cpp:44-125.  n = 2^k, 2..65536 is the reference's
 * range (its stage kernel overflows int32 above that, cl_fft.cpp:32); as an extension n up to 2^24
 * is accepted (two passes up to 2^22, three above; 256 MiB of workspace), and so is any length 2..2^22
 * that is not a power of two (the reference's callers pad those, opcode.cpp:30-35): exact DFT by Bluestein's
 * algorithm around two power-of-two plans, same scaling conventions.
 * On failure *plan is still a valid handle whose clfa_fft_get_error() reports
 * the setup error (the reference's constructors never throw, cl_fft.h:65). */
CLFA_API int clfa_cfft_create(clfa_fft **plan, int device, int n, int forward);
/* Clrfft::Clrfft(device_id, size, fwd), cl_fft.cpp:208-259.  size real points
 * = 2^k, 4..131072 (extension: up to 2^25, and multiples of 4 that are not powers of two up to 2^23);
 * the inner complex length is M = size/2 (cl_fft.cpp:210). */
CLFA_API int clfa_rfft_create(clfa_fft **plan, int device, int size, int forward);
/* Clcfft::~Clcfft / Clrfft::~Clrfft, cl_fft.cpp:127-136, 261-265 */
CLFA_API void clfa_fft_destroy(clfa_fft *plan);
/* Clcfft::get_error(), cl_fft.h:65; Clcfft::get_log(), cl_fft.h:69 */
CLFA_API int clfa_fft_get_error(const clfa_fft *plan);
CLFA_API const char *clfa_fft_get_log(const clfa_fft *plan);
/* Clcfft::transform(c), cl_fft.cpp:153-161: in place on `batch` host arrays of
 * n complex (batch = 1 is the reference call). */
CLFA_API int clfa_cfft_transform(clfa_fft *plan, float *c, long batch);
/* Clrfft::transform(c, r), cl_fft.cpp:267-296.  forward: r (size floats per
 * batch) -> c (M complex per batch); inverse: c -> r.  c and r may alias. */
CLFA_API int clfa_rfft_transform(clfa_fft *plan, float *c, float *r, long batch);
/* Extension to the two host calls above, which replace the reference's blocking clEnqueueWriteBuffer / ReadBuffer around
 * every transform (cl_fft.cpp:155-158, 275-291): a caller that keeps ONE array for the plan's life (the Csound opcodes keep
 * one buffer per instance, csound/opcode.cpp) takes it from the plan — page-locked host memory the device sees.  transform
 * calls on arrays INSIDE such a buffer run on that memory directly: the kernels read and write it over PCIe, one pass each
 * way, no staging copy (up to 8 MiB per call and on the routes that touch source and destination once; otherwise the usual
 * copies, by DMA).  Results are bit-identical to the copying call.  The memory lives until clfa_fft_host_free or the plan's
 * destruction.  Real plans, out of place: both arrays must come from the plan for the direct route.  (Pinning the caller's
 * own heap array — hipHostRegister — was measured and is not offered: clfft_amd.cpp, profiles/host_path_r05.txt.) */
CLFA_API int clfa_fft_host_alloc(clfa_fft *plan, size_t bytes, void **ptr);
CLFA_API int clfa_fft_host_free(clfa_fft *plan, void *ptr);
/* device-resident, in place, asynchronous on `stream`: the body of
 * Clcfft::fft() (cl_fft.cpp:138-151) / the kernel part of Clrfft::transform.
 * data: batch * n complex64 (c2c) or batch * size float32 (r2c, packed in place). */
CLFA_API int clfa_fft_exec_dev(clfa_fft *plan, void *data, long batch, void *stream);
/* the same from `src` to `dst` (extension).  The reference's device side is itself out of place — its `reorder` gathers
 * data1 -> data2 and the stages then run on data2, cl_fft.cpp:138-151.  Every plan runs
 * src -> dst natively, at the cost of the in-place call: the kernels read the source and write the destination, routes of
 * several passes put their first pass there.  src == dst is clfa_fft_exec_dev; partly overlapping buffers, and buffers
 * that are not a whole number of complex values apart, are CLFA_INVALID_VALUE.  `src` is left untouched. */
CLFA_API int clfa_fft_exec_dev_oop(clfa_fft *plan, const void *src, void *dst, long batch, void *stream);
/* The reference's protected members for subclasses (cl_fft.h:35-44): data1 / data2 = the object's own device buffers of one
 * transform each (n complex64; real plans: size floats), allocated on first request and released with the plan;
 * commands = the plan's hipStream_t (the reference's cl_command_queue). */
CLFA_API int clfa_fft_device_buffers(clfa_fft *plan, void **data1, void **data2, void **commands);
/* ... and its tables (cl_fft.h:35; cl_fft.cpp:86-104): w = n complex64 twiddles of the plan's direction, b = n int32
 * bit reversals (c2c plans of the reference's range only, n <= 65536), device copies made on first request */
CLFA_API int clfa_fft_device_tables(clfa_fft *plan, void **w, void **b);
/* clEnqueueWriteBuffer / clEnqueueReadBuffer / clFinish as the reference uses them on its queue (cl_fft.cpp:155-158):
 * host <-> device copies on a hipStream_t (NULL: the default stream), blocking or not */
CLFA_API int clfa_copy_to_device(void *stream, void *dst, const void *src, size_t bytes, int blocking);
CLFA_API int clfa_copy_from_device(void *stream, void *dst, const void *src, size_t bytes, int blocking);
CLFA_API int clfa_stream_synchronize(void *stream);
/* Clcfft::fft(), cl_fft.cpp:138-151: one transform data1 -> data2 on the plan's stream; like the reference it only
 * enqueues (synchronise the stream before reading data2).  Real plans run their whole r2c / c2r. */
CLFA_API int clfa_fft_run_buffers(clfa_fft *plan);
/* bytes of device workspace a plan holds (0 for single-pass sizes) */
CLFA_API size_t clfa_fft_workspace_bytes(const clfa_fft *plan);
/* name of the HIP kernel that does the work for this plan (for profiles) */
CLFA_API const char *clfa_fft_kernel_name(const clfa_fft *plan);

/* measurement aid (nothing of the reference's): sustained device-memory bandwidth in TB/s over
 * `launches` launches on two buffers of `bytes` each (a multiple of 512 KiB, far beyond the 256 MiB
 * Infinity Cache for a meaningful number).  what: 0 read, 1 write, 2 copy (all 16-byte non-temporal
 * accesses, contiguous), 3 copy in the four-step FFT's own shape (256 x 256 matrices of 8-byte elements
 * moved in blocks of 16 columns: 128-byte segments 2 KiB apart).  Copies count bytes read + written. */
CLFA_API int clfa_bandwidth_probe(int device, int what, size_t bytes, int launches, double *tb_per_s);

/* the reference's `reorder` kernel as a stand-alone op (cl_fft.cpp:24-27):
 * out[b*n + k] = in[b*n + bitrev(k)], exact gather of complex64, out != in */
CLFA_API int clfa_reorder_dev(int device, void *out, const void *in, int n, long batch, void *stream);

/* ---- partitioned convolution: cl_conv::Clpconv ------------------------------ */
/* Clpconv::Clpconv(device_id, cvs, pts, ...), cl_conv.cpp:140-320, for
 * `channels` independent instances (channels = 1 is the reference object).
 * bins = pts, nparts = cvs / pts (floor, cl_conv.cpp:143). */
CLFA_API int clfa_pconv_create(clfa_pconv **pc, int device, int cvs, int pts, int channels);
CLFA_API void clfa_pconv_destroy(clfa_pconv *pc);            /* cl_conv.cpp:322-347 */
CLFA_API int clfa_pconv_get_error(const clfa_pconv *pc);     /* Clpconv::get_cl_err, cl_conv.h:187 */
/* state readers (ring indices must match the reference bit for bit, cl_conv.cpp:144,424,385,519) */
CLFA_API int clfa_pconv_nparts(const clfa_pconv *pc);
CLFA_API int clfa_pconv_wp(const clfa_pconv *pc);
CLFA_API int clfa_pconv_wp2(const clfa_pconv *pc);
/* Clpconv::push_ir(ir), cl_conv.cpp:353-388: ir = channels x (nparts*pts) floats */
CLFA_API int clfa_pconv_push_ir(clfa_pconv *pc, const float *ir);
/* device-resident form: channel c's response starts at ir + c * channel_stride floats
 * (channel_stride >= nparts*pts; a (channels, cvs) tensor passes cvs), nparts*pts floats are read of each */
CLFA_API int clfa_pconv_push_ir_dev(clfa_pconv *pc, const void *ir, long channel_stride, void *stream);
/* Clpconv::convolution(out, in), cl_conv.cpp:393-458: channels x pts floats each */
CLFA_API int clfa_pconv_convolution(clfa_pconv *pc, float *out, const float *in);
/* Clpconv::convolution(out, in1, in2), cl_conv.cpp:460-548 (time-varying) */
CLFA_API int clfa_pconv_convolution_tv(clfa_pconv *pc, float *out, const float *in1, const float *in2);
/* device-resident variants; in2 may be NULL (static IR).  `out` must not overlap an input, not even partly, on the one-launch
 * routes (clfa_pconv_kernel_name() = k_pconv_fused or k_pconv_coop: partitions up to 4096 samples), where workgroups of
 * other channels may still be reading: CL_INVALID_VALUE.  The launch chain of larger partitions reads every input before it
 * writes `out`; in place is accepted there. */
CLFA_API int clfa_pconv_process_dev(clfa_pconv *pc, void *out, const void *in1, const void *in2, void *stream);
CLFA_API size_t clfa_pconv_state_bytes(const clfa_pconv *pc);
/* which launch structure a block of this object takes (diagnostics and tests): "k_pconv_fused" (one launch, one
 * workgroup per channel), "k_pconv_coop" (one launch, a few channels), "chain" (forward / MAC / inverse launches) */
CLFA_API const char *clfa_pconv_kernel_name(const clfa_pconv *pc);

/* ---- direct convolution: cl_conv::Cldconv ----------------------------------- */
/* Cldconv::Cldconv(device_id, cvs, vsize, ...), cl_dconv.cpp:46-98 */
CLFA_API int clfa_dconv_create(clfa_dconv **dc, int device, int irsize, int vsize);
CLFA_API void clfa_dconv_destroy(clfa_dconv *dc);            /* cl_dconv.cpp:100-107 */
CLFA_API int clfa_dconv_get_error(const clfa_dconv *dc);     /* cl_dconv.h:65 */
CLFA_API int clfa_dconv_push_ir(clfa_dconv *dc, const float *ir);                 /* cl_dconv.cpp:150-153 */
CLFA_API int clfa_dconv_convolution(clfa_dconv *dc, float *out, const float *in); /* cl_dconv.cpp:109-132 */
CLFA_API int clfa_dconv_convolution_tv(clfa_dconv *dc, float *out, const float *in1, const float *in2); /* :134-148 */
/* device-resident variant of both (in2 may be NULL): vsize floats each, asynchronous on `stream`, one launch per
 * block; out must not be in1 or in2 */
CLFA_API int clfa_dconv_process_dev(clfa_dconv *dc, void *out, const void *in1, const void *in2, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CLFFT_AMD_H */
