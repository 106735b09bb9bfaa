// cl_fft.h — drop-in for the reference's cl_fft.h (class surface cl_fft.h:22-112):
// same namespace, class names, constructor and method signatures, integer OpenCL
// status codes; computation by libclfft_amd.so (hand-written HIP for gfx950).
// Link with -lcl_fft -lclfft_amd.
#ifndef __CL_FFT_H__
#define __CL_FFT_H__

#include <complex>
#include <iostream>

#include "clfft_amd/cl_compat.h"

namespace cl_fft {

const double PI = 3.141592653589793;       // cl_fft.h:24
const char *cl_error_string(int err);      // cl_fft.h:25

/** Complex to Complex FFT class (cl_fft.h:29-70) */
class Clcfft {
 protected:
  int N;
  bool forward;
  clfa_fft *plan;   // replaces the reference's context / program / kernels
  int cl_err;
  // The reference's device members a subclass may touch (cl_fft.h:35-37), as HIP objects (cl_compat.h: cl_mem = device
  // pointer, cl_command_queue = hipStream_t; clEnqueueWriteBuffer / clEnqueueReadBuffer / clFinish work on them):
  // w = N twiddles, b = N bit reversals (cl_fft.cpp:86-104), data1 / data2 = N complex numbers each, commands = the
  // object's stream.  They exist for the reference's range of sizes (N <= 65536; NULL above; a failure to create them is
  // a setup error, get_error()).  fft() (cl_fft.h:44, cl_fft.cpp:138-151) transforms data1 -> data2 on that stream and,
  // like the reference, only enqueues; on a Clrfft it is the complex transform of the N = size / 2 points alone, as in the
  // reference (conv / iconv are Clrfft::transform's own kernels, cl_fft.cpp:267-296).
  cl_mem w, b, data1, data2;
  cl_command_queue commands;
  int fft();

 public:
  /** device_id - device handle; size - DFT size (N); fwd - direction */
  Clcfft(cl_device_id device_id, int size, bool fwd = true);
  virtual ~Clcfft();
  /** DFT operation (in-place), c - N complex numbers */
  virtual int transform(std::complex<float> *c);
  /** batched extension: `batch` contiguous arrays of N complex numbers */
  int transform(std::complex<float> *c, long batch);
  /** device-resident extension: in place on device memory, asynchronous on a hipStream_t */
  int transform_device(void *data, long batch, void *stream = 0);
  /** ... from src to dst (the reference's device side is out of place too: data1 -> data2) */
  int transform_device(const void *src, void *dst, long batch, void *stream);
  /** extension: page-locked host memory from the object, for a caller that keeps one array for the object's life;
      transform() calls on arrays inside it run on that memory directly (no staging copies).  NULL if the runtime refuses.
      Lives until free_host() or the destructor. */
  void *alloc_host(size_t bytes);
  int free_host(void *ptr);
  /** Get setup error code */
  int get_error() { return cl_err; }
  /** Get compilation log (setup diagnostics here; nothing is JIT-compiled) */
  const char *get_log();

 private:
  // not copyable (the reference's implicit copy would double-release its OpenCL handles; here it is ruled out)
  Clcfft(const Clcfft &);
  Clcfft &operator=(const Clcfft &);

  void protected_members();

 protected:
  Clcfft(cl_device_id device_id, int size, bool fwd, bool real);
};

/** Real to Complex FFT class (cl_fft.h:74-111) */
class Clrfft : public Clcfft {
 public:
  Clrfft(cl_device_id device_id, int size, bool fwd);
  virtual ~Clrfft();
  /** c - N/2 complex numbers, r - N real numbers; in place if both point to the same memory */
  int transform(std::complex<float> *c, float *r);
  /** in-place form */
  virtual int transform(std::complex<float> *c) {
    float *r = reinterpret_cast<float *>(c);
    return transform(c, r);
  }
  /** batched extension */
  int transform(std::complex<float> *c, float *r, long batch);
};
}  // namespace cl_fft

#endif
