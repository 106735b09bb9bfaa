// cl_conv.h — drop-in for the reference's cl_conv.h (cl_conv.h:23-189).
#ifndef __CL_CONV_H__
#define __CL_CONV_H__
#include <complex>
#include <iostream>
#include <string>

#include "clfft_amd/cl_compat.h"

namespace cl_conv {

/** error string for an OpenCL status code (cl_conv.h:25-122) */
inline const char *cl_string(int err) { return clfa_error_string(err); }

class Clpconv {
  int N, bins;
  int nparts;
  clfa_pconv *pc;
  void (*err)(std::string s, void *uData);
  void *userData;
  int cl_err;

  static void msg(std::string str, void *userData) {   // cl_conv.h:142-145
    if (userData == NULL) std::cout << str << std::endl;
  }
  Clpconv(const Clpconv &);
  Clpconv &operator=(const Clpconv &);

 public:
  /** device_id - device handle; cvs - impulse response size; pts - partition size;
      errs - error message callback; uData - callback user data;
      in1, in2, out - accepted and ignored (host-memory mode is dead code in the
      reference, cl_conv.cpp:151,232-237) */
  Clpconv(cl_device_id device_id, int cvs, int pts, void (*errs)(std::string s, void *d) = NULL,
          void *uData = NULL, void *in1 = NULL, void *in2 = NULL, void *out = 0);
  /** extension: `channels` independent instances in one object (arrays become channel-major) */
  Clpconv(cl_device_id device_id, int cvs, int pts, int channels, void (*errs)(std::string s, void *d),
          void *uData);
  ~Clpconv();

  const char *cl_error_string(int err) { return cl_string(err); }
  /** ir - impulse response of size cvs (nparts*pts samples are read) */
  int push_ir(float *ir);
  /** output, input - partition-size samples */
  int convolution(float *output, float *input);
  /** time-varying convolution */
  int convolution(float *output, float *input1, float *input2);
  /** device-resident extension (in2 may be NULL) */
  int convolution_device(void *out, const void *in1, const void *in2, void *stream = 0);
  int get_cl_err() { return cl_err; }
};
}  // namespace cl_conv
#endif
