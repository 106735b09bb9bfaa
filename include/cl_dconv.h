// cl_dconv.h — drop-in for the reference's cl_dconv.h (cl_dconv.h:15-67).
#ifndef __CL_DCONV_H__
#define __CL_DCONV_H__
#include "cl_conv.h"

namespace cl_conv {

class Cldconv {
  int irsize, vsize;
  clfa_dconv *dc;
  void (*err)(std::string s, void *uData);
  void *userData;
  int cl_err;

  static void msg(std::string str, void *userData) {
    if (userData == NULL) std::cout << str << std::endl;
  }
  Cldconv(const Cldconv &);
  Cldconv &operator=(const Cldconv &);

 public:
  /** cvs - impulse response size; vsize - processing vector size */
  Cldconv(cl_device_id device_id, int cvs, int vsize, void (*errs)(std::string s, void *d) = NULL,
          void *uData = NULL);
  ~Cldconv();
  const char *cl_error_string(int err) { return cl_string(err); }
  int push_ir(float *ir);
  int convolution(float *output, float *input);
  int convolution(float *out, float *in1, float *in2);
  /** device-resident extension (in2 may be NULL; out must not be an input) */
  int convolution_device(void *out, const void *in1, const void *in2, void *stream = 0);
  int get_cl_err() { return cl_err; }
};
}  // namespace cl_conv
#endif
