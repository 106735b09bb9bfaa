// cl_classes.cpp -> libcl_fft.so: the reference's C++ classes (cl_fft.h, cl_conv.h,
// cl_dconv.h) as thin wrappers over the C ABI of libclfft_amd.so.  Plain host C++, no HIP.
#include "../../include/cl_dconv.h"
#include "../../include/cl_fft.h"

namespace cl_fft {

const char *cl_error_string(int err) { return clfa_error_string(err); }   // cl_fft.cpp:298-395

// Clcfft::Clcfft, cl_fft.cpp:44-125
Clcfft::Clcfft(cl_device_id device_id, int size, bool fwd)
    : N(size), forward(fwd), plan(NULL), cl_err(0), w(NULL), b(NULL), data1(NULL), data2(NULL), commands(NULL) {
  cl_err = clfa_cfft_create(&plan, clfa_device_ordinal(device_id), size, fwd ? 1 : 0);
  if (!cl_err && size <= 65536 && !(size & (size - 1))) protected_members();   // the reference's range
}
// the reference's protected members (cl_fft.h:31-44), for subclasses; a failure here is a setup error like any other
// (cl_fft.cpp:79-84: the reference's clCreateBuffer calls feed the same cl_err)
void Clcfft::protected_members() {
  cl_err = clfa_fft_device_buffers(plan, &data1, &data2, &commands);
  if (!cl_err) cl_err = clfa_fft_device_tables(plan, &w, &b);
  if (cl_err) w = b = data1 = data2 = NULL;
}
// Clrfft constructs its base with size/2 (cl_fft.cpp:210): the member N is M = size/2
Clcfft::Clcfft(cl_device_id device_id, int size, bool fwd, bool)
    : N(size / 2), forward(fwd), plan(NULL), cl_err(0), w(NULL), b(NULL), data1(NULL), data2(NULL), commands(NULL) {
  cl_err = clfa_rfft_create(&plan, clfa_device_ordinal(device_id), size, fwd ? 1 : 0);
  if (!cl_err && size <= 131072 && !(size & (size - 1))) protected_members();
}
int Clcfft::fft() { return clfa_fft_run_buffers(plan); }                   // cl_fft.cpp:138-151
Clcfft::~Clcfft() { clfa_fft_destroy(plan); }                              // cl_fft.cpp:127-136
int Clcfft::transform(std::complex<float> *c) {                             // cl_fft.cpp:153-161
  return clfa_cfft_transform(plan, reinterpret_cast<float *>(c), 1);
}
int Clcfft::transform(std::complex<float> *c, long batch) {
  return clfa_cfft_transform(plan, reinterpret_cast<float *>(c), batch);
}
int Clcfft::transform_device(void *data, long batch, void *stream) { return clfa_fft_exec_dev(plan, data, batch, stream); }
int Clcfft::transform_device(const void *src, void *dst, long batch, void *stream) {
  return clfa_fft_exec_dev_oop(plan, src, dst, batch, stream);
}
void *Clcfft::alloc_host(size_t bytes) {
  void *ptr = NULL;
  return clfa_fft_host_alloc(plan, bytes, &ptr) == CLFA_SUCCESS ? ptr : NULL;
}
int Clcfft::free_host(void *ptr) { return clfa_fft_host_free(plan, ptr); }
const char *Clcfft::get_log() { return clfa_fft_get_log(plan); }

Clrfft::Clrfft(cl_device_id device_id, int size, bool fwd) : Clcfft(device_id, size, fwd, true) {}   // cl_fft.cpp:208-259
Clrfft::~Clrfft() {}
int Clrfft::transform(std::complex<float> *c, float *r) {                   // cl_fft.cpp:267-296
  return clfa_rfft_transform(plan, reinterpret_cast<float *>(c), r, 1);
}
int Clrfft::transform(std::complex<float> *c, float *r, long batch) {
  return clfa_rfft_transform(plan, reinterpret_cast<float *>(c), r, batch);
}
}  // namespace cl_fft

namespace cl_conv {

// Clpconv::Clpconv, cl_conv.cpp:140-320.  Setup errors go to the callback as a string by value.
Clpconv::Clpconv(cl_device_id device_id, int cvs, int pts, void (*errs)(std::string s, void *d), void *uData,
                 void *, void *, void *)
    : N(pts << 1), bins(pts), nparts(pts > 0 ? cvs / pts : 0), pc(NULL), err(errs == NULL ? this->msg : errs),
      userData(uData), cl_err(CL_SUCCESS) {
  cl_err = clfa_pconv_create(&pc, clfa_device_ordinal(device_id), cvs, pts, 1);
  if (cl_err != CL_SUCCESS) err(cl_error_string(cl_err), userData);
}
Clpconv::Clpconv(cl_device_id device_id, int cvs, int pts, int channels, void (*errs)(std::string s, void *d),
                 void *uData)
    : N(pts << 1), bins(pts), nparts(pts > 0 ? cvs / pts : 0), pc(NULL), err(errs == NULL ? this->msg : errs),
      userData(uData), cl_err(CL_SUCCESS) {
  cl_err = clfa_pconv_create(&pc, clfa_device_ordinal(device_id), cvs, pts, channels);
  if (cl_err != CL_SUCCESS) err(cl_error_string(cl_err), userData);
}
Clpconv::~Clpconv() { clfa_pconv_destroy(pc); }                             // cl_conv.cpp:322-347
int Clpconv::push_ir(float *ir) { return cl_err = clfa_pconv_push_ir(pc, ir); }                      // :353-388
int Clpconv::convolution(float *output, float *input) {                                              // :393-458
  return cl_err = clfa_pconv_convolution(pc, output, input);
}
int Clpconv::convolution(float *output, float *input1, float *input2) {                              // :460-548
  return cl_err = clfa_pconv_convolution_tv(pc, output, input1, input2);
}
int Clpconv::convolution_device(void *out, const void *in1, const void *in2, void *stream) {
  return cl_err = clfa_pconv_process_dev(pc, out, in1, in2, stream);
}

// Cldconv, cl_dconv.cpp:46-153
Cldconv::Cldconv(cl_device_id device_id, int cvs, int vsiz, void (*errs)(std::string s, void *d), void *uData)
    : irsize(cvs), vsize(vsiz), dc(NULL), err(errs == NULL ? this->msg : errs), userData(uData), cl_err(CL_SUCCESS) {
  cl_err = clfa_dconv_create(&dc, clfa_device_ordinal(device_id), cvs, vsiz);
  if (cl_err != CL_SUCCESS) err(cl_error_string(cl_err), userData);
}
Cldconv::~Cldconv() { clfa_dconv_destroy(dc); }
int Cldconv::push_ir(float *ir) { return clfa_dconv_push_ir(dc, ir); }
int Cldconv::convolution(float *out, float *in) {
  cl_err = clfa_dconv_convolution(dc, out, in);
  if (cl_err) err(cl_error_string(cl_err), userData);                       // cl_dconv.cpp:128-129
  return cl_err;
}
int Cldconv::convolution(float *out, float *in1, float *in2) {
  cl_err = clfa_dconv_convolution_tv(dc, out, in1, in2);
  if (cl_err) err(cl_error_string(cl_err), userData);
  return cl_err;
}
int Cldconv::convolution_device(void *out, const void *in1, const void *in2, void *stream) {
  return cl_err = clfa_dconv_process_dev(dc, out, in1, in2, stream);
}
}  // namespace cl_conv
