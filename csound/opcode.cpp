// csound/opcode.cpp — Csound 7 opcode library over the MI355X backend.
//
// Same four opcodes, names and type strings as the reference plugin (csound/opcode.cpp:347-352,
// csound/README.md:7-10):
//     out:a   clconv   in:a, table:i, partsize:i, dev:i [, skip:i, size:i]        "a"   "aiiioo"
//     out:a   cltvconv in1:a, in2:a, freeze1:k, freeze2:k, partsize:i, size:i, dev:i  "a" "aakkiii"
//     out:k[] clfft    in:k[], fwd:i, dev:i                                        "k[]" "k[]ii"
//     out:k[] clrfft   in:k[], fwd:i, dev:i                                        "k[]" "k[]ii"
// Written against the documented behaviour, not against the reference's defects (SURVEY.md §8b:
// `buf[i]` indexing with an enum constant, k-rate method named perf(), freeze2 reading the wrong
// argument, undersized FFT buffer).  Needs Csound 7's <plugin.h>/<modload.h>, which are not in this
// repository: csound/CMakeLists.txt builds it only when find_package(CSOUND) succeeds.
#include <cl_conv.h>
#include <cl_dconv.h>
#include <cl_fft.h>
#include <modload.h>

#include <algorithm>
#include <complex>
#include <memory>
#include <vector>

namespace {

// next power of two >= n, at least 2 (the transforms are radix-2)
inline uint32_t pow2_ceil(uint32_t n) {
  uint32_t v = 2;
  while (v < n) v <<= 1;
  return v;
}

// error callback handed to the convolution classes: route to the Csound message stream
void to_csound(std::string s, void *user) { static_cast<csnd::Csound *>(user)->message(s); }

// device index -> handle, announcing the device like the reference does
bool pick_device(csnd::Csound *cs, int index, cl_device_id &id) {
  cl_device_id ids[32];
  cl_uint num = 0;
  if (clGetDeviceIDs(NULL, CL_DEVICE_TYPE_ALL, 32, ids, &num) != CL_SUCCESS || index < 0 || (cl_uint)index >= num)
    return false;
  char name[128] = {0};
  clGetDeviceInfo(ids[index], CL_DEVICE_NAME, sizeof(name), name, NULL);
  cs->message(std::string("using device: ") + name);
  id = ids[index];
  return true;
}

// ---- clfft / clrfft: k-rate array in, k-rate array out -------------------------------------
template <bool REAL> struct FftOp : csnd::Plugin<1, 3> {
  cl_fft::Clcfft *plan;
  csnd::AuxMem<float> aux;    // np2 complex (clfft) or np2 reals (clrfft), zero padded — used if the plan has no pinned memory
  float *work;                // the array every k-cycle's transform() runs on: page-locked memory of the plan, else `aux`
  uint32_t np2;

  int init() {
    delete plan;   // reinit without deinit (the struct starts zeroed: Csound runs no constructor)
    plan = nullptr;
    csnd::Vector<MYFLT> &in = inargs.vector_data<MYFLT>(0);
    csnd::Vector<MYFLT> &out = outargs.vector_data<MYFLT>(0);
    out.init(csound, in.len(), this);
    cl_device_id id;
    if (!pick_device(csound, (int)inargs[2], id)) return csound->init_error("failed to find a device!\n");
    const bool fwd = inargs[1] != 0;
    // clfft: the array is interleaved (re, im) -> len/2 complex points; clrfft: len real points
    np2 = pow2_ceil(REAL ? in.len() : std::max<uint32_t>(in.len() / 2, 2));
    plan = REAL ? static_cast<cl_fft::Clcfft *>(new cl_fft::Clrfft(id, (int)np2, fwd)) : new cl_fft::Clcfft(id, (int)np2, fwd);
    if (plan->get_error() != CL_SUCCESS) {
      const char *msg = cl_fft::cl_error_string(plan->get_error());
      delete plan;
      plan = nullptr;
      return csound->init_error(msg);
    }
    // the one array every k-cycle's transform() runs on, for the life of the instance: taken from the plan (page-locked,
    // seen by the device), so the transform needs no staging copies; it dies with the plan.  A refusal is not an error: the
    // calls then run on Csound's own memory and copy, as the reference's do.
    const uint32_t cap = REAL ? np2 : 2 * np2;
    work = static_cast<float *>(plan->alloc_host(sizeof(float) * cap));
    if (!work) {
      aux.allocate(csound, cap);
      work = aux.data();
    }
    return OK;
  }

  int kperf() {
    csnd::Vector<MYFLT> &in = inargs.vector_data<MYFLT>(0);
    csnd::Vector<MYFLT> &out = outargs.vector_data<MYFLT>(0);
    const uint32_t cap = REAL ? np2 : 2 * np2, n = std::min<uint32_t>(in.len(), cap);
    std::fill(work, work + cap, 0.f);
    for (uint32_t k = 0; k < n; k++) work[k] = (float)in[k];
    int err = plan->transform(reinterpret_cast<std::complex<float> *>(work));
    if (err != CL_SUCCESS) return csound->perf_error(cl_fft::cl_error_string(err), this);
    for (uint32_t k = 0; k < std::min<uint32_t>(out.len(), cap); k++) out[k] = (MYFLT)work[k];
    return OK;
  }

  int deinit() {
    delete plan;   // (releases `work` with it)
    plan = nullptr;
    work = nullptr;
    return OK;
  }
};

// ---- clconv / cltvconv ---------------------------------------------------------------------
// Both buffer `parts` samples, run one block through the device and emit it one block later
// (latency = one partition, as the reference); parts == 1 selects direct convolution with
// vsize = ksmps.
// No in-class initialisers: Csound zero-allocates opcode structs and runs no constructor, so a null
// pointer here always comes from that zeroed memory.  init() may run again on the same struct (reinit)
// without an intervening deinit(): it releases whatever a previous init() left.
struct ConvBase {
  cl_conv::Clpconv *pconv;
  cl_conv::Cldconv *dconv;
  int parts, cnt;
  bool direct;
  void release() {
    delete pconv;
    delete dconv;
    pconv = nullptr;
    dconv = nullptr;
  }
};

struct Conv : csnd::Plugin<1, 6>, ConvBase {
  csnd::AuxMem<float> bufin, bufout;

  int init() {
    release();   // reinit without deinit must not leak the previous device objects
    cl_device_id id;
    if (!pick_device(csound, (int)inargs[3], id)) return csound->init_error("failed to find a device!\n");
    csnd::Table ir;
    ir.init(csound, inargs(1));
    parts = (int)inargs[2];
    const int skip = (int)inargs[4];
    int size = (inargs[5] == 0 ? (int)ir.len() : (int)inargs[5]) - skip;
    if (parts < 1 || size < parts) return csound->init_error("bad partition / impulse response size\n");
    const MYFLT scale = csound->_0dbfs();
    std::vector<float> coefs(size);
    for (int k = 0; k < size; k++) coefs[k] = (float)(ir[skip + k] * scale);
    direct = parts == 1;
    int err;
    if (direct) {
      const int ksmps = insdshead->ksmps;
      dconv = new cl_conv::Cldconv(id, size, ksmps, to_csound, (void *)csound);
      err = dconv->get_cl_err() != CL_SUCCESS ? dconv->get_cl_err() : dconv->push_ir(coefs.data());
      bufin.allocate(csound, ksmps);
      bufout.allocate(csound, ksmps);
    } else {
      pconv = new cl_conv::Clpconv(id, size, parts, to_csound, (void *)csound);
      err = pconv->get_cl_err() != CL_SUCCESS ? pconv->get_cl_err() : pconv->push_ir(coefs.data());
      bufin.allocate(csound, parts);
      bufout.allocate(csound, parts);
    }
    cnt = 0;
    if (err != CL_SUCCESS) {
      release();
      return csound->init_error("error initialising the convolution object");
    }
    return OK;
  }

  int deinit() {
    release();
    return OK;
  }

  int aperf() {
    csnd::AudioSig in(this, inargs(0));
    csnd::AudioSig out(this, outargs(0));
    if (direct) {
      for (uint32_t n = offset; n < nsmps; n++) bufin[n] = (float)in[n];
      if (dconv->convolution(bufout.data(), bufin.data()) != CL_SUCCESS)
        return csound->perf_error("error computing convolution\n", this);
      for (uint32_t n = offset; n < nsmps; n++) out[n] = (MYFLT)bufout[n];
      return OK;
    }
    for (uint32_t n = offset; n < nsmps; n++) {
      bufin[cnt] = (float)in[n];
      out[n] = (MYFLT)bufout[cnt];
      if (++cnt == parts) {
        if (pconv->convolution(bufout.data(), bufin.data()) != CL_SUCCESS)
          return csound->perf_error("error computing convolution\n", this);
        cnt = 0;
      }
    }
    return OK;
  }
};

struct TVConv : csnd::Plugin<1, 7>, ConvBase {
  csnd::AuxMem<float> bufin1, bufin2, bufout;

  int init() {
    release();   // see Conv::init
    cl_device_id id;
    if (!pick_device(csound, (int)inargs[6], id)) return csound->init_error("failed to find a device!\n");
    parts = (int)inargs[4];
    const int size = (int)inargs[5];
    if (parts < 1 || size < parts) return csound->init_error("bad partition / filter size\n");
    direct = parts == 1;
    const int block = direct ? (int)insdshead->ksmps : parts;
    int err;
    if (direct) {
      dconv = new cl_conv::Cldconv(id, size, block, to_csound, (void *)csound);
      err = dconv->get_cl_err();
    } else {
      pconv = new cl_conv::Clpconv(id, size, parts, to_csound, (void *)csound);
      err = pconv->get_cl_err();
    }
    if (err != CL_SUCCESS) {
      release();
      return csound->init_error("error initialising the convolution object");
    }
    bufin1.allocate(csound, block);
    bufin2.allocate(csound, block);
    bufout.allocate(csound, block);
    cnt = 0;
    return OK;
  }

  int deinit() {
    release();
    return OK;
  }

  int aperf() {
    csnd::AudioSig in1(this, inargs(0));
    csnd::AudioSig in2(this, inargs(1));
    csnd::AudioSig out(this, outargs(0));
    // freeze flag = 0 holds the previous content of that input's buffer (each input has its own flag)
    const bool run1 = inargs[2] != 0, run2 = inargs[3] != 0;
    const MYFLT scale = csound->_0dbfs();
    if (direct) {
      for (uint32_t n = offset; n < nsmps; n++) {
        if (run1) bufin1[n] = (float)(in1[n] / scale);
        if (run2) bufin2[n] = (float)(in2[n] / scale);
      }
      if (dconv->convolution(bufout.data(), bufin1.data(), bufin2.data()) != CL_SUCCESS)
        return csound->perf_error("error computing convolution\n", this);
      for (uint32_t n = offset; n < nsmps; n++) out[n] = (MYFLT)(bufout[n] * scale);
      return OK;
    }
    for (uint32_t n = offset; n < nsmps; n++) {
      if (run1) bufin1[cnt] = (float)(in1[n] / scale);
      if (run2) bufin2[cnt] = (float)(in2[n] / scale);
      out[n] = (MYFLT)(bufout[cnt] * scale);
      if (++cnt == parts) {
        if (pconv->convolution(bufout.data(), bufin1.data(), bufin2.data()) != CL_SUCCESS)
          return csound->perf_error("error computing convolution\n", this);
        cnt = 0;
      }
    }
    return OK;
  }
};

}  // namespace

namespace csnd {
void on_load(Csound *csound) {
  plugin<Conv>(csound, "clconv", "a", "aiiioo", csnd::thread::ia);
  plugin<TVConv>(csound, "cltvconv", "a", "aakkiii", csnd::thread::ia);
  plugin<FftOp<false>>(csound, "clfft", "k[]", "k[]ii", csnd::thread::ik);
  plugin<FftOp<true>>(csound, "clrfft", "k[]", "k[]ii", csnd::thread::ik);
}
}  // namespace csnd
