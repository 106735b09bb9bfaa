// clfft_amd.cpp — C ABI of libclfft_amd.so (see include/clfft_amd.h).
//
// Host side of the hot path: plan objects (the reference's Clcfft / Clrfft /
// Clpconv / Cldconv instances), exact host tables, H2D/D2H staging for the
// blocking host-pointer entry points, and the mapping hipError_t -> OpenCL
// status numbers.  No CPU compute fallback exists: without a HIP device every
// constructor reports CL_DEVICE_NOT_FOUND and every exec call fails.
#include "../../include/clfft_amd.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "internal.hpp"

using namespace clfa;

namespace {

const double kPI = 3.141592653589793;  // cl_fft.h:24

int map_hip(hipError_t e) {
  switch (e) {
    case hipSuccess: return CLFA_SUCCESS;
    case hipErrorNoDevice: return CLFA_DEVICE_NOT_FOUND;
    case hipErrorInvalidDevice: return CLFA_INVALID_DEVICE;
    case hipErrorOutOfMemory: return CLFA_MEM_OBJECT_ALLOCATION_FAILURE;
    case hipErrorInvalidValue: return CLFA_INVALID_VALUE;
    case hipErrorInvalidDevicePointer: return CLFA_INVALID_MEM_OBJECT;
    case hipErrorInvalidResourceHandle: return CLFA_INVALID_COMMAND_QUEUE;
    case hipErrorNotInitialized:
    case hipErrorInsufficientDriver: return CLFA_DEVICE_NOT_AVAILABLE;
    default: return CLFA_OUT_OF_RESOURCES;
  }
}

#define HIP_TRY(expr)                   \
  do {                                  \
    hipError_t _e = (expr);             \
    if (_e != hipSuccess) {             \
      (void)hipGetLastError();          \
      return map_hip(_e);               \
    }                                   \
  } while (0)

// current-device guard: every entry point works on its object's device and leaves the caller's
// current device as it found it
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t enter(int device) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return e;
    if (prev == device) return hipSuccess;
    e = hipSetDevice(device);
    switched = e == hipSuccess;
    return e;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};
#define ENTER_DEVICE(dev) \
  DeviceGuard _guard;     \
  HIP_TRY(_guard.enter(dev))

// An object owns one device workspace: work on a second stream has to wait for the first.  Switching
// streams is rare (the reference has one queue per object), so the wait is a host-side synchronise
// at the switch instead of an event per launch.
struct StreamOrder {
  hipStream_t last = nullptr;
  bool any = false;
  static bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
      (void)hipGetLastError();   // a stale handle: not capturing
      return false;
    }
    return st != hipStreamCaptureStatusNone;
  }
  hipError_t use(hipStream_t s) {
    hipError_t e = hipSuccess;
    if (any && s != last) {
      // a stream under hipGraph capture must not be waited for (nor may anything else be while it
      // captures): captured launches are ordered by the graph, and whatever the object was doing
      // before the capture has to be complete when the graph is replayed — the caller's contract.
      // The same holds when the PREVIOUS stream is the one under capture.
      if (!capturing(s) && !capturing(last)) {
        e = hipStreamSynchronize(last);
        if (e == hipErrorInvalidHandle || e == hipErrorContextIsDestroyed || e == hipErrorInvalidResourceHandle) {
          // the caller has destroyed its previous stream (we do not own it and cannot keep it alive): its handle is
          // gone, its work may not be — wait for the device instead of the handle, and carry on
          (void)hipGetLastError();
          e = hipDeviceSynchronize();
        }
      }
    }
    last = s;
    any = true;
    return e;
  }
};

int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) l++;
  return l;
}
bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// W_n^k = (cos(2 pi k/n), -sin(2 pi k/n)) rounded from double, the expression of
// cl_fft.cpp:89-90 (`i * 2 * PI / N`) so the float values are bit-identical.
void fill_twiddle(std::vector<cpx> &v, int count, int n, int stride, float sign) {
  v.resize(count > 0 ? count : 1);
  for (int i = 0; i < count; i++) {
    int k = i * stride;
    v[i].x = (float)cos(k * 2 * kPI / n);
    v[i].y = sign * (float)sin(k * 2 * kPI / n);
  }
  if (count <= 0) v[0] = mk(1.f, 0.f);
}
// cl_fft.cpp:236-237 (`i * PI / N`)
void fill_w2(std::vector<cpx> &v, int m, float sign) {
  v.resize(m);
  for (int i = 0; i < m; i++) {
    v[i].x = (float)cos(i * kPI / m);
    v[i].y = sign * (float)sin(i * kPI / m);
  }
}

// host tables of the four-step kernel: [half N1 | half N2 | lo: W_n^k, k < 2^loglo | hi: W_n^(k 2^loglo)]
void fill_fourstep_tables(std::vector<cpx> &all, int logn) {
  int l1, l2, llo;
  fourstep_split(logn, &l1, &l2, &llo);
  const int n = 1 << logn, n1 = 1 << l1, n2 = 1 << l2, lo = 1 << llo, hi = n >> llo;
  std::vector<cpx> part;
  all.clear();
  fill_twiddle(part, n1 / 2, n1, 1, -1.f);
  all.insert(all.end(), part.begin(), part.begin() + n1 / 2);
  fill_twiddle(part, n2 / 2, n2, 1, -1.f);
  all.insert(all.end(), part.begin(), part.begin() + n2 / 2);
  fill_twiddle(part, lo, n, 1, -1.f);
  all.insert(all.end(), part.begin(), part.begin() + lo);
  fill_twiddle(part, hi, n, lo, -1.f);
  all.insert(all.end(), part.begin(), part.begin() + hi);
}

// host tables of the resident n = 65536 kernel (internal.hpp, kRes16TabSize), each value rounded from
// double like the reference's table (cl_fft.cpp:89-90)
void fill_res16_tables(std::vector<cpx> &all) {
  all.clear();
  std::vector<cpx> part;
  for (int t = 0; t < 16; t++)
    for (int j = 0; j < 16; j++) all.push_back(mk((float)cos((t * j) * 2 * kPI / 256), -(float)sin((t * j) * 2 * kPI / 256)));
  fill_twiddle(part, 256, 65536, 1, -1.f);
  all.insert(all.end(), part.begin(), part.begin() + 256);
  fill_twiddle(part, 256, 256, 1, -1.f);
  all.insert(all.end(), part.begin(), part.begin() + 256);
  for (int m = 1; m <= 8; m *= 2)
    for (int k = 0; k < 256; k++) {
      const int idx = (m * k) & 4095;
      all.push_back(mk((float)cos(idx * 2 * kPI / 4096), -(float)sin(idx * 2 * kPI / 4096)));
    }
}

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int ensure(size_t want) {
    if (want <= bytes) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      return map_hip(e);
    }
    bytes = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

// pinned host memory mapped into the device's address space: kernels read / write it directly over
// PCIe.  For the few KiB of one audio block that beats three hipMemcpyAsync calls (10-15 us each).
struct HostBuf {
  void *h = nullptr;   // host pointer
  void *d = nullptr;   // the same memory as the device sees it
  size_t bytes = 0;
  int ensure(size_t want) {
    if (want <= bytes) return 0;
    release();
    hipError_t e = hipHostMalloc(&h, want, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      release();
      return map_hip(e);
    }
    bytes = want;
    return 0;
  }
  void release() {
    if (h) (void)hipHostFree(h);
    h = d = nullptr;
    bytes = 0;
  }
};

int upload(DevBuf &b, const void *src, size_t bytes) {
  int e = b.ensure(bytes);
  if (e) return e;
  HIP_TRY(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
  return 0;
}

int device_info(int device, DeviceInfo &di) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return CLFA_DEVICE_NOT_FOUND;
  }
  if (device < 0 || device >= count) return CLFA_INVALID_DEVICE;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  di.device = device;
  di.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------
// plan objects
// ---------------------------------------------------------------------------------

struct clfa_fft {
  DeviceInfo di;
  bool real = false;     // Clrfft
  bool fwd = true;
  int n = 0;             // complex length (Clrfft: M = size/2, cl_fft.cpp:210)
  int logn = 0;
  int size = 0;          // user-visible size (n, or real points for Clrfft)
  int err = 0;           // Clcfft::cl_err
  char log[2048];
  hipStream_t stream = nullptr;
  DevBuf half, w2, four, scratch, stage, res16;
  DevBuf own1, own2;     // the reference's protected data1 / data2 (cl_fft.h:35), on request: clfa_fft_device_buffers
  DevBuf own_w, own_b;   // ... and w / b: clfa_fft_device_tables
  bool own_tables_ready = false;   // both tables allocated AND filled
  struct Pinned {          // a pinned array the caller got from the plan: clfa_fft_host_alloc
    char *h, *d;           // host address, and the same memory as the device sees it
    size_t bytes;
  };
  std::vector<Pinned> pinned;
  clfa_fft *own_cplx = nullptr;    // Clrfft::fft() = the complex n-point transform alone (cl_fft.cpp:138-151 on N = size / 2): a c2c plan, on demand
  StreamOrder order;
  HostBuf zstage;        // zero-copy staging of small host transforms
  FftTables tabs;
  // n > 65536 (extension): n = N1 x N2; `tabs` then belongs to the N2-point row transform
  BigGeom big{};
  DevBuf bigtabs, scratch2;
  bool rlds15 = false;   // packed real size 65536: k_rfft_2x<14> (two 16384-point runs per transform, one HBM pass)
  bool c2x13 = false;    // complex n = 16384: k_cfft_2x<13> (two 8192-point runs per transform, two workgroups per CU)
  bool r2x13 = false;    // packed real size 32768: k_rfft_2x<13> (the same, with the pair maps in registers)
  bool r16 = false;      // packed real size 131072: k_fft_res16 with the pair map inside (one HBM pass), either direction
  DevBuf half2;          // ... the tables of the last two: the n = 8192 lane tables + W_16384^t, t < 512
  long spread_below = 0; // real sizes 32768 / 65536: batches up to this run the four-step pair + pack kernel instead
  // any other length (extension): Bluestein around two power-of-two plans of length blue_m
  int blue_m = 0;
  clfa_fft *blue_f = nullptr, *blue_i = nullptr;
  DevBuf blue_w, blue_b, blue_work;
};

struct clfa_pconv {
  DeviceInfo di;
  PconvGeom g{};
  int cvs = 0, pts = 0;
  int wp = 0, wp2 = 0;   // cl_conv.cpp:144
  int err = 0;
  hipStream_t stream = nullptr;
  DevBuf half, w2f, w2i;             // tables (cl_conv.cpp:263-287)
  DevBuf ringA, ringB, acc, tail;    // spec1, spec2, in1-as-accumulator, olap tail
  DevBuf in1, in2, out, ir;          // staging for the host entry points
  HostBuf zin1, zin2, zout;          // ... zero-copy staging for small blocks
  DevBuf four, scratch, work;        // partitions above the LDS sizes: large-N tables, scratch, work frames
  StreamOrder order;
  bool fused = false;                // one launch per block (resolved at creation)
  PconvCoop coop{-1, 1};             // few channels: one cooperative launch per block (logs >= 0)
  DevBuf cnt;                        // ... its arrival counters (one per channel)
  FftTables big;
};

struct clfa_dconv {
  DeviceInfo di;
  int irsize = 0, vsize = 0, wp = 0;
  int err = 0;
  hipStream_t stream = nullptr;
  DevBuf del, coefs, out;
  DevBuf in1, in2;     // staging of the host entry points' input blocks
  HostBuf zin1, zin2, zout;   // ... zero-copy staging for small blocks (mapped pinned host memory)
  DevBuf part, cnt;    // partial sums per tap chunk and their arrival counter (plan.G > 1)
  DconvPlan plan{64, 1, 1};
  StreamOrder order;
};

extern "C" {

// ---------------------------------------------------------------------------------
// library / devices
// ---------------------------------------------------------------------------------

const char *clfa_version(void) { return "clfft_amd 0.1 (gfx950)"; }

int clfa_device_count(int *count) {
  if (!count) return CLFA_INVALID_VALUE;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess || c <= 0) {
    (void)hipGetLastError();
    *count = 0;
    return CLFA_DEVICE_NOT_FOUND;
  }
  *count = c;
  return CLFA_SUCCESS;
}

int clfa_device_name(int device, char *buf, size_t len) {
  if (!buf || len == 0) return CLFA_INVALID_VALUE;
  DeviceInfo di;
  int e = device_info(device, di);
  if (e) return e;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  const char *nm = prop.name[0] ? prop.name : prop.gcnArchName;
  snprintf(buf, len, "%s", nm);
  return CLFA_SUCCESS;
}

// message table of cl_fft::cl_error_string (cl_fft.cpp:298-395)
const char *clfa_error_string(int err) {
  static const struct {
    int code;
    const char *msg;
  } tab[] = {{0, "Success!"}, {-1, "Device not found."}, {-2, "Device not available"},
             {-3, "Compiler not available"}, {-4, "Memory object allocation failure"},
             {-5, "Out of resources"}, {-6, "Out of host memory"},
             {-7, "Profiling information not available"}, {-8, "Memory copy overlap"},
             {-9, "Image format mismatch"}, {-10, "Image format not supported"},
             {-11, "Program build failure"}, {-12, "Map failure"}, {-30, "Invalid value"},
             {-31, "Invalid device type"}, {-32, "Invalid platform"}, {-33, "Invalid device"},
             {-34, "Invalid context"}, {-35, "Invalid queue properties"}, {-36, "Invalid command queue"},
             {-37, "Invalid host pointer"}, {-38, "Invalid memory object"},
             {-39, "Invalid image format descriptor"}, {-40, "Invalid image size"},
             {-41, "Invalid sampler"}, {-42, "Invalid binary"}, {-43, "Invalid build options"},
             {-44, "Invalid program"}, {-45, "Invalid program executable"}, {-46, "Invalid kernel name"},
             {-47, "Invalid kernel definition"}, {-48, "Invalid kernel"}, {-49, "Invalid argument index"},
             {-50, "Invalid argument value"}, {-51, "Invalid argument size"},
             {-52, "Invalid kernel arguments"}, {-53, "Invalid work dimension"},
             {-54, "Invalid work group size"}, {-55, "Invalid work item size"},
             {-56, "Invalid global offset"}, {-57, "Invalid event wait list"}, {-58, "Invalid event"},
             {-59, "Invalid operation"}, {-60, "Invalid OpenGL object"}, {-61, "Invalid buffer size"},
             {-62, "Invalid mip-map level"}};
  for (const auto &t : tab)
    if (t.code == err) return t.msg;
  return "Unknown error";
}

// ---------------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------------

int clfa_bitrev_table(int n, int *out) {
  if (!out || !is_pow2(n)) return CLFA_INVALID_VALUE;
  // doubling construction of cl_fft.cpp:96-101
  out[0] = 0;
  for (int i = 1, h = n / 2; i < n; i <<= 1, h >>= 1)
    for (int j = 0; j < i; j++) out[i + j] = out[j] + h;
  return CLFA_SUCCESS;
}

int clfa_twiddle_table(int n, int forward, float *out) {
  if (!out || n < 1) return CLFA_INVALID_VALUE;
  std::vector<cpx> v;
  fill_twiddle(v, n, n, 1, forward ? -1.f : 1.f);
  memcpy(out, v.data(), sizeof(cpx) * n);
  return CLFA_SUCCESS;
}

int clfa_r2c_twiddle_table(int m, int forward, float *out) {
  if (!out || m < 1) return CLFA_INVALID_VALUE;
  std::vector<cpx> v;
  fill_w2(v, m, forward ? -1.f : 1.f);
  memcpy(out, v.data(), sizeof(cpx) * m);
  return CLFA_SUCCESS;
}

// ---------------------------------------------------------------------------------
// FFT plans
// ---------------------------------------------------------------------------------

// host double-precision radix-2 transform (unscaled, forward sign), for the Bluestein filter table
static void host_fft(std::vector<double> &re, std::vector<double> &im) {
  const size_t n = re.size();
  for (size_t i = 1, j = 0; i < n; i++) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      std::swap(re[i], re[j]);
      std::swap(im[i], im[j]);
    }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t h = len / 2;
    std::vector<double> c(h), sn(h);
    for (size_t k = 0; k < h; k++) {
      c[k] = cos(2 * kPI * (double)k / (double)len);
      sn[k] = -sin(2 * kPI * (double)k / (double)len);
    }
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < h; k++) {
        const double xr = re[i + k + h] * c[k] - im[i + k + h] * sn[k], xi = re[i + k + h] * sn[k] + im[i + k + h] * c[k];
        re[i + k + h] = re[i + k] - xr;
        im[i + k + h] = im[i + k] - xi;
        re[i + k] += xr;
        im[i + k] += xi;
      }
  }
}

// any length that is not a power of two (extension): chirp w[j] = exp(-+ i pi j^2 / n) (j^2 reduced mod 2 n
// in integers, then double), filter B = DFT_m(conj(w) wrapped round m) in double, two m-point sub-plans
static int blue_setup(clfa_fft *p, int device, int n, bool real, bool fwd) {
  if (real && (n & 1)) {
    snprintf(p->log, sizeof(p->log), "real sizes that are not powers of two must be multiples of 4 (got %d)", 2 * n);
    return CLFA_INVALID_VALUE;
  }
  int e = device_info(device, p->di);
  if (e) return e;
  ENTER_DEVICE(device);
  HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  p->logn = -1;
  int m = 1;
  while (m < 2 * n - 1) m <<= 1;
  p->blue_m = m;
  const double sgn = fwd ? -1.0 : 1.0;
  std::vector<cpx> w(n), bt(m);
  std::vector<double> br(m, 0.0), bi(m, 0.0);
  for (int j = 0; j < n; j++) {
    const long long q = ((long long)j * j) % (2LL * n);
    const double a = kPI * (double)q / (double)n;
    w[j] = mk((float)cos(a), (float)(sgn * sin(a)));
    br[j] = cos(a);
    bi[j] = -sgn * sin(a);              // conj(w)
    if (j) {
      br[m - j] = br[j];
      bi[m - j] = bi[j];
    }
  }
  host_fft(br, bi);
  for (int k = 0; k < m; k++) bt[k] = mk((float)br[k], (float)bi[k]);
  if ((e = upload(p->blue_w, w.data(), sizeof(cpx) * n))) return e;
  if ((e = upload(p->blue_b, bt.data(), sizeof(cpx) * m))) return e;
  if ((e = clfa_cfft_create(&p->blue_f, device, m, 1))) return e;
  if ((e = clfa_cfft_create(&p->blue_i, device, m, 0))) return e;
  // workspace: as many m-point rows as fit 256 MiB (at least one); exec walks the batch in such chunks
  const size_t per = sizeof(cpx) * (size_t)m, cap = (size_t)256 << 20;
  if (!blue_lds_ok(m) && (e = p->blue_work.ensure(per * (cap / per > 0 ? cap / per : 1)))) return e;   // (m <= 8192: one launch, no workspace)
  if (real) {
    std::vector<cpx> h;
    fill_w2(h, n, fwd ? -1.f : 1.f);
    if ((e = upload(p->w2, h.data(), sizeof(cpx) * n))) return e;
    p->tabs.w2 = (const cpx *)p->w2.p;
  }
  return CLFA_SUCCESS;
}

static int fft_setup(clfa_fft *p, int device, int n, bool real, int size, bool fwd) {
  p->real = real;
  p->fwd = fwd;
  p->n = n;
  p->size = size;
  p->log[0] = 0;
  if (n < 2 || (is_pow2(n) && n > (1 << kBigMaxLog)) || (!is_pow2(n) && n > kBlueMaxN)) {
    snprintf(p->log, sizeof(p->log), "complex length must be 2..%d (powers of two) or 2..%d (other lengths), got %d",
             1 << kBigMaxLog, kBlueMaxN, n);
    return CLFA_INVALID_VALUE;
  }
  if (!is_pow2(n)) return blue_setup(p, device, n, real, fwd);
  p->logn = ilog2(n);
  int e = device_info(device, p->di);
  if (e) return e;
  ENTER_DEVICE(device);
  HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  std::vector<cpx> h;
  int rown = n, rowlog = p->logn;   // the transform the LDS / four-step tables are for
  if (p->logn > kMaxLog) {
    big_split(p->logn, &p->big);
    rowlog = p->big.logn2;
    rown = 1 << rowlog;
    const int n1 = 1 << p->big.logn1;
    std::vector<cpx> all, part;
    fill_twiddle(part, n1 / 2, n1, 1, -1.f);
    all.insert(all.end(), part.begin(), part.begin() + n1 / 2);
    fill_twiddle(part, 128, n, 1, -1.f);              // W_n^e, e = e0 + 128 e1 + 16384 e2 (big_tw(), fft_kernels.hip)
    all.insert(all.end(), part.begin(), part.begin() + 128);
    fill_twiddle(part, 128, n, 128, -1.f);
    all.insert(all.end(), part.begin(), part.begin() + 128);
    fill_twiddle(part, n / 16384, n, 16384, -1.f);
    all.insert(all.end(), part.begin(), part.begin() + n / 16384);
    if ((e = upload(p->bigtabs, all.data(), sizeof(cpx) * all.size()))) return e;
    // workspace: as many whole transforms as fit 256 MiB (at least one); exec walks the batch in such chunks
    size_t per = sizeof(cpx) * (size_t)n, cap = (size_t)256 << 20;
    if (const char *mb = getenv("CLFA_BIG_CHUNK_MB")) cap = (size_t)(atoi(mb) > 0 ? atoi(mb) : 256) << 20;   // tuning switch, read once
    if ((e = p->scratch.ensure(per * (cap / per > 0 ? cap / per : 1)))) return e;
  }
  p->rlds15 = real && p->logn == 15;
  p->r2x13 = real && p->logn == 14;
  p->c2x13 = !real && p->logn == 14;
  p->r16 = real && p->logn == 16;
  // The fused real kernels put one workgroup on a transform (13-23 us for a single one); a few transforms are
  // faster spread over the column / row blocks of the four-step pair plus the pack kernel (11 us): real plans of
  // these two sizes carry both sets of tables and exec picks by batch (p->spread_below).
  const bool lane14 = p->rlds15;     // k_rfft_2x<14> runs on the 16384-point lane tables
  const bool both = p->rlds15 || p->r2x13;
  const int fourlog = rowlog;
  if (lane14) rowlog = kLds14Log;
  if (both) p->spread_below = p->di.num_cus / 8;   // measured crossover: between 32 and 64 transforms
  if (rowlog <= kLdsMaxLog || lane14) {
    if (kLdsTwoLevel(rowlog)) {
      // n = 8192 / 16384: lane-addressed tables (internal.hpp, kLane13Size / kLane14Size), every value rounded from double
      h.clear();
      auto w = [&](long k, long n) { h.push_back(mk((float)cos(k * 2 * kPI / n), -(float)sin(k * 2 * kPI / n))); };
      for (int j = 0; j < 16; j++)
        for (int t = 0; t < 16; t++) w(j * t, 256);
      for (int k = 0; k < 4; k++)
        for (int j = 0; j < 256; j++) w(((1 << k) * j) & 4095, 4096);
      if (rowlog == kLds14Log) {
        for (int m = 1; m <= 3; m++)
          for (int t = 0; t < 1024; t++) w(m * t, 16384);
      } else {
        for (int t = 0; t < 512; t++) w(t, 8192);
      }
    } else {
      fill_twiddle(h, rown / 2, rown, 1, -1.f);
    }
    if ((e = upload(p->half, h.data(), sizeof(cpx) * h.size()))) return e;
    p->tabs.half = (const cpx *)p->half.p;
  }
  if (!(rowlog <= kLdsMaxLog || lane14) || both) {
    rowlog = fourlog;
    rown = 1 << rowlog;
    std::vector<cpx> all;
    fill_fourstep_tables(all, rowlog);
    if ((e = upload(p->four, all.data(), sizeof(cpx) * all.size()))) return e;
    p->tabs.four = (const cpx *)p->four.p;
    if (rowlog == 16) {
      fill_res16_tables(all);
      if ((e = upload(p->res16, all.data(), sizeof(cpx) * all.size()))) return e;
      p->tabs.res16 = (const cpx *)p->res16.p;
    }
    size_t sbytes = (size_t)fourstep_grid(p->di) * rown * sizeof(cpx);
    DevBuf &ws = p->logn > kMaxLog ? p->scratch2 : p->scratch;
    if ((e = ws.ensure(sbytes))) return e;
  }
  if (real) {
    fill_w2(h, n, fwd ? -1.f : 1.f);
    if ((e = upload(p->w2, h.data(), sizeof(cpx) * n))) return e;
    p->tabs.w2 = (const cpx *)p->w2.p;
  }
  if (p->c2x13 || p->r2x13) {
    // the n = 8192 lane tables (as above) + the radix-2 step's lane constants W_16384^t
    h.clear();
    auto w = [&](long k, long nn) { h.push_back(mk((float)cos(k * 2 * kPI / nn), -(float)sin(k * 2 * kPI / nn))); };
    for (int j = 0; j < 16; j++)
      for (int t = 0; t < 16; t++) w(j * t, 256);
    for (int k = 0; k < 4; k++)
      for (int j = 0; j < 256; j++) w(((1 << k) * j) & 4095, 4096);
    for (int t = 0; t < 512; t++) w(t, 8192);
    for (int t = 0; t < 512; t++) w(t, 16384);
    if ((e = upload(p->half2, h.data(), sizeof(cpx) * h.size()))) return e;
  }
  return CLFA_SUCCESS;
}

int clfa_cfft_create(clfa_fft **plan, int device, int n, int forward) {
  if (!plan) return CLFA_INVALID_VALUE;
  clfa_fft *p = new (std::nothrow) clfa_fft();
  if (!p) return CLFA_OUT_OF_HOST_MEMORY;
  p->err = fft_setup(p, device, n, false, n, forward != 0);
  *plan = p;
  return p->err;
}

int clfa_rfft_create(clfa_fft **plan, int device, int size, int forward) {
  if (!plan) return CLFA_INVALID_VALUE;
  clfa_fft *p = new (std::nothrow) clfa_fft();
  if (!p) return CLFA_OUT_OF_HOST_MEMORY;
  if (size < 4 || (size & 1)) {
    p->log[0] = 0;
    snprintf(p->log, sizeof(p->log), "real size must be even, 4..%d (got %d)", 2 << kBigMaxLog, size);
    p->err = CLFA_INVALID_VALUE;
  } else {
    p->err = fft_setup(p, device, size / 2, true, size, forward != 0);
  }
  *plan = p;
  return p->err;
}

void clfa_fft_destroy(clfa_fft *p) {
  if (!p) return;
  DeviceGuard guard;
  (void)guard.enter(p->di.device);
  if (p->stream) {
    (void)hipStreamSynchronize(p->stream);
    (void)hipStreamDestroy(p->stream);
  }
  for (auto &r : p->pinned) (void)hipHostFree(r.h);
  p->pinned.clear();
  if (p->own_cplx) clfa_fft_destroy(p->own_cplx);
  if (p->blue_f) clfa_fft_destroy(p->blue_f);
  if (p->blue_i) clfa_fft_destroy(p->blue_i);
  p->blue_w.release();
  p->blue_b.release();
  p->blue_work.release();
  p->half.release();
  p->half2.release();
  p->w2.release();
  p->four.release();
  p->scratch.release();
  p->stage.release();
  p->own1.release();
  p->own2.release();
  p->own_w.release();
  p->own_b.release();
  p->res16.release();
  p->zstage.release();
  p->bigtabs.release();
  p->scratch2.release();
  delete p;
}

int clfa_fft_get_error(const clfa_fft *p) { return p ? p->err : CLFA_INVALID_VALUE; }
const char *clfa_fft_get_log(const clfa_fft *p) { return p ? p->log : ""; }
size_t clfa_fft_workspace_bytes(const clfa_fft *p) {
  if (!p) return 0;
  size_t sub = p->blue_f ? clfa_fft_workspace_bytes(p->blue_f) + clfa_fft_workspace_bytes(p->blue_i) : 0;
  return p->scratch.bytes + p->scratch2.bytes + p->blue_work.bytes + sub;
}

const char *clfa_fft_kernel_name(const clfa_fft *p) {
  if (!p) return "";
  if (p->logn > kMaxLog) return p->big.logn2 <= 11 ? "k_big2_cols" : "k_big_cols";   // two passes (to 2^22) / three
  if (p->blue_m) return blue_lds_ok(p->blue_m) ? "k_blue_lds" : "bluestein";
  if (p->rlds15 || p->r2x13) return "k_rfft_2x";
  if (p->c2x13) return "k_cfft_2x";
  return p->logn <= kLdsMaxLog ? name_fft_lds(p->logn, p->fwd, !p->real ? MODE_C2C : (p->fwd ? MODE_R2C : MODE_C2R)) : name_fft_4step(p->logn);
}

// The body of every device-resident transform: `d` is read, the results go to d + off complex elements (off = 0: in
// place; otherwise a destination that does not overlap the source, which is then left untouched).  Two-pass routes run
// their FIRST pass from the source to the destination and the rest in place there.
static int fft_exec(clfa_fft *p, cpx *d, long off, long batch, hipStream_t s) {
  const bool scale = p->fwd;  // cl_fft.cpp:39-40: forward plans divide by N, inverse plans do not
  cpx *o = d + off;
  if (p->blue_m) {
    const int n = p->n, m = p->blue_m;
    cpx *work = (cpx *)p->blue_work.p;
    const cpx *w = (const cpx *)p->blue_w.p, *bt = (const cpx *)p->blue_b.p;
    cpx *src = d;
    if (p->real && !p->fwd) {
      HIP_TRY(launch_c2r_unpack(d, p->tabs.w2, n, batch, s, off));   // -> the destination
      src = o;
    }
    if (blue_lds_ok(m)) {   // one launch, one read and one write of the data (fft_kernels.hip, k_blue_lds)
      HIP_TRY(launch_blue_lds(m, src, o, w, bt, p->blue_f->tabs.half, n, (scale ? 1.0f / (float)n : 1.0f) / (float)m, batch, p->di, s));
      if (p->real && p->fwd) HIP_TRY(launch_r2c_pack(o, p->tabs.w2, n, batch, s));
      return CLFA_SUCCESS;
    }
    const long cb = (long)(p->blue_work.bytes / (sizeof(cpx) * (size_t)m));
    for (long b0 = 0; b0 < batch; b0 += cb) {
      const long nb = batch - b0 < cb ? batch - b0 : cb;
      HIP_TRY(launch_blue_pre(src + b0 * (long)n, w, work, n, m, nb, s));
      int e = fft_exec(p->blue_f, work, 0, nb, s);
      if (e) return e;
      HIP_TRY(launch_blue_mul(work, bt, m, nb, s));
      if ((e = fft_exec(p->blue_i, work, 0, nb, s))) return e;
      HIP_TRY(launch_blue_post(work, w, o + b0 * (long)n, n, m, scale ? 1.0f / (float)n : 1.0f, nb, s));
    }
    if (p->real && p->fwd) HIP_TRY(launch_r2c_pack(o, p->tabs.w2, n, batch, s));
    return CLFA_SUCCESS;
  }
  const bool spread = p->real && batch <= p->spread_below;   // a few transforms: one workgroup each would be slower
  if (p->rlds15 && !spread) {
    HIP_TRY(launch_rfft_lds15(p->fwd, d, p->tabs, batch, p->di, s, off));
    return CLFA_SUCCESS;
  }
  if (p->r2x13 && !spread) {
    FftTables t2 = p->tabs;
    t2.half = (const cpx *)p->half2.p;
    HIP_TRY(launch_rfft_2x13(p->fwd, d, t2, batch, p->di, s, off));
    return CLFA_SUCCESS;
  }
  if (p->c2x13 && batch * 4 > p->di.num_cus) {   // (fewer transforms: spread over the four-step column / row kernels)
    FftTables t2 = p->tabs;
    t2.half = (const cpx *)p->half2.p;
    HIP_TRY(launch_cfft_2x13(p->fwd, scale, d, t2, batch, p->di, s, off));
    return CLFA_SUCCESS;
  }
  if (p->logn <= kLdsMaxLog) {
    int mode = !p->real ? MODE_C2C : (p->fwd ? MODE_R2C : MODE_C2R);
    HIP_TRY(launch_fft_lds(p->logn, p->fwd, mode, scale, d, p->tabs, batch, p->di, s, off));
    return CLFA_SUCCESS;
  }
  if (p->r16 && batch * 4 > fourstep_grid(p->di)) {   // (fewer transforms: spread over the column / row kernels + pack)
    if (p->fwd) HIP_TRY(launch_rfft_res16(d, o, (cpx *)p->scratch.p, p->tabs.res16, p->tabs.w2, batch, p->di, s));
    else HIP_TRY(launch_crfft_res16(d, o, (cpx *)p->scratch.p, p->tabs.res16, p->tabs.w2, batch, p->di, s));
    return CLFA_SUCCESS;
  }
  // a complex transform, with the reference's pack / unpack as a pass of its own for the packed real sizes that have no
  // fused kernel (and for a few transforms of those that have one)
  cpx *src = d;
  long toff = off;   // the transform's own offset: the inverse real route is at the destination already
  if (p->real && !p->fwd) {
    HIP_TRY(launch_c2r_unpack(d, p->tabs.w2, p->n, batch, s, off));
    src = o;
    toff = 0;
  }
  if (p->logn > kMaxLog) {
    const long cb = (long)(p->scratch.bytes / (sizeof(cpx) * (size_t)p->n));
    for (long b0 = 0; b0 < batch; b0 += cb) {
      const long nb = batch - b0 < cb ? batch - b0 : cb;
      HIP_TRY(launch_fft_big(p->big, p->fwd, scale, src + b0 * (long)p->n, src + toff + b0 * (long)p->n, (cpx *)p->scratch.p,
                             (cpx *)p->scratch2.p, (const cpx *)p->bigtabs.p, p->tabs, nb, p->di, s));
    }
  } else {
    HIP_TRY(launch_fft_4step(p->logn, p->fwd, scale, src, (cpx *)p->scratch.p, p->tabs, batch, p->di, s, toff));
  }
  if (p->real && p->fwd) HIP_TRY(launch_r2c_pack(o, p->tabs.w2, p->n, batch, s));
  return CLFA_SUCCESS;
}

int clfa_fft_exec_dev(clfa_fft *p, void *data, long batch, void *stream) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  if (!data || batch < 0) return CLFA_INVALID_VALUE;
  if (batch == 0) return CLFA_SUCCESS;
  ENTER_DEVICE(p->di.device);
  hipStream_t s = (hipStream_t)stream;  // NULL is the HIP default stream
  HIP_TRY(p->order.use(s));
  return fft_exec(p, (cpx *)data, 0, batch, s);
}

int clfa_fft_exec_dev_oop(clfa_fft *p, const void *src, void *dst, long batch, void *stream) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  if (!src || !dst || batch < 0) return CLFA_INVALID_VALUE;
  if (src == dst) return clfa_fft_exec_dev(p, dst, batch, stream);
  if (batch == 0) return CLFA_SUCCESS;
  const size_t bytes = sizeof(cpx) * (size_t)p->n * (size_t)batch;   // real plans: n = size / 2 packed bins = size floats
  const char *a = (const char *)src, *b = (const char *)dst;
  if (a < b + bytes && b < a + bytes) return CLFA_INVALID_VALUE;     // partly overlapping
  if ((b - a) % (long)sizeof(cpx)) return CLFA_INVALID_VALUE;         // the two buffers a whole number of complex values apart
  ENTER_DEVICE(p->di.device);
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(p->order.use(s));
  // every kernel reads the source and writes the destination (the kernels take the distance between the two); the source
  // is never written — also not by the routes of several passes, whose first pass already lands in the destination
  return fft_exec(p, (cpx *)const_cast<void *>(src), (long)((b - a) / (long)sizeof(cpx)), batch, s);
}

int clfa_fft_device_buffers(clfa_fft *p, void **data1, void **data2, void **commands) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  ENTER_DEVICE(p->di.device);
  const size_t bytes = sizeof(cpx) * (size_t)p->n;
  int e = p->own1.ensure(bytes);
  if (!e) e = p->own2.ensure(bytes);
  if (e) return e;
  if (data1) *data1 = p->own1.p;
  if (data2) *data2 = p->own2.p;
  if (commands) *commands = (void *)p->stream;
  return CLFA_SUCCESS;
}

int clfa_fft_device_tables(clfa_fft *p, void **w, void **b) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  if (p->blue_m || p->logn < 1 || p->logn > kMaxLog) return CLFA_INVALID_OPERATION;
  ENTER_DEVICE(p->di.device);
  const int n = p->n;
  if (!p->own_tables_ready) {
    std::vector<cpx> tw;
    fill_twiddle(tw, n, n, 1, p->fwd ? -1.f : 1.f);                 // cl_fft.cpp:86-91
    std::vector<int> br((size_t)n);
    int e = clfa_bitrev_table(n, br.data());                        // cl_fft.cpp:96-101
    if (!e) e = p->own_w.ensure(sizeof(cpx) * (size_t)n);
    if (!e) e = p->own_b.ensure(sizeof(int) * (size_t)n);
    if (!e) e = map_hip(hipMemcpy(p->own_w.p, tw.data(), sizeof(cpx) * (size_t)n, hipMemcpyHostToDevice));
    if (!e) e = map_hip(hipMemcpy(p->own_b.p, br.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    if (e) {   // all or nothing: a later call must not hand out a table that was never filled
      (void)hipGetLastError();
      p->own_w.release();
      p->own_b.release();
      return e;
    }
    p->own_tables_ready = true;
  }
  if (w) *w = p->own_w.p;
  if (b) *b = p->own_b.p;
  return CLFA_SUCCESS;
}

int clfa_copy_to_device(void *stream, void *dst, const void *src, size_t bytes, int blocking) {
  if ((!dst || !src) && bytes) return CLFA_INVALID_VALUE;
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  if (blocking) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return CLFA_SUCCESS;
}
int clfa_copy_from_device(void *stream, void *dst, const void *src, size_t bytes, int blocking) {
  if ((!dst || !src) && bytes) return CLFA_INVALID_VALUE;
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (blocking) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return CLFA_SUCCESS;
}
int clfa_stream_synchronize(void *stream) {
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return CLFA_SUCCESS;
}

int clfa_fft_run_buffers(clfa_fft *p) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  if (!p->own1.p || !p->own2.p) return CLFA_INVALID_MEM_OBJECT;
  if (!p->real) return clfa_fft_exec_dev_oop(p, p->own1.p, p->own2.p, 1, p->stream);
  // a Clrfft's fft() is the complex transform of its N = size / 2 points and nothing else: the reference's conv / iconv
  // are kernels of their own, enqueued by Clrfft::transform (cl_fft.cpp:267-296), not by fft()
  if (!p->own_cplx) {
    const int e = clfa_cfft_create(&p->own_cplx, p->di.device, p->n, p->fwd ? 1 : 0);
    if (e) {
      if (p->own_cplx) clfa_fft_destroy(p->own_cplx);
      p->own_cplx = nullptr;
      return e;
    }
  }
  return clfa_fft_exec_dev_oop(p->own_cplx, p->own1.p, p->own2.p, 1, p->stream);
}

// bytes per call up to which the host entry points go zero-copy: the kernels read the input from,
// and write the result to, mapped pinned host memory (one pass each way), instead of two
// hipMemcpyAsync calls of 10-15 us each around a kernel of a few microseconds
#ifndef CLFA_ZEROCOPY_MAX_KIB
#define CLFA_ZEROCOPY_MAX_KIB 512   // (profiles/host_path_r05.txt: one N = 65536 transform, 512 KiB: 59.5 us this way, 73.5 by copies)
#endif
constexpr size_t kZeroCopyMax = (size_t)CLFA_ZEROCOPY_MAX_KIB << 10;
constexpr size_t kZeroCopyMaxConv = (size_t)256 << 10;   // the convolutions' blocks (measured at this size only)

// host staging in chunks of at most ~256 MiB so huge host batches do not need a
// device buffer of their full size
static long chunk_batches(size_t bytes_per_batch, long batch) {
  size_t cap = (size_t)256 << 20;
  long c = (long)(cap / bytes_per_batch);
  if (c < 1) c = 1;
  return c < batch ? c : batch;
}

// ---- pinned arrays for the caller (extension) --------------------------------------------------------------------
// The reference's transform() copies the caller's array to the device and back with two blocking transfers
// (cl_fft.cpp:155-158).  A caller that keeps ONE array for the object's life — the Csound opcodes do: one buffer per
// instance, csound/opcode.cpp — can take that array FROM the plan: page-locked host memory mapped into the device's
// address space.  transform() calls on arrays inside it run on that memory directly (the kernels read and write it over
// PCIe, one pass each way, no staging copy, one synchronisation).
// Why the plan allocates instead of pinning the caller's own array: hipHostRegister on heap arrays was built and measured
// first (round 5) — with arrays registered, unregistered, freed and their addresses reused by arrays of other sizes it
// produced wrong results in 8 of 3000 randomised calls and one GPU memory access fault on a host heap address
// (tools/stress_pinned.py; profiles/host_path_r05.txt) although every array was unregistered before it was freed.
// Memory of hipHostMalloc is the route the staging buffers have used since round 1.
int clfa_fft_host_alloc(clfa_fft *p, size_t bytes, void **ptr) {
  if (!p || !ptr) return CLFA_INVALID_VALUE;
  *ptr = nullptr;
  if (p->err) return p->err;
  if (!bytes) return CLFA_INVALID_VALUE;
  ENTER_DEVICE(p->di.device);
  void *h = nullptr, *d = nullptr;
  hipError_t e = hipHostMalloc(&h, bytes, hipHostMallocMapped);
  if (e == hipSuccess && (e = hipHostGetDevicePointer(&d, h, 0)) != hipSuccess) (void)hipHostFree(h);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return map_hip(e);
  }
  p->pinned.push_back({(char *)h, (char *)d, bytes});
  *ptr = h;
  return CLFA_SUCCESS;
}
int clfa_fft_host_free(clfa_fft *p, void *ptr) {
  if (!p) return CLFA_INVALID_VALUE;
  for (size_t i = 0; i < p->pinned.size(); i++)
    if (p->pinned[i].h == (char *)ptr) {
      ENTER_DEVICE(p->di.device);
      if (p->stream) (void)hipStreamSynchronize(p->stream);
      (void)hipHostFree(ptr);
      p->pinned.erase(p->pinned.begin() + (long)i);
      return CLFA_SUCCESS;
    }
  return CLFA_INVALID_VALUE;
}
// device view of [h, h + bytes) if it lies inside a pinned range, else NULL
static void *pinned_dev(const clfa_fft *p, const void *h, size_t bytes) {
  for (auto &r : p->pinned)
    if ((const char *)h >= r.h && (const char *)h + bytes <= r.h + r.bytes) return r.d + ((const char *)h - r.h);
  return nullptr;
}
// Pinned arrays up to this size run zero-copy when the plan's route touches its source and its destination once each
// (every route of the reference's range except the launch chains with a pack / unpack pass of their own); beyond it, and
// on the other routes, they are copied by DMA like pageable arrays, only faster.
constexpr size_t kPinnedZeroCopyMax = (size_t)8 << 20;
static bool one_touch_route(const clfa_fft *p, long batch) {
  if (p->blue_m || p->logn > kMaxLog) return false;
  if (!p->real) return true;
  if (p->logn <= kLdsMaxLog) return true;
  return batch > p->spread_below && (p->rlds15 || p->r2x13 || (p->r16 && batch * 4 > fourstep_grid(p->di)));
}

int clfa_cfft_transform(clfa_fft *p, float *c, long batch) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  if (!c || batch < 0 || p->real) return CLFA_INVALID_VALUE;
  ENTER_DEVICE(p->di.device);
  const size_t per = sizeof(cpx) * (size_t)p->n;
  if (batch > 0 && per * (size_t)batch <= kPinnedZeroCopyMax && one_touch_route(p, batch)) {
    if (void *dv = pinned_dev(p, c, per * (size_t)batch)) {   // an array of clfa_fft_host_alloc: in place over PCIe
      // (profiles/host_path_r05.txt: N = 65536 42.2 us; with the copy engine bringing the array in first 46.4)
      const int e = clfa_fft_exec_dev(p, dv, batch, p->stream);
      if (e) return e;
      HIP_TRY(hipStreamSynchronize(p->stream));
      return CLFA_SUCCESS;
    }
  }
  if (batch > 0 && per * (size_t)batch <= kZeroCopyMax) {
    int e = p->zstage.ensure(per * batch);
    if (e) return e;
    memcpy(p->zstage.h, c, per * batch);
    if ((e = clfa_fft_exec_dev(p, p->zstage.d, batch, p->stream))) return e;
    HIP_TRY(hipStreamSynchronize(p->stream));
    memcpy(c, p->zstage.h, per * batch);
    return CLFA_SUCCESS;
  }
  const long cb = chunk_batches(per, batch);
  for (long b0 = 0; b0 < batch; b0 += cb) {
    long nb = batch - b0 < cb ? batch - b0 : cb;
    int e = p->stage.ensure(per * nb);
    if (e) return e;
    char *h = (char *)c + per * b0;
    HIP_TRY(hipMemcpyAsync(p->stage.p, h, per * nb, hipMemcpyHostToDevice, p->stream));   // cl_fft.cpp:155
    if ((e = clfa_fft_exec_dev(p, p->stage.p, nb, p->stream))) return e;                   // cl_fft.cpp:157
    HIP_TRY(hipMemcpyAsync(h, p->stage.p, per * nb, hipMemcpyDeviceToHost, p->stream));   // cl_fft.cpp:158
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  return CLFA_SUCCESS;
}

int clfa_rfft_transform(clfa_fft *p, float *c, float *r, long batch) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  if (!c || !r || batch < 0 || !p->real) return CLFA_INVALID_VALUE;
  ENTER_DEVICE(p->di.device);
  const size_t per = sizeof(cpx) * (size_t)p->n;  // size floats == M complex
  if (batch > 0 && per * (size_t)batch <= kPinnedZeroCopyMax && one_touch_route(p, batch)) {
    void *src = p->fwd ? (void *)r : (void *)c, *dst = p->fwd ? (void *)c : (void *)r;
    void *ds = pinned_dev(p, src, per * (size_t)batch), *dd = src == dst ? ds : pinned_dev(p, dst, per * (size_t)batch);
    if (ds && dd) {   // both arrays (or the one, in place) come from clfa_fft_host_alloc
      const int e = clfa_fft_exec_dev_oop(p, ds, dd, batch, p->stream);
      if (e) return e;
      HIP_TRY(hipStreamSynchronize(p->stream));
      return CLFA_SUCCESS;
    }
  }
  if (batch > 0 && per * (size_t)batch <= kZeroCopyMax) {
    int e = p->zstage.ensure(per * batch);
    if (e) return e;
    memcpy(p->zstage.h, p->fwd ? (void *)r : (void *)c, per * batch);
    if ((e = clfa_fft_exec_dev(p, p->zstage.d, batch, p->stream))) return e;
    HIP_TRY(hipStreamSynchronize(p->stream));
    memcpy(p->fwd ? (void *)c : (void *)r, p->zstage.h, per * batch);
    return CLFA_SUCCESS;
  }
  const long cb = chunk_batches(per, batch);
  // forward reads r and writes c; inverse reads c and writes r (cl_fft.cpp:272-294)
  char *src = (char *)(p->fwd ? (void *)r : (void *)c);
  char *dst = (char *)(p->fwd ? (void *)c : (void *)r);
  for (long b0 = 0; b0 < batch; b0 += cb) {
    long nb = batch - b0 < cb ? batch - b0 : cb;
    int e = p->stage.ensure(per * nb);
    if (e) return e;
    HIP_TRY(hipMemcpyAsync(p->stage.p, src + per * b0, per * nb, hipMemcpyHostToDevice, p->stream));
    if ((e = clfa_fft_exec_dev(p, p->stage.p, nb, p->stream))) return e;
    HIP_TRY(hipMemcpyAsync(dst + per * b0, p->stage.p, per * nb, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  return CLFA_SUCCESS;
}

int clfa_reorder_dev(int device, void *out, const void *in, int n, long batch, void *stream) {
  if (!out || !in || out == in || !is_pow2(n) || n < 2 || batch < 0) return CLFA_INVALID_VALUE;
  ENTER_DEVICE(device);
  HIP_TRY(launch_reorder((cpx *)out, (const cpx *)in, ilog2(n), batch, (hipStream_t)stream));
  return CLFA_SUCCESS;
}

// ---------------------------------------------------------------------------------
// partitioned convolution
// ---------------------------------------------------------------------------------

static int pconv_setup(clfa_pconv *p, int device, int cvs, int pts, int channels) {
  p->cvs = cvs;
  p->pts = pts;
  if (!is_pow2(pts) || pts < 2 || pts > (1 << kPconvMaxLogBins) || cvs < pts || channels < 1)
    return CLFA_INVALID_VALUE;
  p->g.bins = pts;                 // cl_conv.cpp:143
  p->g.logb = ilog2(pts);
  p->g.nparts = cvs / pts;         // floor: remainder samples are dropped
  p->g.channels = channels;
  p->wp = 0;
  p->wp2 = p->g.nparts - 1;        // cl_conv.cpp:144
  int e = device_info(device, p->di);
  if (e) return e;
  p->fused = pconv_fused_ok(p->g, p->di) && !getenv("CLFA_PCONV_NO_FUSE");   // tuning switch, read once
  if (!p->fused) p->coop = pconv_coop_plan(p->g, p->di);
  ENTER_DEVICE(device);
  HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  std::vector<cpx> h;
  fill_twiddle(h, pts / 2, pts, 1, -1.f);
  if ((e = upload(p->half, h.data(), sizeof(cpx) * h.size()))) return e;
  fill_w2(h, pts, -1.f);           // cl_conv.cpp:276-281
  if ((e = upload(p->w2f, h.data(), sizeof(cpx) * pts))) return e;
  fill_w2(h, pts, 1.f);            // cl_conv.cpp:282-287
  if ((e = upload(p->w2i, h.data(), sizeof(cpx) * pts))) return e;
  if (p->g.logb > kLdsMaxLog) {
    const int n = pts;
    std::vector<cpx> all;
    fill_fourstep_tables(all, p->g.logb);
    if ((e = upload(p->four, all.data(), sizeof(cpx) * all.size()))) return e;
    p->big.four = (const cpx *)p->four.p;
    if ((e = p->scratch.ensure((size_t)fourstep_grid(p->di) * n * sizeof(cpx)))) return e;
    if ((e = p->work.ensure(sizeof(cpx) * (size_t)channels * n))) return e;
  }
  const size_t ring = sizeof(cpx) * (size_t)channels * p->g.nparts * pts;
  const size_t blk = sizeof(float) * (size_t)channels * pts;
  if ((e = p->ringA.ensure(ring))) return e;
  if ((e = p->ringB.ensure(ring))) return e;
  const int acc_copies = p->coop.logs >= 0 ? p->coop.sparts : pconv_mac_split(p->g);
  if ((e = p->acc.ensure(sizeof(cpx) * (size_t)channels * pts * acc_copies))) return e;
  if ((e = p->tail.ensure(blk))) return e;
  if (p->coop.logs >= 0) {
    if ((e = p->cnt.ensure(sizeof(unsigned) * (size_t)channels))) return e;
    HIP_TRY(hipMemsetAsync(p->cnt.p, 0, sizeof(unsigned) * (size_t)channels, p->stream));
  }
  // zero-initialised state (cl_conv.cpp:303-313)
  HIP_TRY(hipMemsetAsync(p->ringA.p, 0, ring, p->stream));
  HIP_TRY(hipMemsetAsync(p->ringB.p, 0, ring, p->stream));
  HIP_TRY(hipMemsetAsync(p->tail.p, 0, blk, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return CLFA_SUCCESS;
}

int clfa_pconv_create(clfa_pconv **pc, int device, int cvs, int pts, int channels) {
  if (!pc) return CLFA_INVALID_VALUE;
  clfa_pconv *p = new (std::nothrow) clfa_pconv();
  if (!p) return CLFA_OUT_OF_HOST_MEMORY;
  p->err = pconv_setup(p, device, cvs, pts, channels);
  *pc = p;
  return p->err;
}

void clfa_pconv_destroy(clfa_pconv *p) {
  if (!p) return;
  DeviceGuard guard;
  (void)guard.enter(p->di.device);
  if (p->stream) {
    (void)hipStreamSynchronize(p->stream);
    (void)hipStreamDestroy(p->stream);
  }
  for (DevBuf *b : {&p->half, &p->w2f, &p->w2i, &p->ringA, &p->ringB, &p->acc, &p->tail, &p->in1, &p->in2,
                    &p->out, &p->ir, &p->four, &p->scratch, &p->work, &p->cnt})
    b->release();
  p->zin1.release();
  p->zin2.release();
  p->zout.release();
  delete p;
}

int clfa_pconv_get_error(const clfa_pconv *p) { return p ? p->err : CLFA_INVALID_VALUE; }
int clfa_pconv_nparts(const clfa_pconv *p) { return p ? p->g.nparts : 0; }
int clfa_pconv_wp(const clfa_pconv *p) { return p ? p->wp : -1; }
int clfa_pconv_wp2(const clfa_pconv *p) { return p ? p->wp2 : -1; }
const char *clfa_pconv_kernel_name(const clfa_pconv *p) {
  if (!p || p->err) return "";
  return p->fused ? "k_pconv_fused" : (p->coop.logs >= 0 ? "k_pconv_coop" : "chain");
}
size_t clfa_pconv_state_bytes(const clfa_pconv *p) {
  return p ? p->ringA.bytes + p->ringB.bytes + p->acc.bytes + p->tail.bytes : 0;
}

// forward chain of one block for all channels: in -> spectrum frame `frame` of `ring`
static int pconv_forward(clfa_pconv *p, const float *in, long in_stride, cpx *ring, int frame, hipStream_t s) {
  if (p->g.logb <= kLdsMaxLog) {
    HIP_TRY(launch_pconv_forward(p->g, in, in_stride, ring, frame, (const cpx *)p->half.p, (const cpx *)p->w2f.p, s));
    return CLFA_SUCCESS;
  }
  // composed: zero-pad -> large-N forward FFT (unscaled) -> reference r2c -> place the frames in the ring
  const int bins = p->g.bins, ch = p->g.channels;
  cpx *work = (cpx *)p->work.p;
  HIP_TRY(launch_pconv_pad(in, in_stride, work, bins, ch, s));
  HIP_TRY(launch_fft_4step(p->g.logb, true, false, work, (cpx *)p->scratch.p, p->big, ch, p->di, s));
  HIP_TRY(launch_r2c_pack(work, (const cpx *)p->w2f.p, bins, ch, s));
  HIP_TRY(hipMemcpy2DAsync(ring + (size_t)frame * bins, sizeof(cpx) * (size_t)p->g.nparts * bins, work,
                           sizeof(cpx) * (size_t)bins, sizeof(cpx) * (size_t)bins, ch, hipMemcpyDeviceToDevice, s));
  return CLFA_SUCCESS;
}

// inverse chain: accumulator -> c2r -> inverse FFT -> overlap-add
static int pconv_inverse(clfa_pconv *p, float *out, hipStream_t s) {
  if (p->g.logb <= kLdsMaxLog) {
    HIP_TRY(launch_pconv_inverse(p->g, (const cpx *)p->acc.p, (float *)p->tail.p, out, (const cpx *)p->half.p,
                                 (const cpx *)p->w2i.p, s));
    return CLFA_SUCCESS;
  }
  const int bins = p->g.bins, ch = p->g.channels;
  cpx *acc = (cpx *)p->acc.p;
  HIP_TRY(launch_c2r_unpack(acc, (const cpx *)p->w2i.p, bins, ch, s));
  HIP_TRY(launch_fft_4step(p->g.logb, false, false, acc, (cpx *)p->scratch.p, p->big, ch, p->di, s));
  HIP_TRY(launch_pconv_olap((const float *)acc, (float *)p->tail.p, out, bins, ch, s));
  return CLFA_SUCCESS;
}

int clfa_pconv_push_ir_dev(clfa_pconv *p, const void *ir, long channel_stride, void *stream) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  if (!ir || channel_stride < (long)p->g.nparts * p->pts) return CLFA_INVALID_VALUE;
  ENTER_DEVICE(p->di.device);
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(p->order.use(s));
  const long stride = channel_stride;
  // cl_conv.cpp:358-386: partition i -> frame wp2, wp2 counts down from nparts-1
  for (int i = 0; i < p->g.nparts; i++) {
    int e = pconv_forward(p, (const float *)ir + (long)i * p->pts, stride, (cpx *)p->ringB.p, p->wp2, s);
    if (e) return e;
    p->wp2 = p->wp2 == 0 ? p->g.nparts - 1 : p->wp2 - 1;
  }
  return CLFA_SUCCESS;
}

int clfa_pconv_push_ir(clfa_pconv *p, const float *ir) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  if (!ir) return CLFA_INVALID_VALUE;
  ENTER_DEVICE(p->di.device);
  const size_t bytes = sizeof(float) * (size_t)p->g.channels * p->g.nparts * p->pts;
  int e = p->ir.ensure(bytes);
  if (e) return e;
  HIP_TRY(hipMemcpyAsync(p->ir.p, ir, bytes, hipMemcpyHostToDevice, p->stream));
  if ((e = clfa_pconv_push_ir_dev(p, p->ir.p, (long)p->g.nparts * p->pts, p->stream))) return e;
  HIP_TRY(hipStreamSynchronize(p->stream));
  return CLFA_SUCCESS;
}

// [a, a + n) and [b, b + n) share a byte
static bool ranges_overlap(const void *a, const void *b, size_t n) {
  const char *x = (const char *)a, *y = (const char *)b;
  return x < y + n && y < x + n;
}

int clfa_pconv_process_dev(clfa_pconv *p, void *out, const void *in1, const void *in2, void *stream) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  if (!out || !in1) return CLFA_INVALID_VALUE;
  if (p->fused || p->coop.logs >= 0) {
    // the one-launch routes read the inputs of ALL channels while workgroups of other channels may already write their
    // output (the buffers are __restrict__): any overlap of out with an input — not only equal pointers — is refused.
    // (The launch chain below has read every input when its forward launch ends, before the inverse launch writes `out`:
    // in place is fine there, as it was for the reference's host arrays.)
    const size_t blk = sizeof(float) * (size_t)p->pts * (size_t)p->g.channels;
    if (ranges_overlap(out, in1, blk) || (in2 && ranges_overlap(out, in2, blk))) return CLFA_INVALID_VALUE;
  }
  ENTER_DEVICE(p->di.device);
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(p->order.use(s));
  int e;
  if (p->fused || p->coop.logs >= 0) {
    // whole block in one launch; ring indices advance exactly as below — committed only once the launch has been accepted
    // (a rejected launch must not skew the host's ring position against the device's rings)
    const int frame1 = p->wp, frame2 = p->wp2;
    const int wp_next = p->wp != p->g.nparts - 1 ? p->wp + 1 : 0;
    const int wp2_next = in2 ? (p->wp2 == 0 ? p->g.nparts - 1 : p->wp2 - 1) : p->wp2;
    if (!p->fused) {
      HIP_TRY(launch_pconv_coop(p->g, p->coop, (const float *)in1, (const float *)in2, (cpx *)p->ringA.p,
                                (cpx *)p->ringB.p, (float *)p->tail.p, (float *)out, frame1, frame2, wp_next,
                                (const cpx *)p->half.p, (const cpx *)p->w2f.p, (const cpx *)p->w2i.p, (cpx *)p->acc.p,
                                (unsigned *)p->cnt.p, p->di.num_cus, s));
    } else {
      HIP_TRY(launch_pconv_fused(p->g, (const float *)in1, (const float *)in2, (cpx *)p->ringA.p, (cpx *)p->ringB.p,
                                 (float *)p->tail.p, (float *)out, frame1, frame2, wp_next, (const cpx *)p->half.p,
                                 (const cpx *)p->w2f.p, (const cpx *)p->w2i.p, s, p->g.channels < p->di.num_cus));
    }
    p->wp = wp_next;
    p->wp2 = wp2_next;
    return CLFA_SUCCESS;
  }
  const bool lds = p->g.logb <= kLdsMaxLog;
  // forward chain(s): cl_conv.cpp:399-419 / 465-513 (both inputs of a time-varying block in one launch)
  if (lds && in2) {
    HIP_TRY(launch_pconv_forward(p->g, (const float *)in1, p->pts, (cpx *)p->ringA.p, p->wp, (const cpx *)p->half.p,
                                 (const cpx *)p->w2f.p, s, (const float *)in2, (cpx *)p->ringB.p, p->wp2));
  } else {
    if ((e = pconv_forward(p, (const float *)in1, p->pts, (cpx *)p->ringA.p, p->wp, s))) return e;
    if (in2 && (e = pconv_forward(p, (const float *)in2, p->pts, (cpx *)p->ringB.p, p->wp2, s))) return e;
  }
  p->wp = p->wp != p->g.nparts - 1 ? p->wp + 1 : 0;            // cl_conv.cpp:424 / 516
  if (in2) p->wp2 = p->wp2 == 0 ? p->g.nparts - 1 : p->wp2 - 1;  // cl_conv.cpp:519
  // cl_conv.cpp:428-449.  (Adding the partial sums of a split MAC inside the single-workgroup inverse kernel
  // instead of the wide k_pconv_reduce launch was measured: 22 -> 130 us per block for one channel.)
  HIP_TRY(launch_pconv_mac(p->g, (const cpx *)p->ringA.p, (const cpx *)p->ringB.p, p->wp, (cpx *)p->acc.p, s));
  if ((e = pconv_inverse(p, (float *)out, s))) return e;
  return CLFA_SUCCESS;
}

static int pconv_host(clfa_pconv *p, float *out, const float *in1, const float *in2) {
  if (!p) return CLFA_INVALID_VALUE;
  if (p->err) return p->err;
  if (!out || !in1) return CLFA_INVALID_VALUE;
  ENTER_DEVICE(p->di.device);
  const size_t blk = sizeof(float) * (size_t)p->g.channels * p->pts;
  int e;
  if (blk <= kZeroCopyMaxConv) {
    // one audio block of a few channels: the kernels read the input from, and write the output to,
    // mapped pinned host memory — no copy calls, one synchronisation (cl_conv.cpp:399, 455)
    if ((e = p->zin1.ensure(blk)) || (e = p->zout.ensure(blk)) || (in2 && (e = p->zin2.ensure(blk)))) return e;
    memcpy(p->zin1.h, in1, blk);
    if (in2) memcpy(p->zin2.h, in2, blk);
    if ((e = clfa_pconv_process_dev(p, p->zout.d, p->zin1.d, in2 ? p->zin2.d : nullptr, p->stream))) return e;
    HIP_TRY(hipStreamSynchronize(p->stream));
    memcpy(out, p->zout.h, blk);
    return CLFA_SUCCESS;
  }
  if ((e = p->in1.ensure(blk)) || (e = p->out.ensure(blk))) return e;
  HIP_TRY(hipMemcpyAsync(p->in1.p, in1, blk, hipMemcpyHostToDevice, p->stream));
  if (in2) {
    if ((e = p->in2.ensure(blk))) return e;
    HIP_TRY(hipMemcpyAsync(p->in2.p, in2, blk, hipMemcpyHostToDevice, p->stream));
  }
  if ((e = clfa_pconv_process_dev(p, p->out.p, p->in1.p, in2 ? p->in2.p : nullptr, p->stream))) return e;
  HIP_TRY(hipMemcpyAsync(out, p->out.p, blk, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));   // blocking read, cl_conv.cpp:455
  return CLFA_SUCCESS;
}

int clfa_pconv_convolution(clfa_pconv *p, float *out, const float *in) { return pconv_host(p, out, in, nullptr); }
int clfa_pconv_convolution_tv(clfa_pconv *p, float *out, const float *in1, const float *in2) {
  if (!in2) return CLFA_INVALID_VALUE;
  return pconv_host(p, out, in1, in2);
}

// ---------------------------------------------------------------------------------
// direct convolution
// ---------------------------------------------------------------------------------

int clfa_dconv_create(clfa_dconv **dc, int device, int irsize, int vsize) {
  if (!dc) return CLFA_INVALID_VALUE;
  clfa_dconv *d = new (std::nothrow) clfa_dconv();
  if (!d) return CLFA_OUT_OF_HOST_MEMORY;
  *dc = d;
  d->irsize = irsize;
  d->vsize = vsize;
  auto setup = [&]() -> int {
    if (irsize < 1 || vsize < 1 || (long)irsize * vsize > 0x7fffffffL) return CLFA_INVALID_VALUE;
    int e = device_info(device, d->di);
    if (e) return e;
    ENTER_DEVICE(device);
    HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    const size_t ring = sizeof(float) * ((size_t)irsize + vsize), blk = sizeof(float) * (size_t)vsize;
    d->plan = dconv_plan(irsize, vsize);
    if ((e = d->del.ensure(ring)) || (e = d->coefs.ensure(ring)) || (e = d->out.ensure(blk)) || (e = d->in1.ensure(blk)) ||
        (e = d->in2.ensure(blk)) || (e = d->part.ensure(blk * d->plan.G)) ||
        (e = d->cnt.ensure(sizeof(unsigned) * d->plan.VB)))
      return e;
    // the reference leaves these uninitialised (cl_dconv.cpp:87-91); zero is the intent
    HIP_TRY(hipMemsetAsync(d->del.p, 0, ring, d->stream));
    HIP_TRY(hipMemsetAsync(d->coefs.p, 0, ring, d->stream));
    HIP_TRY(hipMemsetAsync(d->cnt.p, 0, sizeof(unsigned) * d->plan.VB, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return CLFA_SUCCESS;
  };
  d->err = setup();
  return d->err;
}

void clfa_dconv_destroy(clfa_dconv *d) {
  if (!d) return;
  DeviceGuard guard;
  (void)guard.enter(d->di.device);
  if (d->stream) {
    (void)hipStreamSynchronize(d->stream);
    (void)hipStreamDestroy(d->stream);
  }
  d->del.release();
  d->coefs.release();
  d->out.release();
  d->in1.release();
  d->in2.release();
  d->zin1.release();
  d->zin2.release();
  d->zout.release();
  d->part.release();
  d->cnt.release();
  delete d;
}

int clfa_dconv_get_error(const clfa_dconv *d) { return d ? d->err : CLFA_INVALID_VALUE; }

int clfa_dconv_push_ir(clfa_dconv *d, const float *ir) {
  if (!d) return CLFA_INVALID_VALUE;
  if (d->err) return d->err;
  if (!ir) return CLFA_INVALID_VALUE;
  ENTER_DEVICE(d->di.device);
  HIP_TRY(d->order.use(d->stream));
  HIP_TRY(hipMemcpyAsync(d->coefs.p, ir, sizeof(float) * d->irsize, hipMemcpyHostToDevice, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return CLFA_SUCCESS;
}

// one block on stream s, everything device-resident: ring write at wp with wrap-around (intent of cl_dconv.cpp:112-122;
// the two-input form writes in2 into the coefficient ring at the same point, :134-147), wp advanced (:124), vsize
// outputs — all in ONE launch (conv_kernels.hip, k_dconv_block)
static int dconv_block(clfa_dconv *d, float *out, const float *in1, const float *in2, hipStream_t s) {
  const int wp = d->wp;
  HIP_TRY(launch_dconv_block(d->plan, out, in1, in2, (float *)d->del.p, (float *)d->coefs.p, (float *)d->part.p,
                             (unsigned *)d->cnt.p, d->irsize, d->vsize, wp, d->di.num_cus, s));
  d->wp = (wp + d->vsize) % (d->irsize + d->vsize);   // committed only once the launch has been accepted
  return CLFA_SUCCESS;
}

static int dconv_host(clfa_dconv *d, float *out, const float *in1, const float *in2) {
  if (!d) return CLFA_INVALID_VALUE;
  if (d->err) return d->err;
  if (!out || !in1) return CLFA_INVALID_VALUE;
  ENTER_DEVICE(d->di.device);
  HIP_TRY(d->order.use(d->stream));
  const size_t blk = sizeof(float) * (size_t)d->vsize;
  if (blk <= (16u << 10)) {
    // an audio block: the kernel reads the input from, and writes the output to, mapped pinned host memory — no copy
    // calls, one synchronisation (as the partitioned convolution's host path; only for blocks of up to 4096 samples:
    // every workgroup whose ring window meets the new block reads it from there)
    int e;
    if ((e = d->zin1.ensure(blk)) || (e = d->zout.ensure(blk)) || (in2 && (e = d->zin2.ensure(blk)))) return e;
    memcpy(d->zin1.h, in1, blk);
    if (in2) memcpy(d->zin2.h, in2, blk);
    if ((e = dconv_block(d, (float *)d->zout.d, (const float *)d->zin1.d, in2 ? (const float *)d->zin2.d : nullptr,
                         d->stream)))
      return e;
    HIP_TRY(hipStreamSynchronize(d->stream));
    memcpy(out, d->zout.h, blk);
    return CLFA_SUCCESS;
  }
  HIP_TRY(hipMemcpyAsync(d->in1.p, in1, blk, hipMemcpyHostToDevice, d->stream));
  if (in2) HIP_TRY(hipMemcpyAsync(d->in2.p, in2, blk, hipMemcpyHostToDevice, d->stream));
  int e = dconv_block(d, (float *)d->out.p, (const float *)d->in1.p, in2 ? (const float *)d->in2.p : nullptr, d->stream);
  if (e) return e;
  HIP_TRY(hipMemcpyAsync(out, d->out.p, blk, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return CLFA_SUCCESS;
}

int clfa_dconv_convolution(clfa_dconv *d, float *out, const float *in) { return dconv_host(d, out, in, nullptr); }

int clfa_dconv_convolution_tv(clfa_dconv *d, float *out, const float *in1, const float *in2) {
  if (d && !d->err && !in2) return CLFA_INVALID_VALUE;
  return dconv_host(d, out, in1, in2);
}

int clfa_dconv_process_dev(clfa_dconv *d, void *out, const void *in1, const void *in2, void *stream) {
  if (!d) return CLFA_INVALID_VALUE;
  if (d->err) return d->err;
  if (!out || !in1) return CLFA_INVALID_VALUE;
  // the last-arriving workgroup writes out while others may still stage their in1 / in2 windows: no overlap at all
  const size_t blk = sizeof(float) * (size_t)d->vsize;
  if (ranges_overlap(out, in1, blk) || (in2 && ranges_overlap(out, in2, blk))) return CLFA_INVALID_VALUE;
  ENTER_DEVICE(d->di.device);
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(d->order.use(s));
  return dconv_block(d, (float *)out, (const float *)in1, (const float *)in2, s);
}

}  // extern "C"
