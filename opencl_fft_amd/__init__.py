"""opencl_fft_amd — MI355X-native drop-in for the hot path of vlazzarini/opencl_fft.

Host-side mirror of the reference's class surface (same names, argument
meaning and integer OpenCL status codes), over the C ABI of libclfft_amd.so:

    Clcfft   cl_fft.h:29-70     complex FFT, forward scaled 1/N, inverse unscaled
    Clrfft   cl_fft.h:74-111    packed real FFT
    Clpconv  cl_conv.h:124-188  uniformly partitioned overlap-add convolution
    Cldconv  cl_dconv.h:17-66   direct convolution

Like the reference, constructors never raise: a failed setup is read back with
``get_error()`` / ``get_cl_err()`` and every method returns the status code.
Extensions the reference does not have: batches (leading array axes), many
independent channels per Clpconv object, and ``*_device`` methods that work in
place on device memory (raw pointers or torch tensors) on a given HIP stream.

PyTorch is only used by callers for device memory and streams; nothing here
imports it.
"""
import ctypes as C

import numpy as np

from ._lib import ClError, check, lib

__all__ = ["Clcfft", "Clrfft", "Clpconv", "Cldconv", "ClError", "cl_error_string", "device_count",
           "device_name", "bitrev_table", "twiddle_table", "r2c_twiddle_table", "reorder_device", "PI"]

PI = 3.141592653589793  # cl_fft.h:24
CL_SUCCESS = 0
CL_INVALID_VALUE = -30


def cl_error_string(err):
    """cl_fft::cl_error_string (cl_fft.cpp:298-395)"""
    return lib().clfa_error_string(int(err)).decode()


def device_count():
    """number of devices, as clGetDeviceIDs would report (test_cfft.cpp:31)"""
    n = C.c_int(0)
    lib().clfa_device_count(C.byref(n))
    return n.value


def device_name(device=0):
    """clGetDeviceInfo(CL_DEVICE_NAME) (test_cfft.cpp:37)"""
    buf = C.create_string_buffer(256)
    check(lib().clfa_device_name(device, buf, 256), "device_name")
    return buf.value.decode()


def bitrev_table(n):
    """cl_fft.cpp:96-101"""
    out = np.empty(n, dtype=np.int32)
    check(lib().clfa_bitrev_table(n, out.ctypes.data_as(C.POINTER(C.c_int))), "bitrev_table")
    return out


def twiddle_table(n, forward=True):
    """cl_fft.cpp:86-91"""
    out = np.empty(2 * n, dtype=np.float32)
    check(lib().clfa_twiddle_table(n, int(forward), out.ctypes.data_as(C.POINTER(C.c_float))), "twiddle_table")
    return out.view(np.complex64)


def r2c_twiddle_table(m, forward=True):
    """cl_fft.cpp:233-238"""
    out = np.empty(2 * m, dtype=np.float32)
    check(lib().clfa_r2c_twiddle_table(m, int(forward), out.ctypes.data_as(C.POINTER(C.c_float))),
          "r2c_twiddle_table")
    return out.view(np.complex64)


def _ptr_stream(obj, stream):
    """(device pointer, hip stream handle) from a torch tensor or a raw int pointer"""
    if hasattr(obj, "data_ptr"):
        if not obj.is_contiguous():
            raise ValueError("device tensor must be contiguous")
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(obj.device).cuda_stream
        return obj.data_ptr(), stream
    return int(obj), stream


def _host(a, dtype):
    if not (isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous and a.flags.writeable):
        raise ValueError("expected a writable C-contiguous numpy array of %s" % np.dtype(dtype).name)
    return a


def reorder_device(device, out, inp, n, batch, stream=None):
    """the reference's reorder kernel (cl_fft.cpp:24-27) as an op on device memory"""
    po, stream = _ptr_stream(out, stream)
    pi, _ = _ptr_stream(inp, stream)
    return lib().clfa_reorder_dev(device, po, pi, n, batch, stream)


class _Plan:
    _h = None

    def __del__(self):
        h, self._h = self._h, None
        if h and lib is not None:      # module globals are already gone at interpreter shutdown
            lib().clfa_fft_destroy(h)

    def get_error(self):
        """cl_fft.h:65"""
        return lib().clfa_fft_get_error(self._h)

    def get_log(self):
        """cl_fft.h:69"""
        return lib().clfa_fft_get_log(self._h).decode()

    def workspace_bytes(self):
        return lib().clfa_fft_workspace_bytes(self._h)

    def kernel_name(self):
        return lib().clfa_fft_kernel_name(self._h).decode()

    def alloc_host(self, shape, dtype):
        """extension (clfa_fft_host_alloc): a numpy array over page-locked host memory of the plan; transform() calls on it
        (or on contiguous slices of it) run on that memory directly, without staging copies.  The array must not outlive the
        plan; free_host(array) releases it earlier."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        ptr = C.c_void_p()
        e = lib().clfa_fft_host_alloc(self._h, n * dt.itemsize, C.byref(ptr))
        if e != CL_SUCCESS:
            return None
        buf = (C.c_char * (n * dt.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

    def free_host(self, array):
        return lib().clfa_fft_host_free(self._h, array.ctypes.data)

    def exec_device(self, data, batch, stream=None):
        """in place on device memory, asynchronous on `stream` (Clcfft::fft(), cl_fft.cpp:138-151)"""
        p, stream = _ptr_stream(data, stream)
        return lib().clfa_fft_exec_dev(self._h, p, batch, stream)

    def exec_device_oop(self, src, dst, batch, stream=None):
        """src -> dst on device memory (extension; the reference's device side is out of place too: data1 -> data2,
        cl_fft.cpp:138-151); src is left untouched"""
        ps, stream = _ptr_stream(src, stream)
        pd, _ = _ptr_stream(dst, stream)
        return lib().clfa_fft_exec_dev_oop(self._h, ps, pd, batch, stream)


class Clcfft(_Plan):
    """cl_fft::Clcfft(device_id, size, fwd=true) (cl_fft.h:29-70, cl_fft.cpp:44-161)"""

    def __init__(self, device_id, size, fwd=True):
        self.N = int(size)
        self.forward = bool(fwd)
        h = C.c_void_p()
        lib().clfa_cfft_create(C.byref(h), int(device_id), int(size), int(bool(fwd)))
        self._h = h

    def transform(self, c):
        """in place on complex64[..., N]; leading axes are batches (cl_fft.cpp:153-161)"""
        c = _host(c, np.complex64)
        if c.shape[-1] != self.N:
            return CL_INVALID_VALUE
        return lib().clfa_cfft_transform(self._h, c.ctypes.data, c.size // self.N)


class Clrfft(_Plan):
    """cl_fft::Clrfft(device_id, size, fwd) (cl_fft.h:74-111, cl_fft.cpp:208-296)"""

    def __init__(self, device_id, size, fwd):
        self.size = int(size)
        self.N = self.size // 2          # the inherited member N is size/2 (cl_fft.cpp:210)
        self.forward = bool(fwd)
        h = C.c_void_p()
        lib().clfa_rfft_create(C.byref(h), int(device_id), int(size), int(bool(fwd)))
        self._h = h

    def transform(self, c, r=None):
        """transform(c, r): forward reads r (float32[..., size]) and writes c
        (complex64[..., size/2]); inverse reads c and writes r.  transform(c) is
        the in-place form (cl_fft.h:104-109)."""
        c = _host(c, np.complex64)
        if c.shape[-1] != self.N:
            return CL_INVALID_VALUE
        if r is None:
            rp = c.ctypes.data
        else:
            r = _host(r, np.float32)
            if r.shape[-1] != self.size or r.size // self.size != c.size // self.N:
                return CL_INVALID_VALUE
            rp = r.ctypes.data
        return lib().clfa_rfft_transform(self._h, c.ctypes.data, rp, c.size // self.N)


def bandwidth_probe(device_id=0, nbytes=1 << 30, launches=100):
    """sustained device-memory bandwidth in TB/s: {"read", "write", "copy", "copy_colblock"} — the
    yardsticks for the roofline fractions (clfa_bandwidth_probe in clfft_amd.h)"""
    out = {}
    for what, name in enumerate(("read", "write", "copy", "copy_colblock")):
        v = C.c_double(0.0)
        e = lib().clfa_bandwidth_probe(int(device_id), what, int(nbytes), int(launches), C.byref(v))
        if e != CL_SUCCESS:
            raise RuntimeError("bandwidth probe: " + cl_error_string(e))
        out[name] = v.value
    return out


class Clpconv:
    """cl_conv::Clpconv(device_id, cvs, pts, errs=NULL, uData=NULL, ...) (cl_conv.h:124-188)

    `channels` (extension) runs that many independent instances in one object;
    arrays then carry a leading channel axis."""

    def __init__(self, device_id, cvs, pts, errs=None, uData=None, channels=1):
        self.pts = int(pts)
        self.channels = int(channels)
        self._errs, self._udata = errs, uData
        h = C.c_void_p()
        e = lib().clfa_pconv_create(C.byref(h), int(device_id), int(cvs), int(pts), int(channels))
        self._h = h
        if e != CL_SUCCESS:
            self._report(e)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib().clfa_pconv_destroy(h)

    def _report(self, e):
        # error callback by value, default prints unless user data is given (cl_conv.h:142-145)
        msg = cl_error_string(e)
        if self._errs is not None:
            self._errs(msg, self._udata)
        elif self._udata is None:
            print(msg)

    def cl_error_string(self, err):
        return cl_error_string(err)

    def get_cl_err(self):
        """cl_conv.h:187"""
        return lib().clfa_pconv_get_error(self._h)

    nparts = property(lambda s: lib().clfa_pconv_nparts(s._h))
    wp = property(lambda s: lib().clfa_pconv_wp(s._h))
    wp2 = property(lambda s: lib().clfa_pconv_wp2(s._h))

    def state_bytes(self):
        return lib().clfa_pconv_state_bytes(self._h)

    def kernel_name(self):
        return lib().clfa_pconv_kernel_name(self._h).decode()

    def push_ir(self, ir):
        """cl_conv.cpp:353-388; ir: float32[channels, nparts*pts] (or 1-D for one channel)"""
        ir = np.ascontiguousarray(ir, dtype=np.float32)
        need = self.nparts * self.pts
        if ir.ndim == 1:
            ir = ir[None, :]
        if ir.shape[0] != self.channels or ir.shape[1] < need:
            return CL_INVALID_VALUE
        ir = np.ascontiguousarray(ir[:, :need])
        return lib().clfa_pconv_push_ir(self._h, ir.ctypes.data)

    def convolution(self, output, input1, input2=None):
        """convolution(out, in) (cl_conv.cpp:393-458) or the time-varying
        convolution(out, in1, in2) (cl_conv.cpp:460-548); float32[channels, pts]"""
        output = _host(output, np.float32)
        a = np.ascontiguousarray(input1, dtype=np.float32)
        n = self.channels * self.pts
        if output.size != n or a.size != n:
            return CL_INVALID_VALUE
        if input2 is None:
            return lib().clfa_pconv_convolution(self._h, output.ctypes.data, a.ctypes.data)
        b = np.ascontiguousarray(input2, dtype=np.float32)
        if b.size != n:
            return CL_INVALID_VALUE
        return lib().clfa_pconv_convolution_tv(self._h, output.ctypes.data, a.ctypes.data, b.ctypes.data)

    def push_ir_device(self, ir, stream=None):
        """ir: device tensor (channels, >= nparts*pts) of float32, rows contiguous; a (channels, cvs)
        tensor with cvs not a multiple of pts is fine (the remainder of every row is ignored, like the
        reference's floor(cvs / pts), cl_conv.cpp:143)"""
        need = self.nparts * self.pts
        shape, strides = tuple(ir.shape), tuple(ir.stride())
        if ir.dim() == 1:
            shape, strides = (1,) + shape, (shape[0],) + strides
        if (len(shape) != 2 or shape[0] != self.channels or shape[1] < need or strides[1] != 1
                or (shape[0] > 1 and strides[0] < need) or str(ir.dtype) != "torch.float32"):
            return -30   # CL_INVALID_VALUE
        p, stream = _ptr_stream(ir, stream)
        return lib().clfa_pconv_push_ir_dev(self._h, p, strides[0], stream)

    def process_device(self, out, in1, in2=None, stream=None):
        po, stream = _ptr_stream(out, stream)
        p1, _ = _ptr_stream(in1, stream)
        p2 = _ptr_stream(in2, stream)[0] if in2 is not None else None
        return lib().clfa_pconv_process_dev(self._h, po, p1, p2, stream)


class Cldconv:
    """cl_conv::Cldconv(device_id, cvs, vsize, errs=NULL, uData=NULL) (cl_dconv.h:17-66)"""

    def __init__(self, device_id, cvs, vsize, errs=None, uData=None):
        self.irsize, self.vsize = int(cvs), int(vsize)
        h = C.c_void_p()
        e = lib().clfa_dconv_create(C.byref(h), int(device_id), int(cvs), int(vsize))
        self._h = h
        if e != CL_SUCCESS:
            msg = cl_error_string(e)
            if errs is not None:
                errs(msg, uData)
            elif uData is None:
                print(msg)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib().clfa_dconv_destroy(h)

    def cl_error_string(self, err):
        return cl_error_string(err)

    def get_cl_err(self):
        return lib().clfa_dconv_get_error(self._h)

    def push_ir(self, ir):
        """cl_dconv.cpp:150-153"""
        ir = np.ascontiguousarray(ir, dtype=np.float32)
        if ir.size < self.irsize:
            return CL_INVALID_VALUE
        return lib().clfa_dconv_push_ir(self._h, ir.ctypes.data)

    def convolution(self, out, in1, in2=None):
        """cl_dconv.cpp:109-148"""
        out = _host(out, np.float32)
        a = np.ascontiguousarray(in1, dtype=np.float32)
        if out.size != self.vsize or a.size != self.vsize:
            return CL_INVALID_VALUE
        if in2 is None:
            return lib().clfa_dconv_convolution(self._h, out.ctypes.data, a.ctypes.data)
        b = np.ascontiguousarray(in2, dtype=np.float32)
        if b.size != self.vsize:
            return CL_INVALID_VALUE
        return lib().clfa_dconv_convolution_tv(self._h, out.ctypes.data, a.ctypes.data, b.ctypes.data)

    def process_device(self, out, in1, in2=None, stream=None):
        """device-resident block (extension): vsize float32 each, asynchronous on `stream`; out must not be an input"""
        for t in (out, in1) + ((in2,) if in2 is not None else ()):
            if hasattr(t, "numel") and (t.numel() != self.vsize or not t.is_contiguous() or str(t.dtype) != "torch.float32"):
                return CL_INVALID_VALUE
        po, stream = _ptr_stream(out, stream)
        p1, _ = _ptr_stream(in1, stream)
        p2 = _ptr_stream(in2, stream)[0] if in2 is not None else None
        return lib().clfa_dconv_process_dev(self._h, po, p1, p2, stream)
