"""ctypes front end of oracle/liboracle.so (CPU restatement of the reference).

TEST INFRASTRUCTURE.  Import only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this.
Each wrapper names the reference function it restates (file:line relative to
/root/reference); the C side carries the per-statement citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        L.orc_bitrev_table.argtypes = [C.c_int, ip]
        L.orc_twiddle_table.argtypes = [C.c_int, C.c_int, fp]
        L.orc_r2c_twiddle_table.argtypes = [C.c_int, C.c_int, fp]
        L.orc_reorder.argtypes = [fp, fp, ip, C.c_int]
        L.orc_cfft.argtypes = [fp, C.c_int, C.c_int]
        L.orc_rfft.argtypes = [fp, C.c_int, C.c_int]
        L.orc_cfft_batched.argtypes = [fp, C.c_int, C.c_long, C.c_int, C.c_int]
        L.orc_rfft_batched.argtypes = [fp, C.c_int, C.c_long, C.c_int, C.c_int]
        L.orc_pconv_create.restype = C.c_void_p
        L.orc_pconv_create.argtypes = [C.c_int, C.c_int]
        L.orc_pconv_destroy.argtypes = [C.c_void_p]
        L.orc_pconv_push_ir.argtypes = [C.c_void_p, fp]
        L.orc_pconv_convolution.argtypes = [C.c_void_p, fp, fp]
        L.orc_pconv_convolution_tv.argtypes = [C.c_void_p, fp, fp, fp]
        for f in (L.orc_pconv_wp, L.orc_pconv_wp2, L.orc_pconv_nparts):
            f.argtypes = [C.c_void_p]
        L.orc_dconv_create.restype = C.c_void_p
        L.orc_dconv_create.argtypes = [C.c_int, C.c_int]
        L.orc_dconv_destroy.argtypes = [C.c_void_p]
        L.orc_dconv_push_ir.argtypes = [C.c_void_p, fp]
        L.orc_dconv_convolution.argtypes = [C.c_void_p, fp, fp]
        L.orc_dconv_convolution_tv.argtypes = [C.c_void_p, fp, fp, fp]
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def bitrev_table(n):
    """cl_fft.cpp:96-101"""
    b = np.empty(n, dtype=np.int32)
    lib().orc_bitrev_table(n, b.ctypes.data_as(C.POINTER(C.c_int)))
    return b


def twiddle_table(n, forward=True):
    """cl_fft.cpp:86-91 -> complex64[n]"""
    w = np.empty(2 * n, dtype=np.float32)
    lib().orc_twiddle_table(n, int(forward), _fp(w))
    return w.view(np.complex64)


def r2c_twiddle_table(m, forward=True):
    """cl_fft.cpp:233-238 -> complex64[m]"""
    w = np.empty(2 * m, dtype=np.float32)
    lib().orc_r2c_twiddle_table(m, int(forward), _fp(w))
    return w.view(np.complex64)


def reorder(x, b):
    """cl_fft.cpp:24-27: out[k] = in[b[k]] on complex64"""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    b = np.ascontiguousarray(b, dtype=np.int32)
    out = np.empty_like(x)
    lib().orc_reorder(_fp(out.view(np.float32)), _fp(x.view(np.float32)),
                      b.ctypes.data_as(C.POINTER(C.c_int)), x.shape[-1])
    return out


def cfft(x, forward=True, nthreads=0):
    """Clcfft::transform (cl_fft.cpp:153-161) on the last axis; leading axes are batches."""
    x = np.array(x, dtype=np.complex64, order="C", copy=True)
    n = x.shape[-1]
    batch = x.size // n
    e = lib().orc_cfft_batched(_fp(x.view(np.float32)), n, batch, int(forward), nthreads)
    if e:
        raise ValueError("orc_cfft_batched error %d" % e)
    return x


def rfft_forward(x, nthreads=0):
    """Clrfft::transform forward (cl_fft.cpp:272-282): real[..., size] -> packed complex64[..., size/2]"""
    x = np.array(x, dtype=np.float32, order="C", copy=True)
    size = x.shape[-1]
    e = lib().orc_rfft_batched(_fp(x), size, x.size // size, 1, nthreads)
    if e:
        raise ValueError("orc_rfft_batched error %d" % e)
    return x.view(np.complex64)


def rfft_inverse(c, nthreads=0):
    """Clrfft::transform inverse (cl_fft.cpp:283-294): packed complex64[..., M] -> real[..., 2M]"""
    c = np.array(c, dtype=np.complex64, order="C", copy=True)
    m = c.shape[-1]
    r = c.view(np.float32)
    e = lib().orc_rfft_batched(_fp(r), 2 * m, c.size // m, 0, nthreads)
    if e:
        raise ValueError("orc_rfft_batched error %d" % e)
    return r


def num_threads():
    return lib().orc_num_threads()


class Pconv:
    """cl_conv::Clpconv (cl_conv.cpp:140-548)"""

    def __init__(self, cvs, pts):
        self.h = lib().orc_pconv_create(cvs, pts)
        if not self.h:
            raise ValueError("bad Clpconv geometry")
        self.pts = pts

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pconv_destroy(self.h)
            self.h = None

    nparts = property(lambda s: lib().orc_pconv_nparts(s.h))
    wp = property(lambda s: lib().orc_pconv_wp(s.h))
    wp2 = property(lambda s: lib().orc_pconv_wp2(s.h))

    def push_ir(self, ir):
        ir = _f32(ir)
        assert ir.size >= self.nparts * self.pts
        return lib().orc_pconv_push_ir(self.h, _fp(ir))

    def convolution(self, inp, in2=None):
        inp = _f32(inp)
        out = np.empty(self.pts, dtype=np.float32)
        if in2 is None:
            lib().orc_pconv_convolution(self.h, _fp(out), _fp(inp))
        else:
            in2 = _f32(in2)
            lib().orc_pconv_convolution_tv(self.h, _fp(out), _fp(inp), _fp(in2))
        return out


class Dconv:
    """cl_conv::Cldconv (cl_dconv.cpp:46-153)"""

    def __init__(self, irsize, vsize):
        self.h = lib().orc_dconv_create(irsize, vsize)
        if not self.h:
            raise ValueError("bad Cldconv geometry")
        self.vsize = vsize

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_dconv_destroy(self.h)
            self.h = None

    def push_ir(self, ir):
        return lib().orc_dconv_push_ir(self.h, _fp(_f32(ir)))

    def convolution(self, inp, in2=None):
        inp = _f32(inp)
        out = np.empty(self.vsize, dtype=np.float32)
        if in2 is None:
            lib().orc_dconv_convolution(self.h, _fp(out), _fp(inp))
        else:
            lib().orc_dconv_convolution_tv(self.h, _fp(out), _fp(inp), _fp(_f32(in2)))
        return out
