"""Shared test helpers: golden-vector loader, fixture PRNG, parity metric."""
import json
import os

import numpy as np

GOLDEN_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref")
_MANIFEST = None


def manifest():
    global _MANIFEST
    if _MANIFEST is None:
        with open(os.path.join(GOLDEN_REF, "manifest.json")) as f:
            _MANIFEST = json.load(f)
    return _MANIFEST


def golden(name):
    """Load one vector written by oracle/ref_driver.cpp (the unmodified
    reference run on the MI355X through OpenCL). (..., 2) float32 -> complex64."""
    m = manifest()[name]
    dt = {"f32": np.float32, "f64": np.float64, "i32": np.int32}[m["dtype"]]
    a = np.fromfile(os.path.join(GOLDEN_REF, name + ".bin"), dtype=dt).reshape(m["shape"])
    if dt is np.float32 and a.ndim == 2 and a.shape[1] == 2:
        a = np.ascontiguousarray(a).view(np.complex64).reshape(-1)
    return a


def lcg_u32(seed, count):
    """fixture PRNG of SURVEY.md §8c: s = s*1664525 + 1013904223 mod 2^32"""
    out = np.empty(count, dtype=np.uint32)
    s = np.uint64(seed)
    a, c, mask = np.uint64(1664525), np.uint64(1013904223), np.uint64(0xFFFFFFFF)
    for i in range(count):
        s = (s * a + c) & mask
        out[i] = s
    return out


def lcg_sym(seed, count):
    """uniform [-1,1) float32: (s>>8)/2^23 - 1"""
    return ((lcg_u32(seed, count) >> 8).astype(np.float32) / np.float32(8388608.0) - np.float32(1.0)).astype(np.float32)


def lcg_half(seed, count):
    """uniform [-0.5,0.5) float32: (s>>8)/2^24 - 0.5"""
    return ((lcg_u32(seed, count) >> 8).astype(np.float32) / np.float32(16777216.0) - np.float32(0.5)).astype(np.float32)


def lcg_complex(seed, n):
    """re then im per element (ref_driver.cpp G3)"""
    return lcg_sym(seed, 2 * n).view(np.complex64)


def decimate(v):
    """first 64, last 64, every 16th (ref_driver.cpp decimate())"""
    v = np.asarray(v).reshape(-1)
    return np.concatenate([v[:64], v[-64:], v[::16]])


# parity criterion of SURVEY.md §8d ("within 1e-6 relative"):
#   ||y - ref||2 / ||ref||2 <= tol  AND  max|y - ref| / max|ref| <= tol
TOL = 1e-6


def rel_err(y, ref):
    y = np.asarray(y).reshape(-1)
    ref = np.asarray(ref).reshape(-1)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    d = (y.astype(np.complex128) - ref.astype(np.complex128))
    nrm = np.linalg.norm(ref.astype(np.complex128))
    mx = np.max(np.abs(ref))
    l2 = np.linalg.norm(d) / nrm if nrm > 0 else np.linalg.norm(d)
    mxe = np.max(np.abs(d)) / mx if mx > 0 else np.max(np.abs(d))
    return float(l2), float(mxe)


def assert_parity(y, ref, tol=TOL, what=""):
    l2, mx = rel_err(y, ref)
    assert l2 <= tol and mx <= tol, "%s: relL2=%.3g max/max=%.3g (tol %.1g)" % (what, l2, mx, tol)
    return l2, mx


# ---- float64 evaluations of the reference's convolution formulas (tests/test_gpu_conv_accuracy.py) ------------------
def _block_spectra64(x, pts):
    """rfft of every pts-sample block zero-padded to 2 pts (float64): the spectra the reference's forward chain produces
    (cl_conv.cpp:399-419), up to its packing of bins 0 and pts"""
    blocks = x.size // pts
    z = np.zeros((blocks, 2 * pts), np.float64)
    z[:, :pts] = np.asarray(x, np.float64).reshape(blocks, pts)
    return np.fft.rfft(z, axis=1)


def _olap64(Y, pts):
    """c2r + inverse transform + overlap-add (cl_conv_kernels.h:87-100, 120-124) of block spectra Y; the packed bin 0 of the
    reference carries HALF the DC / Nyquist values on both operands of the product, one of which its inverse map undoes:
    bins 0 and pts of every product have gain 1/2"""
    Y = Y.copy()
    Y[:, 0] *= 0.5
    Y[:, pts] *= 0.5
    y = np.fft.irfft(Y, n=2 * pts, axis=1)
    out = y[:, :pts].copy()
    out[1:] += y[:-1, pts:]
    return out.reshape(-1)


def pconv_f64(ir, x, pts):
    """Clpconv::push_ir + convolution(out, in) (cl_conv.cpp:353-458) in float64: block t = sum over partitions p of
    X[t - p] H[p], overlap-added"""
    X, H = _block_spectra64(x, pts), _block_spectra64(ir[:(ir.size // pts) * pts], pts)
    Y = np.zeros_like(X)
    for p in range(H.shape[0]):
        Y[p:] += X[:X.shape[0] - p] * H[p] if p else X * H[0]
    return _olap64(Y, pts)


def pconv_tv_f64(x1, x2, pts, nparts):
    """Clpconv::convolution(out, in1, in2) (cl_conv.cpp:460-548) in float64: block T = sum over a < nparts of
    X1[T - a] X2[t'(a)], t'(a) = the most recent block index <= T congruent to a mod nparts — the second input's block t
    overwrites "partition t mod nparts" """
    X1, X2 = _block_spectra64(x1, pts), _block_spectra64(x2, pts)
    Y = np.zeros_like(X1)
    for T in range(X1.shape[0]):
        for a in range(min(nparts, T + 1)):
            tp = T - ((T - a) % nparts)
            if tp >= 0:
                Y[T] += X1[T - a] * X2[tp]
    return _olap64(Y, pts)


def dconv_last_block(ir, x, vsize, b, pick):
    """Cldconv::convolution (cl_dconv.cpp:32-43, 109-132), outputs `pick` of block b (the delay line full): out[i] =
    sum_k ir[k] x[i - 1 - k].  Returns (float64 value, float32 products added one by one in the kernel's tap order h =
    irsize - 1 - k ascending — the oracle's arithmetic, oracle/clfft_oracle.c orc_dconv_convolution)"""
    irsize = ir.size
    rev = np.ascontiguousarray(ir[::-1])
    truth, seq = np.empty(len(pick), np.float64), np.empty(len(pick), np.float32)
    for j, n in enumerate(pick):
        i0 = b * vsize - irsize + int(n)
        assert i0 >= 0
        seg = x[i0:i0 + irsize]
        truth[j] = np.dot(seg.astype(np.float64), rev.astype(np.float64))
        seq[j] = np.cumsum(seg * rev, dtype=np.float32)[-1]
    return truth, seq
