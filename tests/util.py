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
