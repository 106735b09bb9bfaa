"""Why the convolution tolerances are what they are: HIP and oracle against FLOAT64 truth.

The reference sums the partitions' products (cl_conv_kernels.h:102-118) and the taps (cl_dconv.cpp:32-43) with CAS-loop
float atomics, i.e. in no fixed order; the oracle restates ONE order (ascending), the HIP kernels another (registers /
chunks in ascending order).  Two float32 sums of the same terms in different orders differ by their rounding, which grows
with the number of terms — so "HIP vs oracle <= tol" alone does not say which of the two is off.  These tests measure
both against a float64 evaluation of the same formulas and require the HIP result to be AT LEAST AS ACCURATE as the
oracle's (within 20 %); the tolerances of tests/test_gpu_conv.py are set from the figures printed here
(profiles/conv_accuracy_r05.txt)."""
import numpy as np
import pytest

import opencl_fft_amd as fa
from oracle import oracle
from tests import util
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _report(what, hip, orc, truth):
    eh, eo, d = rel_err(hip, truth), rel_err(orc, truth), rel_err(hip, orc)
    print("ACCURACY %-44s HIP vs f64 relL2 %.3g max %.3g | oracle vs f64 relL2 %.3g max %.3g | HIP vs oracle relL2 %.3g max %.3g"
          % (what, eh[0], eh[1], eo[0], eo[1], d[0], d[1]))
    return eh, eo, d


def test_pconv_config4_hip_at_least_as_accurate_as_oracle():
    """config 4's geometry (pts 1024, 94 partitions), three channels, 110 blocks (the ring wraps), static response"""
    pts, nparts, channels, blocks = 1024, 94, 3, 110
    cvs = pts * nparts
    rng = np.random.default_rng(5)
    ir = ((rng.random((channels, cvs), dtype=np.float32) - 0.5) / np.float32(np.sqrt(cvs))).astype(np.float32)
    x = (rng.random((channels, blocks * pts), dtype=np.float32) * 2 - 1).astype(np.float32)
    p = fa.Clpconv(0, cvs, pts, channels=channels)
    assert p.get_cl_err() == 0 and p.push_ir(ir) == 0
    got = np.zeros((channels, blocks * pts), np.float32)
    out = np.zeros((channels, pts), np.float32)
    for b in range(blocks):
        assert p.convolution(out, np.ascontiguousarray(x[:, b * pts:(b + 1) * pts])) == 0
        got[:, b * pts:(b + 1) * pts] = out
    for c in range(channels):
        o = oracle.Pconv(cvs, pts)
        o.push_ir(ir[c])
        want = np.concatenate([o.convolution(x[c, b * pts:(b + 1) * pts]) for b in range(blocks)])
        truth = util.pconv_f64(ir[c], x[c], pts)
        eh, eo, d = _report("pconv 1024 x 94, channel %d" % c, got[c], want, truth)
        assert eh[0] <= 1.2 * eo[0] + 1e-9 and eh[1] <= 1.2 * eo[1] + 1e-9, (eh, eo)
        assert d[0] <= 1e-6 and d[1] <= 1e-6, d


def test_pconv_time_varying_hip_at_least_as_accurate_as_oracle():
    """the second input ring (cl_conv.cpp:460-548), pts 256, 16 partitions, 40 blocks"""
    pts, nparts, blocks = 256, 16, 40
    rng = np.random.default_rng(6)
    x1 = (rng.random(blocks * pts, dtype=np.float32) * 2 - 1).astype(np.float32)
    x2 = ((rng.random(blocks * pts, dtype=np.float32) - 0.5) / np.float32(np.sqrt(pts * nparts))).astype(np.float32)
    p, o = fa.Clpconv(0, pts * nparts, pts), oracle.Pconv(pts * nparts, pts)
    got, want = [], []
    out = np.zeros((1, pts), np.float32)
    for b in range(blocks):
        sl = slice(b * pts, (b + 1) * pts)
        assert p.convolution(out, x1[sl], x2[sl]) == 0
        got.append(out[0].copy())
        want.append(o.convolution(x1[sl], x2[sl]))
    truth = util.pconv_tv_f64(x1, x2, pts, nparts)
    eh, eo, d = _report("tv pconv 256 x 16", np.concatenate(got), np.concatenate(want), truth)
    assert eh[0] <= 1.2 * eo[0] + 1e-9 and eh[1] <= 1.2 * eo[1] + 1e-9, (eh, eo)
    assert d[0] <= 1e-6 and d[1] <= 1e-6, d


@pytest.mark.parametrize("irsize,vsize", [(1024, 64), (96000, 500), (1 << 20, 1024)])
def test_dconv_hip_at_least_as_accurate_as_oracle(irsize, vsize):
    """Cldconv (cl_dconv.cpp:32-43, 109-132): the last block after the delay line has filled, every output (64 of them at
    irsize 2^20) against float64 and against the oracle's arithmetic — float32 products added one by one in tap order, which
    is what oracle/clfft_oracle.c does (checked against it at the small size)."""
    blocks = irsize // vsize + 3
    rng = np.random.default_rng(irsize)
    ir = ((rng.random(irsize, dtype=np.float32) - 0.5) / np.float32(np.sqrt(irsize))).astype(np.float32)
    x = (rng.random(blocks * vsize, dtype=np.float32) * 2 - 1).astype(np.float32)
    d = fa.Cldconv(0, irsize, vsize)
    assert d.get_cl_err() == 0 and d.push_ir(ir) == 0
    out = np.zeros(vsize, np.float32)
    for b in range(blocks):
        assert d.convolution(out, x[b * vsize:(b + 1) * vsize]) == 0
    pick = np.arange(vsize) if irsize < (1 << 20) else np.arange(0, vsize, vsize // 64)
    truth, seq = util.dconv_last_block(ir, x, vsize, blocks - 1, pick)
    if irsize == 1024:   # the numpy restatement of the oracle's order IS the oracle
        o = oracle.Dconv(irsize, vsize)
        o.push_ir(ir)
        for b in range(blocks):
            w = o.convolution(x[b * vsize:(b + 1) * vsize])
        assert np.array_equal(w[pick], seq)
    eh, eo, df = _report("dconv irsize %d" % irsize, out[pick], seq, truth)
    assert eh[0] <= 1.2 * eo[0] + 1e-9 and eh[1] <= 1.2 * eo[1] + 1e-9, (eh, eo)
    # the bound tests/test_gpu_conv.py uses for HIP vs oracle: two float32 sums of irsize terms in different orders
    bound = max(1e-6, 2 * float(np.sqrt(irsize)) * 2.0 ** -24)
    assert df[0] <= bound and df[1] <= bound, (df, bound)
