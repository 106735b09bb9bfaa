"""Pin the CPU oracle (oracle/clfft_oracle.c) before trusting it.

Golden vectors under tests/golden/ref/ are outputs of the UNMODIFIED reference
classes run on an MI355X through the AMD OpenCL runtime (oracle/ref_driver.cpp,
oracle/Makefile target `ref`).  The GPU OpenCL compiler may contract a*b+c to
fma, the oracle is compiled with -ffp-contract=off, so float results agree to
rounding (tolerance TOL = 1e-6, criterion of SURVEY.md §8d) while integer
tables and double-derived twiddle tables agree bit for bit.
"""
import numpy as np
import pytest

from oracle import oracle
from tests import util
from tests.util import TOL, assert_parity, golden

POW2_SMALL = [1 << k for k in range(1, 13)]     # 2..4096 full vectors
POW2_LARGE = [1 << k for k in range(13, 17)]    # 8192..65536 decimated


# ---- a1/a2/a3: tables, bit-exact -------------------------------------------

@pytest.mark.parametrize("n", [16, 1024, 65536])
def test_bitrev_table_bit_exact(n):
    b = oracle.bitrev_table(n)
    assert np.array_equal(b, golden("g6_bitrev%d" % n))
    # it is the bit-reversal permutation
    lg = n.bit_length() - 1
    ref = np.array([int(format(i, "0%db" % lg)[::-1], 2) for i in range(n)], dtype=np.int32)
    assert np.array_equal(b, ref)


def test_bitrev_n16_literal():
    # SURVEY.md §8a a1
    assert oracle.bitrev_table(16).tolist() == [0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15]


@pytest.mark.parametrize("n", [16, 1024])
def test_twiddle_table_bit_exact(n):
    w = oracle.twiddle_table(n, True)
    assert np.array_equal(w.view(np.uint32), golden("g6_twiddle%d" % n).view(np.uint32))
    wi = oracle.twiddle_table(n, False)
    assert np.array_equal(wi.real, w.real) and np.array_equal(wi.imag, -w.imag)


# ---- known-answer programs of the reference ---------------------------------

def test_kat_test_cfft_n16():
    """test_cfft.cpp:54-56: sin(2 pi i/16) -> spec[1]=(0,-0.5), spec[15]=(0,0.5)"""
    n = 16
    x = np.sin(np.arange(n) * 2 * np.pi / n).astype(np.float32).astype(np.complex64)
    assert np.array_equal(x, golden("g1_cfft16_in"))
    y = oracle.cfft(x, True)
    want = np.zeros(n, np.complex64)
    want[1], want[15] = -0.5j, 0.5j
    assert np.max(np.abs(y - want)) < 1e-7
    assert_parity(y, golden("g1_cfft16_fwd"), what="g1 fwd")
    z = oracle.cfft(y, False)
    assert np.max(np.abs(z - x)) < 2e-7
    assert_parity(z, golden("g1_cfft16_inv"), what="g1 inv")


def test_kat_test_rfft_n16():
    """test_rfft.cpp:54-57: 0.5 + sin + 0.5 cos(pi i) -> (0.5,0.5),(0,-1),0..."""
    n = 16
    i = np.arange(n)
    x = (0.5 + np.sin(i * 2 * np.pi / n) + 0.5 * np.cos(i * np.pi)).astype(np.float32)
    assert np.allclose(x, golden("g2_rfft16_in"), atol=1e-7)
    spec = oracle.rfft_forward(golden("g2_rfft16_in"))
    want = np.zeros(n // 2, np.complex64)
    want[0], want[1] = 0.5 + 0.5j, -1.0j
    assert np.max(np.abs(spec - want)) < 2e-7
    assert_parity(spec, golden("g2_rfft16_fwd"), what="g2 fwd")
    back = oracle.rfft_inverse(spec)
    assert_parity(back, golden("g2_rfft16_inv"), what="g2 inv")
    assert np.max(np.abs(back - x)) < 5e-7


# ---- a4: reorder is an exact gather ------------------------------------------

@pytest.mark.parametrize("n", [2, 16, 1024])
def test_reorder_bit_exact(n):
    x = util.lcg_complex(99, n)
    b = oracle.bitrev_table(n)
    assert np.array_equal(oracle.reorder(x, b).view(np.uint32), x[b].view(np.uint32))


# ---- a5/a6: c2c against the reference's outputs -------------------------------

@pytest.mark.parametrize("n", POW2_SMALL)
def test_cfft_small_vs_reference(n):
    x = util.lcg_complex(12345, n)
    y = oracle.cfft(x, True)
    z = oracle.cfft(x, False)
    assert_parity(y, golden("g3_cfft%d_fwd" % n), what="fwd")
    assert_parity(z, golden("g3_cfft%d_inv" % n), what="inv")
    assert_parity(oracle.cfft(y, False), golden("g3_cfft%d_rt" % n), what="rt")


@pytest.mark.parametrize("n", POW2_LARGE)
def test_cfft_large_vs_reference(n):
    x = util.lcg_complex(12345, n)
    y = oracle.cfft(x, True)
    z = oracle.cfft(x, False)
    assert_parity(util.decimate(y), golden("g4_cfft%d_fwd_dec" % n), what="fwd")
    assert_parity(util.decimate(z), golden("g4_cfft%d_inv_dec" % n), what="inv")
    assert_parity(util.decimate(oracle.cfft(y, False)), golden("g4_cfft%d_rt_dec" % n), what="rt")
    chk = golden("g4_cfft%d_fwd_chk" % n)
    e = float(np.sum(np.abs(y.astype(np.complex128)) ** 2))
    assert abs(e - chk[2]) <= 1e-5 * chk[2]


@pytest.mark.parametrize("n", [16, 1024, 65536])
def test_cfft_vs_float64_dft(n):
    """fact 1 + fact 5 of SURVEY.md §8a: forward = DFT/N, inverse unscaled"""
    x = util.lcg_complex(12345, n)
    X = np.fft.fft(x.astype(np.complex128))
    l2, _ = util.rel_err(oracle.cfft(x, True), X / n)
    assert l2 < 4e-7
    l2, _ = util.rel_err(oracle.cfft(x, False), np.conj(np.fft.fft(np.conj(x.astype(np.complex128)))))
    assert l2 < 4e-7


def test_cfft_batched_equals_sequential():
    x = util.lcg_complex(5, 8 * 256).reshape(8, 256)
    yb = oracle.cfft(x, True)
    for b in range(8):
        assert np.array_equal(yb[b].view(np.uint32), oracle.cfft(x[b], True).view(np.uint32))


def test_cfft_rejects_bad_sizes():
    for n in (0, 1, 3, 12, 1000):
        with pytest.raises(ValueError):
            oracle.cfft(np.zeros(max(n, 1), np.complex64), True)


# ---- a7/a8/a9: r2c / c2r --------------------------------------------------------

RSIZES_SMALL = [1 << k for k in range(2, 13)]
RSIZES_LARGE = [1 << k for k in range(13, 18)]


@pytest.mark.parametrize("size", RSIZES_SMALL)
def test_rfft_small_vs_reference(size):
    x = util.lcg_sym(12345, size)
    spec = oracle.rfft_forward(x)
    assert_parity(spec, golden("g5_rfft%d_fwd" % size), what="fwd")
    assert_parity(oracle.rfft_inverse(spec), golden("g5_rfft%d_rt" % size), what="rt")
    arb = util.lcg_complex(777, size // 2)
    assert_parity(oracle.rfft_inverse(arb), golden("g5_rfft%d_invarb" % size), what="invarb")


@pytest.mark.parametrize("size", RSIZES_LARGE)
def test_rfft_large_vs_reference(size):
    x = util.lcg_sym(12345, size)
    spec = oracle.rfft_forward(x)
    m = size // 2
    got = np.concatenate([util.decimate(spec), spec[m // 2:m // 2 + 1]])
    assert_parity(got, golden("g5_rfft%d_fwd_dec" % size), what="fwd")
    rt = oracle.rfft_inverse(spec)
    assert_parity(util.decimate(rt.view(np.complex64)), golden("g5_rfft%d_rt_dec" % size), what="rt")
    arb = util.lcg_complex(777, m)
    assert_parity(util.decimate(oracle.rfft_inverse(arb).view(np.complex64)),
                  golden("g5_rfft%d_invarb_dec" % size), what="invarb")


@pytest.mark.parametrize("size", [16, 1024, 16384])
def test_rfft_packing_quirks(size):
    """fact 2 of SURVEY.md §8a: spec[0]=(X0/size, X[size/2]/size); spec[k]=2X[k]/size;
    spec[M/2] = conj(2 X[M/2]/size) (the pack kernel never visits the self-paired bin)."""
    x = util.lcg_sym(12345, size)
    m = size // 2
    X = np.fft.fft(x.astype(np.float64))
    want = 2 * X[:m] / size
    want[0] = complex(X[0].real / size, X[m].real / size)
    want[m // 2] = np.conj(want[m // 2])
    l2, mx = util.rel_err(oracle.rfft_forward(x), want)
    assert l2 < 5e-7 and mx < 5e-7
    # and the round trip is the identity
    l2, mx = util.rel_err(oracle.rfft_inverse(oracle.rfft_forward(x)), x)
    assert l2 < 5e-7


# ---- a10-a13: partitioned convolution ---------------------------------------------

def _run_pconv(pts, nparts, blocks, ir, inp):
    p = oracle.Pconv(pts * nparts, pts)
    p.push_ir(ir)
    return np.concatenate([p.convolution(inp[b * pts:(b + 1) * pts]) for b in range(blocks)])


@pytest.mark.parametrize("tag,pts,nparts,blocks", [
    ("g7_pconv_p8_n4", 8, 4, 12), ("g7_pconv_p8_n4_ones", 8, 4, 12),
    ("g7_pconv_p2_n3", 2, 3, 9), ("g7_pconv_p64_n1", 64, 1, 4),
    ("g8_pconv_p1024_n8", 1024, 8, 24)])
def test_pconv_vs_reference(tag, pts, nparts, blocks):
    ir, inp = golden(tag + "_ir"), golden(tag + "_in")
    out = _run_pconv(pts, nparts, blocks, ir, inp)
    assert_parity(out, golden(tag + "_out"), what=tag)


def test_pconv_fixture_inputs_are_the_lcg():
    ir, inp = golden("g7_pconv_p8_n4_ir"), golden("g7_pconv_p8_n4_in")
    s = util.lcg_half(7, ir.size + inp.size)
    assert np.array_equal(ir, s[:ir.size]) and np.array_equal(inp, s[ir.size:])


def test_pconv_dc_nyquist_half_gain():
    """fact 3 of SURVEY.md §8a: all-ones through an IR summing to 1 settles at 0.5"""
    out = golden("g7_pconv_p8_n4_ones_out")
    assert np.allclose(out[-8:], 0.5, atol=1e-6)
    mine = _run_pconv(8, 4, 12, golden("g7_pconv_p8_n4_ones_ir"), golden("g7_pconv_p8_n4_ones_in"))
    assert np.allclose(mine[-8:], 0.5, atol=1e-6)


def _pconv_model(ir, x, pts, nparts, blocks):
    """closed form: textbook overlap-add with Y[0]*=0.5, Y[pts]*=0.5 per block product"""
    H = [np.fft.rfft(np.concatenate([ir[i * pts:(i + 1) * pts], np.zeros(pts)])) for i in range(nparts)]
    Xs, tail, out = [], np.zeros(pts), []
    for t in range(blocks):
        Xs.append(np.fft.rfft(np.concatenate([x[t * pts:(t + 1) * pts], np.zeros(pts)])))
        Y = sum(Xs[t - a] * H[a] for a in range(nparts) if t - a >= 0)
        Y[0] *= 0.5
        Y[pts] *= 0.5
        y = np.fft.irfft(Y, 2 * pts)
        out.append(y[:pts] + tail)
        tail = y[pts:]
    return np.concatenate(out)


@pytest.mark.parametrize("pts,nparts,blocks", [(8, 4, 12), (64, 3, 10), (1024, 8, 24)])
def test_pconv_closed_form_model(pts, nparts, blocks):
    s = util.lcg_half(7, pts * nparts + pts * blocks).astype(np.float64)
    ir, x = s[:pts * nparts], s[pts * nparts:]
    out = _run_pconv(pts, nparts, blocks, ir, x)
    l2, mx = util.rel_err(out, _pconv_model(ir, x, pts, nparts, blocks))
    assert l2 < 2e-6 and mx < 2e-6


def test_pconv_ring_indices_bit_exact():
    """a13: wp increments mod nparts, wp2 decrements; push_ir leaves wp2 = nparts-1"""
    p = oracle.Pconv(32, 8)
    assert (p.nparts, p.wp, p.wp2) == (4, 0, 3)
    p.push_ir(np.zeros(32, np.float32))
    assert p.wp2 == 3
    seq = []
    for _ in range(9):
        p.convolution(np.zeros(8, np.float32))
        seq.append(p.wp)
    assert seq == [1, 2, 3, 0, 1, 2, 3, 0, 1]
    q = oracle.Pconv(32, 8)
    seq = []
    for _ in range(6):
        q.convolution(np.zeros(8, np.float32), np.zeros(8, np.float32))
        seq.append((q.wp, q.wp2))
    assert seq == [(1, 2), (2, 1), (3, 0), (0, 3), (1, 2), (2, 1)]


def test_pconv_nparts_floor():
    """cl_conv.cpp:143: remainder samples of the IR are dropped"""
    assert oracle.Pconv(96000, 1024).nparts == 93
    assert oracle.Pconv(96256, 1024).nparts == 94


@pytest.mark.parametrize("tag,pts,nparts,blocks", [("g9_tvconv_p8_n4", 8, 4, 12), ("g9_tvconv_p256_n5", 256, 5, 14)])
def test_tvconv_vs_reference(tag, pts, nparts, blocks):
    in1, in2 = golden(tag + "_in1"), golden(tag + "_in2")
    p = oracle.Pconv(pts * nparts, pts)
    out = np.concatenate([p.convolution(in1[b * pts:(b + 1) * pts], in2[b * pts:(b + 1) * pts])
                          for b in range(blocks)])
    assert_parity(out, golden(tag + "_out"), what=tag)


def test_tvconv_closed_form_model():
    """fact 4 of SURVEY.md §8a: X1[T-a] * X2[t'(a)], t'(a) = T - ((T-a) mod nparts)"""
    pts, nparts, blocks = 8, 4, 12
    s = util.lcg_half(7, 2 * pts * blocks).astype(np.float64)
    x1, x2 = s[:pts * blocks], s[pts * blocks:]
    F = lambda v, t: np.fft.rfft(np.concatenate([v[t * pts:(t + 1) * pts], np.zeros(pts)]))
    tail, want = np.zeros(pts), []
    for T in range(blocks):
        Y = np.zeros(pts + 1, complex)
        for a in range(nparts):
            if T - a < 0:
                continue
            Y += F(x1, T - a) * F(x2, T - ((T - a) % nparts))
        Y[0] *= 0.5
        Y[pts] *= 0.5
        y = np.fft.irfft(Y, 2 * pts)
        want.append(y[:pts] + tail)
        tail = y[pts:]
    p = oracle.Pconv(pts * nparts, pts)
    out = np.concatenate([p.convolution(x1[b * pts:(b + 1) * pts], x2[b * pts:(b + 1) * pts])
                          for b in range(blocks)])
    l2, mx = util.rel_err(out, np.concatenate(want))
    assert l2 < 2e-6 and mx < 2e-6


# ---- a14: direct convolution ----------------------------------------------------------

def test_dconv_vs_reference_first_blocks():
    ir, inp = golden("g10_dconv_ir"), golden("g10_dconv_in")
    d = oracle.Dconv(16, 8)
    d.push_ir(ir)
    out = np.concatenate([d.convolution(inp[b * 8:(b + 1) * 8]) for b in range(2)])
    ref = golden("g10_dconv_out")
    # The reference's device buffers are uninitialised (cl_dconv.cpp:87-91);
    # the fixture is only meaningful if that memory happened to be zero.
    if not np.all(np.isfinite(ref)) or np.max(np.abs(ref)) > 1e3:
        pytest.skip("reference fixture polluted by uninitialised device memory")
    assert_parity(out, ref, tol=2e-6, what="dconv")


G11 = [("g11_dconv_i16_v8", 16, 8, False), ("g11_dconv_i1024_v64", 1024, 64, False), ("g11_dconv_i64_v64", 64, 64, False),
       ("g11_tvdconv_i16_v8", 16, 8, True), ("g11_tvdconv_i256_v32", 256, 32, True)]


def dconv_tol(irsize):
    """The reference sums its irsize taps by float CAS atomics in arbitrary order (cl_dconv.cpp:17-31,42):
    its own result moves by ~sqrt(irsize) ulp between runs, so the bar is 2 sqrt(irsize) ulp
    (1e-6 up to 64 taps)."""
    return max(1e-6, 2.0 * np.sqrt(irsize) * 2.0 ** -24)


@pytest.mark.parametrize("tag,irsize,vsize,tv", G11)
def test_dconv_vs_reference_over_ring_cycles(tag, irsize, vsize, tv):
    """G11: the unmodified reference with irsize % vsize == 0 (its defective wrap branch never runs) over
    more than three ring cycles; from block irsize / vsize + 1 on nothing depends on the uninitialised
    device memory of cl_dconv.cpp:87-91 any more, and the oracle must agree block for block"""
    ir, x, ref = golden(tag + "_ir"), golden(tag + "_in"), golden(tag + "_out")
    x2 = golden(tag + "_in2") if tv else None
    d = oracle.Dconv(irsize, vsize)
    d.push_ir(ir)
    first, blocks = irsize // vsize + 1, x.size // vsize
    assert blocks >= 3 * first or irsize == 1024
    for b in range(blocks):
        sl = slice(b * vsize, (b + 1) * vsize)
        out = d.convolution(x[sl], x2[sl]) if tv else d.convolution(x[sl])
        if b >= first:
            assert_parity(out, ref[sl], tol=dconv_tol(irsize), what="%s block %d" % (tag, b))


def test_dconv_is_fir_with_one_sample_latency():
    """kernel cl_dconv.cpp:32-43: y[n] = sum_k coefs[k] x[n-1-k] (ring read point)"""
    irsize, vsize, blocks = 16, 8, 9     # crosses the ring wrap several times
    s = util.lcg_half(3, irsize + vsize * blocks)
    ir, x = s[:irsize], s[irsize:]
    d = oracle.Dconv(irsize, vsize)
    d.push_ir(ir)
    out = np.concatenate([d.convolution(x[b * vsize:(b + 1) * vsize]) for b in range(blocks)])
    full = np.convolve(x.astype(np.float64), ir.astype(np.float64))
    want = np.concatenate([[0.0], full])[:out.size]
    assert np.max(np.abs(out - want)) < 1e-6
