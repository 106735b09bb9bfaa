"""GPU parity tests for the FFT path: HIP kernels (through the C ABI of
libclfft_amd.so) against the CPU oracle and the reference's golden vectors.

Bar: integer/index work bit-exact; float32 spectra within
TOL = 1e-6 in both ||y-ref||2/||ref||2 and max|y-ref|/max|ref|.
"""
import os

import numpy as np
import pytest

import opencl_fft_amd as fa
from oracle import oracle
from tests import util
from tests.util import TOL, assert_parity, golden

pytestmark = pytest.mark.gpu

CSIZES = [1 << k for k in range(1, 17)]
RSIZES = [1 << k for k in range(2, 18)]


def _c2c(n, fwd, x):
    plan = fa.Clcfft(0, n, fwd)
    assert plan.get_error() == 0, plan.get_log()
    y = np.array(x, dtype=np.complex64, copy=True)
    assert plan.transform(y) == 0
    return y


def test_device_visible():
    assert fa.device_count() >= 1
    assert fa.device_name(0)


# ---- tables (host side of the ABI, bit-exact) ---------------------------------

@pytest.mark.parametrize("n", [16, 1024, 65536])
def test_tables_bit_exact(n):
    assert np.array_equal(fa.bitrev_table(n), oracle.bitrev_table(n))
    assert np.array_equal(fa.bitrev_table(n), golden("g6_bitrev%d" % n))
    for fwd in (True, False):
        assert np.array_equal(fa.twiddle_table(n, fwd).view(np.uint32), oracle.twiddle_table(n, fwd).view(np.uint32))
        assert np.array_equal(fa.r2c_twiddle_table(n, fwd).view(np.uint32),
                              oracle.r2c_twiddle_table(n, fwd).view(np.uint32))


# ---- a4: reorder kernel, bit-exact ------------------------------------------------

@pytest.mark.parametrize("n,batch", [(2, 3), (16, 5), (1024, 7), (65536, 3)])
def test_reorder_bit_exact(n, batch):
    import torch
    x = util.lcg_complex(99, n * batch).reshape(batch, n)
    d_in = torch.from_numpy(x.view(np.float32).copy()).cuda()
    d_out = torch.zeros_like(d_in)
    assert fa.reorder_device(0, d_out, d_in, n, batch) == 0
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(np.complex64).reshape(batch, n)
    b = oracle.bitrev_table(n)
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(x[:, b]).view(np.uint32))


# ---- a5/a6: c2c -------------------------------------------------------------------------

def test_kat_test_cfft_n16():
    x = golden("g1_cfft16_in")
    y = _c2c(16, True, x)
    want = np.zeros(16, np.complex64)
    want[1], want[15] = -0.5j, 0.5j
    assert np.max(np.abs(y - want)) < 1e-7
    assert_parity(_c2c(16, False, y), golden("g1_cfft16_inv"), what="inv")


@pytest.mark.parametrize("n", CSIZES)
def test_cfft_vs_oracle_and_reference(n):
    x = util.lcg_complex(12345, n)
    for fwd in (True, False):
        y = _c2c(n, fwd, x)
        assert_parity(y, oracle.cfft(x, fwd), what="n=%d fwd=%s vs oracle" % (n, fwd))
        tag = "fwd" if fwd else "inv"
        if n <= 4096:
            assert_parity(y, golden("g3_cfft%d_%s" % (n, tag)), what="vs reference")
        else:
            assert_parity(util.decimate(y), golden("g4_cfft%d_%s_dec" % (n, tag)), what="vs reference")


@pytest.mark.parametrize("n,batch", [(2, 1000), (8, 333), (64, 37), (256, 17), (1024, 9), (4096, 5), (8192, 3),
                                     (16384, 3), (32768, 2), (65536, 3)])
def test_cfft_batched_ragged(n, batch):
    """batch counts that do not divide the transforms-per-workgroup packing"""
    x = util.lcg_complex(4242 + n, n * batch).reshape(batch, n)
    for fwd in (True, False):
        y = _c2c(n, fwd, x)
        assert_parity(y, oracle.cfft(x, fwd), what="n=%d batch=%d" % (n, batch))


@pytest.mark.parametrize("n", [2, 4])
@pytest.mark.parametrize("batch", [1, 2, 3, 127, 255, 257, 1025, 100003, 2099201])
def test_cfft_tiny_kernel_ragged(n, batch):
    """k_fft_tiny (n = 2 in one lane, n = 4 in a lane pair through DPP): batch counts around every boundary of its tiling
    (a lane pair, a 512-piece run of a workgroup, the grid), in place with a guard behind the batch, out of place with the
    source untouched"""
    import torch
    x = util.lcg_complex(31 * n + batch, n * batch).reshape(batch, n)
    for fwd in (True, False):
        plan = fa.Clcfft(0, n, fwd)
        assert plan.kernel_name() == "k_fft_tiny"
        buf = torch.full((batch + 64, n, 2), 7.0, device="cuda")
        buf[:batch] = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2))
        src = buf[:batch].clone()
        assert plan.exec_device(buf, batch) == 0
        torch.cuda.synchronize()
        assert bool(torch.all(buf[batch:] == 7.0)), "wrote behind the batch"
        got = buf[:batch].cpu().numpy().view(np.complex64).reshape(batch, n)
        pick = np.unique(np.concatenate([np.arange(min(batch, 600)), np.arange(max(0, batch - 600), batch), np.arange(0, batch, 4099)]))
        want = oracle.cfft(x[pick], fwd)
        assert_parity(got[pick], want, what="n=%d fwd=%s batch=%d" % (n, fwd, batch))
        keep = src.clone()
        dst = torch.full_like(src, float("nan"))
        assert plan.exec_device_oop(src, dst, batch) == 0
        torch.cuda.synchronize()
        assert torch.equal(src.view(torch.int32), keep.view(torch.int32)), "source modified"
        assert torch.equal(dst.view(torch.int32), buf[:batch].view(torch.int32)), "out of place differs from in place"


@pytest.mark.parametrize("n,batch", [(2, 3), (2, 100001), (4, 1), (4, 100001), (8, 1000), (16384, 5), (32768, 3)])
def test_cfft_buffer_aligned_to_eight_bytes_only(n, batch):
    """a complex buffer owes the library 8-byte alignment, not 16 (a view one sample into a larger buffer): the kernels that
    move 16 bytes per lane (k_fft_tiny, the two-run kernels) must cope, and nothing outside the view may change"""
    import torch
    x = util.lcg_complex(n + batch, n * batch).reshape(batch, n)
    big = torch.zeros((batch * n + 3, 2), device="cuda")
    view = big[1:1 + batch * n].view(batch, n, 2)
    assert view.data_ptr() % 16 == 8
    view.copy_(torch.from_numpy(x.view(np.float32).reshape(batch, n, 2)))
    plan = fa.Clcfft(0, n, True)
    assert plan.exec_device(view, batch) == 0
    torch.cuda.synchronize()
    got = view.cpu().numpy().view(np.complex64).reshape(batch, n)
    pick = sorted({0, batch // 2, batch - 1})
    assert_parity(got[pick], oracle.cfft(x[pick], True), what="n=%d batch=%d (%s)" % (n, batch, plan.kernel_name()))
    assert float(big[0].abs().sum()) == 0 and float(big[1 + batch * n:].abs().sum()) == 0


def test_cfft_empty_batch_and_bad_sizes():
    plan = fa.Clcfft(0, 64, True)
    assert plan.transform(np.zeros((0, 64), np.complex64)) == 0
    assert plan.transform(np.zeros(32, np.complex64)) == -30
    for n in (0, 1, (1 << 22) + 1, 1 << 25):            # (other lengths up to 2^22: Bluestein, below)
        bad = fa.Clcfft(0, n, True)
        assert bad.get_error() == -30 and fa.cl_error_string(bad.get_error()) == "Invalid value"
        assert bad.transform(np.zeros(max(n, 1), np.complex64)) == -30
    assert fa.Clcfft(99, 64, True).get_error() == -33


def test_cfft_roundtrip_and_linearity_65536():
    n = 65536
    x = util.lcg_complex(1, n)
    z = util.lcg_complex(2, n)
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    X, Z, S = x.copy(), z.copy(), (x + 2 * z).astype(np.complex64)
    assert f.transform(X) == 0 and f.transform(Z) == 0 and f.transform(S) == 0
    assert_parity(S, X + 2 * Z, tol=2e-6, what="linearity")
    back = X.copy()
    assert i.transform(back) == 0
    assert_parity(back, x, what="inverse(forward(x)) == x")


def test_cfft_device_resident_batch():
    """device-pointer entry point on a torch buffer, checked on a few batches + Parseval on all"""
    import torch
    n, batch = 65536, 96
    g = torch.Generator(device="cuda").manual_seed(7)
    d = torch.rand((batch, n, 2), generator=g, device="cuda", dtype=torch.float32) * 2 - 1
    x = d.cpu().numpy().view(np.complex64).reshape(batch, n)
    plan = fa.Clcfft(0, n, True)
    assert plan.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    y = d.cpu().numpy().view(np.complex64).reshape(batch, n)
    for b in (0, 1, batch // 2, batch - 1):
        assert_parity(y[b], oracle.cfft(x[b], True), what="batch %d" % b)
    e_in = np.sum(np.abs(x.astype(np.complex128)) ** 2, axis=1)
    e_out = np.sum(np.abs(y.astype(np.complex128)) ** 2, axis=1) * n
    assert np.max(np.abs(e_out / e_in - 1)) < 1e-5


def test_cfft_resident_kernel_both_directions():
    """n = 65536 with more transforms than CUs / 4: the resident kernel (fft_resident.hip), forward and
    inverse, every transform against the oracle"""
    n, batch = 65536, 70
    x = util.lcg_complex(31, n * batch).reshape(batch, n)
    for fwd in (True, False):
        plan = fa.Clcfft(0, n, fwd)
        assert plan.kernel_name() == "k_fft_res16"
        y = x.copy()
        assert plan.transform(y) == 0
        assert_parity(y, oracle.cfft(x, fwd), what="resident kernel fwd=%s" % fwd)


@pytest.mark.parametrize("n", [16384, 32768, 65536])
def test_cfft_batched_kernels_vs_reference_vectors(n):
    """The reference's own outputs (Clcfft::transform, cl_fft.cpp:153-161, run on the MI355X through OpenCL) against the
    BATCHED kernels: a single transform of these lengths goes to the column / row kernel pair, so the golden input
    LCG(12345) is replicated 70 times — the batch that selects k_fft_res16 (n = 65536), the persistent four-step kernel
    (n = 32768) and k_cfft_2x (n = 16384) — and EVERY transform of the batch is compared with the reference's vector."""
    batch = 70
    x = np.tile(util.lcg_complex(12345, n), (batch, 1))
    kernel = {65536: "k_fft_res16", 32768: "k_fft_4step", 16384: "k_cfft_2x"}[n]
    for fwd in (True, False):
        plan = fa.Clcfft(0, n, fwd)
        assert plan.kernel_name() == kernel
        y = x.copy()
        assert plan.transform(y) == 0
        ref = golden("g4_cfft%d_%s_dec" % (n, "fwd" if fwd else "inv"))
        for b in range(batch):
            assert_parity(util.decimate(y[b]), ref, what="n=%d fwd=%s %s transform %d vs reference" % (n, fwd, kernel, b))
        assert np.array_equal(y.view(np.uint32), np.tile(y[0], (batch, 1)).view(np.uint32)), "transforms of one batch differ"


@pytest.mark.parametrize("n,batch", [(65536, 70), (65536, 300), (1024, 9), (16384, 70), (65536, 3)])
def test_cfft_out_of_place(n, batch):
    """clfa_fft_exec_dev_oop (extension; the reference's device side is out of place too, cl_fft.cpp:138-151): the resident
    kernel src -> dst for n = 65536 in batches, every other kernel with its stores redirected; the source stays untouched, and the
    result is the reference's vector (golden input replicated) in every transform, forward and inverse"""
    import torch
    x = np.tile(util.lcg_complex(12345, n), (batch, 1))
    for fwd in (True, False):
        plan = fa.Clcfft(0, n, fwd)
        src = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2).copy()).cuda()
        dst = torch.full_like(src, float("nan"))
        assert plan.exec_device_oop(src, dst, batch) == 0
        torch.cuda.synchronize()
        assert np.array_equal(src.cpu().numpy().view(np.uint32), x.view(np.uint32).reshape(batch, n, 2)), "source modified"
        y = dst.cpu().numpy().view(np.complex64).reshape(batch, n)
        name = "g4_cfft%d_%s_dec" % (n, "fwd" if fwd else "inv") if n >= 16384 else "g3_cfft1024_%s" % ("fwd" if fwd else "inv")
        ref = golden(name)
        for b in range(batch):
            assert_parity(util.decimate(y[b]) if n >= 16384 else y[b], ref, what="oop n=%d fwd=%s transform %d vs reference" % (n, fwd, b))
        # the same bits as the in-place entry point
        assert plan.exec_device(src, batch) == 0
        torch.cuda.synchronize()
        assert torch.equal(src.view(torch.int32), dst.view(torch.int32)), "out of place and in place differ"


def test_cfft_out_of_place_arguments():
    import torch
    n, batch = 65536, 70
    plan = fa.Clcfft(0, n, True)
    buf = torch.zeros((2 * batch, n, 2), device="cuda")
    assert plan.exec_device_oop(buf, buf, batch) == 0                       # src == dst: in place
    assert plan.exec_device_oop(buf, buf[1:], batch) == fa.CL_INVALID_VALUE     # partly overlapping
    assert plan.exec_device_oop(buf[1:], buf, batch) == fa.CL_INVALID_VALUE
    assert plan.exec_device_oop(buf, buf[batch:], batch) == 0               # adjacent, disjoint
    assert plan.exec_device_oop(buf, buf[batch:], 0) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("size,batch", [(16384, 11), (65536, 40)])
def test_rfft_out_of_place(size, batch):
    """real plans through the out-of-place entry point: oracle parity, source untouched"""
    import torch
    r = (np.random.default_rng(size).random((batch, size), dtype=np.float32) * 2 - 1)
    src = torch.from_numpy(r.copy()).cuda()
    dst = torch.empty_like(src)
    assert fa.Clrfft(0, size, True).exec_device_oop(src, dst, batch) == 0
    torch.cuda.synchronize()
    assert np.array_equal(src.cpu().numpy(), r)
    assert_parity(dst.cpu().numpy().view(np.complex64), oracle.rfft_forward(r), what="rfft oop size %d" % size)


@pytest.mark.parametrize("real,n,batch", [(False, 2, 1001), (False, 4, 77), (False, 64, 333), (False, 256, 17), (False, 4096, 5),
                                          (False, 8192, 70), (False, 16384, 3), (False, 16384, 70), (False, 32768, 3),
                                          (False, 32768, 70), (False, 65536, 3), (False, 1 << 17, 2), (False, 1 << 19, 1),
                                          (False, 1 << 21, 1), (False, 1000, 9), (False, 44100, 2),
                                          (True, 8, 1000), (True, 256, 130), (True, 512, 9), (True, 8192, 37), (True, 16384, 70),
                                          (True, 32768, 3), (True, 32768, 70), (True, 65536, 3), (True, 65536, 40),
                                          (True, 131072, 5), (True, 131072, 70), (True, 1 << 18, 2), (True, 1000, 7)])
def test_out_of_place_equals_in_place(real, n, batch):
    """Every route of the library through clfa_fft_exec_dev_oop (complex and packed real; one-, two- and three-pass
    sizes; the few-transform routes; other lengths): forward and inverse, the destination holds bit for bit what the
    in-place entry point produces, and the source is left untouched (routes of several passes must put their FIRST pass
    into the destination)."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(n + batch)
    shape = (batch, n) if real else (batch, n, 2)
    for fwd in (True, False):
        plan = (fa.Clrfft if real else fa.Clcfft)(0, n, fwd)
        assert plan.get_error() == 0, plan.get_log()
        src = torch.rand(shape, generator=g, device="cuda", dtype=torch.float32) * 2 - 1
        keep = src.clone()
        dst = torch.full_like(src, float("nan"))
        assert plan.exec_device_oop(src, dst, batch) == 0
        torch.cuda.synchronize()
        assert torch.equal(src.view(torch.int32), keep.view(torch.int32)), "source modified (%s)" % plan.kernel_name()
        assert plan.exec_device(src, batch) == 0
        torch.cuda.synchronize()
        assert not torch.isnan(dst).any()
        assert torch.equal(src.view(torch.int32), dst.view(torch.int32)), "out of place differs from in place (%s)" % plan.kernel_name()


@pytest.mark.parametrize("size,kernel", [(8192, "k_fft_lds"), (32768, "k_rfft_2x"), (65536, "k_rfft_2x"), (131072, None)])
def test_rfft_batched_kernels_vs_reference_vectors(size, kernel):
    """Clrfft::transform (cl_fft.cpp:267-296) vectors against the one-pass real kernels: more than 32 transforms select
    k_rfft_2x (two 8192- / 16384-point runs for sizes 32768 / 65536) instead of the spread path a single transform takes; every
    transform of the batch against the reference's forward, round-trip and arbitrary-spectrum inverse vectors, bin M/2
    (the reference's never-conjugated self-paired bin, cl_fft.cpp:278) included."""
    batch, m = 70, size // 2
    f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
    if kernel is not None:
        assert f.kernel_name() == kernel and i.kernel_name() == kernel
    x = np.tile(util.lcg_sym(12345, size), (batch, 1))
    buf = x.copy().view(np.complex64)
    assert f.transform(buf) == 0                                   # in place, batch x m packed bins
    ref = golden("g5_rfft%d_fwd_dec" % size)
    for b in range(batch):
        got = np.concatenate([util.decimate(buf[b]), buf[b, m // 2:m // 2 + 1]])
        assert_parity(got, ref, what="size=%d fwd transform %d" % (size, b))
    assert i.transform(buf) == 0
    ref = golden("g5_rfft%d_rt_dec" % size)
    for b in range(batch):
        assert_parity(util.decimate(buf[b]), ref, what="size=%d round trip %d" % (size, b))
    arb = np.tile(util.lcg_complex(777, m), (batch, 1))
    assert i.transform(arb) == 0
    ref = golden("g5_rfft%d_invarb_dec" % size)
    for b in range(batch):
        assert_parity(util.decimate(arb[b]), ref, what="size=%d invarb %d" % (size, b))


@pytest.mark.parametrize("batch", [65, 257, 300, 777])
def test_rfft131072_forward_one_pass_ragged(batch):
    """Clrfft forward (cl_fft.cpp:272-282) of size 131072 with more than CUs / 4 transforms: ONE launch — the resident
    65536-point kernel with the reference's `conv` pair map (cl_fft.cpp:178-191) inside its second phase.  Ragged batches
    (not multiples of the grid, one workgroup with one transform more than its neighbour); transforms picked across the
    batch against the oracle, with the map's special bins (0: DC / Nyquist packed, M/2: left as the complex transform made
    it) and the rows that pair across lanes (k1 = 0, 128) checked by name; out of place bit for bit the same."""
    import torch
    size, m = 131072, 65536
    rng = np.random.default_rng(batch)
    r = (rng.random((batch, size), dtype=np.float32) * 2 - 1).astype(np.float32)
    f = fa.Clrfft(0, size, True)
    assert f.get_error() == 0 and f.kernel_name() == "k_fft_res16"
    d = torch.from_numpy(r.copy()).cuda()
    assert f.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(np.complex64).reshape(batch, m)
    pick = sorted(set([0, 1, batch // 2, 255 % batch, 256 % batch, batch - 2, batch - 1]))
    want = oracle.rfft_forward(r[pick])
    assert_parity(got[pick], want, what="rfft 131072 batch %d" % batch)
    scale = np.abs(want).max()
    for name, idx in (("bin 0", [0]), ("bin M/2", [m // 2]), ("row k1 = 0", np.arange(0, m, 256)), ("row k1 = 128", np.arange(128, m, 256)),
                      ("lanes c = 0", np.arange(0, m, 16))):
        err = np.abs(got[pick][:, idx] - want[:, idx]).max() / scale
        assert err < 1e-6, (name, err)
    src = torch.from_numpy(r).cuda()
    dst = torch.full_like(src, float("nan"))
    assert f.exec_device_oop(src, dst, batch) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst.view(torch.int32), d.view(torch.int32))
    assert np.array_equal(src.cpu().numpy(), r)


@pytest.mark.parametrize("batch", [65, 257, 300, 777])
def test_rfft131072_inverse_one_pass_ragged(batch):
    """Clrfft inverse (cl_fft.cpp:283-294) of size 131072 with more than CUs / 4 transforms: ONE launch — the reference's
    `iconv` pair map (cl_fft.cpp:192-205) inside the FIRST phase of the resident kernel (column blocks in mirrored pairs).
    Arbitrary packed spectra (not only those of real signals: bin 0's imaginary part and bin M/2 are free) against the
    oracle; single-bin spectra on the bins the map treats specially (0, M/2) and on the columns that pair across lanes
    (bins 256 k, 128 + 256 k) and one step further (16 q, 16 (16 - q)); out of place bit for bit the same."""
    import torch
    size, m = 131072, 65536
    rng = np.random.default_rng(1000 + batch)
    c = ((rng.random((batch, m), dtype=np.float32) * 2 - 1) + 1j * (rng.random((batch, m), dtype=np.float32) * 2 - 1)).astype(np.complex64)
    c *= np.float32(1 / 256)
    bins = [0, m // 2, 256, 128, 128 + 256 * 77, 256 * 255, 16, 240, 112, 144, 65535, 1, 4096 + 16 * 5, m - 4096 - 16 * 5]
    for k, bin_ in enumerate(bins):   # the last transforms of the batch: one bin each
        c[batch - 1 - k] = 0
        c[batch - 1 - k, bin_] = 1 - 0.5j
    i = fa.Clrfft(0, size, False)
    assert i.get_error() == 0 and i.kernel_name() == "k_fft_res16"
    d = torch.from_numpy(c.view(np.float32).reshape(batch, size).copy()).cuda()
    assert i.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    got = d.cpu().numpy().reshape(batch, size)
    pick = sorted(set([0, 1, batch // 2, 255 % batch, 256 % batch] + [batch - 1 - k for k in range(len(bins))]))
    want = oracle.rfft_inverse(c[pick])
    for j, b in enumerate(pick):
        assert_parity(got[b], want[j], what="irfft 131072 batch %d transform %d" % (batch, b))
    src = torch.from_numpy(c.view(np.float32).reshape(batch, size).copy()).cuda()
    keep = src.clone()
    dst = torch.full_like(src, float("nan"))
    assert i.exec_device_oop(src, dst, batch) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst.view(torch.int32), d.view(torch.int32))
    assert torch.equal(src.view(torch.int32), keep.view(torch.int32))


@pytest.mark.parametrize("n,batch", [(16384, 70), (16384, 300), (32768, 70), (32768, 300), (65536, 70), (65536, 300)])
def test_cfft_persistent_kernel_ragged(n, batch):
    """the persistent four-step kernel (intermediate in LDS + registers) with batch counts that are
    neither small enough for the small-batch kernels nor multiples of the grid: every transform
    against the oracle, and the round trip"""
    x = util.lcg_complex(977 + n + batch, n * batch).reshape(batch, n)
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    y = x.copy()
    assert f.transform(y) == 0
    assert_parity(y, oracle.cfft(x, True), what="fwd n=%d batch=%d" % (n, batch))
    worst = max(util.rel_err(y[b], oracle.cfft(x[b:b + 1], True)[0])[0] for b in (0, 1, batch // 2, batch - 1))
    assert worst < 1e-6, worst
    assert i.transform(y) == 0
    assert_parity(y, x, what="round trip n=%d batch=%d" % (n, batch))


# ---- a7/a8/a9: r2c / c2r -----------------------------------------------------------------

def test_kat_test_rfft_n16():
    x = golden("g2_rfft16_in").copy()
    spec = np.zeros(8, np.complex64)
    assert fa.Clrfft(0, 16, True).transform(spec, x) == 0
    want = np.zeros(8, np.complex64)
    want[0], want[1] = 0.5 + 0.5j, -1.0j
    assert np.max(np.abs(spec - want)) < 2e-7
    back = np.zeros(16, np.float32)
    assert fa.Clrfft(0, 16, False).transform(spec, back) == 0
    assert_parity(back, golden("g2_rfft16_inv"), what="inv")


@pytest.mark.parametrize("size", RSIZES)
def test_rfft_vs_oracle_and_reference(size):
    m = size // 2
    x = util.lcg_sym(12345, size)
    f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
    assert f.get_error() == 0 and i.get_error() == 0
    spec = np.zeros(m, np.complex64)
    assert f.transform(spec, x.copy()) == 0
    ospec = oracle.rfft_forward(x)
    assert_parity(spec, ospec, what="fwd vs oracle")
    # the self-paired bin keeps the reference's quirk (never conjugated, cl_fft.cpp:278)
    assert abs(spec[m // 2] - ospec[m // 2]) <= TOL * np.max(np.abs(ospec))
    back = np.zeros(size, np.float32)
    assert i.transform(spec.copy(), back) == 0
    assert_parity(back, oracle.rfft_inverse(ospec), what="inv vs oracle")
    arb = util.lcg_complex(777, m)
    arbout = np.zeros(size, np.float32)
    assert i.transform(arb.copy(), arbout) == 0
    if size <= 4096:
        assert_parity(spec, golden("g5_rfft%d_fwd" % size), what="fwd vs reference")
        assert_parity(back, golden("g5_rfft%d_rt" % size), what="rt vs reference")
        assert_parity(arbout, golden("g5_rfft%d_invarb" % size), what="invarb vs reference")
    else:
        got = np.concatenate([util.decimate(spec), spec[m // 2:m // 2 + 1]])
        assert_parity(got, golden("g5_rfft%d_fwd_dec" % size), what="fwd vs reference")
        assert_parity(util.decimate(back.view(np.complex64)), golden("g5_rfft%d_rt_dec" % size), what="rt")
        assert_parity(util.decimate(arbout.view(np.complex64)), golden("g5_rfft%d_invarb_dec" % size), what="invarb")


@pytest.mark.parametrize("size,batch", [(4, 777), (64, 130), (2048, 9), (16384, 11), (65536, 3)])
def test_rfft_batched_in_place(size, batch):
    x = util.lcg_sym(55 + size, size * batch).reshape(batch, size)
    buf = x.copy().view(np.complex64)
    assert fa.Clrfft(0, size, True).transform(buf) == 0      # in-place form, cl_fft.h:104-109
    assert_parity(buf, oracle.rfft_forward(x), what="fwd in place")
    assert fa.Clrfft(0, size, False).transform(buf) == 0
    assert_parity(buf.view(np.float32), x, what="round trip")



@pytest.mark.parametrize("size,batch", [(32768, 1), (32768, 32), (32768, 33), (32768, 259), (32768, 1030),
                                        (65536, 1), (65536, 32), (65536, 33), (65536, 257), (65536, 700)])
def test_rfft_fused_big_sizes_ragged_batches(size, batch):
    """real sizes 32768 (k_rfft_2x<13>, two 512-lane workgroups per CU) and 65536 (k_rfft_2x<14>, one of 1024 lanes):
    fewer transforms than CUs, one more than a whole number of rounds, several rounds; up to 32 transforms run
    spread over the four-step pair + pack kernel instead (both sides of that switch are here); device-resident,
    a few transforms against the oracle, every transform through the round trip"""
    import torch
    g = torch.Generator(device="cuda").manual_seed(size + batch)
    d = torch.rand((batch, size), generator=g, device="cuda", dtype=torch.float32) * 2 - 1
    x = d.clone()
    f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
    assert f.kernel_name() == "k_rfft_2x"
    assert f.exec_device(d, batch) == 0
    pick = sorted({0, batch // 2, batch - 1})
    spec = d[pick].cpu().numpy().view(np.complex64)
    assert_parity(spec, oracle.rfft_forward(x[pick].cpu().numpy()), what="fwd size=%d batch=%d" % (size, batch))
    assert i.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    err = (d - x).reshape(batch, -1).norm(dim=1) / x.reshape(batch, -1).norm(dim=1)
    assert float(err.max()) < 1e-6, float(err.max())


def test_rfft_bad_sizes():
    for s in (0, 2, 3, 6, 1 << 26):                      # odd, or not a power of two and not a multiple of 4
        assert fa.Clrfft(0, s, True).get_error() == -30


# ---- full BASELINE sizes: size-independent properties -------------------------------------

def test_config2_full_size_roundtrip_parseval():
    """config 2 of BASELINE.json: 4096 x 65536 c2c in HBM; inverse(forward(x)) == x,
    Parseval on every batch, four batches against the oracle"""
    import torch
    n, batch = 65536, 4096
    g = torch.Generator(device="cuda").manual_seed(2024)
    d = torch.rand((batch, n, 2), generator=g, device="cuda", dtype=torch.float32) * 2 - 1
    keep = {b: d[b].cpu().numpy().view(np.complex64).reshape(n) for b in (0, 1, 2047, 4095)}
    e_in = (d.double() ** 2).sum(dim=(1, 2))
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    assert f.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    e_out = (d.double() ** 2).sum(dim=(1, 2)) * n
    assert float(((e_out / e_in) - 1).abs().max()) < 1e-5
    for b, x in keep.items():
        assert_parity(d[b].cpu().numpy().view(np.complex64).reshape(n), oracle.cfft(x, True), what="batch %d" % b)
    assert i.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    for b, x in keep.items():
        assert_parity(d[b].cpu().numpy().view(np.complex64).reshape(n), x, what="round trip batch %d" % b)
    e_rt = (d.double() ** 2).sum(dim=(1, 2))
    assert float(((e_rt / e_in) - 1).abs().max()) < 1e-5


def test_config3_full_size_roundtrip():
    """config 3: r2c + c2r, 16384 real x 8192 batches, packed in place"""
    import torch
    size, batch = 16384, 8192
    g = torch.Generator(device="cuda").manual_seed(3)
    d = torch.rand((batch, size), generator=g, device="cuda", dtype=torch.float32) * 2 - 1
    orig = d.clone()
    keep = {b: d[b].cpu().numpy() for b in (0, 4095, 8191)}
    f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
    assert f.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    for b, x in keep.items():
        assert_parity(d[b].cpu().numpy().view(np.complex64), oracle.rfft_forward(x), what="fwd batch %d" % b)
    assert i.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    err = float((d - orig).double().norm() / orig.double().norm())
    assert err < TOL


@pytest.mark.parametrize("real,n", [(False, 8), (False, 16), (False, 32), (False, 64), (False, 128), (False, 256), (False, 512), (False, 1024),
                                    (False, 2048), (False, 4096), (False, 8192), (True, 8), (True, 16), (True, 32), (True, 64), (True, 256), (True, 512),
                                    (True, 1024), (True, 2048), (True, 4096), (True, 8192)])
def test_streaming_batches_on_the_reduced_grid(real, n):
    """1 GiB of transforms per launch and one more (ragged): from there the persistent grids of k_fft_small / k_fft_lds put
    one or two workgroups on a CU instead of all that fit (wgs_per_cu(), fft_kernels.hip) — same transforms, longer
    grid-stride loops: picked transforms against the oracle, all of them through the round trip"""
    import torch
    per = n * (4 if real else 8)
    batch = (1 << 30) // per + 1
    g = torch.Generator(device="cuda").manual_seed(n)
    d = torch.rand((batch, n) if real else (batch, n, 2), generator=g, device="cuda", dtype=torch.float32) * 2 - 1
    orig = d.clone()
    pick = [0, 1, batch // 3, batch - 2, batch - 1]
    keep = {b: d[b].cpu().numpy() for b in pick}
    f, i = (fa.Clrfft(0, n, True), fa.Clrfft(0, n, False)) if real else (fa.Clcfft(0, n, True), fa.Clcfft(0, n, False))
    assert f.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    for b, x in keep.items():
        got = d[b].cpu().numpy().reshape(-1).view(np.complex64)
        want = oracle.rfft_forward(x) if real else oracle.cfft(x.reshape(-1).view(np.complex64), True)
        assert_parity(got, want, what="%s n=%d transform %d of %d" % ("real" if real else "complex", n, b, batch))
    assert i.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    err = float((d - orig).double().norm() / orig.double().norm())
    worst = float((d - orig).abs().max())
    assert err < TOL and worst < 1e-5, (err, worst)


# ---- beyond the reference: n = 2^17 .. 2^24 ---------------------------
# The reference's stage kernel overflows int32 above 65536 (cl_fft.cpp:32), so there is nothing of its
# own to compare with: the yardstick is numpy's float64 FFT under the reference's conventions
# (forward scaled by 1/n, inverse unscaled), same norm-relative 1e-6 criterion.

@pytest.mark.parametrize("logn,batch", [(17, 3), (18, 2), (19, 1), (19, 2), (20, 2), (20, 1), (21, 1), (21, 3), (22, 1), (22, 2), (23, 1), (24, 1)])
def test_cfft_big_sizes(logn, batch):
    n = 1 << logn
    rng = np.random.default_rng(logn)
    x = (rng.uniform(-1, 1, (batch, n)) + 1j * rng.uniform(-1, 1, (batch, n))).astype(np.complex64)
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    assert f.get_error() == 0 and i.get_error() == 0
    assert f.workspace_bytes() >= 8 * n
    assert f.kernel_name() == i.kernel_name() == ("k_big2_cols" if logn <= 22 else "k_big_cols")   # two passes / three
    y = x.copy()
    assert f.transform(y) == 0
    want = (np.fft.fft(x.astype(np.complex128), axis=-1) / n)
    assert_parity(y, want.astype(np.complex64), what="fwd 2^%d" % logn)
    z = x.copy()
    assert i.transform(z) == 0
    want = np.fft.ifft(x.astype(np.complex128), axis=-1) * n
    assert_parity(z, want.astype(np.complex64), what="inv 2^%d" % logn)
    assert i.transform(y) == 0                       # round trip
    assert_parity(y, x, what="round trip 2^%d" % logn)



# ---- lengths that are not powers of two (extension) -----------------------
# The reference itself only runs powers of two (its callers pad, opcode.cpp:30-35); Bluestein's algorithm
# around two power-of-two plans gives the exact DFT of any length.  Yardstick: numpy's float64 FFT under the
# reference's conventions, same norm-relative 1e-6 criterion.

@pytest.mark.parametrize("n,batch", [(3, 7), (5, 1), (6, 3), (12, 100), (100, 33), (129, 70), (300, 17), (1000, 9), (1536, 5),
                                     (2049, 600), (4095, 3), (4096 - 7, 1), (4097, 2),
                                     (44100, 2), (48000, 3), (65537, 1), (100000, 2), (3 << 20, 1)])
def test_cfft_any_length(n, batch):
    """(convolution lengths 256 .. 8192, i.e. n = 65 .. 4096, run in ONE launch: k_blue_lds; ragged groups, more transforms
    than workgroups)"""
    rng = np.random.default_rng(n)
    x = (rng.uniform(-1, 1, (batch, n)) + 1j * rng.uniform(-1, 1, (batch, n))).astype(np.complex64)
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    assert f.get_error() == 0 and i.get_error() == 0, (f.get_log(), i.get_log())
    m = 1 << int(np.ceil(np.log2(2 * n - 1)))
    assert f.kernel_name() == ("k_blue_lds" if 256 <= m <= 8192 else "bluestein")
    assert (f.workspace_bytes() > 0) == (f.kernel_name() == "bluestein")
    y = x.copy()
    assert f.transform(y) == 0
    assert_parity(y, (np.fft.fft(x.astype(np.complex128), axis=-1) / n).astype(np.complex64), what="fwd n=%d" % n)
    z = x.copy()
    assert i.transform(z) == 0
    assert_parity(z, (np.fft.ifft(x.astype(np.complex128), axis=-1) * n).astype(np.complex64), what="inv n=%d" % n)
    assert i.transform(y) == 0
    assert_parity(y, x, what="round trip n=%d" % n)


def test_cfft_any_length_out_of_place():
    """the one-launch form (k_blue_lds) from src to dst on device memory: the source is left untouched, ragged groups"""
    import torch
    for n, batch in ((1000, 9), (100, 37), (4095, 5)):
        f = fa.Clcfft(0, n, True)
        assert f.kernel_name() == "k_blue_lds"
        src = torch.rand((batch, n, 2), device="cuda") * 2 - 1
        keep = src.clone()
        dst = torch.zeros_like(src)
        assert f.exec_device_oop(src, dst, batch) == 0
        torch.cuda.synchronize()
        assert torch.equal(src, keep)
        x = keep.cpu().numpy().view(np.complex64).reshape(batch, n)
        y = dst.cpu().numpy().view(np.complex64).reshape(batch, n)
        assert_parity(y, (np.fft.fft(x.astype(np.complex128), axis=-1) / n).astype(np.complex64), what="oop fwd n=%d" % n)


def test_cfft_any_length_chunks():
    """more rows than the 256 MiB convolution workspace holds (m = 2^18 per row of n = 100000): the batch is walked in chunks"""
    import torch
    n, batch = 100000, 200
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    x = d.clone()
    assert f.exec_device(d, batch) == 0
    pick = [0, 127, 128, 199]
    y = d[pick].cpu().numpy().view(np.complex64).reshape(len(pick), n)
    xp = x[pick].cpu().numpy().view(np.complex64).reshape(len(pick), n)
    assert_parity(y, (np.fft.fft(xp.astype(np.complex128), axis=-1) / n).astype(np.complex64), what="chunked fwd")
    assert i.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    assert float((d - x).norm() / x.norm()) < 1e-6


@pytest.mark.parametrize("size,batch", [(12, 50), (1000, 7), (48000, 3), (96000, 2), (3 << 17, 1)])
def test_rfft_any_length(size, batch):
    """packed real transforms of sizes that are multiples of 4: the reference's packing rules (amplitude
    scaling, DC / Nyquist in bin 0, bin M/2 left un-conjugated) on a float64 FFT"""
    m = size // 2
    rng = np.random.default_rng(size)
    x = rng.uniform(-1, 1, (batch, size)).astype(np.float32)
    f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
    assert f.get_error() == 0 and i.get_error() == 0, (f.get_log(), i.get_log())
    c = np.zeros((batch, m), np.complex64)
    assert f.transform(c, x) == 0
    X = np.fft.fft(x.astype(np.float64), axis=-1)
    want = np.empty((batch, m), np.complex128)
    want[:, 0] = X[:, 0].real / size + 1j * X[:, m].real / size
    want[:, 1:] = 2 * X[:, 1:m] / size
    want[:, m // 2] = np.conj(want[:, m // 2])
    assert_parity(c, want.astype(np.complex64), what="r2c size=%d" % size)
    r = np.zeros((batch, size), np.float32)
    assert i.transform(c, r) == 0
    # forward and inverse chained: two Bluestein transforms (four power-of-two transforms of the convolution length, eight chirp
    # products).  Measured over sizes 96000 .. 786432 and three seeds: relL2 3.5-4.0e-7, worst sample 0.89-1.19e-6 (1.07e-6 in
    # round 4's build as well, profiles/big_two_pass_r05.txt) — each direction alone is held to 1e-6 above, the pair to twice that,
    # like the round trips of tests/fuzz_parity.py
    assert_parity(r, x, tol=2e-6, what="rfft round trip size=%d" % size)


def test_cfft_big_batch_chunks():
    """more transforms than the 256 MiB workspace holds at once: exec walks the batch in chunks"""
    import torch
    n, batch = 1 << 17, 300                              # 1 MiB each, workspace = 256 of them
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    x = d.clone()
    assert f.exec_device(d, batch) == 0
    y = d[[0, 255, 256, 299]].cpu().numpy().view(np.complex64).reshape(4, n)
    x4 = x[[0, 255, 256, 299]].cpu().numpy().view(np.complex64).reshape(4, n)
    assert_parity(y, (np.fft.fft(x4.astype(np.complex128), axis=-1) / n).astype(np.complex64), what="chunked fwd")
    assert i.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    err = float((d - x).norm() / x.norm())
    assert err < 1e-6, err


def test_rfft_big_size():
    """packed real transform above 131072 points: same packing rules as the reference's (bin M/2 left
    un-conjugated, amplitude scaling), checked against the oracle's pack applied to a float64 FFT"""
    size = 1 << 19
    m = size // 2
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, size).astype(np.float32)
    f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
    assert f.get_error() == 0 and i.get_error() == 0
    c = np.zeros(m, np.complex64)
    assert f.transform(c, x) == 0
    X = np.fft.fft(x.astype(np.float64))
    want = np.empty(m, np.complex128)
    want[0] = X[0].real / size + 1j * X[m].real / size
    want[1:] = 2 * X[1:m] / size
    want[m // 2] = np.conj(want[m // 2])
    assert_parity(c, want.astype(np.complex64), what="big r2c")
    r = np.zeros(size, np.float32)
    assert i.transform(c, r) == 0
    assert_parity(r, x, what="big rfft round trip")
