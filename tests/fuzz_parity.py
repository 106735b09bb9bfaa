"""Randomised parity sweep on the GPU box: `python tests/fuzz_parity.py [seconds] [seed]`, and — with a fixed seed and a
bounded budget — a `-m gpu` test (tests/test_gpu_misc.py::test_randomised_parity_sweep calls run()).
Random geometries of every object of the path — complex and packed real plans (powers of two with ragged batches, a
few other lengths, the sizes above 65536), partitioned convolutions (channels, partitions, static / time-varying, host and device entry
points), direct convolutions — each checked against the CPU oracle (numpy fp64 for the lengths the reference does not
have).  Prints one line per case and a summary; exits non-zero on the first failure."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import opencl_fft_amd as fa  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.util import rel_err  # noqa: E402

rng = np.random.default_rng(1)
counts = {}
quiet = False


def check(kind, desc, got, want, tol):
    l2, mx = rel_err(got, want)
    counts[kind] = counts.get(kind, 0) + 1
    ok = l2 <= tol and mx <= tol
    line = "%-6s %-62s relL2 %.2e max %.2e %s" % (kind, desc, l2, mx, "" if ok else "FAIL (tol %.1e)" % tol)
    if not quiet or not ok:
        print(line, flush=True)
    assert ok, line


def sym(shape):
    return (rng.random(shape, dtype=np.float32) * 2 - 1).astype(np.float32)


def case_cfft():
    logn = int(rng.integers(1, 18))
    n = 1 << logn
    cap = max(1, min(300, (1 << 22) // n))
    batch = int(rng.integers(1, cap + 1))
    fwd = bool(rng.integers(0, 2))
    x = (sym((batch, n)) + 1j * sym((batch, n))).astype(np.complex64)
    p = fa.Clcfft(0, n, fwd)
    assert p.get_error() == 0, p.get_log()
    d = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2).copy()).cuda()
    oop = bool(rng.integers(0, 3) == 0)          # a third of the cases through the out-of-place entry point
    if oop:
        dst = torch.full_like(d, float("nan"))
        assert p.exec_device_oop(d, dst, batch) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy().view(np.uint32), x.view(np.uint32).reshape(batch, n, 2)), "source modified"
        d = dst
    else:
        assert p.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(np.complex64).reshape(batch, n)
    pick = sorted(set([0, batch - 1, batch // 2] + [int(v) for v in rng.integers(0, batch, 3)]))
    if logn <= 16:
        want = oracle.cfft(x[pick], fwd)
    else:   # beyond the reference's range: numpy fp64 with the reference's scaling
        want = (np.fft.fft(x[pick].astype(np.complex128)) / n) if fwd else np.fft.ifft(x[pick].astype(np.complex128)) * n
    check("cfft", "n=2^%d batch=%d %s%s kernel=%s" % (logn, batch, "fwd" if fwd else "inv", " oop" if oop else "", p.kernel_name()), got[pick], want, 1e-6)


def case_cfft_big():
    """beyond the reference's range: two passes to 2^22 (three above), against numpy float64 under the reference's scaling"""
    logn = int(rng.integers(17, 23))
    n = 1 << logn
    batch = int(rng.integers(1, max(2, min(6, (1 << 24) // n) + 1)))
    fwd = bool(rng.integers(0, 2))
    x = (sym((batch, n)) + 1j * sym((batch, n))).astype(np.complex64)
    p = fa.Clcfft(0, n, fwd)
    assert p.get_error() == 0, p.get_log()
    d = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2).copy()).cuda()
    oop = bool(rng.integers(0, 3) == 0)
    if oop:
        dst = torch.full_like(d, float("nan"))
        assert p.exec_device_oop(d, dst, batch) == 0
        d = dst
    else:
        assert p.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(np.complex64).reshape(batch, n)
    b = int(rng.integers(0, batch))
    want = (np.fft.fft(x[b].astype(np.complex128)) / n) if fwd else np.fft.ifft(x[b].astype(np.complex128)) * n
    check("big", "n=2^%d batch=%d %s%s kernel=%s" % (logn, batch, "fwd" if fwd else "inv", " oop" if oop else "", p.kernel_name()), got[b], want, 1e-6)


def case_cfft_any():
    n = int(rng.integers(3, 5000))
    if n & (n - 1) == 0:
        n += 1
    batch = int(rng.integers(1, 40))
    fwd = bool(rng.integers(0, 2))
    x = (sym((batch, n)) + 1j * sym((batch, n))).astype(np.complex64)
    p = fa.Clcfft(0, n, fwd)
    assert p.get_error() == 0, p.get_log()
    d = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2).copy()).cuda()
    assert p.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(np.complex64).reshape(batch, n)
    want = (np.fft.fft(x.astype(np.complex128)) / n) if fwd else np.fft.ifft(x.astype(np.complex128)) * n
    check("any", "n=%d batch=%d %s" % (n, batch, "fwd" if fwd else "inv"), got, want, 1e-6)


def case_rfft():
    logs = int(rng.integers(2, 18))
    size = 1 << logs
    cap = max(1, min(300, (1 << (24 if logs == 17 else 22)) // size))   # (size 131072: past CUs / 4 transforms, the one-pass kernel)
    batch = int(rng.integers(1, cap + 1))
    r = sym((batch, size))
    f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
    assert f.get_error() == 0 and i.get_error() == 0
    d = torch.from_numpy(r.copy()).cuda()
    assert f.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    spec = d.cpu().numpy().view(np.complex64).reshape(batch, size // 2)
    pick = sorted(set([0, batch - 1] + [int(v) for v in rng.integers(0, batch, 2)]))
    check("rfft", "size=2^%d batch=%d fwd kernel=%s" % (logs, batch, f.kernel_name()), spec[pick], oracle.rfft_forward(r[pick]), 1e-6)
    assert i.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    back = d.cpu().numpy().reshape(batch, size)
    check("rfft", "size=2^%d batch=%d inverse of the forward" % (logs, batch), back[pick], oracle.rfft_inverse(oracle.rfft_forward(r[pick])), 2e-6)


def case_pconv():
    pts = 1 << int(rng.integers(1, 12))
    nparts = int(rng.integers(1, 40)) if pts >= 512 else int(rng.integers(1, 300))
    channels = int(rng.choice([1, 1, 1, 2, 3, 5, 17, 40, 150, 256]))   # (137 and more: k_pconv_fused)
    if pts * nparts * channels > (1 << 21):
        channels = 1
    tv = bool(rng.integers(0, 2))
    dev = bool(rng.integers(0, 2))
    cvs = pts * nparts + int(rng.integers(0, pts))   # the reference floors cvs / pts
    blocks = int(rng.integers(2, 2 * nparts + 4)) if nparts < 20 else int(rng.integers(2, 12))
    p = fa.Clpconv(0, cvs, pts, channels=channels)
    assert p.get_cl_err() == 0 and p.nparts == nparts
    os_ = [oracle.Pconv(cvs, pts) for _ in range(channels)]
    ir = (sym((channels, cvs)) * (0.5 / np.sqrt(cvs))).astype(np.float32)
    if not tv:
        assert p.push_ir(ir) == 0
        for c in range(channels):
            os_[c].push_ir(ir[c])
    x1 = sym((blocks, channels, pts))
    x2 = (sym((blocks, channels, pts)) * (0.5 / np.sqrt(cvs))).astype(np.float32)
    got = np.zeros((blocks, channels, pts), np.float32)
    if dev:
        d1, d2 = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
        dout = torch.zeros((blocks, channels, pts), device="cuda")
        for b in range(blocks):
            assert p.process_device(dout[b], d1[b], d2[b] if tv else None) == 0
        torch.cuda.synchronize()
        got = dout.cpu().numpy()
    else:
        for b in range(blocks):
            out = np.zeros((channels, pts), np.float32)
            assert p.convolution(out, x1[b], x2[b] if tv else None) == 0
            got[b] = out
    want = np.zeros_like(got)
    for b in range(blocks):
        for c in range(channels):
            want[b, c] = os_[c].convolution(x1[b, c], x2[b, c]) if tv else os_[c].convolution(x1[b, c])
    check("pconv", "pts=%d parts=%d ch=%d %s %s blocks=%d kernel=%s" % (pts, nparts, channels, "tv" if tv else "static", "dev" if dev else "host", blocks, p.kernel_name()),
          got, want, max(1e-6, 1e-7 * float(np.sqrt(nparts))))   # float32 sums of nparts products in two orders: 3.5e-7 at 94 partitions (profiles/conv_accuracy_r05.txt)


def case_pconv_deep():
    """k_pconv_fused with fewer channels than CUs and 32 partitions or more (static response): the first eight partitions are
    requested in front of the forward chain; a few channels against the oracle, the ring wrapping when there are enough blocks"""
    pts = int(rng.choice([512, 1024]))
    nparts = int(rng.integers(32, 49))
    channels = int(rng.integers(137, 256))
    blocks = int(rng.choice([3, 5, nparts + 3]))
    cvs = pts * nparts
    p = fa.Clpconv(0, cvs, pts, channels=channels)
    assert p.get_cl_err() == 0 and p.nparts == nparts
    ir = (sym((channels, cvs)) * (0.5 / np.sqrt(cvs))).astype(np.float32)
    assert p.push_ir(ir) == 0
    x1 = sym((blocks, channels, pts))
    d1 = torch.from_numpy(x1).cuda()
    dout = torch.zeros((blocks, channels, pts), device="cuda")
    for b in range(blocks):
        assert p.process_device(dout[b], d1[b], None) == 0
    torch.cuda.synchronize()
    got = dout.cpu().numpy()
    pick = sorted({0, channels - 1, int(rng.integers(0, channels)), int(rng.integers(0, channels))})
    want = np.zeros((blocks, len(pick), pts), np.float32)
    for k, c in enumerate(pick):
        o = oracle.Pconv(cvs, pts)
        o.push_ir(ir[c])
        for b in range(blocks):
            want[b, k] = o.convolution(x1[b, c])
    check("pconv", "pts=%d parts=%d ch=%d static dev blocks=%d kernel=%s (deep queue)" % (pts, nparts, channels, blocks, p.kernel_name()),
          got[:, pick], want, max(1e-6, 1e-7 * float(np.sqrt(nparts))))


def case_dconv():
    irsize = int(rng.choice([int(rng.integers(1, 64)), int(rng.integers(64, 5000)), int(rng.integers(5000, 200000))]))
    vsize = int(rng.choice([int(rng.integers(1, 130)), int(rng.integers(130, 3000))]))
    if irsize * vsize > (1 << 27):
        vsize = max(1, (1 << 27) // irsize)
    tv = bool(rng.integers(0, 2))
    dev = bool(rng.integers(0, 2))
    blocks = int(rng.integers(2, 8)) + (irsize // vsize + 2 if irsize // vsize < 30 else 0)
    d, o = fa.Cldconv(0, irsize, vsize), oracle.Dconv(irsize, vsize)
    assert d.get_cl_err() == 0
    ir = (sym(irsize) * (0.5 / np.sqrt(irsize))).astype(np.float32)
    assert d.push_ir(ir) == 0
    o.push_ir(ir)
    x1 = sym((blocks, vsize))
    x2 = (sym((blocks, vsize)) * (0.5 / np.sqrt(irsize))).astype(np.float32)
    got = np.zeros((blocks, vsize), np.float32)
    if dev:
        d1, d2 = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
        dout = torch.zeros((blocks, vsize), device="cuda")
        for b in range(blocks):
            assert d.process_device(dout[b], d1[b], d2[b] if tv else None) == 0
        torch.cuda.synchronize()
        got = dout.cpu().numpy()
    else:
        for b in range(blocks):
            out = np.zeros(vsize, np.float32)
            assert d.convolution(out, x1[b], x2[b] if tv else None) == 0
            got[b] = out
    want = np.stack([o.convolution(x1[b], x2[b]) if tv else o.convolution(x1[b]) for b in range(blocks)])
    tol = max(1e-6, 2 * float(np.sqrt(irsize)) * 2.0 ** -24)   # tests/test_gpu_conv.py dconv_tol()
    check("dconv", "irsize=%d vsize=%d %s %s blocks=%d" % (irsize, vsize, "tv" if tv else "static", "dev" if dev else "host", blocks), got, want, tol)


cases = [case_cfft, case_cfft, case_rfft, case_rfft, case_pconv, case_pconv, case_pconv, case_dconv, case_dconv, case_cfft_any, case_cfft_big,
         case_pconv_deep]


def run(budget, seed, silent=False):
    """random cases of all five object kinds until `budget` seconds are spent; AssertionError on the first failure;
    -> (number of cases, per-kind counts)"""
    global rng, quiet
    rng = np.random.default_rng(seed)
    quiet = silent
    counts.clear()
    t0 = time.time()
    k = 0
    while time.time() - t0 < budget:
        cases[k % len(cases)]()
        k += 1
    print("fuzz ok: %d cases in %.0f s (seed %d): %s" % (k, time.time() - t0, seed, counts))
    return k, dict(counts)


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
