"""GPU parity tests for partitioned and direct convolution (through the C ABI)
against the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest

import opencl_fft_amd as fa
from oracle import oracle
from tests import util
from tests.util import assert_parity, golden

pytestmark = pytest.mark.gpu

# The reference sums partitions with float CAS atomics in arbitrary order
# (cl_conv_kernels.h:116-117); ours sums in ascending partition order in
# registers.  Same tolerance as the spectra (north_star's 1e-6), applied to the time-domain output: measured against a
# float64 evaluation of the same formulas (tests/test_gpu_conv_accuracy.py, profiles/conv_accuracy_r05.txt) the HIP result
# is 2.3e-7 off, the oracle 2.9e-7, the two 3.3-3.7e-7 apart at config 4's 94 partitions.
CTOL = 1e-6


def dconv_tol(irsize):
    """HIP vs oracle for Cldconv: two float32 sums of `irsize` products in different orders (cl_dconv.cpp:32-43 leaves the
    order to its atomics).  Against float64 the HIP sums (chunks, then a tree) stay at 3-5e-7 for every length while the
    oracle's one-by-one sum grows like sqrt(irsize): 4.6e-7 at 1024 taps, 5.8e-6 at 96000, 1.6e-5 at 2^20 — the difference
    between the two IS the oracle's rounding (profiles/conv_accuracy_r05.txt), which this bound follows."""
    return max(1e-6, 2 * float(np.sqrt(irsize)) * 2.0 ** -24)


def _run(p, blocks, pts, in1, in2=None):
    outs = []
    for b in range(blocks):
        o = np.zeros((p.channels, pts), np.float32)
        a = in1[..., b * pts:(b + 1) * pts]
        if in2 is None:
            assert p.convolution(o, a) == 0
        else:
            assert p.convolution(o, a, in2[..., b * pts:(b + 1) * pts]) == 0
        outs.append(o)
    return np.concatenate(outs, axis=-1)


@pytest.mark.parametrize("tag,pts,nparts,blocks", [
    ("g7_pconv_p8_n4", 8, 4, 12), ("g7_pconv_p8_n4_ones", 8, 4, 12),
    ("g7_pconv_p2_n3", 2, 3, 9), ("g7_pconv_p64_n1", 64, 1, 4),
    ("g8_pconv_p1024_n8", 1024, 8, 24)])
def test_pconv_vs_reference(tag, pts, nparts, blocks):
    ir, inp = golden(tag + "_ir"), golden(tag + "_in")
    p = fa.Clpconv(0, pts * nparts, pts)
    assert p.get_cl_err() == 0 and p.nparts == nparts
    assert p.push_ir(ir) == 0
    out = _run(p, blocks, pts, inp[None, :])[0]
    assert_parity(out, golden(tag + "_out"), tol=CTOL, what=tag)


def test_pconv_ring_indices_bit_exact():
    p, o = fa.Clpconv(0, 32, 8), oracle.Pconv(32, 8)
    assert (p.nparts, p.wp, p.wp2) == (o.nparts, o.wp, o.wp2) == (4, 0, 3)
    z = np.zeros(32, np.float32)
    p.push_ir(z)
    o.push_ir(z)
    assert p.wp2 == o.wp2 == 3
    out = np.zeros((1, 8), np.float32)
    for _ in range(9):
        p.convolution(out, z[:8])
        o.convolution(z[:8])
        assert (p.wp, p.wp2) == (o.wp, o.wp2)
    q, oq = fa.Clpconv(0, 32, 8), oracle.Pconv(32, 8)
    for _ in range(7):
        q.convolution(out, z[:8], z[:8])
        oq.convolution(z[:8], z[:8])
        assert (q.wp, q.wp2) == (oq.wp, oq.wp2)


def test_pconv_device_overlap_contract():
    """clfa_pconv_process_dev: `out` overlapping an input is refused on the one-launch routes (other channels' workgroups may
    still be reading); the launch chain of partitions above 4096 samples has read every input before it writes and runs in
    place, block for block what the oracle gives (the reference's own calls are on host arrays, cl_conv.cpp:393-458)"""
    import torch
    for pts, nparts in ((1024, 3), (256, 2)):
        p = fa.Clpconv(0, pts * nparts, pts)
        assert p.kernel_name() in ("k_pconv_fused", "k_pconv_coop")
        buf = torch.zeros(2 * pts, device="cuda")
        assert p.process_device(buf[:pts], buf[:pts]) == -30
        assert p.process_device(buf[pts // 2:pts // 2 + pts], buf[:pts]) == -30
        assert p.process_device(buf[pts:], buf[:pts], buf[1:pts + 1]) == -30
        assert (p.wp, p.wp2) == (0, nparts - 1)            # a refused call moves no ring index
        assert p.process_device(buf[pts:], buf[:pts]) == 0
    pts, nparts, blocks = 8192, 2, 4
    s = util.lcg_half(99, pts * nparts + pts * blocks)
    ir, x = s[:pts * nparts], s[pts * nparts:]
    p, o = fa.Clpconv(0, pts * nparts, pts), oracle.Pconv(pts * nparts, pts)
    assert p.kernel_name() not in ("k_pconv_fused", "k_pconv_coop")
    assert p.push_ir(ir) == 0
    o.push_ir(ir)
    d = torch.from_numpy(x.copy()).cuda()
    for b in range(blocks):
        blk = d[b * pts:(b + 1) * pts]
        assert p.process_device(blk, blk) == 0             # in place
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    for b in range(blocks):
        assert_parity(got[b * pts:(b + 1) * pts], o.convolution(x[b * pts:(b + 1) * pts]), tol=CTOL, what="in-place block %d" % b)


@pytest.mark.parametrize("pts,nparts,channels,blocks", [(2, 1, 3, 4), (16, 5, 7, 12), (256, 3, 5, 8), (1024, 6, 3, 9),
                                                        (4096, 2, 2, 5), (8192, 2, 1, 4), (16384, 2, 2, 4),
                                                        (32768, 3, 1, 5)])
def test_pconv_multichannel_vs_oracle(pts, nparts, channels, blocks):
    """`channels` independent instances in one object == that many oracle objects"""
    s = util.lcg_half(7 + pts, channels * (pts * nparts + pts * blocks))
    ir = s[:channels * pts * nparts].reshape(channels, pts * nparts)
    x = s[channels * pts * nparts:].reshape(channels, pts * blocks)
    p = fa.Clpconv(0, pts * nparts, pts, channels=channels)
    assert p.get_cl_err() == 0
    assert p.push_ir(ir) == 0
    out = _run(p, blocks, pts, x)
    for c in range(channels):
        o = oracle.Pconv(pts * nparts, pts)
        o.push_ir(ir[c])
        want = np.concatenate([o.convolution(x[c, b * pts:(b + 1) * pts]) for b in range(blocks)])
        assert_parity(out[c], want, tol=CTOL, what="channel %d" % c)


def test_pconv_nparts_floor_and_bad_geometry():
    assert fa.Clpconv(0, 96000, 1024).nparts == 93        # remainder dropped, cl_conv.cpp:143
    msgs = []
    bad = fa.Clpconv(0, 100, 24, errs=lambda s, d: msgs.append((s, d)), uData="u")
    assert bad.get_cl_err() == -30 and msgs == [("Invalid value", "u")]
    assert bad.convolution(np.zeros((1, 24), np.float32), np.zeros(24, np.float32)) == -30


@pytest.mark.parametrize("tag,pts,nparts,blocks", [("g9_tvconv_p8_n4", 8, 4, 12), ("g9_tvconv_p256_n5", 256, 5, 14)])
def test_tvconv_vs_reference(tag, pts, nparts, blocks):
    in1, in2 = golden(tag + "_in1"), golden(tag + "_in2")
    p = fa.Clpconv(0, pts * nparts, pts)
    out = _run(p, blocks, pts, in1[None, :], in2[None, :])[0]
    assert_parity(out, golden(tag + "_out"), tol=CTOL, what=tag)


def test_tvconv_multichannel_vs_oracle():
    pts, nparts, channels, blocks = 64, 4, 3, 11
    s = util.lcg_half(9, 2 * channels * pts * blocks)
    x1 = s[:channels * pts * blocks].reshape(channels, -1)
    x2 = s[channels * pts * blocks:].reshape(channels, -1)
    p = fa.Clpconv(0, pts * nparts, pts, channels=channels)
    out = _run(p, blocks, pts, x1, x2)
    for c in range(channels):
        o = oracle.Pconv(pts * nparts, pts)
        want = np.concatenate([o.convolution(x1[c, b * pts:(b + 1) * pts], x2[c, b * pts:(b + 1) * pts])
                               for b in range(blocks)])
        assert_parity(out[c], want, tol=CTOL, what="channel %d" % c)


def test_pconv_device_resident_config4_shape():
    """config 4 geometry at reduced channel count: pts=1024, cvs=96256 (94 partitions)"""
    import torch
    pts, nparts, channels, blocks = 1024, 94, 4, 6
    s = util.lcg_half(11, channels * (pts * nparts + pts * blocks))
    ir = (s[:channels * pts * nparts] / np.float32(np.sqrt(pts * nparts))).reshape(channels, -1)
    x = (2 * s[channels * pts * nparts:]).reshape(channels, -1)
    p = fa.Clpconv(0, 96256, pts, channels=channels)
    assert p.nparts == nparts
    d_ir = torch.from_numpy(ir).cuda()
    assert p.push_ir_device(d_ir) == 0
    d_x = torch.from_numpy(x).cuda()
    outs = []
    for b in range(blocks):
        d_in = d_x[:, b * pts:(b + 1) * pts].contiguous()
        d_out = torch.empty((channels, pts), device="cuda")
        assert p.process_device(d_out, d_in) == 0
        outs.append(d_out)
    torch.cuda.synchronize()
    out = torch.cat(outs, dim=1).cpu().numpy()
    for c in range(channels):
        o = oracle.Pconv(96256, pts)
        o.push_ir(ir[c])
        want = np.concatenate([o.convolution(x[c, b * pts:(b + 1) * pts]) for b in range(blocks)])
        assert_parity(out[c], want, tol=CTOL, what="channel %d" % c)


# ---- direct convolution ----------------------------------------------------------------------

def test_dconv_vs_reference_and_oracle():
    ir, inp = golden("g10_dconv_ir"), golden("g10_dconv_in")
    d, o = fa.Cldconv(0, 16, 8), oracle.Dconv(16, 8)
    assert d.get_cl_err() == 0
    assert d.push_ir(ir) == 0
    o.push_ir(ir)
    outs, wants = [], []
    for b in range(2):
        out = np.zeros(8, np.float32)
        assert d.convolution(out, inp[b * 8:(b + 1) * 8]) == 0
        outs.append(out)
        wants.append(o.convolution(inp[b * 8:(b + 1) * 8]))
    assert_parity(np.concatenate(outs), np.concatenate(wants), tol=CTOL, what="vs oracle")
    assert_parity(np.concatenate(outs), golden("g10_dconv_out"), tol=CTOL, what="vs reference")


@pytest.mark.parametrize("tag,irsize,vsize,tv", [("g11_dconv_i16_v8", 16, 8, False), ("g11_dconv_i1024_v64", 1024, 64, False),
                                                 ("g11_dconv_i64_v64", 64, 64, False), ("g11_tvdconv_i16_v8", 16, 8, True),
                                                 ("g11_tvdconv_i256_v32", 256, 32, True)])
def test_dconv_vs_reference_over_ring_cycles(tag, irsize, vsize, tv):
    """G11 (tests/test_oracle_golden.py): HIP Cldconv against the unmodified reference over several ring
    cycles, every block from the first one that no longer depends on the reference's uninitialised memory"""
    ir, x, ref = golden(tag + "_ir"), golden(tag + "_in"), golden(tag + "_out")
    x2 = golden(tag + "_in2") if tv else None
    d = fa.Cldconv(0, irsize, vsize)
    assert d.get_cl_err() == 0 and d.push_ir(ir) == 0
    tol = dconv_tol(irsize)   # the reference's CAS-atomic sum is order-dependent
    first = irsize // vsize + 1
    for b in range(x.size // vsize):
        sl = slice(b * vsize, (b + 1) * vsize)
        out = np.zeros(vsize, np.float32)
        assert (d.convolution(out, x[sl], x2[sl]) if tv else d.convolution(out, x[sl])) == 0
        if b >= first:
            assert_parity(out, ref[sl], tol=tol, what="%s block %d" % (tag, b))


@pytest.mark.parametrize("irsize,vsize,blocks", [(16, 8, 9), (1000, 64, 40), (5, 7, 6), (4096, 32, 10)])
def test_dconv_ring_wrap_vs_oracle(irsize, vsize, blocks):
    s = util.lcg_half(3, irsize + vsize * blocks)
    ir, x = s[:irsize], s[irsize:]
    d, o = fa.Cldconv(0, irsize, vsize), oracle.Dconv(irsize, vsize)
    d.push_ir(ir)
    o.push_ir(ir)
    for b in range(blocks):
        out = np.zeros(vsize, np.float32)
        assert d.convolution(out, x[b * vsize:(b + 1) * vsize]) == 0
        assert_parity(out, o.convolution(x[b * vsize:(b + 1) * vsize]), tol=dconv_tol(irsize), what="block %d" % b)


@pytest.mark.parametrize("irsize,vsize,blocks,tv", [(96000, 64, 6, False), (5000, 100, 8, True), (3, 200, 4, False),
                                                    (70000, 256, 5, True), (64, 64, 5, False), (300000, 16, 3, False),
                                                    (4500, 40000, 2, True)])   # (more tiles than output blocks)
def test_dconv_device_resident_blocks_vs_oracle(irsize, vsize, blocks, tv):
    """the device-resident entry point (one launch per block; responses of more than one tap chunk hand their partial
    sums to the last workgroup to arrive): a stream of blocks issued back to back, checked afterwards; the host entry
    point then continues on the same object"""
    import torch
    s = util.lcg_half(11, irsize + 2 * vsize * (blocks + 1))
    ir, x1, x2 = s[:irsize], s[irsize:irsize + vsize * (blocks + 1)], s[irsize + vsize * (blocks + 1):]
    d, o = fa.Cldconv(0, irsize, vsize), oracle.Dconv(irsize, vsize)
    assert d.get_cl_err() == 0 and d.push_ir(ir) == 0
    o.push_ir(ir)
    dx1, dx2 = torch.from_numpy(x1.copy()).cuda(), torch.from_numpy(x2.copy()).cuda()
    dout = torch.zeros((blocks, vsize), device="cuda")
    for b in range(blocks):
        sl = slice(b * vsize, (b + 1) * vsize)
        assert d.process_device(dout[b], dx1[sl], dx2[sl] if tv else None) == 0
    torch.cuda.synchronize()
    got = dout.cpu().numpy()
    tol = dconv_tol(irsize)
    for b in range(blocks):
        sl = slice(b * vsize, (b + 1) * vsize)
        want = o.convolution(x1[sl], x2[sl]) if tv else o.convolution(x1[sl])
        assert_parity(got[b], want, tol=tol, what="block %d" % b)
    sl = slice(blocks * vsize, (blocks + 1) * vsize)
    out = np.zeros(vsize, np.float32)
    assert (d.convolution(out, x1[sl], x2[sl]) if tv else d.convolution(out, x1[sl])) == 0
    assert_parity(out, o.convolution(x1[sl], x2[sl]) if tv else o.convolution(x1[sl]), tol=tol, what="host call after")
    assert d.process_device(dout[0], dout[0]) == -30   # out must not be an input
    if vsize >= 4:                                     # ... nor overlap one partly (one caller buffer, out = in + k)
        buf = torch.zeros(2 * vsize, device="cuda")
        assert d.process_device(buf[vsize // 2:vsize // 2 + vsize], buf[:vsize]) == -30
        assert d.process_device(buf[:vsize], buf[vsize // 2:vsize // 2 + vsize]) == -30
        assert d.process_device(buf[vsize:], buf[:vsize], buf[1:vsize + 1] if tv else None) == (-30 if tv else 0)


def test_dconv_handoff_under_load():
    """k_dconv_block's hand-over of the tap chunks' partial sums (agent-scope stores, arrival counter per output block,
    the last workgroup adds in fixed order) under UNEVEN load: the same blocks once on an idle GPU and once while another
    stream keeps every CU busy with large transforms — the two runs must agree bit for bit."""
    import torch
    blocks = 120
    # the last geometry launches 256 x 4 workgroups — more than the chip has CUs, where the last workgroup also runs an
    # agent-scope acquire (conv_kernels.hip, handover_acquire)
    for irsize, vsize, tv in ((96000, 64, False), (70000, 256, True), (300000, 16, False), (1 << 20, 256, False)):
        blocks = 120 if irsize < (1 << 20) else 40
        g = torch.Generator(device="cuda").manual_seed(irsize + vsize)
        ir = ((torch.rand(irsize, generator=g, device="cuda") - 0.5) / irsize ** 0.5).cpu().numpy()
        x1 = torch.rand((blocks, vsize), generator=g, device="cuda") * 2 - 1
        x2 = (torch.rand((blocks, vsize), generator=g, device="cuda") - 0.5) / irsize ** 0.5
        big = torch.rand((512, 65536, 2), device="cuda") * 2 - 1
        f, i = fa.Clcfft(0, 65536, True), fa.Clcfft(0, 65536, False)
        outs = []
        for loaded in (False, True):
            d = fa.Cldconv(0, irsize, vsize)
            assert d.get_cl_err() == 0 and d.push_ir(ir) == 0
            y = torch.empty((blocks, vsize), device="cuda")
            torch.cuda.synchronize()
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            for b in range(blocks):
                if loaded and b % 4 == 0:
                    assert (f if (b // 4) % 2 == 0 else i).exec_device(big, 512, sb.cuda_stream) == 0
                assert d.process_device(y[b], x1[b], x2[b] if tv else None, sa.cuda_stream) == 0
            torch.cuda.synchronize()
            outs.append(y)
        assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), (irsize, vsize, tv)
        o = oracle.Dconv(irsize, vsize)
        o.push_ir(ir)
        xs1, xs2 = x1.cpu().numpy(), x2.cpu().numpy()
        want = np.stack([o.convolution(xs1[b], xs2[b]) if tv else o.convolution(xs1[b]) for b in range(6)])
        assert_parity(outs[1][:6].cpu().numpy(), want, tol=dconv_tol(irsize), what="first blocks vs oracle")


def test_dconv_time_varying_vs_oracle():
    irsize, vsize, blocks = 32, 8, 12
    s = util.lcg_half(5, 2 * vsize * blocks)
    x1, x2 = s[:vsize * blocks], s[vsize * blocks:]
    d, o = fa.Cldconv(0, irsize, vsize), oracle.Dconv(irsize, vsize)
    for b in range(blocks):
        out = np.zeros(vsize, np.float32)
        sl = slice(b * vsize, (b + 1) * vsize)
        assert d.convolution(out, x1[sl], x2[sl]) == 0
        want = o.convolution(x1[sl], x2[sl])
        assert np.max(np.abs(out - want)) <= dconv_tol(irsize) * max(1e-3, np.max(np.abs(want)))


def test_config4_full_size_properties():
    """config 4 of BASELINE.json at full size: 256 channels, pts 1024, 94 partitions, device-resident.
    Three channels against the oracle, linearity (conv(a x) = a conv(x)) and channel independence on all."""
    import torch
    pts, cvs, channels, blocks = 1024, 96256, 256, 4
    g = torch.Generator(device="cuda").manual_seed(4)
    ir = (torch.rand((channels, cvs), generator=g, device="cuda") - 0.5) / (cvs ** 0.5)
    x = torch.rand((blocks, channels, pts), generator=g, device="cuda") * 2 - 1
    gain = torch.linspace(0.5, 2.0, channels, device="cuda").reshape(1, channels, 1)

    def run(inp):
        p = fa.Clpconv(0, cvs, pts, channels=channels)
        assert p.get_cl_err() == 0 and p.nparts == 94
        assert p.push_ir_device(ir) == 0
        outs = []
        for b in range(blocks):
            o = torch.empty((channels, pts), device="cuda")
            assert p.process_device(o, inp[b].contiguous()) == 0
            outs.append(o)
        torch.cuda.synchronize()
        return torch.stack(outs)

    y = run(x)
    y2 = run(x * gain)
    rel = float(((y2 - y * gain).double().norm() / (y * gain).double().norm()))
    assert rel < 1e-6, rel
    # channel independence: permuting the channels permutes the outputs
    perm = torch.randperm(channels, generator=torch.Generator().manual_seed(1)).cuda()
    ir_saved = ir
    ir = ir_saved[perm].contiguous()
    y3 = run(x[:, perm].contiguous())
    ir = ir_saved
    assert float((y3 - y[:, perm]).abs().max()) <= CTOL * float(y.abs().max())
    ir_h, x_h, y_h = ir.cpu().numpy(), x.cpu().numpy(), y.cpu().numpy()
    for c in (0, 127, 255):
        o = oracle.Pconv(cvs, pts)
        o.push_ir(ir_h[c])
        want = np.stack([o.convolution(x_h[b, c]) for b in range(blocks)])
        assert_parity(y_h[:, c], want, tol=CTOL, what="channel %d" % c)


@pytest.mark.parametrize("pts,nparts,channels,blocks,tv", [(512, 5, 162, 7, False), (1024, 4, 160, 6, True),
                                                          (2048, 3, 170, 5, False), (4096, 2, 160, 4, True),
                                                          (1024, 6, 260, 5, False),
                                                          # fewer channels than CUs, static response, 32 partitions and more: the
                                                          # first eight are requested in front of the forward chain (the ring wraps)
                                                          (1024, 40, 160, 47, False), (512, 33, 200, 40, False),
                                                          (1024, 32, 140, 5, False), (1024, 36, 170, 6, True)])
def test_pconv_fused_block_kernel_vs_oracle(pts, nparts, channels, blocks, tv):
    """enough channels to select the one-launch-per-block kernel (forward + MAC + inverse per channel)"""
    s = util.lcg_half(21 + pts, channels * (pts * nparts + 2 * pts * blocks))
    ir = s[:channels * pts * nparts].reshape(channels, pts * nparts)
    x1 = s[channels * pts * nparts:channels * (pts * nparts + pts * blocks)].reshape(channels, pts * blocks)
    x2 = s[channels * (pts * nparts + pts * blocks):].reshape(channels, pts * blocks)
    p = fa.Clpconv(0, pts * nparts, pts, channels=channels)
    assert p.get_cl_err() == 0
    if not tv:
        assert p.push_ir(ir) == 0
    out = _run(p, blocks, pts, x1, x2 if tv else None)
    for c in (0, 1, channels // 2, channels - 1):
        o = oracle.Pconv(pts * nparts, pts)
        if not tv:
            o.push_ir(ir[c])
        want = np.concatenate([o.convolution(x1[c, b * pts:(b + 1) * pts], x2[c, b * pts:(b + 1) * pts] if tv else None)
                               for b in range(blocks)])
        assert_parity(out[c], want, tol=CTOL, what="channel %d" % c)
    assert (p.wp, p.wp2) == (blocks % nparts, (nparts - 1 - (blocks if tv else 0)) % nparts)


def test_config4_full_size_ring_wraps():
    """config 4 at full size for 110 blocks — the ring of 94 spectra wraps — eight channels against the
    oracle on every block, ring indices as the reference's (cl_conv.cpp:424)"""
    import torch
    pts, cvs, channels, blocks = 1024, 96256, 256, 110
    g = torch.Generator(device="cuda").manual_seed(44)
    ir = (torch.rand((channels, cvs), generator=g, device="cuda") - 0.5) / (cvs ** 0.5)
    x = torch.rand((blocks, channels, pts), generator=g, device="cuda") * 2 - 1
    p = fa.Clpconv(0, cvs, pts, channels=channels)
    assert p.get_cl_err() == 0 and p.nparts == 94
    assert p.push_ir_device(ir) == 0
    y = torch.empty((blocks, channels, pts), device="cuda")
    for b in range(blocks):
        assert p.process_device(y[b], x[b]) == 0
    torch.cuda.synchronize()
    assert (p.wp, p.wp2) == (blocks % 94, 93)
    pick = [0, 1, 31, 100, 128, 199, 254, 255]
    ir_h, x_h, y_h = ir[pick].cpu().numpy(), x[:, pick].cpu().numpy(), y[:, pick].cpu().numpy()
    for k, c in enumerate(pick):
        o = oracle.Pconv(cvs, pts)
        o.push_ir(ir_h[k])
        want = np.stack([o.convolution(x_h[b, k]) for b in range(blocks)])
        assert_parity(y_h[:, k], want, tol=CTOL, what="channel %d, all blocks" % c)
        assert_parity(y_h[94:, k], want[94:], tol=CTOL, what="channel %d, blocks after the wrap" % c)


def test_pconv_push_ir_device_ragged_cvs():
    """a (channels, cvs) device tensor with cvs % pts != 0 (96000 / 1024 -> 93 partitions, cl_conv.cpp:143):
    every channel is read at its own row stride — identical to the host push_ir of the same rows"""
    import torch
    pts, cvs, channels, blocks = 1024, 96000, 3, 4
    rng = np.random.default_rng(17)
    ir = ((rng.random((channels, cvs), dtype=np.float32) - 0.5) / np.float32(np.sqrt(cvs))).astype(np.float32)
    x = (rng.random((blocks, channels, pts), dtype=np.float32) * 2 - 1).astype(np.float32)
    a, b = fa.Clpconv(0, cvs, pts, channels=channels), fa.Clpconv(0, cvs, pts, channels=channels)
    assert a.nparts == 93 and b.nparts == 93
    assert a.push_ir(ir) == 0
    d_ir = torch.from_numpy(ir).cuda()
    assert b.push_ir_device(d_ir) == 0
    torch.cuda.synchronize()
    for blk in range(blocks):
        oa, ob = np.zeros((channels, pts), np.float32), np.zeros((channels, pts), np.float32)
        assert a.convolution(oa, x[blk]) == 0 and b.convolution(ob, x[blk]) == 0
        assert np.array_equal(oa.view(np.uint32), ob.view(np.uint32)), "block %d" % blk
    # shapes the object cannot take are refused, not read from the wrong offsets
    assert b.push_ir_device(d_ir[:, :1024]) == -30
    assert b.push_ir_device(d_ir[:2]) == -30
    assert b.push_ir_device(d_ir.double()) == -30


@pytest.mark.parametrize("pts,nparts,channels,blocks,tv", [(512, 5, 1, 13, False), (512, 128, 1, 6, True), (512, 33, 4, 70, False),
                                                          (1024, 8, 3, 20, True), (2048, 7, 2, 16, False),
                                                          (4096, 3, 1, 8, True), (1024, 94, 2, 100, False),
                                                          (4096, 2, 40, 5, False),
                                                          # long filters: the partition axis cut into segments as well
                                                          (512, 600, 1, 8, True), (1024, 400, 2, 6, False),
                                                          (512, 2048, 1, 5, False),
                                                          # partitions below 512 samples (low-latency audio blocks)
                                                          (32, 5, 1, 12, False), (64, 9, 2, 20, True), (128, 16, 1, 10, False),
                                                          (256, 5, 3, 14, True), (64, 1, 1, 4, False),
                                                          # many channels: fewer, wider bin slices per channel
                                                          (512, 6, 100, 5, False), (2048, 3, 9, 6, True), (1024, 5, 33, 7, True),
                                                          # ... streaming their whole share of long rings (no segments)
                                                          (1024, 94, 24, 3, False), (1024, 94, 100, 2, True)])
def test_pconv_cooperative_block_kernel_vs_oracle(pts, nparts, channels, blocks, tv):
    """few channels: one cooperative launch per block (k_pconv_coop: the bins of the multiply-accumulate split over
    the workgroups of a channel, the last workgroup to arrive runs the inverse chain).  Static and time-varying,
    partition counts that are not multiples of the partition rows, rings that wrap more than once, every block of
    every channel against the oracle; twice the same input gives bit-identical output (fixed summation order)."""
    s = util.lcg_half(31 + pts + nparts, channels * (pts * nparts + 2 * pts * blocks))
    ir = s[:channels * pts * nparts].reshape(channels, pts * nparts)
    x1 = s[channels * pts * nparts:channels * (pts * nparts + pts * blocks)].reshape(channels, pts * blocks)
    x2 = s[channels * (pts * nparts + pts * blocks):].reshape(channels, pts * blocks)
    outs = []
    for rep in range(2):
        p = fa.Clpconv(0, pts * nparts, pts, channels=channels)
        assert p.get_cl_err() == 0 and p.kernel_name() == "k_pconv_coop"
        if not tv:
            assert p.push_ir(ir) == 0
        outs.append(_run(p, blocks, pts, x1, x2 if tv else None))
        assert (p.wp, p.wp2) == (blocks % nparts, (nparts - 1 - (blocks if tv else 0)) % nparts)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    for c in sorted({0, channels // 2, channels - 1}):
        o = oracle.Pconv(pts * nparts, pts)
        if not tv:
            o.push_ir(ir[c])
        want = np.concatenate([o.convolution(x1[c, b * pts:(b + 1) * pts], x2[c, b * pts:(b + 1) * pts] if tv else None)
                               for b in range(blocks)])
        assert_parity(outs[0][c], want, tol=CTOL, what="channel %d" % c)


def test_pconv_cooperative_kernel_device_stream_of_blocks():
    """the device-pointer path: 300 blocks queued back to back on one stream without any host synchronisation
    (every launch finds its counters returned to zero by the launch before it), every block against the oracle"""
    import torch
    pts, nparts, blocks = 512, 128, 300
    g = torch.Generator(device="cuda").manual_seed(5)
    ir = (torch.rand((1, pts * nparts), generator=g, device="cuda") - 0.5) / (pts * nparts) ** 0.5
    x = torch.rand((blocks, 1, pts), generator=g, device="cuda") * 2 - 1
    p = fa.Clpconv(0, pts * nparts, pts)
    assert p.kernel_name() == "k_pconv_coop"
    assert p.push_ir_device(ir) == 0
    y = torch.empty((blocks, 1, pts), device="cuda")
    for b in range(blocks):
        assert p.process_device(y[b], x[b]) == 0
    torch.cuda.synchronize()
    o = oracle.Pconv(pts * nparts, pts)
    o.push_ir(ir[0].cpu().numpy())
    xs = x.cpu().numpy()
    want = np.stack([o.convolution(xs[b, 0]) for b in range(blocks)])
    assert_parity(y[:, 0].cpu().numpy(), want, tol=CTOL, what="300 blocks back to back")


def test_pconv_cooperative_handoff_under_load():
    """The hand-over of the accumulator slices between workgroups (agent-scope stores, arrival counter, the last
    workgroup reads) under UNEVEN load: the same 200 blocks once on an idle GPU and once while another stream keeps the
    chip busy with large transforms, so that the cooperative block's workgroups start at different times and share
    their CUs.  The summation order is fixed, so the two runs must agree bit for bit in every output word."""
    import torch
    blocks = 200
    # the last two geometries launch MORE workgroups than the chip has CUs (136 channels x 2 bin slices = 272; 100 x 4 =
    # 400): there the last workgroup also runs an agent-scope acquire (conv_kernels.hip, handover_acquire)
    for pts, nparts, channels, tv in ((512, 128, 1, False), (1024, 94, 3, True), (256, 600, 1, False), (1024, 94, 136, False),
                                      (2048, 24, 100, True)):
        blocks = 200 if channels < 100 else 60
        g = torch.Generator(device="cuda").manual_seed(pts + nparts)
        ir = (torch.rand((channels, pts * nparts), generator=g, device="cuda") - 0.5) / (pts * nparts) ** 0.5
        x1 = torch.rand((blocks, channels, pts), generator=g, device="cuda") * 2 - 1
        x2 = (torch.rand((blocks, channels, pts), generator=g, device="cuda") - 0.5) * 0.05
        big = torch.rand((512, 65536, 2), device="cuda") * 2 - 1
        f, i = fa.Clcfft(0, 65536, True), fa.Clcfft(0, 65536, False)
        outs = []
        for loaded in (False, True):
            p = fa.Clpconv(0, pts * nparts, pts, channels=channels)
            assert p.kernel_name() == "k_pconv_coop"
            if not tv:
                assert p.push_ir_device(ir) == 0
            y = torch.empty((blocks, channels, pts), device="cuda")
            torch.cuda.synchronize()
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            for b in range(blocks):
                if loaded and b % 4 == 0:      # ~0.1 ms of transforms on every CU, overlapping the next blocks
                    assert (f if (b // 4) % 2 == 0 else i).exec_device(big, 512, sb.cuda_stream) == 0
                assert p.process_device(y[b], x1[b], x2[b] if tv else None, sa.cuda_stream) == 0
            torch.cuda.synchronize()
            outs.append(y)
        assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), (pts, nparts, channels, tv)
        for c in sorted({0, channels - 1}):
            o = oracle.Pconv(pts * nparts, pts)
            if not tv:
                o.push_ir(ir[c].cpu().numpy())
            xs1, xs2 = x1[:, c].cpu().numpy(), x2[:, c].cpu().numpy()
            want = np.stack([o.convolution(xs1[b], xs2[b] if tv else None) for b in range(12)])
            assert_parity(outs[1][:12, c].cpu().numpy(), want, tol=CTOL, what="first blocks of channel %d vs oracle" % c)
