"""The float64 evaluations of the convolution formulas (tests/util.py) that tests/test_gpu_conv_accuracy.py measures
HIP and oracle against: checked here against the oracle — itself pinned by the reference's vectors — so that the yardstick
is the reference's formula (ring rules, the half gain of bins 0 and pts, Cldconv's one-sample latency) and not a textbook one."""
import numpy as np

from oracle import oracle
from tests import util
from tests.util import rel_err


def test_pconv_f64_models_follow_the_oracle():
    pts, nparts, blocks = 64, 5, 17          # three times round the ring
    rng = np.random.default_rng(1)
    ir = ((rng.random(pts * nparts, dtype=np.float32) - 0.5) / np.float32(np.sqrt(pts * nparts))).astype(np.float32)
    x = (rng.random(blocks * pts, dtype=np.float32) * 2 - 1).astype(np.float32)
    x2 = ((rng.random(blocks * pts, dtype=np.float32) - 0.5) / 8).astype(np.float32)
    o = oracle.Pconv(pts * nparts, pts)
    o.push_ir(ir)
    w = np.concatenate([o.convolution(x[b * pts:(b + 1) * pts]) for b in range(blocks)])
    assert max(rel_err(w, util.pconv_f64(ir, x, pts))) < 5e-7
    o = oracle.Pconv(pts * nparts, pts)
    w = np.concatenate([o.convolution(x[b * pts:(b + 1) * pts], x2[b * pts:(b + 1) * pts]) for b in range(blocks)])
    assert max(rel_err(w, util.pconv_tv_f64(x, x2, pts, nparts))) < 5e-7
    # the reference's vectors themselves (G7: pts 8, 4 partitions, 12 blocks; G9 time-varying)
    g = util.golden
    assert max(rel_err(g("g7_pconv_p8_n4_out"), util.pconv_f64(g("g7_pconv_p8_n4_ir"), g("g7_pconv_p8_n4_in"), 8))) < 1e-6
    assert max(rel_err(g("g7_pconv_p8_n4_ones_out"), util.pconv_f64(g("g7_pconv_p8_n4_ones_ir"), g("g7_pconv_p8_n4_ones_in"), 8))) < 1e-6
    assert max(rel_err(g("g9_tvconv_p8_n4_out"), util.pconv_tv_f64(g("g9_tvconv_p8_n4_in1"), g("g9_tvconv_p8_n4_in2"), 8, 4))) < 1e-6
    assert max(rel_err(g("g9_tvconv_p256_n5_out"), util.pconv_tv_f64(g("g9_tvconv_p256_n5_in1"), g("g9_tvconv_p256_n5_in2"), 256, 5))) < 1e-6


def test_dconv_f64_model_and_tap_order_follow_the_oracle():
    irsize, vsize = 1024, 64
    blocks = irsize // vsize + 3
    rng = np.random.default_rng(2)
    ir = ((rng.random(irsize, dtype=np.float32) - 0.5) / np.float32(np.sqrt(irsize))).astype(np.float32)
    x = (rng.random(blocks * vsize, dtype=np.float32) * 2 - 1).astype(np.float32)
    o = oracle.Dconv(irsize, vsize)
    o.push_ir(ir)
    for b in range(blocks):
        w = o.convolution(x[b * vsize:(b + 1) * vsize])
    truth, seq = util.dconv_last_block(ir, x, vsize, blocks - 1, np.arange(vsize))
    assert np.array_equal(w, seq)                      # float32 products added one by one in tap order IS the oracle
    assert max(rel_err(seq, truth)) < 2e-6
