"""dev check of the fused real-131072 forward kernel against the oracle (GPU box)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opencl_fft_amd as fa
from oracle import oracle
from tests.util import rel_err

rng = np.random.default_rng(5)
size = 131072
for batch in (65, 256, 300, 1037):
    r = (rng.random((batch, size), dtype=np.float32) * 2 - 1).astype(np.float32)
    f = fa.Clrfft(0, size, True)
    d = torch.from_numpy(r.copy()).cuda()
    assert f.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    spec = d.cpu().numpy().view(np.complex64).reshape(batch, size // 2)
    pick = sorted(set([0, 1, batch - 1, batch // 2, 255 % batch, 256 % batch]))
    want = oracle.rfft_forward(r[pick])
    l2, mx = rel_err(spec[pick], want)
    print("batch", batch, f.kernel_name(), "relL2 %.2e max %.2e" % (l2, mx), flush=True)
    if not (l2 < 1e-6 and mx < 1e-6):
        diff = np.abs(spec[pick[0]] - want[0])
        bad = np.nonzero(diff > 1e-5 * np.abs(want[0]).max())[0]
        print("bad bins", len(bad), bad[:40])
        print("k1 of bad", sorted(set((bad % 256).tolist()))[:64])
        print("k2 of bad", sorted(set((bad // 256).tolist()))[:64])
        sys.exit(1)
    # out of place
    src = torch.from_numpy(r.copy()).cuda()
    dst = torch.full_like(src, float("nan"))
    assert f.exec_device_oop(src, dst, batch) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst, d), "out of place differs"
# timing
batch = 4096
d = torch.rand((batch, size), device="cuda") * 2 - 1
f = fa.Clrfft(0, size, True)
for _ in range(3):
    f.exec_device(d, batch)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    f.exec_device(d, batch)
torch.cuda.synchronize()
dt = (time.time() - t0) / 10
print("real 131072 fwd batch %d: %.3f ms  %.2f TB/s alg (frac %.3f)" % (batch, dt * 1e3, batch * size * 8 / dt / 1e12, batch * size * 8 / dt / 8e12))
