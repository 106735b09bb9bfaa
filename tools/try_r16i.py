"""dev check of the fused real-131072 inverse kernel against the oracle (GPU box)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opencl_fft_amd as fa
from oracle import oracle
from tests.util import rel_err

rng = np.random.default_rng(6)
size, m = 131072, 65536
for batch in (65, 256, 300, 1037):
    c = ((rng.random((batch, m), dtype=np.float32) * 2 - 1) + 1j * (rng.random((batch, m), dtype=np.float32) * 2 - 1)).astype(np.complex64)
    c *= np.float32(1.0 / 256)
    f = fa.Clrfft(0, size, False)
    d = torch.from_numpy(c.view(np.float32).reshape(batch, size).copy()).cuda()
    assert f.exec_device(d, batch) == 0
    torch.cuda.synchronize()
    got = d.cpu().numpy().reshape(batch, size)
    pick = sorted(set([0, 1, batch - 1, batch // 2, 255 % batch, 256 % batch]))
    want = oracle.rfft_inverse(c[pick])
    l2, mx = rel_err(got[pick], want)
    print("batch", batch, f.kernel_name(), "relL2 %.2e max %.2e" % (l2, mx), flush=True)
    if not (l2 < 1e-6 and mx < 1e-6):
        # which input bins matter is not visible in the output; compare against a transform with single bins instead
        print("FAIL"); sys.exit(1)
    src = torch.from_numpy(c.view(np.float32).reshape(batch, size).copy()).cuda()
    dst = torch.full_like(src, float("nan"))
    assert f.exec_device_oop(src, dst, batch) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst, d), "out of place differs"
batch = 4096
d = torch.rand((batch, size), device="cuda") * 2 - 1
f = fa.Clrfft(0, size, False)
d.mul_(1.0 / 256)
for _ in range(2):
    f.exec_device(d, batch); d.mul_(1.0 / 131072)
torch.cuda.synchronize()
ts = []
for _ in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f.exec_device(d, batch); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b)); d.mul_(1.0 / 131072)
dt = sorted(ts)[len(ts) // 2] * 1e-3
print("real 131072 inv batch %d: %.3f ms  %.2f TB/s alg (frac %.3f)" % (batch, dt * 1e3, batch * size * 8 / dt / 1e12, batch * size * 8 / dt / 8e12))
