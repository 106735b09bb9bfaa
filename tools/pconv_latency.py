"""dev tool: per-block latency of a single-channel partitioned convolution (the Csound use case):
host-pointer calls (Clpconv::convolution as the opcode makes them) and device-resident calls."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import opencl_fft_amd as fa

def run(pts, cvs, tv, iters=300):
    c = fa.Clpconv(0, cvs, pts)
    rng = np.random.default_rng(1)
    ir = (rng.standard_normal(c.nparts * pts) * 0.01).astype(np.float32)
    if not tv:
        assert c.push_ir(ir) == 0
    x = rng.standard_normal(pts).astype(np.float32)
    x2 = rng.standard_normal(pts).astype(np.float32) * 0.01
    out = np.zeros(pts, np.float32)
    call = (lambda: c.convolution(out, x, x2)) if tv else (lambda: c.convolution(out, x))
    for _ in range(20): call()
    t0 = time.perf_counter()
    for _ in range(iters): call()
    host = (time.perf_counter() - t0) / iters * 1e6
    dx, dx2, dout = torch.from_numpy(x).cuda(), torch.from_numpy(x2).cuda(), torch.zeros(pts, device="cuda")
    dcall = (lambda: c.process_device(dout, dx, dx2)) if tv else (lambda: c.process_device(dout, dx))
    for _ in range(20): dcall()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): dcall()
    torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / iters * 1e6
    print("pts %5d  parts %4d  %s: host call %.1f us   device-resident %.1f us per block" % (pts, c.nparts, "tv    " if tv else "static", host, dev), flush=True)

for pts, cvs in [(64, 1 << 13), (128, 1 << 15), (256, 1 << 16), (512, 1 << 16), (512, 1 << 20), (2048, 1 << 18), (8192, 1 << 20)]:
    for tv in (False, True):
        run(pts, cvs, tv)
