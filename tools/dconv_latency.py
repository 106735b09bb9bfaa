"""dev tool: per-block latency of the direct convolution (Cldconv, what the opcodes use when parts == 1):
host-pointer calls as the opcode makes them; device time per block from back-to-back device-resident calls
where the library offers them."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import opencl_fft_amd as fa

def run(irsize, vsize, tv, iters=300):
    c = fa.Cldconv(0, irsize, vsize)
    rng = np.random.default_rng(1)
    assert c.push_ir((rng.standard_normal(irsize) * 0.01).astype(np.float32)) == 0
    x = rng.standard_normal(vsize).astype(np.float32)
    x2 = rng.standard_normal(vsize).astype(np.float32) * 0.01
    out = np.zeros(vsize, np.float32)
    call = (lambda: c.convolution(out, x, x2)) if tv else (lambda: c.convolution(out, x))
    for _ in range(20): assert call() == 0
    t0 = time.perf_counter()
    for _ in range(iters): call()
    host = (time.perf_counter() - t0) / iters * 1e6
    dev = float("nan")
    if hasattr(c, "process_device"):
        dx, dx2, dout = torch.from_numpy(x).cuda(), torch.from_numpy(x2).cuda(), torch.zeros(vsize, device="cuda")
        dcall = (lambda: c.process_device(dout, dx, dx2)) if tv else (lambda: c.process_device(dout, dx))
        for _ in range(20): assert dcall() == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters): dcall()
        torch.cuda.synchronize()
        dev = (time.perf_counter() - t0) / iters * 1e6
    print("irsize %6d  vsize %4d  %s: host call %.1f us   device-resident %.1f us per block" % (irsize, vsize, "tv    " if tv else "static", host, dev), flush=True)

for irsize, vsize in [(256, 64), (1024, 64), (16384, 64), (96000, 64), (96000, 1024), (1 << 20, 256)]:
    for tv in (False, True):
        run(irsize, vsize, tv)
