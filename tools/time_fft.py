"""dev tool: time the kernels with HIP events (run on the GPU box).  Every loop alternates forward and
inverse plans so the data stay O(1): repeating one direction drives the values to zero (forward is
scaled by 1/n) or to inf, and such data run measurably faster than real ones (lower power)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import opencl_fft_amd as fa

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

def alt(f, i, d, batch):
    k = [0]
    def step():
        (f if k[0] % 2 == 0 else i).exec_device(d, batch)
        k[0] += 1
    return step

def main():
    n, batch = 65536, 4096
    d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    src = d.clone()
    ms = timeit(lambda: d.copy_(src))
    print("copy 2GiB->2GiB: %.3f ms = %.2f TB/s (r+w)" % (ms, 2 * d.numel() * 4 / ms / 1e9))
    plan, inv = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    d.copy_(src)
    ms = timeit(alt(plan, inv, d, batch), iters=20)
    gs = batch * n / ms / 1e6
    print("c2c %d x %d (%s): %.3f ms  %.1f Gsamples/s  alg %.2f TB/s" % (n, batch, plan.kernel_name(), ms, gs, gs * 16 / 1e3), flush=True)
    for size, batch in [(16384, 8192)]:
        x = torch.rand((batch, size), device="cuda") * 2 - 1
        f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
        ms = timeit(alt(f, i, x, batch), iters=20)
        print("r2c/c2r %d x %d: %.3f ms alg %.2f TB/s" % (size, batch, ms, batch * size * 8 / ms / 1e9))
    for n, batch in [(1024, 262144), (4096, 65536), (8192, 32768), (256, 1 << 20), (16384, 16384), (32768, 8192)]:
        x = torch.rand((batch, n, 2), device="cuda") * 2 - 1
        f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
        ms = timeit(alt(f, i, x, batch), iters=20)
        print("c2c %d x %d: %.3f ms alg %.2f TB/s" % (n, batch, ms, batch * n * 16 / ms / 1e9))

main()
