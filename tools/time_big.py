"""dev tool: time the n > 65536 path (run on the GPU box); 1 GiB of data per size"""
import sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

for logn in [int(a) for a in sys.argv[1:]] or [17, 18, 19, 20, 21, 22, 23, 24]:
    n = 1 << logn
    batch = max(1, (1 << 27) // n)
    d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    k = [0]
    def step():
        (f if k[0] % 2 == 0 else i).exec_device(d, batch); k[0] += 1
    ms = timeit(step)
    print("c2c 2^%d x %d: %.3f ms  %.1f Gsamples/s  alg %.2f TB/s" % (logn, batch, ms, batch * n / ms / 1e6, batch * n * 16 / ms / 1e9), flush=True)
