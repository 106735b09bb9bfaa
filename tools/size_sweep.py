"""Throughput of every supported complex length (2^1 .. 2^24), 1 GiB of data per size, steps alternating
forward / inverse plans on O(1) data; and the packed real transforms of the same lengths."""
import sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa

def timeit(step, iters=30, warm=6):
    for _ in range(warm): step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): step()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

# leave the chip's start-up clock ramp (~20 ms of work) behind before the first row is timed
_w = torch.rand((1024, 65536, 2), device="cuda")
_p = [fa.Clcfft(0, 65536, True), fa.Clcfft(0, 65536, False)]
for _k in range(120): _p[_k & 1].exec_device(_w, 1024)
torch.cuda.synchronize()
del _w, _p

print("| n | batch | c2c ms | c2c Gsamples/s | c2c TB/s (16 B/sample) | r2c+c2r (size 2n) TB/s (8 B/real sample) |")
print("|---|---|---|---|---|---|")
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 24)   # optional range of log2(n)
for logn in range(lo, hi + 1):
    n = 1 << logn
    batch = max(1, (1 << 27) // n)
    d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    k = [0]
    def step():
        (f if k[0] % 2 == 0 else i).exec_device(d, batch); k[0] += 1
    ms = timeit(step)
    rf, ri = fa.Clrfft(0, 2 * n, True), fa.Clrfft(0, 2 * n, False)
    rtb = float("nan")
    if rf.get_error() == 0:
        x = d.view(batch, 2 * n)
        k[0] = 0
        def rstep():
            (rf if k[0] % 2 == 0 else ri).exec_device(x, batch); k[0] += 1
        rms = timeit(rstep)
        rtb = batch * 2 * n * 8 / rms / 1e9
    print("| 2^%d | %d | %.3f | %.1f | %.2f | %.2f |" % (logn, batch, ms, batch * n / ms / 1e6, batch * n * 16 / ms / 1e9, rtb), flush=True)
