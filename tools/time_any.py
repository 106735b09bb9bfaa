"""dev tool: lengths that are not powers of two — the one-launch form (k_blue_lds, convolution length up to 8192) and the
composed form (pre, plan, multiply, plan, post), 256 MiB of complex data per length"""
import sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa
s = torch.cuda.current_stream().cuda_stream
for n in (100, 1000, 1536, 3000, 4095, 4097, 6000, 48000, 100000):
    batch = max(1, (1 << 25) // n)
    d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    for k in range(4):
        (f, i)[k & 1].exec_device(d, batch, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(10):
        (f, i)[k & 1].exec_device(d, batch, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("n = %6d x %6d  %-10s %.3f ms  %.2f TB/s algorithmic (16 B per sample)" % (n, batch, f.kernel_name(), ms, batch * n * 16 / ms / 1e9), flush=True)
