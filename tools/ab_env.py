"""interleaved A/B of a plan-time environment switch: python ab_env.py <VAR>[=value] <n> [real]
("off" = variable unset, "on" = variable set to value, default 1)"""
import os, sys, statistics
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa
var, n = sys.argv[1], int(sys.argv[2])
var, val = (var.split("=") + ["1"])[:2]
real = len(sys.argv) > 3
batch = (1 << 27) // n
d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
def mk():
    if real:
        return [fa.Clrfft(0, 2 * n, True), fa.Clrfft(0, 2 * n, False)]
    return [fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)]
os.environ.pop(var, None)
off = mk()
os.environ[var] = val
on = mk()
os.environ.pop(var, None)
print("kernels:", off[0].kernel_name() if hasattr(off[0], "kernel_name") else "?", on[0].kernel_name() if hasattr(on[0], "kernel_name") else "?")
x = d.view(batch, 2 * n) if real else d
def run(ps, k):
    for j in range(k):
        assert ps[j % 2].exec_device(x, batch) == 0
for ps in (off, on): run(ps, 20)
torch.cuda.synchronize()
t = {"off": [], "on": []}
for r in range(11):
    for k, ps in (("off", off), ("on", on)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(ps, 10); b.record(); torch.cuda.synchronize()
        t[k].append(a.elapsed_time(b) / 10)
for k, v in t.items():
    m = statistics.median(v)
    print("%s %s: median %.4f ms  min %.4f  alg %.2f TB/s" % (var, k, m, min(v), batch * n * 16 / m / 1e9))
