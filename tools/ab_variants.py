"""interleaved A/B of large-N kernel variants in ONE process (median / min over rounds)"""
import sys, statistics
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa
n, batch = 65536, 4096
variants = [int(a) for a in sys.argv[1:]] or [0, 3]
rounds, per = 7, 6
d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
plans = {}
for v in variants:
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    assert f.set_variant(v) == 0 and i.set_variant(v) == 0
    plans[v] = (f, i)
    for _ in range(2):
        f.exec_device(d, batch); i.exec_device(d, batch)
torch.cuda.synchronize()
times = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        f, i = plans[v]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(per):
            (f if k % 2 == 0 else i).exec_device(d, batch)
        b.record(); torch.cuda.synchronize()
        times[v].append(a.elapsed_time(b) / per)
for v in variants:
    t = times[v]
    print("variant %2d: median %.3f ms  min %.3f ms  (alg %.2f TB/s at median)" % (v, statistics.median(t), min(t), batch * n * 16 / statistics.median(t) / 1e9))
