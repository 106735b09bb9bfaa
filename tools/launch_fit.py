"""dev tool: launch time vs batch for one plan (fixed cost a and per-transform cost b of t = a + b * batch)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import opencl_fft_amd as fa

kind, n = sys.argv[1], int(sys.argv[2])          # rfft 16384 | cfft 65536
bmax = int(sys.argv[3])
mk = (lambda f: fa.Clrfft(0, n, f)) if kind == "rfft" else (lambda f: fa.Clcfft(0, n, f))
pf, pi = mk(True), mk(False)
per = n if kind == "rfft" else 2 * n
x = torch.rand((bmax, per), device="cuda") * 2 - 1
rows = []
b = bmax
while b >= max(64, bmax // 64):
    k = [0]
    def step():
        (pf if k[0] % 2 == 0 else pi).exec_device(x, b); k[0] += 1
    for _ in range(20): step()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): step()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 40)
    rows.append((b, float(np.median(ts))))
    print("batch %6d  %.4f ms  (%.2f TB/s)" % (b, rows[-1][1], b * per * 8 / rows[-1][1] / 1e9), flush=True)
    b //= 2
B = np.array([r[0] for r in rows], float); T = np.array([r[1] for r in rows])
bb, aa = np.polyfit(B[:4], T[:4], 1)
print("fit over the 4 largest: t = %.1f us + %.3f us * batch  (asymptotic %.2f TB/s)" % (aa * 1e3, bb * 1e3, per * 8 / bb / 1e9))
