"""dev tool: interleaved A/B of two builds of libclfft_amd.so in ONE process, on realistic data
(steps alternate forward / inverse plans so the values stay O(1): all-zero or inf data draw less
power and run at higher clocks, which flatters repeated same-direction loops).
usage: python tools/ab_libs.py <old.so> [rfft|rfft<size>|c2c|c2c<n>] [swap]"""
import ctypes as C, statistics, sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd._lib as L

new = L.lib()
old = C.CDLL(sys.argv[1])
for name, res, args in L.SYMBOLS:
    if not hasattr(old, name):      # an older build: entry points added since are simply absent
        continue
    f = getattr(old, name); f.restype = res; f.argtypes = args
what = sys.argv[2] if len(sys.argv) > 2 else "rfft"
rsize = int(what[4:]) if what.startswith("rfft") and len(what) > 4 else 16384   # rfft<size>: another packed real size

def plans(lib):
    out = []
    for fwd in (1, 0):
        h = C.c_void_p()
        if what.startswith("rfft"):
            e = lib.clfa_rfft_create(C.byref(h), 0, rsize, fwd)
        else:
            e = lib.clfa_cfft_create(C.byref(h), 0, int(what[3:]) if len(what) > 3 else 65536, fwd)
        assert e == 0
        out.append(h)
    return out

if what.startswith("rfft"):
    batch = (1 << 27) // rsize
    d = torch.rand((batch, rsize), device="cuda") * 2 - 1
    unit = batch * rsize * 8
elif len(what) > 3:
    n = int(what[3:])
    batch = (1 << 28) // n
    d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    unit = batch * n * 16
else:
    batch = int(__import__("os").environ.get("AB_BATCH", "4096"))      # AB_BATCH: another batch of N = 65536 transforms
    d = torch.rand((batch, 65536, 2), device="cuda") * 2 - 1
    unit = batch * 65536 * 16
libs = {"new": (new, plans(new)), "old": (old, plans(old))}
if len(sys.argv) > 3 and sys.argv[3] == "swap":
    libs = dict(reversed(list(libs.items())))
s = torch.cuda.current_stream().cuda_stream
def run(lib, ps, k):
    for j in range(k):
        assert lib.clfa_fft_exec_dev(ps[j % 2], d.data_ptr(), batch, s) == 0
for lib, ps in libs.values():
    run(lib, ps, 4)
torch.cuda.synchronize()
times = {k: [] for k in libs}
for r in range(9):
    for k, (lib, ps) in libs.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(lib, ps, 10); b.record(); torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 10)
for k, t in times.items():
    m = statistics.median(t)
    print("%s: median %.4f ms  min %.4f ms  alg %.2f TB/s" % (k, m, min(t), unit / m / 1e9))
# per direction (an event after every launch keeps launches from running back to back: slower than the figures above)
per = {k: ([], []) for k in libs}
for r in range(5):
    for k, (lib, ps) in libs.items():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        ev[0].record()
        for j in range(10):
            assert lib.clfa_fft_exec_dev(ps[j % 2], d.data_ptr(), batch, s) == 0
            ev[j + 1].record()
        torch.cuda.synchronize()
        for j in range(10):
            per[k][j % 2].append(ev[j].elapsed_time(ev[j + 1]))
for k, (f, i) in per.items():
    print("%s: forward median %.4f ms   inverse median %.4f ms  (event per launch)" % (k, statistics.median(f), statistics.median(i)))
