"""experiment: k_fft_lds with its stores redirected behind the batch (variant library built with -DCLFA_OOP_TEST) against
the in-place library; interleaved, steps alternate directions.  usage: ab_oop_lds.py variant.so [rfft<size>|c2c<n>]"""
import ctypes as C, statistics, sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd._lib as L
new = L.lib()
old = C.CDLL(sys.argv[1])
for name, res, args in L.SYMBOLS:
    f = getattr(old, name); f.restype = res; f.argtypes = args
what = sys.argv[2]
real = what.startswith("rfft")
n = int(what[4:]) // 2 if real else int(what[3:])
batch = (1 << 27) // n
d = torch.rand((2 * batch, n, 2), device="cuda") * 2 - 1      # the variant writes into the second half
def plans(lib):
    out = []
    for fwd in (1, 0):
        h = C.c_void_p()
        assert (lib.clfa_rfft_create(C.byref(h), 0, 2 * n, fwd) if real else lib.clfa_cfft_create(C.byref(h), 0, n, fwd)) == 0
        out.append(h)
    return out
libs = {"in place": (new, plans(new)), "out of place": (old, plans(old))}
s = torch.cuda.current_stream().cuda_stream
def run(lib, ps, k):
    for j in range(k):
        assert lib.clfa_fft_exec_dev(ps[j % 2], d.data_ptr(), batch, s) == 0
for lib, ps in libs.values(): run(lib, ps, 4)
torch.cuda.synchronize()
t = {k: [] for k in libs}
for r in range(9):
    for k, (lib, ps) in libs.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(lib, ps, 10); b.record(); torch.cuda.synchronize()
        t[k].append(a.elapsed_time(b) / 10)
for k, v in t.items():
    print("%s %-14s median %.4f ms  min %.4f" % (what, k, statistics.median(v), min(v)))
