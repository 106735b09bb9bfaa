"""dev tool: interleaved comparison of SEVERAL builds of libclfft_amd.so in ONE process (tools/build_variant.sh makes
them), on realistic data: steps alternate forward / inverse plans so that the values stay O(1).
usage: python tools/ab_multi.py <c2c|c2c<n>|rfft|rfft<size>> <name=path.so> [<name=path.so> ...]   [AB_BATCH=... AB_ROUNDS=...]
The in-tree library is always there as "tree"."""
import ctypes as C, os, statistics, sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd._lib as L

what = sys.argv[1]
libs = {"tree": L.lib()}
for a in sys.argv[2:]:
    name, path = a.split("=", 1)
    lib = C.CDLL(path)
    for sym, res, args in L.SYMBOLS:
        if hasattr(lib, sym):
            f = getattr(lib, sym); f.restype = res; f.argtypes = args
    libs[name] = lib
rsize = int(what[4:]) if what.startswith("rfft") and len(what) > 4 else 16384


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
    batch = int(os.environ.get("AB_BATCH", str((1 << 27) // rsize)))
    d = torch.rand((batch, rsize), device="cuda") * 2 - 1
    unit = batch * rsize * 8
else:
    n = int(what[3:]) if len(what) > 3 else 65536
    batch = int(os.environ.get("AB_BATCH", str((1 << 28) // n)))
    d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    unit = batch * n * 16
P = {k: plans(lib) for k, lib in libs.items()}
s = torch.cuda.current_stream().cuda_stream


def run(k, reps):
    lib, ps = libs[k], P[k]
    for j in range(reps):
        assert lib.clfa_fft_exec_dev(ps[j % 2], d.data_ptr(), batch, s) == 0


for k in libs:
    run(k, 4)
torch.cuda.synchronize()
times = {k: [] for k in libs}
rounds = int(os.environ.get("AB_ROUNDS", "9"))
for r in range(rounds):
    order = list(libs)
    if r & 1:
        order.reverse()
    for k in order:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(k, 10); b.record(); torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 10)
base = statistics.median(times[list(libs)[1]] if len(libs) > 1 else times["tree"])
print("%s batch %d (%.0f MB algorithmic per launch); relative to %s" % (what, batch, unit / 1e6, list(libs)[1] if len(libs) > 1 else "tree"))
for k, t in times.items():
    m = statistics.median(t)
    print("%-12s median %.4f ms  min %.4f ms  alg %.3f TB/s  frac %.3f  %+.2f %%" % (k, m, min(t), unit / m / 1e9, unit / m / 1e9 / 8, (m / base - 1) * 100))
