"""dev tool: real size 131072 with few transforms — the spread route (column / row kernels + pack) against the one-pass
kernel (a build that takes it for every batch: tools/build_variant.sh r16all -DCLFA_R16_DIV=1000)
usage: python tools/small_batch_r16.py tools/ab/libclfft_r16all.so"""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd._lib as L
new = L.lib()
old = C.CDLL(sys.argv[1])
for name, res, args in L.SYMBOLS:
    if hasattr(old, name):
        f = getattr(old, name); f.restype = res; f.argtypes = args
size = 131072
def plan(lib, fwd):
    h = C.c_void_p(); assert lib.clfa_rfft_create(C.byref(h), 0, size, fwd) == 0; return h
P = {(k, f): plan(lib, f) for k, lib in (("lib", new), ("onepass", old)) for f in (1, 0)}
libs = {"lib": new, "onepass": old}
s = torch.cuda.current_stream().cuda_stream
print("batch | library (spread route up to 64) fwd / inv us | one-pass kernel fwd / inv us")
for batch in (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 256):
    d = torch.rand((batch, size), device="cuda") * 2 - 1
    row = []
    for k in ("lib", "onepass"):
        for fwd in (1, 0):
            for _ in range(5): libs[k].clfa_fft_exec_dev(P[(k, fwd)], d.data_ptr(), batch, s)
            ts = []
            for _ in range(15):
                d.uniform_(-1, 1)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); libs[k].clfa_fft_exec_dev(P[(k, fwd)], d.data_ptr(), batch, s); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            row.append(sorted(ts)[len(ts) // 2])
    print("%5d | %8.1f / %8.1f | %8.1f / %8.1f" % (batch, *row))
