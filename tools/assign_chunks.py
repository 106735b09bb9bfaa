"""dev tool: the XCD-interleave granularity of k_fft_lds's transform -> workgroup assignment: workgroup i = x + 8 c (XCD x,
number c inside it) takes transform index bits [c low j bits][x][c high bits][iteration]: j = 0 is "workgroup i takes i"
(the library), j = log2(grid / 8) is XCD-compact.  Library built with -DCLFA_ASSIGN_SEARCH.
usage: python tools/assign_chunks.py tools/ab/libclfft_search.so [rfft<size>|c2c<n>] [grid]"""
import ctypes as C, statistics, sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd._lib as L
lib = C.CDLL(sys.argv[1])
for name, res, args in L.SYMBOLS:
    f = getattr(lib, name); f.restype = res; f.argtypes = args
what = sys.argv[2] if len(sys.argv) > 2 else "rfft16384"
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 512
real = what.startswith("rfft")
n = int(what[4:]) // 2 if real else int(what[3:])
batch = (1 << 27) // n
d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
ps = []
for fwd in (1, 0):
    h = C.c_void_p()
    assert (lib.clfa_rfft_create(C.byref(h), 0, 2 * n, fwd) if real else lib.clfa_cfft_create(C.byref(h), 0, n, fwd)) == 0
    ps.append(h)
s = torch.cuda.current_stream().cuda_stream
lg = grid.bit_length() - 1
nb = batch.bit_length() - 1
def setp(perm):
    a = (C.c_int * 40)(*([len(perm), lg] + list(perm) + [0] * (38 - len(perm)))) if perm else (C.c_int * 40)()
    assert lib.clfa_debug_set_assign(a, 40) == 0
def run(k):
    for j in range(k):
        assert lib.clfa_fft_exec_dev(ps[j % 2], d.data_ptr(), batch, s) == 0
def timeit(perm, reps=30):
    setp(perm); run(6)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(reps); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
run(40)
res = {}
for rnd in range(3):
    for j in range(0, lg - 2):
        cbits = list(range(3, lg))
        perm = cbits[:j] + [0, 1, 2] + cbits[j:] + list(range(lg, nb))
        res.setdefault(j, []).append(timeit(perm))
print("%s, %d transforms, grid %d" % (what, batch, grid))
for j, v in res.items():
    print("  XCD chunks of %3d transforms: median %.4f ms  min %.4f" % (1 << j, statistics.median(v), min(v)))
