"""dev tool: which transform should a workgroup of k_fft_lds take?  Local search over bit permutations of
(workgroup index | iteration << log2(grid)) -> transform index, on a library built with -DCLFA_ASSIGN_SEARCH
(tools/build_variant.sh search -DCLFA_ASSIGN_SEARCH).
usage: python tools/assign_search.py tools/ab/libclfft_search.so [rfft<size>|c2c<n>] [grid]"""
import ctypes as C, random, statistics, sys
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
assert 1 << lg == grid and batch % grid == 0
nb = (batch.bit_length() - 1)
def setp(perm):
    a = (C.c_int * 40)(*([len(perm), lg] + list(perm) + [0] * (38 - len(perm)))) if perm else (C.c_int * 40)()
    assert lib.clfa_debug_set_assign(a, 40) == 0
def run(k):
    for j in range(k):
        assert lib.clfa_fft_exec_dev(ps[j % 2], d.data_ptr(), batch, s) == 0
def timeit(perm, reps=20):
    setp(perm); run(4)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(reps); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
def name(perm):
    return " ".join(("w%d" % p) if p < lg else ("k%d" % (p - lg)) for p in perm)
run(40)
ident = list(range(nb))
print("%s, %d transforms, grid %d (the kernel's own grid must equal it), %d bits" % (what, batch, grid, nb))
print("library (off)     %.4f ms" % timeit(None, 40))
print("identity          %.4f ms" % timeit(ident, 40))
xcd = list(range(3, lg)) + [0, 1, 2] + list(range(lg, nb))
print("XCD-compact       %.4f ms   %s" % (timeit(xcd, 40), name(xcd)))
random.seed(1)
cands = [ident, xcd]
for r in range(40):
    p = ident[:]; random.shuffle(p); cands.append(p)
res = sorted(((timeit(p), p) for p in cands), key=lambda t: t[0])
for t, p in res[:5] + res[-3:]:
    print("  %.4f  %s" % (t, name(p)))
for start in [res[0][1], res[1][1], ident]:
    cur, best = start[:], timeit(start, 30)
    for sweep in range(3):
        improved = False
        for a in range(nb):
            for b in range(a + 1, nb):
                t = cur[:]; t[a], t[b] = t[b], t[a]
                ms = timeit(t, 12)
                if ms < best * 0.995:
                    ms, again = timeit(t, 30), timeit(cur, 30)
                    if ms < again * 0.997:
                        cur, best, improved = t, ms, True
        if not improved:
            break
    print("local optimum %.4f ms (library %.4f)   %s" % (timeit(cur, 40), timeit(None, 40), name(cur)), flush=True)
