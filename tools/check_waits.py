"""dev tool (GPU box): the packed real 131072 kernels wait for their hand-issued loads with COUNTED s_waitcnt vmcnt(N).
A count that is too large would read registers before the data are there — and might pass a parity test by luck.
This compares the library bit for bit with a build whose waits are all vmcnt(0) (tools/build_variant.sh wait0
-DCLFA_C2R_WAIT=0), over many batch sizes and repetitions, with a second stream keeping the memory system busy.
usage: python tools/check_waits.py tools/ab/libclfft_wait0.so [seconds]"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import torch
import opencl_fft_amd._lib as L

new = L.lib()
old = C.CDLL(sys.argv[1])
for name, res, args in L.SYMBOLS:
    if hasattr(old, name):
        f = getattr(old, name); f.restype = res; f.argtypes = args
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
size = 131072
def plan(lib, fwd):
    h = C.c_void_p()
    assert lib.clfa_rfft_create(C.byref(h), 0, size, fwd) == 0
    return h
plans = {(k, fwd): plan(lib, fwd) for k, lib in (("new", new), ("old", old)) for fwd in (1, 0)}
libs = {"new": new, "old": old}
s = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream()
noise = torch.rand((1 << 28,), device="cuda")
g = torch.Generator(device="cuda").manual_seed(7)
t0, n, k = time.time(), 0, 0
while time.time() - t0 < budget:
    batch = [65, 100, 255, 256, 257, 300, 511, 513, 777, 1024, 2048][k % 11]
    k += 1
    x = torch.rand((batch, size), generator=g, device="cuda") * 2 - 1
    for fwd in (1, 0):
        if k % 2:
            with torch.cuda.stream(side):            # background traffic on another stream
                noise.mul_(1.0001)
        out = {}
        for name in ("new", "old"):
            d = x.clone()
            assert libs[name].clfa_fft_exec_dev(plans[(name, fwd)], d.data_ptr(), batch, s) == 0
            out[name] = d
        torch.cuda.synchronize()
        assert torch.equal(out["new"].view(torch.int32), out["old"].view(torch.int32)), ("differs", batch, fwd)
        n += 1
print("check_waits ok: %d comparisons (batches 65..2048, both directions) in %.0f s, bit-equal to the vmcnt(0) build" % (n, time.time() - t0))
