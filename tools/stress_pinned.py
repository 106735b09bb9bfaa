"""dev tool: stress of the plan's page-locked host arrays (clfa_fft_host_alloc): random sizes and batches, arrays taken from
the plan, used and given back in a churning heap, in place and out of place, every result compared bit for bit with the
device-resident call on the same input.  (Written for the FIRST form of the feature — hipHostRegister on the caller's own
heap arrays, unregistered before they were freed: 8 wrong results in 3000 calls and, in a second run, a GPU memory access
fault on a host heap address; all at 192 KiB arrays whose addresses had been used by arrays of other sizes before.  That
form is gone; see clfft_amd.cpp.)
usage: python tools/stress_pinned.py [iterations = 1500]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import opencl_fft_amd as fa



def run(iters, seed=11, verbose=True):
    rng = np.random.default_rng(seed)
    plans = {}

    def plan(kind, n, fwd):
        k = (kind, n, fwd)
        if k not in plans:
            plans[k] = (fa.Clcfft if kind == "c" else fa.Clrfft)(0, n, fwd)
            assert plans[k].get_error() == 0
        return plans[k]
    junk, bad, done = [], 0, 0
    for it in range(iters):
        # churn: allocations of many sizes come and go between the pinned arrays
        for _ in range(int(rng.integers(0, 6))):
            junk.append(np.ones(int(rng.integers(1, 1 << 19)), np.float32))
        while len(junk) > 24:
            junk.pop(int(rng.integers(0, len(junk))))
        kind = "c" if rng.random() < 0.4 else "r"
        fwd = bool(rng.integers(0, 2))
        if kind == "c":
            n = 1 << int(rng.integers(8, 17))
            batch = int(rng.integers(1, max(2, min(9, (1 << 20) // n + 1))))
            x = (rng.random((batch, n, 2), dtype=np.float32) * 2 - 1)
            p = plan("c", n, fwd)
            d = torch.from_numpy(x.copy()).cuda()
            assert p.exec_device(d, batch) == 0
            want = d.cpu().numpy().view(np.complex64).reshape(batch, n)
            buf = p.alloc_host((batch, n), np.complex64)
            buf[:] = x.view(np.complex64).reshape(batch, n)
            assert p.transform(buf) == 0
            ok = np.array_equal(buf.view(np.uint32), want.view(np.uint32))
            assert p.free_host(buf) == 0
            del buf
        else:
            size = 1 << int(rng.integers(9, 18))
            batch = int(rng.integers(1, max(2, min(9, (1 << 21) // size + 1))))
            p = plan("r", size, fwd)
            src = (rng.random((batch, size), dtype=np.float32) * 2 - 1)
            d = torch.from_numpy(src.copy()).cuda()
            assert p.exec_device(d, batch) == 0
            want = d.cpu().numpy()
            oop = rng.random() < 0.6
            a = p.alloc_host((batch, size), np.float32)
            a[:] = src
            b = p.alloc_host((batch, size), np.float32) if oop else a
            if oop:
                b[:] = 0
            c_arr, r_arr = (b, a) if fwd else (a, b)          # forward reads r writes c; inverse reads c writes r
            assert p.transform(c_arr.view(np.complex64), r_arr) == 0
            ok = np.array_equal(b.view(np.uint32), want.view(np.uint32)) and (not oop or np.array_equal(a, src))
            assert p.free_host(a) == 0 and (not oop or p.free_host(b) == 0)
            del a, b, c_arr, r_arr
        done += 1
        if not ok:
            bad += 1
            print("MISMATCH it %d kind %s fwd %s n/size %d batch %d" % (it, kind, fwd, n if kind == "c" else size, batch), flush=True)
        if verbose and it % 250 == 0:
            print("  ... %d iterations, %d mismatches" % (it, bad), flush=True)
    if verbose:
        print("stress_pinned: %d iterations, %d mismatches" % (done, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 1500) else 0)
