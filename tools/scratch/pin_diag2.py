import sys
sys.path.insert(0, ".")
import numpy as np, torch
import opencl_fft_amd as fa
rng = np.random.default_rng(3)
def cfft_part():
    for n, batch in ((65536, 1), (1024, 1), (65536, 3), (8192, 5)):
        x = (rng.random((batch, n, 2), dtype=np.float32) * 2 - 1).view(np.complex64).reshape(batch, n)
        for fwd in (True, False):
            p = fa.Clcfft(0, n, fwd)
            want = x.copy()
            assert p.transform(want) == 0
            buf = np.zeros((batch + 1, n), np.complex64)
            assert p.pin_host(buf) == 0
            buf[1:] = x
            assert p.transform(buf[1:]) == 0
            if not np.array_equal(buf[1:].view(np.uint32), want.view(np.uint32)):
                print("CFFT mismatch", n, batch, fwd, np.argwhere(buf[1:] != want)[:3].tolist())
            assert p.unpin_host(buf) == 0
fails = 0
for rep in range(40):
    cfft_part()
    for size, batch in ((16384, 1), (16384, 4), (131072, 1)):
        r = (rng.random((batch, size), dtype=np.float32) * 2 - 1)
        f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
        spec = np.zeros((batch, size // 2), np.complex64)
        assert f.transform(spec, r.copy()) == 0
        back = np.zeros((batch, size), np.float32)
        assert i.transform(spec.copy(), back) == 0
        a = r.copy()
        assert f.pin_host(a) == 0
        assert f.transform(a.view(np.complex64), a) == 0
        if not np.array_equal(a.view(np.uint32), spec.view(np.uint32).reshape(batch, size)):
            print("rep", rep, size, batch, "forward pinned in place differs")
        s2, b2 = spec.copy(), np.zeros((batch, size), np.float32)
        assert i.pin_host(s2) == 0 and i.pin_host(b2) == 0
        assert i.transform(s2, b2) == 0
        if not np.array_equal(b2.view(np.uint32), back.view(np.uint32)):
            fails += 1
            d = np.argwhere(b2 != back)
            # a third opinion: the device path
            dd = torch.from_numpy(spec.view(np.float32).reshape(batch, size).copy()).cuda()
            assert i.exec_device(dd, batch) == 0
            third = dd.cpu().numpy()
            print("rep", rep, size, batch, "OOP MISMATCH: %d elements, first %s last %s; b2==device %s, back==device %s; addr s2 %x b2 %x back %x"
                  % (len(d), d[0].tolist(), d[-1].tolist(), np.array_equal(b2, third), np.array_equal(back, third), s2.ctypes.data, b2.ctypes.data, back.ctypes.data), flush=True)
            flat = (d[:, 0] * size + d[:, 1]) * 4 + b2.ctypes.data
            print("     byte addresses of mismatches: first %x last %x, distinct 4K pages %d" % (flat[0], flat[-1], len(set(flat >> 12))))
print("oop mismatches:", fails, "of", 40 * 3)
