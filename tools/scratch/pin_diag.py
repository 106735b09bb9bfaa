import sys
sys.path.insert(0, ".")
import numpy as np
import opencl_fft_amd as fa
rng = np.random.default_rng(3)
for size, batch in ((16384, 1), (16384, 4), (16384, 16), (4096, 8), (131072, 1), (32768, 4)):
    r = (rng.random((batch, size), dtype=np.float32) * 2 - 1)
    f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
    spec = np.zeros((batch, size // 2), np.complex64)
    assert f.transform(spec, r.copy()) == 0
    back = np.zeros((batch, size), np.float32)
    assert i.transform(spec.copy(), back) == 0
    back2 = np.zeros((batch, size), np.float32)
    assert i.transform(spec.copy(), back2) == 0
    print(size, batch, "unpinned twice equal:", np.array_equal(back, back2), i.kernel_name())
    s2, b2 = spec.copy(), np.zeros((batch, size), np.float32)
    assert i.pin_host(s2) == 0 and i.pin_host(b2) == 0
    bad = 0
    for k in range(50):
        b2[:] = 0
        assert i.transform(s2, b2) == 0
        if not np.array_equal(b2.view(np.uint32), back.view(np.uint32)):
            d = np.argwhere(b2 != back)
            if bad < 3:
                print("   iter", k, "mismatches", len(d), "first", d[:3].tolist(), "last", d[-1].tolist(), "b2", b2[tuple(d[0])], "want", back[tuple(d[0])])
            bad += 1
        if not np.array_equal(s2, spec):
            print("   source changed!")
    print("   pinned oop: %d of 50 runs differ" % bad)
    # in place pinned
    a = spec.copy().view(np.float32).reshape(batch, size).copy()
    assert i.unpin_host(s2) == 0 and i.unpin_host(b2) == 0
    assert i.pin_host(a) == 0
    bad = 0
    for k in range(50):
        a[:] = spec.view(np.float32).reshape(batch, size)
        assert i.transform(a.view(np.complex64), a) == 0
        bad += not np.array_equal(a.view(np.uint32), back.view(np.uint32))
    print("   pinned in place: %d of 50 runs differ" % bad)
