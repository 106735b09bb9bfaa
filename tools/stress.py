"""dev tool: long round-trip stress of the persistent four-step kernels (intermediate handed over through
LDS + registers): any race in the hand-over would show up as a sporadic round-trip error."""
import sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa

for n, batch, iters in [(65536, 4096, 300), (65536, 777, 300), (32768, 8192, 200), (16384, 16384, 200), (16384, 333, 300)]:
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    x = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    d = x.clone()
    worst = 0.0
    for k in range(iters):
        assert f.exec_device(d, batch) == 0 and i.exec_device(d, batch) == 0
        if k % 10 == 9:
            err = float((d - x).norm() / x.norm())
            worst = max(worst, err)
            assert err < 2e-5, (n, batch, k, err)   # error grows slowly with the number of round trips
            d.copy_(x)
    print("n=%d batch=%d: %d round trips, worst relL2 after 10 round trips %.2e" % (n, batch, iters, worst), flush=True)
print("OK")

# the one-workgroup-per-CU real kernels (k_rfft_2x<13>, k_rfft_2x<14>): r2c then c2r is the identity
for size, batch, iters in [(32768, 8192, 200), (32768, 259, 300), (65536, 4096, 200), (65536, 257, 300)]:
    f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
    x = torch.rand((batch, size), device="cuda") * 2 - 1
    d = x.clone()
    worst = 0.0
    for k in range(iters):
        assert f.exec_device(d, batch) == 0 and i.exec_device(d, batch) == 0
        if k % 10 == 9:
            err = float((d - x).norm() / x.norm())
            worst = max(worst, err)
            assert err < 2e-5, (size, batch, k, err)
            d.copy_(x)
    print("real size=%d batch=%d: %d round trips, worst relL2 after 10 round trips %.2e" % (size, batch, iters, worst), flush=True)
print("real sizes OK")

# small sizes through the staged kernel, ragged batch counts, complex and packed real
import numpy as np
rng = np.random.default_rng(3)
for n in (4, 8, 16, 32, 64, 128, 256):
    for batch in (1, 255, 257, 1000, 4097):
        f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
        x = torch.rand((batch, n, 2), device="cuda") * 2 - 1
        d = x.clone()
        assert f.exec_device(d, batch) == 0
        ref = torch.fft.fft(torch.view_as_complex(x.double()), dim=-1) / n
        err = float((torch.view_as_complex(d.double()) - ref).norm() / ref.norm())
        assert err < 1e-6, ("c2c", n, batch, err)
        assert i.exec_device(d, batch) == 0
        err = float((d - x).norm() / x.norm())
        assert err < 1e-6, ("c2c round trip", n, batch, err)
        size = 2 * n
        rf, ri = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
        r = torch.rand((batch, size), device="cuda") * 2 - 1
        rd = r.clone()
        assert rf.exec_device(rd, batch) == 0 and ri.exec_device(rd, batch) == 0
        err = float((rd - r).norm() / r.norm())
        assert err < 1e-6, ("rfft round trip", size, batch, err)
print("small sizes OK")
