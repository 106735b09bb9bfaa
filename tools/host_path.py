"""PCIe-inclusive rate of the drop-in host-pointer entry point (Clcfft::transform), N=65536"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import opencl_fft_amd as fa
# single small transforms, as the clfft / clrfft opcodes issue them
for n in (256, 1024, 4096, 16384):
    g = fa.Clcfft(0, n, True)
    x = (np.random.default_rng(0).random((n, 2), dtype=np.float32) * 2 - 1).view(np.complex64).reshape(n)
    for _ in range(20): g.transform(x)
    t0 = time.perf_counter()
    for _ in range(500): g.transform(x)
    print("host transform N=%5d single: %.1f us per call" % (n, (time.perf_counter() - t0) / 500 * 1e6), flush=True)
n = 65536
f = fa.Clcfft(0, n, True)
for batch in (1, 16, 256, 4096):
    x = (np.random.default_rng(0).random((batch, n, 2), dtype=np.float32) * 2 - 1).view(np.complex64).reshape(batch, n)
    f.transform(x)
    reps = max(2, 200 // batch)
    t0 = time.perf_counter()
    for _ in range(reps):
        f.transform(x)
    dt = (time.perf_counter() - t0) / reps
    print("host transform N=65536 batch %4d: %.1f us per call, %.1f us per transform, %.3f Gsamples/s" % (batch, dt * 1e6, dt * 1e6 / batch, batch * n / dt / 1e9))
