"""dev tool: one host-pointer Clcfft::transform per call — the caller's pageable array, an array taken from the plan
(clfa_fft_host_alloc: page-locked, seen by the device), and whether such an array elsewhere in the process changes what the
other calls cost.  usage: python tools/host_path2.py [name=lib.so ...]  (other builds: tools/build_variant.sh)"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np
import opencl_fft_amd._lib as L

def t(fn, reps=400):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6

libs = {"tree": L.lib()}
for a in sys.argv[1:]:
    name, path = a.split("=", 1)
    lib = C.CDLL(path)
    for sym, res, args in L.SYMBOLS:
        if hasattr(lib, sym):
            f = getattr(lib, sym); f.restype = res; f.argtypes = args
    libs[name] = lib
for n in (65536, 32768, 16384, 4096):
    print("N = %d (%d KiB each way), microseconds per Clcfft::transform call" % (n, n * 8 // 1024))
    for name, lib in libs.items():
        f, g = C.c_void_p(), C.c_void_p()
        assert lib.clfa_cfft_create(C.byref(f), 0, n, 1) == 0 and lib.clfa_cfft_create(C.byref(g), 0, n, 1) == 0
        x = np.ones((1, n), np.complex64)
        a = t(lambda: lib.clfa_cfft_transform(f, x.ctypes.data, 1))
        hp = C.c_void_p()
        assert lib.clfa_fft_host_alloc(g, x.nbytes, C.byref(hp)) == 0
        C.memset(hp, 0, x.nbytes)
        b = t(lambda: lib.clfa_cfft_transform(g, hp, 1))
        c = t(lambda: lib.clfa_cfft_transform(f, x.ctypes.data, 1))
        assert lib.clfa_fft_host_free(g, hp) == 0
        print("  %-8s caller's pageable array %6.1f   array from the plan %6.1f   pageable while another plan holds such an array %6.1f" % (name, a, b, c))
        lib.clfa_fft_destroy(f); lib.clfa_fft_destroy(g)
