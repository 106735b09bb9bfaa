"""dev tool: where the time of ONE host-pointer transform goes (Clcfft::transform, N = 65536: 512 KiB each way), and the
routes that were tried for it.  usage: python tools/host_breakdown.py [other-lib.so ...]  (builds with another
CLFA_ZEROCOPY_MAX_KIB, tools/build_variant.sh)"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import opencl_fft_amd as fa
import opencl_fft_amd._lib as L

def t(fn, reps=300):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6

rows = []
for n in (65536, 16384, 4096):
    f = fa.Clcfft(0, n, True)
    d = torch.zeros((1, n, 2), device="cuda")
    s = torch.cuda.Stream()
    def launch_sync():
        f.exec_device(d, 1, s.cuda_stream); s.synchronize()
    h = torch.zeros((1, n, 2)); hp = torch.zeros((1, n, 2)).pin_memory()
    def h2d_page(): d.copy_(h); torch.cuda.synchronize()
    def h2d_pin(): d.copy_(hp, non_blocking=True); torch.cuda.synchronize()
    def d2h_page(): h.copy_(d); torch.cuda.synchronize()
    def d2h_pin(): hp.copy_(d, non_blocking=True); torch.cuda.synchronize()
    def pinned_staging():
        hp.copy_(h)                                   # CPU copy into a pinned staging buffer
        with torch.cuda.stream(s):
            d.copy_(hp, non_blocking=True)
            f.exec_device(d, 1, s.cuda_stream)
            hp.copy_(d, non_blocking=True)
        s.synchronize()
        h.copy_(hp)
    x = np.zeros((1, n), np.complex64)
    g = fa.Clcfft(0, n, True)
    xp = g.alloc_host((1, n), np.complex64)
    xp[:] = 0
    print("N = %d (%d KiB each way)" % (n, n * 8 // 1024))
    print("  launch + stream sync (device-resident)      : %6.1f us" % t(launch_sync))
    print("  H2D pageable + sync / pinned + sync         : %6.1f / %6.1f us" % (t(h2d_page), t(h2d_pin)))
    print("  D2H pageable + sync / pinned + sync         : %6.1f / %6.1f us" % (t(d2h_page), t(d2h_pin)))
    print("  memcpy -> pinned staging, DMA, kernel, DMA, memcpy back (torch) : %6.1f us" % t(pinned_staging))
    print("  Clcfft.transform, caller's pageable array (library route)        : %6.1f us" % t(lambda: f.transform(x)))
    print("  Clcfft.transform, array taken from the plan (alloc_host)         : %6.1f us" % t(lambda: g.transform(xp)))
    for path in sys.argv[1:]:
        lib = C.CDLL(path)
        for sym, res, args in L.SYMBOLS:
            if hasattr(lib, sym):
                fn = getattr(lib, sym); fn.restype = res; fn.argtypes = args
        hnd = C.c_void_p()
        assert lib.clfa_cfft_create(C.byref(hnd), 0, n, 1) == 0
        print("  Clcfft.transform, pageable, %-36s : %6.1f us" % (path.split("/")[-1], t(lambda: lib.clfa_cfft_transform(hnd, x.ctypes.data, 1))))
        lib.clfa_fft_destroy(hnd)
