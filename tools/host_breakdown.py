import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import opencl_fft_amd as fa
n = 65536
f = fa.Clcfft(0, n, True)
d = torch.zeros((1, n, 2), device="cuda")
def t(fn, reps=200):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
s = torch.cuda.Stream()
def launch_sync():
    f.exec_device(d, 1, s.cuda_stream); s.synchronize()
print("launch + stream sync          : %.1f us" % t(launch_sync))
h = torch.zeros((1, n, 2)); hp = torch.zeros((1, n, 2)).pin_memory()
def h2d_page(): d.copy_(h); torch.cuda.synchronize()
def h2d_pin(): d.copy_(hp, non_blocking=True); torch.cuda.synchronize()
def d2h_page(): h.copy_(d); torch.cuda.synchronize()
def d2h_pin(): hp.copy_(d, non_blocking=True); torch.cuda.synchronize()
print("H2D 512 KiB pageable + sync   : %.1f us" % t(h2d_page))
print("H2D 512 KiB pinned + sync     : %.1f us" % t(h2d_pin))
print("D2H 512 KiB pageable + sync   : %.1f us" % t(d2h_page))
print("D2H 512 KiB pinned + sync     : %.1f us" % t(d2h_pin))
x = np.zeros((1, n), np.complex64)
print("Clcfft.transform (host)       : %.1f us" % t(lambda: f.transform(x)))
def pinned_path():
    hp.copy_(h)                                   # CPU copy into pinned
    with torch.cuda.stream(s):
        d.copy_(hp, non_blocking=True)
        f.exec_device(d, 1, s.cuda_stream)
        hp.copy_(d, non_blocking=True)
    s.synchronize()
    h.copy_(hp)
print("pinned path via torch          : %.1f us" % t(pinned_path))
