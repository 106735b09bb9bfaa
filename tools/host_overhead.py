import sys, time
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa
n, batch = 65536, 4096
d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
s = torch.cuda.Stream()
for k in range(4): (f if k % 2 == 0 else i).exec_device(d, batch, s.cuda_stream)
torch.cuda.synchronize()
for K in (10, 20, 50):
    t0 = time.perf_counter()
    for k in range(K): (f if k % 2 == 0 else i).exec_device(d, batch, s.cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("K=%d  launch loop %.2f ms  sync %.2f ms  total/step %.3f ms" % (K, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3 / K))
# same with events
with torch.cuda.stream(s):
    K = 50
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter(); ev[0].record(s)
    for k in range(K):
        (f if k % 2 == 0 else i).exec_device(d, batch, s.cuda_stream); ev[k + 1].record(s)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("with events: loop %.2f ms sync %.2f ms; event total %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, ev[0].elapsed_time(ev[K])))
