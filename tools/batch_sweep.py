"""dev tool: launch time per transform against the batch, for the resident kernel's complex (n = 65536) and packed real
(size 131072) instantiations — is the per-transform cost a function of the in-place footprint (TLB reach, the XCD window of
xcd_first())?  Steps alternate forward / inverse plans on random data.  usage: python tools/batch_sweep.py [batches...]"""
import sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa

batches = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096, 8192]
s = torch.cuda.current_stream().cuda_stream
for kind, n, per in (("c2c n=65536", 65536, 16 * 65536), ("real size=131072", 131072, 8 * 131072)):
    plans = (fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)) if kind.startswith("c2c") else (fa.Clrfft(0, n, True), fa.Clrfft(0, n, False))
    print("%s (%s)" % (kind, plans[0].kernel_name()))
    for b in batches:
        d = torch.rand((b, n, 2) if kind.startswith("c2c") else (b, n), device="cuda") * 2 - 1
        for k in range(8):
            plans[k & 1].exec_device(d, b, s)
        torch.cuda.synchronize()
        best = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(20):
                plans[k & 1].exec_device(d, b, s)
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / 20)
        ms = sorted(best)[len(best) // 2]
        print("  batch %5d (%5.2f GiB in place): %.4f ms per launch, %.4f us per transform, %.2f TB/s alg, frac %.3f"
              % (b, b * per / 2 / 2 ** 30, ms, ms * 1e3 / b, b * per / ms / 1e9, b * per / ms / 1e9 / 8), flush=True)
        del d
        torch.cuda.empty_cache()
