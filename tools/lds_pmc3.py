"""dev tool: a few launches of the kernels of other sizes for rocprofv3 --pmc passes (tools/lds_pmc3.sh)"""
import sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa
for kind, n in (("r", 65536), ("r", 32768), ("c", 16384), ("c", 32768), ("r", 8192), ("c", 4096), ("r", 131072)):
    batch = (1 << 27) // n if kind == "r" else (1 << 26) // n
    x = torch.rand((batch, n) if kind == "r" else (batch, n, 2), device="cuda") * 2 - 1
    f, i = (fa.Clrfft(0, n, True), fa.Clrfft(0, n, False)) if kind == "r" else (fa.Clcfft(0, n, True), fa.Clcfft(0, n, False))
    for _ in range(3):
        f.exec_device(x, batch)
        i.exec_device(x, batch)
    torch.cuda.synchronize()
    del x
