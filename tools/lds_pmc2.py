"""dev tool: a few launches of the headline kernel and of config 3's kernels, for rocprofv3 --pmc passes
(tools/lds_pmc2.sh); the library is the in-tree one or CLFA_LIB_PATH"""
import sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa
what = sys.argv[1:] or ["c2c", "rfft"]
if "c2c" in what:
    x = torch.rand((4096, 65536, 2), device="cuda") * 2 - 1
    f, i = fa.Clcfft(0, 65536, True), fa.Clcfft(0, 65536, False)
    for _ in range(3):
        f.exec_device(x, 4096)
        i.exec_device(x, 4096)
    torch.cuda.synchronize()
    del x
if "rfft" in what:
    y = torch.rand((8192, 16384), device="cuda") * 2 - 1
    f, i = fa.Clrfft(0, 16384, True), fa.Clrfft(0, 16384, False)
    for _ in range(3):
        f.exec_device(y, 8192)
        i.exec_device(y, 8192)
    torch.cuda.synchronize()
