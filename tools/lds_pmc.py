import sys
sys.path.insert(0, ".")
import torch
import opencl_fft_amd as fa
for n, batch in ((4096, 65536), (8192, 32768)):
    x = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    f = fa.Clcfft(0, n, True)
    for _ in range(3):
        f.exec_device(x, batch)
    torch.cuda.synchronize()
