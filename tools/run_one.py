"""dev tool: a few launches of one plan for rocprofv3 counters: python tools/run_one.py c2c|r2c|c2r [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opencl_fft_amd as fa
what = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
if what == "c2c":
    p = fa.Clcfft(0, 65536, True)
    d = torch.rand((batch, 65536, 2), device="cuda") * 2 - 1
else:
    p = fa.Clrfft(0, 131072, what == "r2c")
    d = torch.rand((batch, 131072), device="cuda") * 2 - 1
for _ in range(3):
    assert p.exec_device(d, batch) == 0
    d.mul_(65536.0 if what != "c2r" else 1.0 / 131072)
torch.cuda.synchronize()
print(what, p.kernel_name())
