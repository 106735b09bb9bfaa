"""Real-time-ratio sweep of time-varying partitioned convolution — the measurement of the
reference's own harness (csound/tests.py:10-36 + tests.csd:8-20: cltvconv over partition size
M = 2^{9,11,13,15} x filter length L = 2^{16..22}, RT ratio = audio duration / elapsed), here
without Csound: one Clpconv.convolution(out, in1, in2) call per partition of audio.

  host   : the drop-in entry point (host pointers, blocking H2D + kernels + D2H per call, like the
           reference's clEnqueueWrite/ReadBuffer around every block)
  device : device-resident blocks on a stream (no PCIe), one synchronisation at the end
"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import opencl_fft_amd as fa

SR = 44100.0            # Csound default sample rate (tests.csd sets none)

def run(M, L, seconds, mode):
    pc = fa.Clpconv(0, L, M)
    assert pc.get_cl_err() == 0, (M, L, pc.get_cl_err())
    blocks = max(8, int(seconds * SR / M))
    rng = np.random.default_rng(0)
    if mode == "host":
        a = (rng.random((blocks, M), dtype=np.float32) - 0.5)
        b = (rng.random((blocks, M), dtype=np.float32) - 0.5)
        out = np.zeros((1, M), np.float32)
        for k in range(3):
            pc.convolution(out, a[k], b[k])
        t0 = time.perf_counter()
        for k in range(blocks):
            pc.convolution(out, a[k], b[k])
        dt = time.perf_counter() - t0
    else:
        a = torch.rand((blocks, 1, M), device="cuda") - 0.5
        b = torch.rand((blocks, 1, M), device="cuda") - 0.5
        out = torch.empty((1, M), device="cuda")
        for k in range(3):
            pc.process_device(out, a[k], b[k])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(blocks):
            pc.process_device(out, a[k], b[k])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return blocks * M / SR / dt

def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    Ms, Ls = [9, 11, 13, 15], [16, 17, 18, 19, 20, 21, 22]
    for mode in ("host", "device"):
        print("\nRT ratio, cltvconv-equivalent, 1 channel, mode=%s, %.0f s of audio at %.0f Hz" % (mode, seconds, SR))
        print("| M \\ log2 L | " + " | ".join(str(l) for l in Ls) + " |")
        print("|---|" + "---|" * len(Ls))
        for m in Ms:
            row = []
            for l in Ls:
                row.append("%.1f" % run(1 << m, 1 << l, seconds, mode))
            print("| %d | " % (1 << m) + " | ".join(row) + " |", flush=True)

main()
