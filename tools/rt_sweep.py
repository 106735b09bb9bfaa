"""Real-time-ratio sweep of time-varying partitioned convolution — the measurement of the
reference's own harness (csound/tests.py:10-36 + tests.csd:8-20: cltvconv over partition size
M = 2^{9,11,13,15} x filter length L = 2^{16..22}, RT ratio = audio duration / elapsed), here
without Csound: one Clpconv.convolution(out, in1, in2) call per partition of audio, ONE instance.

  reference : the UNMODIFIED reference's Clpconv::convolution(out, in1, in2) (cl_conv.cpp:460-548) on the same GPU
              through OpenCL, timed by oracle/_ref/ref_driver rt-sweep (test infrastructure; skipped where the binary
              or an OpenCL device is missing)
  host      : our drop-in entry point (host pointers, blocking, like the reference's call)
  device    : device-resident blocks on a stream (no PCIe), one synchronisation at the end

usage: python tools/rt_sweep.py [seconds of audio per cell for our side = 5] [blocks per cell for the reference = 60]
"""
import os, subprocess, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import opencl_fft_amd as fa

SR = 44100.0            # Csound default sample rate (tests.csd sets none)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Ms, Ls = [9, 11, 13, 15], [16, 17, 18, 19, 20, 21, 22]


def run(M, L, seconds, mode):
    """microseconds per block"""
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
    return dt / blocks * 1e6


def reference(blocks):
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(exe):
        return None, "oracle/_ref/ref_driver is not built"
    try:
        bus = "pci:%02x" % int(torch.cuda.get_device_properties(0).pci_bus_id)
        out = subprocess.run([exe, "/tmp", bus, "rt-sweep", str(blocks)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        if out.returncode != 0:
            out = subprocess.run([exe, "/tmp", "0", "rt-sweep", str(blocks)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        if out.returncode != 0:
            return None, "ref_driver: " + out.stderr.decode().strip().split("\n")[-1]
    except (OSError, subprocess.SubprocessError) as e:
        return None, "ref_driver: %s" % e
    cells, dev = {}, "?"
    for ln in out.stdout.decode().split("\n"):
        f = ln.split()
        if ln.startswith("#"):
            dev = ln[1:].strip()
        elif len(f) == 3:
            cells[(int(f[0]), int(f[1]))] = float(f[2])
    return cells, dev


def table(title, us):
    print("\n" + title)
    print("| M \\ log2 L | " + " | ".join(str(l) for l in Ls) + " |")
    print("|---|" + "---|" * len(Ls))
    for m in Ms:
        print("| %d | " % (1 << m) + " | ".join("%.1f" % us[(1 << m, l)] if us.get((1 << m, l), -1) > 0 else "-" for l in Ls) + " |", flush=True)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    rblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    ref, dev = reference(rblocks)
    ours = {}
    for mode in ("host", "device"):
        ours[mode] = {(1 << m, l): run(1 << m, 1 << l, seconds, mode) for m in Ms for l in Ls}
    print("Time-varying partitioned convolution (cltvconv-equivalent), ONE instance, per block of M samples; RT ratio = (M / %.0f Hz) / time per block" % SR)
    if ref is None:
        print("reference: not measured (%s)" % dev)
    else:
        print("reference = the unmodified reference's Clpconv::convolution(out, in1, in2) on OpenCL device %s, %d blocks per cell" % (dev, rblocks))
        table("microseconds per block, reference (OpenCL, host pointers)", ref)
    table("microseconds per block, ours, host pointers (%.0f s of audio per cell)" % seconds, ours["host"])
    table("microseconds per block, ours, device-resident", ours["device"])
    rt = lambda us: {k: (k[0] / SR) / (v * 1e-6) for k, v in us.items() if v > 0}
    if ref is not None:
        table("RT ratio, reference", rt(ref))
    table("RT ratio, ours, host pointers", rt(ours["host"]))
    table("RT ratio, ours, device-resident", rt(ours["device"]))
    if ref is not None:
        table("speed-up over the reference, host pointers against host pointers", {k: ref[k] / ours["host"][k] for k in ref if ref[k] > 0})


main()
