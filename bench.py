#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: batched 1-D c2c FFT, N=65536,
float32, 4096 batches per GPU, device-resident, in place.

    python bench.py --gpus 1 --steps 200 --warmup 20      (the defaults)
    python bench.py --gpus N ...                           (starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      (the driver's launch line)

A step = one pass of the hot path (one batched transform) over the rank's 4096 x 65536
complex samples already resident in HBM.  Steps alternate forward / inverse plans on the
same buffer (inverse(forward(x)) == x, cl_fft.cpp:39-40) so the data stay O(1) instead of
shrinking by 1/N per step into denormals; both directions are the same kernel.
The path shards by batches: every rank owns its own 4096 transforms, no data-path
collective ("weak" scaling, config 5 of BASELINE.json); RCCL only carries the end-of-run
checksum and the max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      algorithmic bytes (16 B per complex sample) / average launch
                duration of the dominant kernel, measured here with HIP events on the launch stream
  cpu_baseline  the CPU restatement of the reference algorithm (oracle/, "port") timed on the
                host cores over a bounded sample — a reported baseline, not the target
  config.other_workloads   (N = 1 only) configs[2] and configs[3] of BASELINE.json — r2c + c2r of size
                16384 x 8192 and the 256-channel partitioned convolution — timed in the same run with the
                same machinery, each with ms_per_step and its own roofline block; plus (not a BASELINE config)
                r2c + c2r of size 131072 x 2048, the largest real size of the reference's range
  ms_per_step_cold / roofline.frac_cold   the same K steps after the same W warm-up steps taken FIRST, before
                the full-size self-check (the chip is still inside its ~20 ms start-up clock ramp then);
                config.effective_warmup_launches counts what ran before the headline's timed region;
                config.settle_launches of them are untimed steps worth --settle-ms (60) of device time right in front of
                the W warm-up steps, so that the K timed steps read the chip's sustained clocks, not the tail of its ramp
`--workload rfft | pconv` make one of the other configurations the headline of the line instead.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # idle OpenMP workers of the CPU baseline sleep, not spin

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2c", choices=["c2c", "rfft", "pconv", "rfft131072"])
    ap.add_argument("--batch", type=int, default=0, help="override batches / channels per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bandwidth", action="store_true", help="skip the device copy/read/write yardsticks")
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the full-size property check before the warm-up")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold timed region taken before the self-check")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="device time of untimed steps in front of every workload's warm-up steps (0: none)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="headline workload only (profiling runs); default: N = 1 runs also time the other two configs")
    ap.add_argument("--series-out", default="", help="write the per-launch times (ms) of the timed region to this file")
    return ap.parse_args()


def self_launch(a):
    """`bench.py --gpus N` without an outer launcher: start the N ranks ourselves (torch.distributed.run as a
    CHILD process, before this process has imported torch or touched a GPU), relay their output and exit
    with their code.  Never an exec: a process that has initialised the GPU must not be replaced."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def traffic_from_profiles(key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json,
    written by tools/pmc_traffic.py from FETCH_SIZE / WRITE_SIZE with the gfx950 corrections
    of MI355X_MICROARCH.md); (None, None) when no measurement is committed."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            e = json.load(f).get(key, {})
            return e.get("hbm_bytes_per_launch"), e.get("source")
    except (OSError, ValueError):
        return None, None


def under_profiler():
    """rocprofv3 preloads its tool library into this process AND into every child: the reference's OpenCL kernels would land in
    the same trace directories (and its timing would carry the profiler's overhead), so the reference legs are skipped there"""
    return "rocprofiler" in os.environ.get("LD_PRELOAD", "") or "ROCPROF_OUTPUT_PATH" in os.environ or "ROCPROFILER_LIBRARY_CTOR" in os.environ


def reference_timing(kind, pci_bus=None):
    """The UNMODIFIED reference timed on this machine's OpenCL device by oracle/_ref/ref_driver (built from the reference's
    sources in place by `make -C oracle ref`; the binary travels with the repository, the sources do not) — live when the
    binary and an OpenCL device are there; the c2c leg falls back to the measurement committed with the golden vectors.
      cfft : Clcfft::transform N = 65536 (cl_fft.cpp:153-161: one transform per call, 17 launches, two PCIe copies)
      rfft : Clrfft::transform size 16384, forward / inverse alternating (cl_fft.cpp:267-296)
      pconv: Clpconv::convolution(out, in), pts 1024, 94 partitions, ONE instance per object as the reference has it
             (cl_conv.cpp:393-458: 26 launches and two copies per block) — BASELINE configs[3] is 256 such instances
    The reference has no CPU path of its own: its "CPU path" would be these same kernels on a CPU OpenCL device, and no CPU
    ICD exists on these machines.  The OpenCL device is picked by the PCI bus of the GPU this rank runs on (OpenCL's
    enumeration need not follow HIP's ordinals)."""
    import numpy as np
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    what = {"cfft": "Clcfft::transform N=65536, one transform per call, PCIe copies included",
            "rfft": "Clrfft::transform size=16384, r2c / c2r alternating, one transform per call, PCIe copies included",
            "pconv": "Clpconv::convolution(out, in) pts=1024, 94 partitions, ONE instance, per block, PCIe copies included"}[kind]
    args = {"cfft": ["time"], "rfft": ["time-rfft", "16384"], "pconv": ["time-pconv", "1024", "96256", "300"]}[kind]
    index = str(int(os.environ.get("LOCAL_RANK", "0")))
    for devspec in (["pci:%02x" % pci_bus] if pci_bus is not None else []) + [index]:   # (the index only if the bus finds nothing)
        if not os.path.exists(exe) or under_profiler():
            break
        try:
            out = subprocess.run([exe, "/tmp", devspec] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120)
            f = out.stdout.decode().split()
            if out.returncode == 0 and len(f) >= 3:
                r = {"value": float(f[1]), "unit": "Gsamples/s", "kind": "reference", "what": what,
                     "source": "measured in this run: oracle/_ref/ref_driver %s (%d calls, OpenCL device %s, selected by %s)"
                     % (" ".join(args), int(f[2]), " ".join(f[3:]) or "?", devspec)}
                r["us_per_block" if kind == "pconv" else "us_per_transform"] = float(f[0])
                if kind == "pconv":   # what 256 channels cost the reference: 256 objects called one after the other
                    r["us_per_block_256_instances"] = 256 * float(f[0])
                    r["realtime_ratio_256_instances"] = (1024 / 48000.0) / (256 * float(f[0]) * 1e-6)
                return r
        except (OSError, subprocess.SubprocessError, ValueError):
            pass
    if kind != "cfft":
        return None
    try:
        t_ref = np.fromfile(os.path.join(ROOT, "tests", "golden", "ref", "timing_ref_cfft65536.bin"), dtype=np.float64)
        return {"value": float(t_ref[1]), "unit": "Gsamples/s", "us_per_transform": float(t_ref[0]), "kind": "reference",
                "what": what, "source": "tests/golden/ref/timing_ref_cfft65536.bin (committed measurement of the same binary on an MI355X of this pool)"}
    except OSError:
        return None


def cpu_baseline_c2c(n, sample):
    """oracle (CPU restatement of the reference's reorder + log2N radix-2 passes) on all host
    cores over `sample` transforms of the same workload"""
    import numpy as np
    from oracle import oracle
    rng = np.random.default_rng(0)
    x = (rng.random((sample, n, 2), dtype=np.float32) * 2 - 1).view(np.complex64).reshape(sample, n)
    oracle.cfft(x[:8], True)                     # warm the thread pool
    t0 = time.perf_counter()
    oracle.cfft(x, True)
    dt = time.perf_counter() - t0
    return {"value": sample * n / dt / 1e9, "unit": "Gsamples/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "%d transforms of N=%d (%.1f s wall, OpenMP over batches)" % (sample, n, dt)}


def cpu_baseline_rfft(size, sample):
    """oracle (Clrfft::transform forward, cl_fft.cpp:267-282, restated) on all host cores over `sample` transforms"""
    import numpy as np
    from oracle import oracle
    x = np.random.default_rng(0).random((sample, size), dtype=np.float32) * 2 - 1
    oracle.rfft_forward(x[:8])
    t0 = time.perf_counter()
    oracle.rfft_forward(x)
    dt = time.perf_counter() - t0
    return {"value": sample * size / dt / 1e9, "unit": "Gsamples/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "%d r2c transforms of size %d (%.1f s wall, OpenMP over batches)" % (sample, size, dt)}


def cpu_baseline_pconv(cvs, pts, blocks):
    """oracle (Clpconv::convolution, cl_conv.cpp:393-458, restated: one instance, scalar) over `blocks` blocks of one channel"""
    import numpy as np
    from oracle import oracle
    rng = np.random.default_rng(0)
    o = oracle.Pconv(cvs, pts)
    o.push_ir((rng.random(cvs, dtype=np.float32) - 0.5) / np.float32(cvs ** 0.5))
    x = rng.random((blocks, pts), dtype=np.float32) * 2 - 1
    o.convolution(x[0])
    t0 = time.perf_counter()
    for b in range(blocks):
        o.convolution(x[b])
    dt = time.perf_counter() - t0
    return {"value": blocks * pts / dt / 1e9, "unit": "Gsamples/s", "cores": 1, "kind": "port",
            "sample": "%d blocks of one channel, pts=%d, %d partitions (%.1f s wall, one thread)" % (blocks, pts, cvs // pts, dt)}


class Workload:
    """one configuration of BASELINE.json, device-resident: step(k) launches one pass on `stream`"""

    def __init__(self, name, fa, torch, dev, local, rank, world, batch_override, stream):
        self.name, self.torch, self.stream = name, torch, stream
        self.extra = {}
        self.plans = None
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        self.gen = g
        if name == "c2c":
            from opencl_fft_amd.dist import ShardedBatch
            n = 65536
            # weak scaling: 4096 transforms per GPU (config 5 = 32768 over 8 GPUs); the global batch is
            # split into contiguous blocks, rank r owns [start, start+count)
            shard = ShardedBatch((batch_override or 4096) * world, rank, world)
            self.batch = batch = shard.count
            self.extra["global_batch"], self.extra["shard_start"] = shard.total, shard.start
            self.data = torch.rand((batch, n, 2), generator=g, device=dev, dtype=torch.float32) * 2 - 1
            self.plans = [fa.Clcfft(local, n, True), fa.Clcfft(local, n, False)]
            for p in self.plans:
                assert p.get_error() == 0, p.get_log()
            self.n = n
            self.units = batch * n                       # complex samples per step per rank
            self.alg_bytes = 16.0 * self.units           # 8 B read + 8 B write per sample
            self.step = lambda k: self.plans[k & 1].exec_device(self.data, batch, stream.cuda_stream)
            self.workload = ("c2c N=65536 x %d batches per GPU, float32, in place, device-resident "
                             "(BASELINE configs[1])" % batch)
            self.kernel, self.tkey = self.plans[0].kernel_name(), "c2c65536"
            self.metric = "Gsamples/s for batched 1D FFT (N=65536, float32) + achieved HBM GB/s vs peak"
            self.direction = "steps alternate forward/inverse plans"
        elif name == "rfft131072":
            # not a BASELINE config: the largest packed real size of the reference's range (cl_fft.cpp:32), in ONE HBM pass
            # since round 4 (the resident kernel with the reference's pair maps inside)
            size, batch = 131072, batch_override or 4096   # 2 GiB in place, the headline's footprint (profiles/r05_rfft131072/ is taken at this batch)
            self.batch, self.n = batch, size
            self.data = torch.rand((batch, size), generator=g, device=dev, dtype=torch.float32) * 2 - 1
            self.plans = [fa.Clrfft(local, size, True), fa.Clrfft(local, size, False)]
            for p in self.plans:
                assert p.get_error() == 0, p.get_log()
            self.units = batch * size
            self.alg_bytes = 8.0 * self.units
            self.step = lambda k: self.plans[k & 1].exec_device(self.data, batch, stream.cuda_stream)
            self.workload = "r2c then c2r, size=131072 x %d batches per GPU, packed in place (largest real size of the reference's range)" % batch
            self.kernel, self.tkey = self.plans[0].kernel_name(), "rfft131072"
            self.metric = "Gsamples/s (real samples) for batched r2c/c2r FFT size=131072"
            self.direction = "steps alternate r2c / c2r plans"
        elif name == "rfft":
            size, batch = 16384, batch_override or 8192
            self.batch, self.n = batch, size
            self.data = torch.rand((batch, size), generator=g, device=dev, dtype=torch.float32) * 2 - 1
            self.plans = [fa.Clrfft(local, size, True), fa.Clrfft(local, size, False)]
            for p in self.plans:
                assert p.get_error() == 0, p.get_log()
            self.units = batch * size
            self.alg_bytes = 8.0 * self.units            # 4 B real + 4 B packed complex per real sample, each way
            self.step = lambda k: self.plans[k & 1].exec_device(self.data, batch, stream.cuda_stream)
            self.workload = "r2c then c2r, size=16384 x %d batches per GPU, packed in place (BASELINE configs[2])" % batch
            self.kernel, self.tkey = self.plans[0].kernel_name(), "rfft16384"
            self.metric = "Gsamples/s (real samples) for batched r2c/c2r FFT size=16384"
            self.direction = "steps alternate r2c / c2r plans"
        else:
            pts, cvs, ch = 1024, 96256, batch_override or 256
            self.pc = pc = fa.Clpconv(local, cvs, pts, channels=ch)
            assert pc.get_cl_err() == 0
            ir = (torch.rand((ch, cvs), generator=g, device=dev) - 0.5) / (cvs ** 0.5)
            assert pc.push_ir_device(ir) == 0
            torch.cuda.synchronize()
            self.inp = torch.rand((ch, pts), generator=g, device=dev) * 2 - 1
            self.out = torch.empty((ch, pts), device=dev)
            self.batch = ch
            self.units = ch * pts                        # channel-samples per block
            nparts = pc.nparts
            self.alg_bytes = ch * (2.0 * nparts * pts * 8 + 4 * pts + 8 * pts + 4 * pts + 16 * pts)
            self.step = lambda k: pc.process_device(self.out, self.inp, None, stream.cuda_stream)
            self.workload = ("partitioned convolution, %d channels per GPU, pts=1024, IR 96256 (94 partitions), 48 kHz "
                             "(BASELINE configs[3])" % ch)
            self.kernel, self.tkey = pc.kernel_name(), "pconv1024x94"
            self.metric = "channel-samples/s for partitioned convolution (x1e9)"
            self.direction = "one block of 1024 samples per channel per step"
        self.launches_before_timed = 0

    def run(self, count):
        """`count` untimed steps on the launch stream"""
        for k in range(count):
            assert self.step(k) == 0
        self.launches_before_timed += count

    def settle(self, ms):
        """untimed steps worth `ms` of device time (counted at 5 TB/s of algorithmic traffic, an even number of them, the same
        on every rank) right in front of a workload's W warm-up steps: the chip takes 20-60 ms of continuous work to reach its
        sustained clocks (profiles/README.md, series_cold_start.txt), and the cold reading, the self-check's host round trips
        and the construction of the next workload all leave it below them"""
        count = 0 if ms <= 0 else (int(ms / (self.alg_bytes / 5e9)) + 2) & ~1
        self.run(count)
        return count

    def selfcheck(self):
        """Full-size guard BEFORE the headline is timed (a broken kernel must not get a number): on the benchmark's
        own buffer, at the benchmark's own size — the size-independent properties of the transform: round trip,
        Parseval, linearity (c2c); round trip (packed real).  The data are restored bit for bit afterwards."""
        if self.plans is None:
            return None
        torch, stream, data, batch, plans = self.torch, self.stream, self.data, self.batch, self.plans
        checks = {}
        with torch.cuda.stream(stream):
            x0 = data.clone()
            e0 = (x0.double() ** 2).sum()
            assert plans[0].exec_device(data, batch, stream.cuda_stream) == 0          # forward (scaled 1/n)
            launches = 2
            if self.name == "c2c":
                fx = data.clone()
                checks["parseval_rel"] = abs(float(((fx.double() ** 2).sum() * self.n / e0).item()) - 1.0)
            assert plans[1].exec_device(data, batch, stream.cuda_stream) == 0          # inverse
            checks["roundtrip_max_abs"] = float((data - x0).abs().max().item())
            if self.name == "c2c":
                y = torch.rand(data.shape, generator=self.gen, device=data.device, dtype=torch.float32) * 2 - 1
                z = 0.5 * x0 + y
                assert plans[0].exec_device(y, batch, stream.cuda_stream) == 0
                assert plans[0].exec_device(z, batch, stream.cuda_stream) == 0
                launches += 2
                z -= 0.5 * fx + y
                checks["linearity_max_abs"] = float(z.abs().max().item())
                checks["linearity_ref_max"] = float(fx.abs().max().item())
                del fx, y, z
            data.copy_(x0)
            del x0
            stream.synchronize()
        assert checks["roundtrip_max_abs"] < 2e-5, checks          # |x| <= 1, float32, log2(n) = 16 stages each way
        if self.name == "c2c":
            assert checks["parseval_rel"] < 1e-5, checks
            assert checks["linearity_max_abs"] < 1e-5 * max(checks["linearity_ref_max"], 1e-30) + 1e-7, checks
        self.launches_before_timed += launches
        checks["transform_launches"] = launches
        return checks

    def timed(self, K, barrier, series=False):
        """K steps between one pair of HIP events on the launch stream and host clocks behind barrier +
        synchronize on both sides -> (host seconds, mean ms per launch, per-launch ms or [])"""
        torch, stream = self.torch, self.stream
        with torch.cuda.stream(stream):
            stream.synchronize()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            # One event pair for the whole region — an event after EVERY launch (--series-out) keeps the stream
            # from running launches back to back and costs a 0.2 ms kernel 5-8 % (a 0.9 ms one 1-2 %).
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1 if series else 2)]
            t0 = time.perf_counter()
            ev[0].record(stream)
            rc = 0
            for k in range(K):
                rc |= self.step(k)
                if series:
                    ev[k + 1].record(stream)
            if not series:
                ev[1].record(stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            barrier()
        assert rc == 0
        avg_ms = ev[0].elapsed_time(ev[-1]) / K
        per = [ev[k].elapsed_time(ev[k + 1]) for k in range(K)] if series else []
        self.launches_before_timed += K
        return t1 - t0, avg_ms, per

    def roofline(self, avg_ms):
        achieved = self.alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic, source = traffic_from_profiles(self.tkey)
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
                "kernel": self.kernel, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": self.alg_bytes}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))          # nothing below has run: no torch import, no GPU call in this process

    import numpy as np
    import torch
    import torch.distributed as dist
    import opencl_fft_amd as fa
    from opencl_fft_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    if os.environ.get("CLFA_BENCH_REHEARSE") == "launch":
        # tests only (tests/test_dist_cpu.py, no GPU there): the launch plumbing alone — ranks rendezvous over gloo,
        # reduce like the real run does, rank 0 prints one line; no transform is run and no number is reported
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"rehearsal": "launch", "n_gpus": world, "max_over_ranks": float(t.item()),
                              "steps": a.steps, "warmup": a.warmup, "value": None}), flush=True)
        dist.destroy_process_group()
        return
    # rehearsal hooks (tests only): CLFA_BENCH_BACKEND=gloo CLFA_BENCH_DEVICE=0 run several ranks on ONE GPU
    backend = os.environ.get("CLFA_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("CLFA_BENCH_DEVICE", local))
    ndev = torch.cuda.device_count()
    if ndev > 0:
        local %= ndev      # a launcher that narrows each rank to one visible device leaves only ordinal 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI: one rank per GPU
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()

    stream = torch.cuda.Stream(device=dev)
    K, W = a.steps, a.warmup
    wl = Workload(a.workload, fa, torch, dev, local, rank, world, a.batch, stream)
    extra = wl.extra

    # 1. the contract's reading taken cold: W warm-up steps, K timed steps, nothing before them
    cold = None
    if not a.no_cold and not a.series_out:
        with torch.cuda.stream(stream):
            wl.run(W)
        cold_s, cold_ms, _ = wl.timed(K, barrier)
        cold = {"ms_per_step": cold_s / K * 1e3, "avg_launch_ms": cold_ms}
    elif not a.no_selfcheck and not a.series_out:
        # (profiling runs, --no-cold: the guard's transforms would otherwise be the first launches of the process, at
        # start-up clocks, and weigh on the per-kernel averages of rocprofv3 --stats)
        with torch.cuda.stream(stream):
            wl.run(W)
    # 2. the full-size guard (~15 launch-equivalents of device work with its elementwise kernels)
    if not a.no_selfcheck:
        chk = wl.selfcheck()
        if chk is not None:
            extra["full_size_selfcheck"] = chk
    # 3. the headline: W warm-up steps, K timed steps — at the chip's sustained clocks (settle(): config.settle_launches)
    with torch.cuda.stream(stream):
        extra["settle_launches"] = wl.settle(a.settle_ms)
        wl.run(W)
    extra["effective_warmup_launches"] = wl.launches_before_timed
    elapsed, avg_ms, per_launch_ms = wl.timed(K, barrier, series=bool(a.series_out))
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    per_rank = None
    if world > 1:
        # every rank's own figures, so that the N > 1 line describes itself: time, kernel-only time, device, kernel
        mine = {"rank": rank, "device": fa.device_name(local), "kernel": wl.kernel, "elapsed_s": elapsed, "avg_launch_ms": avg_ms,
                "gsamples_per_s": wl.units * K / elapsed / 1e9, "shard_start": extra.get("shard_start"), "batch": wl.batch}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    if cold is not None:
        cm = torch.tensor([cold["ms_per_step"]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(cm, op=dist.ReduceOp.MAX)
        cold["ms_per_step"] = float(cm.item())
    if per_launch_ms and a.workload != "pconv" and K > 1:
        fwd_ms, inv_ms = per_launch_ms[0::2], per_launch_ms[1::2]
        extra["launch_ms"] = {"fwd_avg": sum(fwd_ms) / len(fwd_ms), "fwd_min": min(fwd_ms), "fwd_max": max(fwd_ms),
                              "inv_avg": sum(inv_ms) / len(inv_ms), "inv_min": min(inv_ms), "inv_max": max(inv_ms)}

    # 3b. the same workload OUT OF PLACE (extension clfa_fft_exec_dev_oop; the reference's device side is out of place
    # too, data1 -> data2, cl_fft.cpp:138-151): forward A -> B, inverse B -> A, same machinery.  Never the headline.
    oop = None
    if a.workload == "c2c" and world == 1 and not a.series_out and not a.no_other_workloads:   # (not in profiling runs: same kernel symbol)
        other = torch.empty_like(wl.data)
        bufs = (wl.data, other)
        in_place_step = wl.step
        wl.step = lambda k: wl.plans[k & 1].exec_device_oop(bufs[k & 1], bufs[1 - (k & 1)], wl.batch, stream.cuda_stream)
        with torch.cuda.stream(stream):
            wl.run(max(W, 4) & ~1)
        before = wl.launches_before_timed
        K2 = (max(K, 20) + 1) & ~1                 # an even count: the data end where they started
        o_s, o_ms, _ = wl.timed(K2, barrier)
        wl.step = in_place_step
        wl.launches_before_timed = before           # (bookkeeping of the in-place headline only)
        oop = {"what": "the same transforms through clfa_fft_exec_dev_oop: forward A -> B, inverse B -> A (two 2 GiB buffers)",
               "steps": K2, "ms_per_step": o_s / K2 * 1e3, "value": wl.units * K2 / o_s / 1e9, "unit": "Gsamples/s",
               "roofline": {"bound": "hbm", "achieved": wl.alg_bytes / (o_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": wl.alg_bytes / (o_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": o_ms,
                            "kernel": wl.kernel}}
        del other, bufs
        torch.cuda.empty_cache()

    # 4. configs[2] and configs[3] in the same run (N = 1 only: the multi-GPU line stays the sharded headline)
    others = None
    if world == 1 and not a.no_other_workloads and not a.series_out and a.batch == 0:
        others, other_names = {}, []
        for name in ("c2c", "rfft", "pconv", "rfft131072"):
            if name == a.workload:
                continue
            o = Workload(name, fa, torch, dev, local, rank, world, 0, stream)
            k2, w2 = max(K, 100), max(W, 20)
            with torch.cuda.stream(stream):
                o.run(w2)
            chk = None if a.no_selfcheck else o.selfcheck()
            with torch.cuda.stream(stream):
                settled = o.settle(a.settle_ms)
                o.run(w2)
            s2, ms2, _ = o.timed(k2, barrier)
            rec = {"workload": o.workload, "metric": o.metric, "value": o.units * k2 / s2 / 1e9, "unit": "Gsamples/s",
                   "steps": k2, "warmup": w2, "settle_launches": settled, "effective_warmup_launches": o.launches_before_timed - k2,
                   "ms_per_step": s2 / k2 * 1e3, "direction": o.direction, "roofline": o.roofline(ms2)}
            if chk is not None:
                rec["full_size_selfcheck"] = chk
            if name == "pconv":
                rec["realtime_ratio"] = (1024 / 48000.0) / (ms2 * 1e-3)   # one block = pts / 48 kHz of audio per channel
            others[name] = rec
            other_names.append(name)
            del o
            torch.cuda.empty_cache()

    # The yardsticks the roofline fraction is read against: sustained device
    # read / write / copy bandwidth, and a copy in the FFT's own access shape (column blocks of 16
    # columns), measured here, on this device, AFTER the timed regions (run before them, 150 ms of copy
    # kernels leave the chip in a lower clock state and the first ~20 FFT launches read 5-15 % slower).
    membench = None
    if not a.no_bandwidth and rank == 0:
        torch.cuda.synchronize()
        membench = {k: round(v, 3) for k, v in fa.bandwidth_probe(local, 1 << 30, 100).items()}
        membench["unit"] = "TB/s"
        membench["how"] = ("2 x 1 GiB buffers, 100 launches each after 27 warm-up launches; 16-byte non-temporal accesses; "
                           "copies count read + write; copy_colblock = 128-byte row segments 2 KiB apart (the "
                           "four-step FFT's global access shape)")
    # parity guard, AFTER the timed region (the oracle's OpenMP pool must not be spinning on the
    # host cores while the HIP runtime threads drive the timed launches): transforms vs the oracle
    if a.workload == "c2c":
        from oracle import oracle
        gq = torch.Generator(device=dev).manual_seed(99)
        nprobe = 72      # more transforms than CUs / 4: the same kernel as the timed launches
        probe = torch.rand((nprobe, 65536, 2), generator=gq, device=dev, dtype=torch.float32) * 2 - 1
        pick = [0, nprobe - 1]
        x0 = probe[pick].cpu().numpy().view(np.complex64).reshape(len(pick), 65536)
        # (the INVERSE plan: a different kernel instantiation than the forward one whose launches the
        # roofline block and the rocprofv3 stats average, so this short launch does not dilute them)
        assert wl.plans[1].exec_device(probe, nprobe, stream.cuda_stream) == 0
        stream.synchronize()
        y0 = probe[pick].cpu().numpy().view(np.complex64).reshape(len(pick), 65536)
        ref = oracle.cfft(x0, False, nthreads=1)
        extra["parity_relL2_vs_oracle"] = float(np.linalg.norm(y0.astype(np.complex128) - ref) /
                                                np.linalg.norm(ref.astype(np.complex128)))
        assert extra["parity_relL2_vs_oracle"] < 1e-6

    # end-of-run checksum over all ranks (validates the sharded run; RCCL only here)
    chk = (wl.data.double() ** 2).sum().reshape(1) if a.workload != "pconv" else (wl.out.double() ** 2).sum().reshape(1)
    if world > 1:
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
    extra["energy_checksum_all_ranks"] = float(chk.item())

    if rank == 0:
        value = wl.units * K * world / elapsed / 1e9
        rec = {
            "metric": wl.metric, "value": value, "unit": "Gsamples/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": dict({"workload": wl.workload, "direction": wl.direction,
                            "sharding": "batches per rank, no data-path collective", "kernel": wl.kernel,
                            }, **extra),
            "roofline": wl.roofline(avg_ms),
        }
        rec["config"]["library"] = os.path.relpath(_lib.LIB_PATH, ROOT)   # which build produced the number (CLFA_LIB_PATH)
        if world > 1:
            rec["config"]["rccl_world_size"] = dist.get_world_size()
            rec["config"]["backend"] = dist.get_backend()
            rec["config"]["per_gpu_gsamples"] = [r["gsamples_per_s"] for r in per_rank]
            rec["config"]["per_rank"] = per_rank
        if oop is not None:
            rec["config"]["out_of_place"] = oop
        # the unmodified reference on the same GPU (child processes, after every timed region)
        try:
            pci_bus = int(torch.cuda.get_device_properties(local).pci_bus_id)
        except (AttributeError, TypeError, ValueError):
            pci_bus = None
        rec["reference_opencl_same_gpu"] = reference_timing({"c2c": "cfft", "rfft": "rfft", "pconv": "pconv", "rfft131072": "rfft"}[a.workload], pci_bus) if a.workload != "rfft131072" else None
        if others is not None:
            for leg in ("rfft", "pconv"):
                if leg in others:
                    others[leg]["reference_opencl_same_gpu"] = reference_timing(leg, pci_bus)
        if cold is not None:
            # the contract's W + K launches read cold (before the self-check): inside the chip's start-up clock ramp
            rec["ms_per_step_cold"] = cold["ms_per_step"]
            rec["roofline"]["avg_launch_ms_cold"] = cold["avg_launch_ms"]
            rec["roofline"]["frac_cold"] = wl.alg_bytes / (cold["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if others is not None:
            rec["config"]["other_workloads"] = others
        if membench is not None:
            rec["membench"] = membench
            # the same achieved rate against what this device sustains for a plain copy and for a copy
            # in the FFT's access shape (8000 GB/s, the peak of the roofline block, is the HBM3E spec)
            rec["roofline"]["frac_of_measured_copy"] = rec["roofline"]["achieved"] / (membench["copy"] * 1e3)
            rec["roofline"]["frac_of_measured_colblock_copy"] = rec["roofline"]["achieved"] / (membench["copy_colblock"] * 1e3)
        if a.workload == "pconv":
            # one block = pts / 48 kHz of audio for every channel
            rec["config"]["realtime_ratio"] = (1024 / 48000.0) / (avg_ms * 1e-3)
        if a.series_out:
            with open(a.series_out, "w") as f:
                f.write("# per-launch ms between HIP events on the launch stream, timed region of: bench.py "
                        "--workload %s --steps %d --warmup %d\n" % (a.workload, K, W))
                f.write("\n".join("%.4f" % t for t in per_launch_ms) + "\n")
        if world == 1 and not a.no_cpu_baseline and a.workload == "c2c":
            rec["cpu_baseline"] = cpu_baseline_c2c(65536, 2048)
            if others is not None:     # the other two legs against their own restatements (bounded samples, after every timed region)
                if "rfft" in others:
                    others["rfft"]["cpu_baseline"] = cpu_baseline_rfft(16384, 32768)
                if "pconv" in others:
                    others["pconv"]["cpu_baseline"] = cpu_baseline_pconv(96256, 1024, 4000)
                if "rfft131072" in others:
                    others["rfft131072"]["cpu_baseline"] = cpu_baseline_rfft(131072, 2048)
        else:
            rec["cpu_baseline"] = None
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
