#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: batched 1-D c2c FFT, N=65536,
float32, 4096 batches per GPU, device-resident, in place.

    python bench.py --gpus 1 --steps 200 --warmup 20      (the defaults)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (one batched transform) over the rank's 4096 x 65536
complex samples already resident in HBM.  Steps alternate forward / inverse plans on the
same buffer (inverse(forward(x)) == x, cl_fft.cpp:39-40) so the data stay O(1) instead of
shrinking by 1/N per step into denormals; both directions are the same kernel.
The path shards by batches: every rank owns its own 4096 transforms, no data-path
collective ("weak" scaling, config 5 of BASELINE.json); RCCL only carries the end-of-run
checksum and the max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      algorithmic bytes (16 B per complex sample) / average launch
                duration of the dominant kernel, measured here with HIP events on the launch stream
  cpu_baseline  the CPU restatement of the reference algorithm (oracle/, "port") timed on the
                host cores over a bounded sample — a reported baseline, not the target
Other workloads (--workload rfft | pconv) time configs 3 and 4 with the same machinery.
"""
import argparse
import json
import os
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
    ap.add_argument("--workload", default="c2c", choices=["c2c", "rfft", "pconv"])
    ap.add_argument("--batch", type=int, default=0, help="override batches / channels per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bandwidth", action="store_true", help="skip the device copy/read/write yardsticks")
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the full-size property check before the warm-up")
    ap.add_argument("--series-out", default="", help="write the per-launch times (ms) of the timed region to this file")
    return ap.parse_args()


def traffic_from_profiles(key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json,
    written by tools/pmc_traffic.py from FETCH_SIZE / WRITE_SIZE with the gfx950 corrections
    of MI355X_MICROARCH.md); None when no measurement is committed."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(key, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
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


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import opencl_fft_amd as fa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % a.gpus)
        a.gpus = world
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

    from opencl_fft_amd.dist import ShardedBatch

    def barrier():
        if world > 1:
            dist.barrier()

    stream = torch.cuda.Stream(device=dev)
    K, W = a.steps, a.warmup
    extra = {}

    if a.workload == "c2c":
        n = 65536
        # weak scaling: 4096 transforms per GPU (config 5 = 32768 over 8 GPUs); the global batch is
        # split into contiguous blocks, rank r owns [start, start+count)
        shard = ShardedBatch((a.batch or 4096) * world, rank, world)
        batch = shard.count
        extra["global_batch"], extra["shard_start"] = shard.total, shard.start
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        data = torch.rand((batch, n, 2), generator=g, device=dev, dtype=torch.float32) * 2 - 1
        plans = [fa.Clcfft(local, n, True), fa.Clcfft(local, n, False)]
        for p in plans:
            assert p.get_error() == 0, p.get_log()
        units = batch * n                       # complex samples per step per rank
        alg_bytes = 16.0 * units                # 8 B read + 8 B write per sample
        step = lambda k: plans[k & 1].exec_device(data, batch, stream.cuda_stream)
        workload = "c2c N=65536 x %d batches per GPU, float32, in place, device-resident (BASELINE configs[1])" % batch
        kernel, tkey = plans[0].kernel_name(), "c2c65536"
        metric, unit = "Gsamples/s for batched 1D FFT (N=65536, float32) + achieved HBM GB/s vs peak", "Gsamples/s"
    elif a.workload == "rfft":
        size, batch = 16384, a.batch or 8192
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        data = torch.rand((batch, size), generator=g, device=dev, dtype=torch.float32) * 2 - 1
        plans = [fa.Clrfft(local, size, True), fa.Clrfft(local, size, False)]
        units = batch * size
        alg_bytes = 8.0 * units                 # 4 B real + 4 B packed complex per real sample, each way
        step = lambda k: plans[k & 1].exec_device(data, batch, stream.cuda_stream)
        workload = "r2c then c2r, size=16384 x %d batches per GPU, packed in place (BASELINE configs[2])" % batch
        kernel, tkey = plans[0].kernel_name(), "rfft16384"
        metric, unit = "Gsamples/s (real samples) for batched r2c/c2r FFT size=16384", "Gsamples/s"
    else:
        pts, cvs, ch = 1024, 96256, a.batch or 256
        pc = fa.Clpconv(local, cvs, pts, channels=ch)
        assert pc.get_cl_err() == 0
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        ir = (torch.rand((ch, cvs), generator=g, device=dev) - 0.5) / (cvs ** 0.5)
        assert pc.push_ir_device(ir) == 0
        torch.cuda.synchronize()
        inp = torch.rand((ch, pts), generator=g, device=dev) * 2 - 1
        out = torch.empty((ch, pts), device=dev)
        units = ch * pts                        # channel-samples per block
        nparts = pc.nparts
        alg_bytes = ch * (2.0 * nparts * pts * 8 + 4 * pts + 8 * pts + 4 * pts + 16 * pts)
        step = lambda k: pc.process_device(out, inp, None, stream.cuda_stream)
        workload = ("partitioned convolution, %d channels per GPU, pts=1024, IR 96256 (94 partitions), 48 kHz "
                    "(BASELINE configs[3])" % ch)
        kernel, tkey = "k_pconv_fused", "pconv1024x94"
        metric, unit = "channel-samples/s for partitioned convolution (x1e9)", "Gsamples/s"

    # Full-size guard BEFORE anything is timed (a broken kernel must not get a number): on the benchmark's own
    # buffer, at the benchmark's own size — the size-independent properties of the transform: round trip,
    # Parseval, linearity (c2c); round trip (packed real).  ~15 launch-equivalents of device work; the data are
    # restored bit for bit afterwards.  (Side effect, stated in DESIGN.md section 5: the chip's ~20 ms start-up
    # clock ramp is over when the W warm-up steps begin.)
    if a.workload in ("c2c", "rfft") and not a.no_selfcheck:
        with torch.cuda.stream(stream):
            x0 = data.clone()
            e0 = (x0.double() ** 2).sum()
            assert plans[0].exec_device(data, batch, stream.cuda_stream) == 0          # forward (scaled 1/n)
            checks = {}
            if a.workload == "c2c":
                fx = data.clone()
                checks["parseval_rel"] = abs(float(((fx.double() ** 2).sum() * n / e0).item()) - 1.0)
            assert plans[1].exec_device(data, batch, stream.cuda_stream) == 0          # inverse
            checks["roundtrip_max_abs"] = float((data - x0).abs().max().item())
            if a.workload == "c2c":
                y = torch.rand(data.shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1
                z = 0.5 * x0 + y
                assert plans[0].exec_device(y, batch, stream.cuda_stream) == 0
                assert plans[0].exec_device(z, batch, stream.cuda_stream) == 0
                z -= 0.5 * fx + y
                checks["linearity_max_abs"] = float(z.abs().max().item())
                checks["linearity_ref_max"] = float(fx.abs().max().item())
                del fx, y, z
            data.copy_(x0)
            del x0
            stream.synchronize()
        assert checks["roundtrip_max_abs"] < 2e-5, checks          # |x| <= 1, float32, log2(n) = 16 stages each way
        if a.workload == "c2c":
            assert checks["parseval_rel"] < 1e-5, checks
            assert checks["linearity_max_abs"] < 1e-5 * max(checks["linearity_ref_max"], 1e-30) + 1e-7, checks
        extra["full_size_selfcheck"] = checks

    with torch.cuda.stream(stream):
        for k in range(W):
            assert step(k) == 0
        stream.synchronize()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        # HIP events on the launch stream bracket the K launches: roofline.achieved divides by their mean.
        # One event pair for the whole region — an event after EVERY launch (--series-out) keeps the stream
        # from running launches back to back and costs a 0.2 ms kernel 5-8 % (a 0.9 ms one 1-2 %).
        series = bool(a.series_out)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1 if series else 2)]
        t0 = time.perf_counter()
        ev[0].record(stream)
        for k in range(K):
            rc = step(k)
            if series:
                ev[k + 1].record(stream)
        if not series:
            ev[1].record(stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
    assert rc == 0
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    avg_ms = ev[0].elapsed_time(ev[-1]) / K          # mean launch duration over the timed region (both directions)
    per_launch_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(K)] if series else []
    if series and a.workload != "pconv" and K > 1:
        fwd_ms, inv_ms = per_launch_ms[0::2], per_launch_ms[1::2]
        extra["launch_ms"] = {"fwd_avg": sum(fwd_ms) / len(fwd_ms), "fwd_min": min(fwd_ms), "fwd_max": max(fwd_ms),
                              "inv_avg": sum(inv_ms) / len(inv_ms), "inv_min": min(inv_ms), "inv_max": max(inv_ms)}

    # The yardsticks the roofline fraction is read against: sustained device
    # read / write / copy bandwidth, and a copy in the FFT's own access shape (column blocks of 16
    # columns), measured here, on this device, AFTER the timed region (run before it, 150 ms of copy
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
    # host cores while the HIP runtime threads drive the timed launches): one transform vs the oracle
    if a.workload == "c2c":
        from oracle import oracle
        gq = torch.Generator(device=dev).manual_seed(99)
        nprobe = 72      # more transforms than CUs / 4: the same kernel as the timed launches
        probe = torch.rand((nprobe, 65536, 2), generator=gq, device=dev, dtype=torch.float32) * 2 - 1
        pick = [0, nprobe - 1]
        x0 = probe[pick].cpu().numpy().view(np.complex64).reshape(len(pick), 65536)
        # (the INVERSE plan: a different kernel instantiation than the forward one whose launches the
        # roofline block and the rocprofv3 stats average, so this short launch does not dilute them)
        assert plans[1].exec_device(probe, nprobe, stream.cuda_stream) == 0
        stream.synchronize()
        y0 = probe[pick].cpu().numpy().view(np.complex64).reshape(len(pick), 65536)
        ref = oracle.cfft(x0, False, nthreads=1)
        extra["parity_relL2_vs_oracle"] = float(np.linalg.norm(y0.astype(np.complex128) - ref) /
                                                np.linalg.norm(ref.astype(np.complex128)))
        assert extra["parity_relL2_vs_oracle"] < 1e-6

    # end-of-run checksum over all ranks (validates the sharded run; RCCL only here)
    chk = (data.double() ** 2).sum().reshape(1) if a.workload != "pconv" else (out.double() ** 2).sum().reshape(1)
    if world > 1:
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
    extra["energy_checksum_all_ranks"] = float(chk.item())

    if rank == 0:
        value = units * K * world / elapsed / 1e9
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        rec = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": dict({"workload": workload, "direction": "steps alternate forward/inverse plans",
                            "sharding": "batches per rank, no data-path collective", "kernel": kernel,
                            }, **extra),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profiles(tkey),
                         "kernel": kernel, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if membench is not None:
            rec["membench"] = membench
            # the same achieved rate against what this device sustains for a plain copy and for a copy
            # in the FFT's access shape (8000 GB/s, the peak of the roofline block, is the HBM3E spec)
            rec["roofline"]["frac_of_measured_copy"] = achieved / (membench["copy"] * 1e3)
            rec["roofline"]["frac_of_measured_colblock_copy"] = achieved / (membench["copy_colblock"] * 1e3)
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
        else:
            rec["cpu_baseline"] = None
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
