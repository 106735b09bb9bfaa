"""Boundary behaviour beyond numerics: independent objects on different host threads (the reference's
objects share no globals), plan churn, and the bench.py JSON contract."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import opencl_fft_amd as fa
from oracle import oracle
from tests import util
from tests.util import assert_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_independent_objects_on_threads():
    sizes = [256, 1024, 4096, 65536, 16384, 8192]
    results, errors = {}, []

    def work(k, n):
        try:
            x = util.lcg_complex(100 + k, n * 3).reshape(3, n)
            f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
            for _ in range(5):
                y = x.copy()
                assert f.transform(y) == 0
                z = y.copy()
                assert i.transform(z) == 0
            results[k] = (x, y, z)
        except Exception as e:      # noqa: BLE001 - reported below
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k, n)) for k, n in enumerate(sizes)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    for k, n in enumerate(sizes):
        x, y, z = results[k]
        assert_parity(y, oracle.cfft(x, True), what="thread %d n=%d" % (k, n))
        assert_parity(z, x, what="round trip thread %d" % k)


def test_plan_churn_does_not_leak_or_fail():
    x = util.lcg_complex(5, 65536)
    want = oracle.cfft(x, True)
    for _ in range(40):                      # each plan owns a stream, tables and 256 MiB of scratch
        p = fa.Clcfft(0, 65536, True)
        assert p.get_error() == 0
        y = x.copy()
        assert p.transform(y) == 0
        del p
    assert_parity(y, want, what="after churn")


def test_device_tensor_must_be_contiguous():
    import torch
    d = torch.zeros((4, 1024, 2), device="cuda")[:, ::2]
    with pytest.raises(ValueError):
        fa.Clcfft(0, 512, True).exec_device(d, 4)


@pytest.mark.parametrize("workload,extra", [("c2c", ["--batch", "64"]), ("rfft", ["--batch", "128"]),
                                            ("pconv", ["--batch", "8"])])
def test_bench_contract(workload, extra):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                                   "--workload", workload] + extra, stderr=subprocess.DEVNULL, timeout=300).decode()
    line = [l for l in out.splitlines() if l.startswith("{")]
    assert len(line) == 1
    r = json.loads(line[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 4 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["higher_is_better"] is True and r["vs_baseline"] is None and r["dtype"] == "f32" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["achieved"] > 0
    if workload == "c2c":
        cb = r["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
        assert r["config"]["parity_relL2_vs_oracle"] < 1e-6


def test_bench_two_rank_rehearsal():
    """the N>1 launch line of the driver (torch.distributed.run, one process per rank), rehearsed on one
    GPU: gloo backend, both ranks on device 0.  Checks the max-over-ranks / whole-job aggregation."""
    env = dict(os.environ, CLFA_BENCH_BACKEND="gloo", CLFA_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--batch", "256"]
    out = subprocess.check_output(cmd, env=env, stderr=subprocess.DEVNULL, timeout=600).decode()
    line = [l for l in out.splitlines() if l.startswith("{")]
    assert len(line) == 1, out            # only rank 0 prints
    r = json.loads(line[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["cpu_baseline"] is None
    assert r["config"]["global_batch"] == 512 and r["config"]["shard_start"] == 0
    # whole-job value = samples of BOTH ranks over the max time
    expect = 2 * 256 * 65536 * 4 / (r["ms_per_step"] * 4 * 1e-3) / 1e9
    assert abs(r["value"] - expect) / expect < 1e-6
    # the N > 1 line describes itself: every rank's own figures, the world size the process group reports
    c = r["config"]
    assert c["rccl_world_size"] == 2 and c["backend"] == "gloo"
    assert len(c["per_gpu_gsamples"]) == 2 and all(v > 0 for v in c["per_gpu_gsamples"])
    assert [q["rank"] for q in c["per_rank"]] == [0, 1] and [q["shard_start"] for q in c["per_rank"]] == [0, 256]
    assert all(q["kernel"] == "k_fft_res16" and q["device"] and q["batch"] == 256 for q in c["per_rank"])
    assert r["value"] <= sum(c["per_gpu_gsamples"]) * (1 + 1e-9)      # whole job = all samples over the SLOWEST rank's time


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without an outer launcher: bench.py spawns torch.distributed.run itself (before it
    touches the GPU) and relays rank 0's line; rehearsed on one GPU (gloo, both ranks on device 0)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CLFA_BENCH_BACKEND="gloo", CLFA_BENCH_DEVICE="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--batch", "256"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    assert p.returncode == 0
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout.decode()
    r = json.loads(line[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 512 and r["cpu_baseline"] is None
    assert "other_workloads" not in r["config"] and r["ms_per_step_cold"] > 0


def test_randomised_parity_sweep():
    """tests/fuzz_parity.py with a fixed seed and a bounded budget: random geometries of all five object kinds (complex and
    packed real plans incl. the out-of-place entry point, partitioned / time-varying / direct convolution, host and device
    entry points), every case against the oracle"""
    from tests import fuzz_parity
    cases, counts = fuzz_parity.run(45.0, 20261004, silent=True)
    assert cases >= 20 and set(counts) >= {"cfft", "rfft", "pconv", "dconv", "any"}, counts


def test_bench_default_line_carries_every_single_gpu_config():
    """the driver's command line (no --workload, no --batch): the headline (configs[1]) with its cold figure beside it,
    configs[2] and configs[3] under config.other_workloads, each with its own roofline block"""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2"],
                                  stderr=subprocess.DEVNULL, timeout=900).decode()
    line = [l for l in out.splitlines() if l.startswith("{")]
    assert len(line) == 1
    r = json.loads(line[0])
    assert r["steps"] == 6 and r["warmup"] == 2 and "4096 batches" in r["config"]["workload"]
    assert r["ms_per_step_cold"] > 0 and 0 < r["roofline"]["frac_cold"] < 1 and 0 < r["roofline"]["frac"] < 1
    assert r["config"]["effective_warmup_launches"] >= 2 * 2 + 6 + 4
    ow = r["config"]["other_workloads"]
    assert set(ow) == {"rfft", "pconv", "rfft131072"}
    assert ow["rfft131072"]["roofline"]["kernel"] == "k_fft_res16" and ow["rfft131072"]["full_size_selfcheck"]["roundtrip_max_abs"] < 2e-5
    for k, v in ow.items():
        assert v["ms_per_step"] > 0 and v["steps"] >= 100 and 0 < v["roofline"]["frac"] < 1 and v["roofline"]["kernel"]
        assert v["roofline"]["bound"] == "hbm" and v["roofline"]["peak"] == 8000.0
    assert ow["pconv"]["realtime_ratio"] > 50 and "BASELINE configs[3]" in ow["pconv"]["workload"]
    assert "BASELINE configs[2]" in ow["rfft"]["workload"] and ow["rfft"]["full_size_selfcheck"]["roundtrip_max_abs"] < 2e-5
    assert r["cpu_baseline"]["kind"] == "port"
    for k in ("rfft", "pconv"):       # every leg against its own CPU restatement
        assert ow[k]["cpu_baseline"]["kind"] == "port" and ow[k]["cpu_baseline"]["value"] > 0 and ow[k]["cpu_baseline"]["cores"] >= 1
    oop = r["config"]["out_of_place"]       # the same transforms through the out-of-place entry point, never the headline
    assert oop["steps"] % 2 == 0 and 0 < oop["roofline"]["frac"] < 1 and oop["roofline"]["kernel"] == "k_fft_res16"
    assert "in place" in r["config"]["workload"] and r["config"]["library"].endswith("libclfft_amd.so")
    ref = r["reference_opencl_same_gpu"]
    assert ref["kind"] == "reference" and 0.1 < ref["value"] < 10
    assert ref["source"].startswith(("tests/golden/ref/", "measured in this run"))


def test_device_entry_points_capture_into_a_hip_graph():
    """the launch path neither allocates nor synchronises, so a sequence of transforms on a caller
    stream can be captured once and replayed (hipGraph via torch.cuda.CUDAGraph)"""
    import torch
    n, batch = 4096, 64
    size, rbatch = 2048, 32
    d = torch.rand((batch, n, 2), device="cuda") * 2 - 1
    r = torch.rand((rbatch, size), device="cuda") * 2 - 1
    big = torch.rand((8, 65536, 2), device="cuda") * 2 - 1
    x0, r0, b0 = d.clone(), r.clone(), big.clone()
    f, i = fa.Clcfft(0, n, True), fa.Clcfft(0, n, False)
    rf = fa.Clrfft(0, size, True)
    bf = fa.Clcfft(0, 65536, True)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):           # warm up outside the capture (module load, occupancy query)
        for p, buf, nb in ((f, d, batch), (i, d, batch), (rf, r, rbatch), (bf, big, 8)):
            assert p.exec_device(buf, nb, side.cuda_stream) == 0
    side.synchronize()
    d.copy_(x0); r.copy_(r0); big.copy_(b0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        assert f.exec_device(d, batch, s) == 0
        assert i.exec_device(d, batch, s) == 0      # back to the input
        assert f.exec_device(d, batch, s) == 0
        assert rf.exec_device(r, rbatch, s) == 0
        assert bf.exec_device(big, 8, s) == 0
    for _ in range(3):                               # replays see fresh inputs
        d.copy_(x0); r.copy_(r0); big.copy_(b0)
        g.replay()
    torch.cuda.synchronize()
    y = d.cpu().numpy().view(np.complex64).reshape(batch, n)
    assert_parity(y, oracle.cfft(x0.cpu().numpy().view(np.complex64).reshape(batch, n), True), what="c2c in graph")
    assert_parity(r.cpu().numpy().view(np.complex64), oracle.rfft_forward(r0.cpu().numpy()), what="r2c in graph")
    assert_parity(big.cpu().numpy().view(np.complex64).reshape(8, 65536),
                  oracle.cfft(b0.cpu().numpy().view(np.complex64).reshape(8, 65536), True), what="N=65536 in graph")


def test_plan_on_two_streams_is_ordered():
    """one plan, two caller streams, no synchronisation by the caller: the plan owns one workspace, so
    the library orders the second stream's work behind the first (clfft_amd.h, conventions)"""
    import torch
    from oracle import oracle
    from tests import util
    n, batch = 32768, 96   # four-step kernel with a scratch workspace
    x = util.lcg_complex(5, n * batch).reshape(batch, n)
    plan = fa.Clcfft(0, n, True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d1 = torch.from_numpy(x.view(np.float32).copy()).cuda()
    d2 = d1.clone()
    torch.cuda.synchronize()
    assert plan.exec_device(d1, batch, s1.cuda_stream) == 0
    assert plan.exec_device(d2, batch, s2.cuda_stream) == 0
    torch.cuda.synchronize()
    want = oracle.cfft(x[::7], True)
    for d in (d1, d2):
        got = d.cpu().numpy().view(np.complex64).reshape(batch, n)[::7]
        util.assert_parity(got, want, what="two streams")


def test_plan_survives_a_destroyed_caller_stream():
    """the caller's previous stream is not the library's to keep alive: use stream A, destroy it, then call on stream B
    (and on the host entry point, which runs on the plan's own stream) — no spurious failure, right results"""
    import ctypes
    import torch
    n, batch = 32768, 96
    x = util.lcg_complex(6, n * batch).reshape(batch, n)
    hip = ctypes.CDLL("libamdhip64.so")
    plan = fa.Clcfft(0, n, True)
    d1 = torch.from_numpy(x.view(np.float32).copy()).cuda()
    d2 = d1.clone()
    torch.cuda.synchronize()
    sa = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(sa)) == 0
    assert plan.exec_device(d1, batch, sa.value) == 0
    assert hip.hipStreamDestroy(sa) == 0
    sb = torch.cuda.Stream()
    assert plan.exec_device(d2, batch, sb.cuda_stream) == 0
    torch.cuda.synchronize()
    y = x[:3].copy()
    assert plan.transform(y) == 0                       # host entry point: the plan's own stream
    want = oracle.cfft(x[::7], True)
    for d in (d1, d2):
        assert_parity(d.cpu().numpy().view(np.complex64).reshape(batch, n)[::7], want, what="after stream destroy")
    assert_parity(y, oracle.cfft(x[:3], True), what="host call after stream destroy")


def test_plan_host_arrays_bit_identical():
    """clfa_fft_host_alloc (extension): transform() on an array the caller took from the plan (page-locked, seen by the
    device) runs on that memory directly; results are bit-identical to the copying call (the reference's, cl_fft.cpp:153-161,
    267-296), for one transform and for a batch, complex and packed real (in place, and out of place with both arrays from
    the plan), repeated with arrays coming and going; errors"""
    rng = np.random.default_rng(3)
    for rep in range(3):
        for n, batch in ((65536, 1), (1024, 1), (65536, 3), (8192, 5), (4096, 6)):
            x = (rng.random((batch, n, 2), dtype=np.float32) * 2 - 1).view(np.complex64).reshape(batch, n)
            for fwd in (True, False):
                p = fa.Clcfft(0, n, fwd)
                want = x.copy()
                assert p.transform(want) == 0                      # the copying route
                buf = p.alloc_host((batch + 1, n), np.complex64)   # the transform sits INSIDE the plan's buffer
                assert buf is not None
                buf[0] = 0
                buf[1:] = x
                assert p.transform(buf[1:]) == 0
                assert np.array_equal(buf[1:].view(np.uint32), want.view(np.uint32)), (n, batch, fwd)
                assert np.all(buf[0] == 0)
                other = p.alloc_host((batch, n), np.complex64)     # a second buffer of the same plan
                other[:] = x
                assert p.transform(other) == 0 and np.array_equal(other.view(np.uint32), want.view(np.uint32))
                assert p.free_host(buf) == 0 and p.free_host(buf) == -30
                del buf
                y = x.copy()
                assert p.transform(y) == 0 and np.array_equal(y.view(np.uint32), want.view(np.uint32))   # copies as before
        for size, batch in ((16384, 1), (16384, 4), (16384, 3), (8192, 6), (131072, 1)):
            r = (rng.random((batch, size), dtype=np.float32) * 2 - 1)
            f, i = fa.Clrfft(0, size, True), fa.Clrfft(0, size, False)
            spec = np.zeros((batch, size // 2), np.complex64)
            assert f.transform(spec, r.copy()) == 0
            back = np.zeros((batch, size), np.float32)
            assert i.transform(spec.copy(), back) == 0
            # in place on one array of the plan (the opcode's use)
            a = f.alloc_host((batch, size), np.float32)
            a[:] = r
            assert f.transform(a.view(np.complex64), a) == 0
            assert np.array_equal(a.view(np.uint32), spec.view(np.uint32).reshape(batch, size))
            # out of place, both arrays from the plan
            s2, b2 = i.alloc_host((batch, size // 2), np.complex64), i.alloc_host((batch, size), np.float32)
            s2[:] = spec
            b2[:] = 0
            assert i.transform(s2, b2) == 0
            assert np.array_equal(b2.view(np.uint32), back.view(np.uint32)), (size, batch)
            assert np.array_equal(s2, spec)                        # the source is left as it was
            # one array from the plan, the other the caller's own: the copying route, same result
            b3 = np.zeros((batch, size), np.float32)
            assert i.transform(s2, b3) == 0 and np.array_equal(b3.view(np.uint32), back.view(np.uint32))
    p = fa.Clcfft(0, 64, True)
    assert p.alloc_host((0,), np.complex64) is None
    assert p.free_host(np.zeros(4, np.complex64)) == -30


def test_plan_host_arrays_randomised():
    """tools/stress_pinned.py with a fixed seed: arrays of random sizes taken from the plans and given back in a churning heap,
    complex and packed real, in place and out of place, every result bit for bit what the device-resident call gives (the
    stress that the first form of the feature — the caller's own arrays pinned in place — did not survive)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_pinned", os.path.join(ROOT, "tools", "stress_pinned.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(400, seed=5, verbose=False) == 0
