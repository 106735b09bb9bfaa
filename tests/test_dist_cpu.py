"""The N>1 path on CPU: 2 processes, gloo backend, 127.0.0.1.  Each rank takes its contiguous
block of the batch axis (opencl_fft_amd.dist.shard_range), transforms it (with the oracle here —
there is no GPU), and the ranks reduce a checksum and the max time exactly as bench.py does
with RCCL.  The union of the shards must equal the unsharded result bit for bit."""
import os
import socket

import numpy as np
import pytest

from opencl_fft_amd.dist import shard_range


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 8, 4096, 32768, 1000003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert spans[-1][0] + spans[-1][1] == total
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    assert shard_range(32768, 3, 8) == (12288, 4096)          # config 5: 4096 per GPU
    with pytest.raises(ValueError):
        shard_range(8, 8, 8)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, n, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opencl_fft_amd.dist import ShardedBatch
    from oracle import oracle
    from tests import util
    sb = ShardedBatch(total)
    assert (sb.rank, sb.world) == (rank, world)
    x = util.lcg_complex(2024, total * n).reshape(total, n)         # every rank can derive any batch
    sb.barrier()
    y = oracle.cfft(x[sb.slice()], True) if sb.count else np.zeros((0, n), np.complex64)
    energy = float(np.sum(np.abs(y.astype(np.complex128)) ** 2))
    tot = float(sb.reduce_sum(energy).item())
    tmax = sb.reduce_max(1.0 + rank)
    np.save(os.path.join(out_dir, "y%d.npy" % rank), y)
    np.save(os.path.join(out_dir, "meta%d.npy" % rank), np.array([tot, tmax, sb.start, sb.count]))
    # the optional exchange of SURVEY.md section 8e: every rank ends up with the whole (ragged) batch
    import torch
    full = sb.all_gather(torch.from_numpy(np.ascontiguousarray(y).view(np.float32).reshape(sb.count, 2 * n)))
    np.save(os.path.join(out_dir, "full%d.npy" % rank), full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 7])
def test_two_rank_gloo_sharded_fft(tmp_path, total):
    import torch.multiprocessing as mp
    from oracle import oracle
    from tests import util
    world, n = 2, 256
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, n, str(tmp_path)), nprocs=world, join=True)
    x = util.lcg_complex(2024, total * n).reshape(total, n)
    want = oracle.cfft(x, True)
    parts = [np.load(tmp_path / ("y%d.npy" % r)) for r in range(world)]
    metas = [np.load(tmp_path / ("meta%d.npy" % r)) for r in range(world)]
    got = np.concatenate(parts, axis=0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    e = float(np.sum(np.abs(want.astype(np.complex128)) ** 2))
    for m in metas:
        assert abs(m[0] - e) <= 1e-9 * e          # checksum of checksums agrees on every rank
        assert m[1] == 2.0                        # max over ranks of (1 + rank)
    assert [int(m[3]) for m in metas] == [c for _, c in (shard_range(total, r, world) for r in range(world))]
    for r in range(world):                        # all_gather: the whole result on every rank, bit for bit
        full = np.load(tmp_path / ("full%d.npy" % r))
        assert np.array_equal(full.view(np.uint32).reshape(-1), want.view(np.uint32).reshape(-1))


# ---- eight ranks, the product's own Python path, the library mocked at the launch boundary -----------
class _FakeLib:
    """stands in for libclfft_amd.so below opencl_fft_amd.Clcfft: the same entry points the wrapper
    calls, with the launch (clfa_fft_exec_dev) done by numpy on host memory.  Everything above the C ABI
    — plan objects, status codes, pointer / batch arithmetic, ShardedBatch — is the product's code."""

    def __init__(self):
        self.plans, self.launches = {}, []

    def clfa_cfft_create(self, href, device, n, forward):
        import ctypes
        h = len(self.plans) + 1
        self.plans[h] = (int(n), bool(forward))
        ctypes.cast(href, ctypes.POINTER(ctypes.c_void_p))[0] = h
        return 0

    def clfa_fft_get_error(self, h):
        return 0

    def clfa_fft_get_log(self, h):
        return b""

    def clfa_fft_destroy(self, h):
        pass

    def clfa_fft_exec_dev(self, h, ptr, batch, stream):
        import ctypes
        n, fwd = self.plans[getattr(h, "value", h)]
        buf = (ctypes.c_float * (2 * n * batch)).from_address(int(ptr))
        a = np.frombuffer(buf, dtype=np.complex64).reshape(batch, n)
        a[...] = (np.fft.fft(a, axis=-1) / n if fwd else np.fft.ifft(a, axis=-1) * n).astype(np.complex64)
        self.launches.append((n, batch))
        return 0


def _worker8(rank, world, port, total, n, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import opencl_fft_amd as fa
    from opencl_fft_amd.dist import ShardedBatch
    from tests import util
    fake = _FakeLib()
    fa.lib = lambda: fake                       # the launch boundary
    sb = ShardedBatch(total)
    x = util.lcg_complex(77, total * n).reshape(total, n)
    mine = torch.from_numpy(np.ascontiguousarray(x[sb.slice()]).view(np.float32).copy())
    plan = fa.Clcfft(0, n, True)
    assert plan.get_error() == 0
    sb.barrier()
    if sb.count:
        assert plan.exec_device(mine, sb.count, stream=0) == 0
    y = mine.numpy().view(np.complex64).reshape(sb.count, n)
    energy = float(np.sum(np.abs(y.astype(np.complex128)) ** 2))
    tot = float(sb.reduce_sum(energy).item())
    tmax = sb.reduce_max(0.5 + 0.25 * rank)     # "elapsed" of this rank; bench.py reports the max
    np.save(os.path.join(out_dir, "y%d.npy" % rank), y)
    np.save(os.path.join(out_dir, "meta%d.npy" % rank),
            np.array([tot, tmax, sb.start, sb.count, len(fake.launches), sum(b for _, b in fake.launches)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [32, 37, 5])
def test_eight_rank_gloo_product_path(tmp_path, total):
    """config 5's shape on CPU: 8 ranks, contiguous shards (ragged and fewer-than-ranks totals included),
    no data-path collective, checksum and max-time reductions as bench.py does with RCCL"""
    import torch.multiprocessing as mp
    from tests import util
    world, n = 8, 512
    port = _free_port()
    mp.spawn(_worker8, args=(world, port, total, n, str(tmp_path)), nprocs=world, join=True)
    x = util.lcg_complex(77, total * n).reshape(total, n)
    want = (np.fft.fft(x, axis=-1) / n).astype(np.complex64)
    parts = [np.load(tmp_path / ("y%d.npy" % r)) for r in range(world)]
    metas = [np.load(tmp_path / ("meta%d.npy" % r)) for r in range(world)]
    got = np.concatenate(parts, axis=0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "union of the shards != unsharded result"
    e = float(np.sum(np.abs(want.astype(np.complex128)) ** 2))
    for r, m in enumerate(metas):
        assert abs(m[0] - e) <= 1e-9 * e                      # every rank holds the same global checksum
        assert m[1] == 0.5 + 0.25 * (world - 1)               # max over ranks
        assert (int(m[2]), int(m[3])) == shard_range(total, r, world)
        assert int(m[5]) == int(m[3]) and int(m[4]) == (1 if m[3] else 0)   # one launch over the rank's own batches


def test_bench_launches_its_own_ranks():
    """`python3 bench.py --gpus 2` with no outer launcher (WORLD_SIZE unset): bench.py starts torch.distributed.run as a
    child before it imports torch, the two ranks rendezvous over gloo on 127.0.0.1, rank 0 alone prints the line and
    the parent exits with the children's code.  (CLFA_BENCH_REHEARSE=launch: the plumbing only — there is no GPU here,
    and the product path has no CPU fallback to rehearse with.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CLFA_BENCH_REHEARSE"] = "launch"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["max_over_ranks"] == 2.0 and r["steps"] == 3 and r["warmup"] == 1
    # failing ranks must fail the parent: without the rehearsal switch the ranks need a GPU, and there is none here
    import torch
    if torch.cuda.device_count() == 0:
        del env["CLFA_BENCH_REHEARSE"]
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode != 0
        assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
