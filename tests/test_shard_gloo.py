"""CPU tests of the multi-GPU path's host logic: ray-range sharding and the gather of hit
records, with two gloo ranks (the GPU path uses the same code over RCCL)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_batch():
    from rtk_amd import shard
    for n in (0, 1, 7, 64, 1000, 2 ** 24, 2 ** 27 + 5):
        for world in (1, 2, 3, 4, 8):
            r = [shard.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = shard.shard_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_path):
    sys.path.insert(0, ROOT)
    from rtk_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank "traces" its shard: record i is 16 bytes derived from the global ray index
    b, e = shard.shard_range(n, rank, world)
    full = (np.arange(n * 16, dtype=np.int64) % 251).astype(np.uint8)
    local = torch.from_numpy(full[b * 16:e * 16].copy())
    sizes = [s * 16 for s in shard.shard_sizes(n, world)]
    got = shard.gather_records(local, sizes, dst=0)
    dist.barrier()
    if rank == 0:
        ok = got is not None and got.numpy().tobytes() == full.tobytes()
        open(out_path, "w").write("ok" if ok else "bad")
    else:
        assert got is None
    dist.destroy_process_group()


def test_gather_records_two_ranks(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), 1001, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _pipelined_worker(rank, world, port, n, steps, out_path):
    """bench.py's N>1 step loop: two output buffers, gather of step k in flight while step k+1 'traces'."""
    sys.path.insert(0, ROOT)
    from rtk_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [n * 16] * world
    bufs = [torch.empty(n * 16, dtype=torch.uint8) for _ in range(2)]
    gathered = [torch.empty(sum(sizes), dtype=torch.uint8) if rank == 0 else None for _ in range(2)]
    pending = [[], []]
    ok = True
    for k in range(steps):
        b = k % 2
        shard.gather_records_wait(pending[b])
        if rank == 0 and k >= 2:      # the gather that just drained belongs to step k-2
            want = torch.cat([torch.full((n * 16,), (r * 7 + (k - 2)) % 251, dtype=torch.uint8) for r in range(world)])
            ok = ok and bool((gathered[b] == want).all())
        bufs[b].fill_((rank * 7 + k) % 251)          # "trace" of step k on this rank
        _, pending[b] = shard.gather_records_start(bufs[b], sizes, dst=0, out=gathered[b])
    for b in range(2):
        shard.gather_records_wait(pending[b])
    dist.barrier()
    if rank == 0:
        for k in (steps - 2, steps - 1):
            want = torch.cat([torch.full((n * 16,), (r * 7 + k) % 251, dtype=torch.uint8) for r in range(world)])
            ok = ok and bool((gathered[k % 2] == want).all())
        open(out_path, "w").write("ok" if ok else "bad")
    dist.destroy_process_group()


def test_pipelined_gather_two_ranks(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_pipelined_worker, args=(2, _free_port(), 257, 6, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _striped_worker(rank, world, port, n, out_path):
    sys.path.insert(0, ROOT)
    from rtk_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank r's shard: record i of it is 16 bytes holding (r, i)
    def shard_of(r):
        a = np.zeros((n, 4), np.uint32)
        a[:, 0] = r
        a[:, 1] = np.arange(n)
        return a
    local = torch.from_numpy(shard_of(rank).view(np.uint8).reshape(-1).copy())
    out, works, seg = shard.exchange_striped_start(local, 16)
    shard.gather_records_wait(works)
    got = out.numpy().view(np.uint32).reshape(-1, 4)
    ok = True
    at = 0
    for r in range(world):
        b, e = shard.stripe_bounds(n, world)[rank]
        want = shard_of(r)[b:e]
        ok = ok and seg[r] == (at * 16, (at + e - b) * 16) and (got[at:at + e - b] == want).all()
        at += e - b
    ok = ok and at == len(got)
    dist.barrier()
    open(out_path + str(rank), "w").write("ok" if ok else "bad")
    dist.destroy_process_group()


def _striped_unequal_worker(rank, world, port, n, out_path):
    """A batch of n records cut by shard_range over `world` ranks that do not divide it: shards differ by one record."""
    sys.path.insert(0, ROOT)
    from rtk_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.zeros((n, 4), np.uint32)
    full[:, 0] = np.arange(n)
    full[:, 1] = np.arange(n) * 7 + 1
    counts = shard.shard_sizes(n, world)
    b, e = shard.shard_range(n, rank, world)
    local = torch.from_numpy(full[b:e].view(np.uint8).reshape(-1).copy())
    ok = True
    # without the counts the ranks find out that their shards differ and raise -- all of them, nobody is left waiting
    try:
        shard.exchange_striped_start(local, 16)
        ok = False
    except ValueError:
        pass
    out, works, seg = shard.exchange_striped_start(local, 16, counts=counts)
    shard.gather_records_wait(works)
    got = out.numpy().view(np.uint32).reshape(-1, 4)
    at = 0
    for r in range(world):
        rb, _ = shard.shard_range(n, r, world)
        sb, se = shard.stripe_bounds(counts[r], world)[rank]
        ok = ok and seg[r] == (at * 16, (at + se - sb) * 16) and (got[at:at + se - sb] == full[rb + sb:rb + se]).all()
        at += se - sb
    ok = ok and at == len(got)
    dist.barrier()
    open(out_path + str(rank), "w").write("ok" if ok else "bad")
    dist.destroy_process_group()


def test_striped_exchange_unequal_shards(tmp_path):
    """1000 records over 3 ranks (334 / 333 / 333 by shard_range): receives are sized from the SENDER's shard."""
    assert len(set(__import__("rtk_amd.shard", fromlist=["x"]).shard_sizes(1000, 3))) > 1
    out = str(tmp_path / "result")
    mp.spawn(_striped_unequal_worker, args=(3, _free_port(), 1000, out), nprocs=3, join=True)
    assert [open(out + str(r)).read() for r in range(3)] == ["ok"] * 3


def test_striped_exchange_three_ranks(tmp_path):
    """The exchange that replaces the gather onto one root: every rank ends up with its stripe of every shard."""
    out = str(tmp_path / "result")
    mp.spawn(_striped_worker, args=(3, _free_port(), 1001, out), nprocs=3, join=True)
    assert [open(out + str(r)).read() for r in range(3)] == ["ok"] * 3


def test_bench_multi_rank_plumbing_dry_run():
    """bench.py's N=2 control flow (env ranks, gloo, double-buffered gather, barrier/max timing, one
    JSON line from rank 0) with the tracer replaced by a stand-in: catches plumbing errors that the
    one-GPU box cannot (the driver runs N>1 only at round end)."""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-cpu", "--frame", "64"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert "stripe" in d["config"]["gather"] and d["config"]["value_without_gather_mrays_s"] > 0 and d["config"]["value_root_gather_mrays_s"] > 0
    assert "dry-run" in d["data"]
    # N > 1 diagnostics: kernel time per rank, and how much of the exchange is not hidden behind the next trace
    k = d["config"]["per_rank_kernel_ms"]
    assert len(k["per_rank"]) == 2 and k["min"] <= k["max"]
    assert set(d["config"]["exposed_exchange_ms_per_step"]) == {"striped", "root"}


def test_bench_starts_its_own_ranks_when_called_bare():
    """The driver's command form is `python bench.py --gpus N`: without WORLD_SIZE in the environment bench.py must start the N
    ranks itself (a child torch.distributed.run; the parent never touches a GPU) and relay rank 0's one JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-cpu",
                        "--frame", "64"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "dry-run" in d["data"]
    assert "cpu_baseline" not in d


def test_bench_refuses_a_world_size_other_than_gpus():
    """--gpus N under a launcher with another WORLD_SIZE: an error exit, never a line for the wrong number of GPUs."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run-cpu",
                        "--frame", "64"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
