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
