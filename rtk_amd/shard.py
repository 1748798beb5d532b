"""Ray-batch sharding across the GPUs of one node.

Rays are independent and the BVH is read-only, so the scene is replicated on every GPU
and the ray batch is cut into contiguous ranges; the only exchange step is the gather of
the 16-byte hit records, which is a concatenation because hits come back in input order.
Pure index arithmetic + torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the
CPU tests); no device code here.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous range [begin, end) of rank `rank`: floor(r*n/R) .. floor((r+1)*n/R)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def shard_sizes(n, world):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def gather_records_start(local, sizes, dst=0, group=None, out=None):
    """Begin gathering per-rank uint8 record buffers (sizes in bytes per rank) onto rank `dst`.

    Point-to-point into dst, issued as ONE batch (one ncclGroup) so that on an xGMI mesh every
    sender streams over its own link to dst at the same time. Returns (out, works): `out` is the
    concatenated tensor on dst (None elsewhere); call gather_records_wait(works) before touching
    `out` or overwriting `local`. Nothing blocks the host, so the next trace launch overlaps the copy."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local, []
    if rank == dst:
        if out is None:
            out = torch.empty(sum(sizes), dtype=local.dtype, device=local.device)
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        out[offs[dst]:offs[dst + 1]].copy_(local, non_blocking=True)
        ops = [dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group) for r in range(world) if r != dst]
        return out, dist.batch_isend_irecv(ops)
    return None, dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst, group)])


def gather_records_wait(works):
    for w in works:
        w.wait()


def gather_records(local, sizes, dst=0, group=None):
    """Blocking form of gather_records_start/_wait."""
    out, works = gather_records_start(local, sizes, dst, group)
    gather_records_wait(works)
    return out
