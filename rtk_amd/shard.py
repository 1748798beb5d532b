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


def gather_records(local, sizes, dst=0, group=None):
    """Gather per-rank uint8 record buffers (sizes in bytes per rank) onto rank `dst`.

    Returns the concatenated tensor on dst, None elsewhere. Implemented as point-to-point
    sends into dst so that on an xGMI mesh every sender uses its own link to dst."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    if rank == dst:
        out = torch.empty(sum(sizes), dtype=local.dtype, device=local.device)
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        out[offs[dst]:offs[dst + 1]].copy_(local)
        ops = [dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group) for r in range(world) if r != dst]
        for q in dist.batch_isend_irecv(ops):   # one ncclGroup: all seven links carry data at once
            q.wait()
        return out
    for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst, group)]):
        q.wait()
    return None
