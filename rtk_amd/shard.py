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


def stripe_bounds(n_records, world):
    """[begin, end) in records of the world's stripes of one shard of n_records (the same floor rule as shard_range)."""
    return [shard_range(n_records, j, world) for j in range(world)]


def _agreed_counts(n_rec, world, group, device):
    """counts=None: every rank claims all shards are as long as its own. Verified on EVERY such call by an all_gather of one
    integer per rank: whether a rank takes part in that collective must not depend on what the rank itself has seen before (a
    per-rank cache made a step with 334 / 333 / 333 records after an agreed 333 enter the all_gather on rank 0 only). A step
    loop that wants no collective per step passes counts=."""
    mine = torch.tensor([n_rec], dtype=torch.int64, device=device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    got = [int(t.item()) for t in every]
    if any(g != n_rec for g in got):
        raise ValueError("exchange_striped_start: shards differ in size (%r records per rank): pass counts=shard_sizes(n, world)" % (got,))
    return got


def exchange_striped_start(local, record_bytes, group=None, out=None, counts=None):
    """The gather that no single link funnels: every rank ends up with stripe `rank` of EVERY shard (an all-to-all of
    record slices) instead of rank 0 ending up with everything.

    Why: a GPU that traces 16 Grays/s produces ~260 GB/s of 16-byte records, one xGMI link carries ~77 GB/s per
    direction and a root takes in at most its 7 links (~0.54 TB/s): gathered onto ONE GPU, eight shards can never
    arrive faster than ~34 Grays/s in total, however the copies are scheduled. Striped, each rank sends 1/world of
    its shard over each of its links and receives as much: per link and step 1/world of a shard, so the exchange
    hides behind the next step's trace. The whole result then lives striped across the GPUs (out[r-th segment] =
    stripe `rank` of rank r's shard, segments in rank order) -- where a distributed consumer wants it, and from where
    eight PCIe links can take it to the host at once.

    `local`: this rank's records as a flat uint8 tensor (a multiple of record_bytes). `counts`: records in EVERY rank's
    shard, in rank order (shard_sizes(n, world) for a batch cut by shard_range: shards differ by one record whenever
    world does not divide n, and a receive posted for the wrong size hangs or truncates over RCCL); None = every shard
    has as many records as this rank's -- checked against the other ranks' on every such call (an all_gather of one
    integer: every rank enters it, whatever it saw before), so that unequal shards without `counts` raise instead of hanging. Point-to-point, issued as ONE
    batch. Returns (out, works, segments): segments[r] = (begin, end) in bytes of rank r's contribution inside `out`;
    gather_records_wait(works) before touching `out` or overwriting `local`."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_rec = local.numel() // record_bytes
    if world == 1:
        return local, [], [(0, local.numel())]
    if local.numel() != n_rec * record_bytes:
        raise ValueError("exchange_striped_start: %d bytes is not a whole number of %d-byte records" % (local.numel(), record_bytes))
    # every rank's shard size (in records): the segments of `out` and the sizes of the receives need them
    if counts is None:
        counts = _agreed_counts(n_rec, world, group, local.device)
    counts = [int(c) for c in counts]
    if len(counts) != world or counts[rank] != n_rec:
        raise ValueError("exchange_striped_start: counts %r do not describe %d ranks with %d records on rank %d" % (counts, world, n_rec, rank))
    mine = [stripe_bounds(c, world)[rank] for c in counts]          # my stripe of rank r's shard, in records of that shard
    seg, at = [], 0
    for b, e in mine:
        seg.append((at, at + (e - b) * record_bytes))
        at += (e - b) * record_bytes
    if out is None:
        out = torch.empty(at, dtype=local.dtype, device=local.device)
    ops = []
    for j, (b, e) in enumerate(stripe_bounds(n_rec, world)):
        piece = local[b * record_bytes:e * record_bytes]
        if j == rank:
            out[seg[rank][0]:seg[rank][1]].copy_(piece, non_blocking=True)
        elif e > b:
            ops.append(dist.P2POp(dist.isend, piece, j, group))
    for r in range(world):
        if r != rank and seg[r][1] > seg[r][0]:
            ops.append(dist.P2POp(dist.irecv, out[seg[r][0]:seg[r][1]], r, group))
    return out, (dist.batch_isend_irecv(ops) if ops else []), seg
