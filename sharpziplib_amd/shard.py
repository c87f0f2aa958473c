"""Multi-GPU sharding of the hot path: the path shards by independent streams —
zip entries / gzip members / "shard = stream" pieces of a corpus — so ranks never exchange data on
the critical path.  The only collectives are a tiny all-gather of per-stream compressed sizes (to lay
the members out in one archive) and the timing reduction of bench.py.  One process per GPU over
torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests)."""


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) of n_items for `rank` (first n_items % world ranks get one more)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_bytes(total, rank, world, align=1 << 20):
    """Byte range of a `total`-byte corpus stream owned by `rank`, aligned to `align` (corpus block size)."""
    blocks = (total + align - 1) // align
    lo, hi = shard_range(blocks, rank, world)
    return min(lo * align, total), min(hi * align, total)


def gather_sizes(local_sizes, dist=None):
    """All ranks' per-stream compressed sizes, in rank order (a few KB: the all-gather of SURVEY §5)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(local_sizes)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(local_sizes))
    return out


def member_offsets(all_sizes):
    """Exclusive scan over every stream of every rank -> byte offset of each member in the joint output."""
    offs, pos = [], 0
    for sizes in all_sizes:
        row = []
        for s in sizes:
            row.append(pos)
            pos += s
        offs.append(row)
    return offs, pos


def max_over_ranks(value, dist=None, device=None):
    """MAX all-reduce of a python float (bench.py's timing rule)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
