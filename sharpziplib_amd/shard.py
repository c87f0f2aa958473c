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


# ---- shard rebalance (BASELINE configs[4]: "RCCL shard rebalance") -------------------------------------------------------------
# Ranks start with unequal work: the streams a rank holds (zip entries, gzip members, pieces of a log) differ in cost per byte —
# level 9 on repetitive logs takes 3-4x the stage-B time of prose.  The plan keeps the global stream order and cuts
# it into `world` contiguous runs of about equal COST; streams that change owner travel in one all-to-all of bytes (RCCL over xGMI
# with the "nccl" backend, gloo in the CPU tests).  Nothing else is exchanged: every stream is still compressed where it lands.


def rebalance_plan(costs_per_rank):
    """costs_per_rank[r] = cost of every stream rank r holds, in global order.  Returns owner[g] for every global stream index g:
    contiguous runs, run r ends where the running cost passes r+1 shares of the total."""
    flat = [c for row in costs_per_rank for c in row]
    world = len(costs_per_rank)
    total = float(sum(flat))
    owner, acc, r = [], 0.0, 0
    for i, c in enumerate(flat):
        # a stream goes to the run its midpoint falls into (keeps runs contiguous and never leaves a later rank empty-handed
        # when streams remain)
        mid = acc + c / 2.0
        while r < world - 1 and total > 0 and mid >= total * (r + 1) / world:
            r += 1
        r = min(r, world - 1)
        owner.append(r)
        acc += c
    return owner


def rebalance(streams, costs, dist=None, device=None):
    """streams: this rank's byte buffers (1-D uint8 torch tensors on `device`), costs: their estimated costs.
    Returns [(global_index, tensor)] this rank owns after the exchange, in global order."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(enumerate(streams))
    world, rank = dist.get_world_size(), dist.get_rank()
    meta = [None] * world
    dist.all_gather_object(meta, ([int(s.numel()) for s in streams], [float(c) for c in costs]))
    owner = rebalance_plan([m[1] for m in meta])
    first = [0]
    for m in meta:
        first.append(first[-1] + len(m[0]))
    sizes = [n for m in meta for n in m[0]]
    # what I send to each rank / receive from each rank (whole streams, global order inside every pair)
    send_idx = [[g for g in range(first[rank], first[rank + 1]) if owner[g] == r] for r in range(world)]
    recv_idx = [[g for g in range(first[r], first[r + 1]) if owner[g] == rank] for r in range(world)]
    in_split = [sum(sizes[g] for g in idx) for idx in send_idx]
    out_split = [sum(sizes[g] for g in idx) for idx in recv_idx]
    dev = device if device is not None else (streams[0].device if streams else "cpu")
    parts = [streams[g - first[rank]] for idx in send_idx for g in idx]
    send = torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=dev)
    recv = torch.empty(sum(out_split), dtype=torch.uint8, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split)
    out, pos = [], 0
    for idx in recv_idx:
        for g in idx:
            out.append((g, recv[pos:pos + sizes[g]]))
            pos += sizes[g]
    out.sort(key=lambda t: t[0])
    return out
