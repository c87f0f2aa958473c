"""world_size-2 CPU test (gloo) of the multi-GPU path: stream sharding, the size all-gather and the
timing reduction.  No data-path collective exists, so each rank's compression is
replaced here by the oracle on the CPU — what is tested is the distributed plumbing bench.py uses."""
import os
import socket
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import oracle_ffi as O
    from sharpziplib_amd import corpus as C, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = 5 * (1 << 20) + 12345
        lo, hi = shard.shard_bytes(total, rank, world)
        data = C.generate("enwik", 0xE9, lo, hi - lo, threads=1)
        # this rank's shard as independent 1 MiB streams (gzip members / zip entries)
        members = [O.deflate(data[i:i + (1 << 20)], 6) for i in range(0, data.size, 1 << 20)]
        sizes = shard.gather_sizes([len(m) for m in members], dist)
        offs, joint = shard.member_offsets(sizes)
        t = shard.max_over_ranks(1.0 + rank, dist)
        # inflate own members back and check against the corpus bytes
        back = b"".join(zlib.decompress(m, -15) for m in members)
        q.put((rank, lo, hi, [len(m) for m in members], sizes, offs, joint, t, back == data.tobytes(),
               zlib.crc32(data.tobytes())))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, m0, sizes0, offs0, joint0, t0, ok0, crc0), (r1, lo1, hi1, m1, sizes1, offs1, joint1, t1, ok1, crc1) = res
    total = 5 * (1 << 20) + 12345
    assert (lo0, hi1) == (0, total) and hi0 == lo1          # shards tile the corpus exactly
    assert sizes0 == sizes1 == [m0, m1]                     # every rank sees every size
    assert offs0 == offs1 and joint0 == joint1 == sum(m0) + sum(m1)
    assert offs0[1][0] == sum(m0)                           # rank 1's first member starts after rank 0's output
    assert t0 == t1 == 2.0                                  # MAX over ranks
    assert ok0 and ok1
    # the sharded CRCs combine to the CRC of the whole stream (what a GZip trailer over the joint stream would need)
    sys.path.insert(0, ROOT)
    from sharpziplib_amd import corpus as C
    whole = C.generate("enwik", 0xE9, 0, total, threads=2).tobytes()
    assert zlib.crc32(whole[lo1:hi1], crc0) == zlib.crc32(whole)


def test_shard_helpers():
    sys.path.insert(0, ROOT)
    from sharpziplib_amd import shard
    for n in (0, 1, 7, 100000):
        for w in (1, 2, 4, 8):
            cover = []
            for r in range(w):
                lo, hi = shard.shard_range(n, r, w)
                cover += list(range(lo, hi)) if n < 1000 else []
                assert 0 <= lo <= hi <= n
            if n < 1000:
                assert cover == list(range(n))
    assert shard.shard_bytes(10 << 20, 0, 8) == (0, 2 << 20) and shard.shard_bytes(10 << 20, 7, 8) == (9 << 20, 10 << 20)
    offs, joint = shard.member_offsets([[3, 4], [5]])
    assert offs == [[0, 3], [7]] and joint == 12


# ---- shard rebalance (sharpziplib_amd/shard.py: the RCCL all-to-all of BASELINE configs[4], gloo here) ---------------------------
def _rebalance_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from sharpziplib_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 holds six streams, the first two of them expensive; rank 1 holds two cheap ones
        rng = np.random.default_rng(100 + rank)
        sizes = [5000, 7000, 300, 200, 100, 50] if rank == 0 else [400, 600]
        costs = [9.0, 9.0, 1.0, 1.0, 1.0, 1.0] if rank == 0 else [1.0, 1.0]
        streams = [torch.from_numpy(rng.integers(0, 256, n).astype(np.uint8)) for n in sizes]
        mine = shard.rebalance(streams, costs, dist)
        q.put((rank, [(g, bytes(t.numpy().tobytes())) for g, t in mine], [bytes(t.numpy().tobytes()) for t in streams]))
    finally:
        dist.destroy_process_group()


def test_rebalance_moves_whole_streams_by_cost():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rebalance_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, own0, orig0), (_, own1, orig1) = res
    everything = orig0 + orig1                               # global order: rank 0's streams, then rank 1's
    # every stream has exactly one owner afterwards, bytes unchanged, global order kept on each rank
    got = dict(own0 + own1)
    assert sorted(got) == list(range(8)) and all(got[g] == everything[g] for g in range(8))
    assert [g for g, _ in own0] == sorted(g for g, _ in own0) and [g for g, _ in own1] == sorted(g for g, _ in own1)
    # total cost 24, share 12: the second expensive stream straddles the cut with its larger half beyond it -> loads 9 / 15
    # (keeping both on rank 0 would be 18 / 6)
    assert [g for g, _ in own0] == [0] and [g for g, _ in own1] == [1, 2, 3, 4, 5, 6, 7]


def test_rebalance_plan_properties():
    sys.path.insert(0, ROOT)
    from sharpziplib_amd import shard
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 8):
        for trial in range(20):
            rows = [list(rng.uniform(0.1, 10.0, int(rng.integers(0, 12)))) for _ in range(world)]
            owner = shard.rebalance_plan(rows)
            n = sum(len(r) for r in rows)
            assert len(owner) == n and all(0 <= o < world for o in owner) and owner == sorted(owner)   # contiguous runs in rank order
            if n:
                flat = [c for r in rows for c in r]
                loads = [sum(c for c, o in zip(flat, owner) if o == r) for r in range(world)]
                assert max(loads) <= sum(flat) / world + max(flat)          # no rank is more than one stream above its share
    assert shard.rebalance_plan([[], []]) == []
