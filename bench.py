#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: compressed MiB/s + ratio, raw-deflate level 6, 1 GiB input.

A "step" is one pass of the hot path (raw deflate L6 + CRC-32, all on the device) over one 1 GiB
shard of the seeded enwik-style corpus that is already resident in HBM.  N>1: one process per GPU
(launched by torch.distributed.run), every rank compresses its own 1 GiB shard of the corpus, no
data-path collective (the path shards by independent streams) -> weak scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib", type=int, default=1024, help="shard size per GPU in MiB (BASELINE: 1024)")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=384)
    args = ap.parse_args()

    import numpy as np
    import torch
    from sharpziplib_amd import _lib, corpus, shard
    from sharpziplib_amd.batch import Engine

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    torch.cuda.set_device(local_rank)
    _lib.check(_lib.lib().szl_set_device(local_rank), "szl_set_device")
    dev = torch.device("cuda", local_rank)

    n = args.mib << 20
    seed = 0xE9
    lo, hi = shard.shard_bytes(world * n, rank, world)            # this rank's shard of the corpus stream (one stream per GPU)
    assert hi - lo == n
    host = corpus.generate("enwik", seed, lo, n)
    d_in = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    d_in[:n].copy_(torch.from_numpy(host))
    eng = Engine()
    streams, _, out_total = Engine.layout([n])
    d_out = torch.empty(out_total + 64, dtype=torch.uint8, device=dev)
    flags = _lib.F_NOWRAP | _lib.F_CRC32
    hip_stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        eng.deflate_device(d_in.data_ptr(), d_out.data_ptr(), streams, level=args.level, flags=flags, hip_stream=hip_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    kern_ms = []
    stage = {}
    for _ in range(args.steps):
        step()
        tm = eng.timing()
        kern_ms.append(tm["match_ms"])
        for k, v in tm.items():
            if k.endswith("_ms"):
                stage[k] = stage.get(k, 0.0) + v / args.steps
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, dist, dev)
    all_sizes = shard.gather_sizes([int(streams[0].out_len)], dist)   # the only other collective: a few bytes per rank

    out_len = int(streams[0].out_len)
    ratio = out_len / n
    # correctness of what was timed: CRC of the input from the device vs zlib, and the stream inflates back (rank 0, cheap)
    parity = {"checked_bytes": 0, "how": []}
    if rank == 0:
        import hashlib
        import zlib
        comp = d_out[:out_len].cpu().numpy().tobytes()
        assert zlib.crc32(host.tobytes()) == streams[0].crc32, "device CRC-32 mismatch"
        assert zlib.decompress(comp, -15) == host.tobytes(), "device output does not inflate to the input"
        # bit-exactness of the very stream that was timed: sha256 of the oracle's output for this workload, frozen in
        # tests/golden/headline_golden.json (tests/golden/make_headline.py; tests/test_gpu_headline.py also runs the oracle itself)
        gpath = os.path.join(ROOT, "tests", "golden", "headline_golden.json")
        if args.mib == 1024 and args.level == 6 and os.path.exists(gpath):
            g = json.load(open(gpath))["cases"]["cfg2_enwik_1g_l6"]
            assert out_len == g["out_len"] and hashlib.sha256(comp).hexdigest() == g["out_sha256"], \
                "the timed stream's device output differs from the oracle's (golden sha256)"
            assert int(streams[0].crc32) == g["crc32"]
            parity["checked_bytes"] += n
            parity["how"].append("timed 1 GiB stream: sha256 == oracle golden")

    if rank == 0:
        value = world * n * args.steps / elapsed / 2 ** 20
        # roofline of the dominant kernel (k_match4, stage B): algorithmic bytes per launch = input read once +
        # output written once = n*(1+ratio), divided by the kernel's mean duration measured with
        # HIP events on the launch stream inside the timed region.
        k_ms = sum(kern_ms) / len(kern_ms)
        alg_bytes = n * (1.0 + ratio)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        # HBM traffic cannot be counted from inside the process: it comes from the committed rocprofv3 --pmc passes of this
        # same command (tools/gpu_traffic.sh -> profiles/rNN/*traffic_pmc.json); `traffic_source` says which file
        tpath = next((t for t in ([os.path.join(ROOT, "profiles", "r02", f) for f in ("traffic_pmc.json",)] +
                                  [os.path.join(ROOT, "profiles", "r01", f) for f in ("g_traffic_pmc.json",)])
                      if os.path.exists(t)), "")
        if args.mib == 1024 and args.level == 6 and tpath:
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units) of this same command, per launch of k_match;
            # FETCH_SIZE doubled for wide coalesced streaming reads on gfx950 (MI355X_MICROARCH.md §HBM)
            rec = json.load(open(tpath))
            k = next((v for kk, v in rec.items() if "k_match4" in kk or "k_match<" in kk), None)   # the search proper, not the pilot (k_match_lazy)
            if k:
                traffic = int((2 * k["fetch"] + k["write"]) * 1024 / max(1, k.get("dispatches", 1)))
        line = {
            "metric": "raw-deflate level 6 throughput (uncompressed MiB/s consumed), 1 GiB enwik-style input, CRC-32 on device",
            "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "ratio": round(sum(s[0] for s in all_sizes) / (world * n), 5), "compressed_mib_s": round(value * ratio, 1),
            "config": {"workload": "configs[1]: GZip-style raw Deflater level %d + CRC-32 on one %d MiB enwik-style stream per GPU "
                                   "(seed 0xE9, shard = rank), bit-identical to the reference Deflater" % (args.level, args.mib),
                       "level": args.level, "shard_mib": args.mib, "parallelism": "stream-per-gpu x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "k_match4", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                         "traffic_source": (os.path.relpath(tpath, ROOT) + " (rocprofv3 PMC pass of this command; not measured in this run)") if traffic else None,
                         "kernel_ms": round(k_ms, 3), "algorithmic_bytes": int(alg_bytes)},
            "stage_ms": {k: round(v, 3) for k, v in stage.items()},
        }
        if not args.no_cpu_baseline:
            import oracle_ffi as O
            sample = min(args.cpu_sample_mib << 20, n)
            t1 = time.perf_counter()
            ref = O.deflate(host[:sample], args.level)
            dt = time.perf_counter() - t1
            cpu_model = ""
            try:
                cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                pass
            line["cpu_baseline"] = {"value": round(sample / dt / 2 ** 20, 2), "unit": "MiB/s", "cores": 1, "kind": "port",
                                    "cpu": cpu_model, "host_cores": os.cpu_count(),
                                    "sample": "first %d MiB of the same shard through oracle/ (C restatement of the managed Deflater, "
                                              "single thread like the reference); %d bytes out" % (sample >> 20, len(ref))}
            # the same prefix as its own stream on the device: byte-for-byte against what the oracle just produced
            pstreams, _, pout_total = Engine.layout([sample])
            p_out = torch.empty(pout_total + 64, dtype=torch.uint8, device=dev)
            eng.deflate_device(d_in.data_ptr(), p_out.data_ptr(), pstreams, level=args.level, flags=flags, hip_stream=hip_stream)
            got = p_out[:int(pstreams[0].out_len)].cpu().numpy().tobytes()
            assert got == ref, "device output of the cpu_baseline sample differs from the oracle's"
            parity["checked_bytes"] += sample
            parity["how"].append("first %d MiB as its own stream: bytes == oracle run in this process" % (sample >> 20))
            # all host cores: one independent oracle Deflater per core on disjoint 16 MiB slices (what a host-side
            # "shard = stream" run over the same corpus could reach; the reference itself is single-threaded per Deflater)
            import threading
            ncore = min(os.cpu_count() or 1, max(1, n // (16 << 20)))
            outs = [0] * ncore

            def work(i):
                outs[i] = len(O.deflate(host[i * (16 << 20):(i + 1) * (16 << 20)], args.level))
            th = [threading.Thread(target=work, args=(i,)) for i in range(ncore)]
            t2 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt2 = time.perf_counter() - t2
            line["cpu_baseline_all_cores"] = {"value": round(ncore * 16 / dt2, 1), "unit": "MiB/s", "cores": ncore, "kind": "port",
                                              "sample": "%d independent 16 MiB slices, one oracle Deflater per thread" % ncore}
        line["parity_checked_bytes"] = parity["checked_bytes"]
        line["parity"] = parity["how"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
