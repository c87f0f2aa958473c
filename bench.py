#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: compressed MiB/s + ratio, raw-deflate level 6, 1 GiB input.

A "step" is one pass of the hot path (raw deflate L6 + CRC-32, all on the device) over one 1 GiB
shard of the seeded enwik-style corpus that is already resident in HBM.

  --gpus N (default mode "weak"): one process per GPU, every rank compresses its own 1 GiB shard of the corpus, no data-path
      collective (the path shards by independent streams) -> weak scaling.  Started under torch.distributed.run the script
      is one rank; started plainly with N > 1 it launches its own N ranks (127.0.0.1 rendezvous) and relays their line.
  --mode strong: ONE 1 GiB stream over N devices in one process (szl_deflate_batch_multi_host: position-range units taken
      dynamically, tokens gathered on device 0) -> "scaling": "strong".  That entry point takes host buffers, so its time
      includes the PCIe copies; the line says so.
  --stub: no GPU, no library — ranks sleep instead of compressing (gloo).  tests/test_bench_launcher.py drives the launcher,
      the barriers, the MAX reduction and the JSON contract with it on the CPU.
Prints ONE JSON line on rank 0.  At N=1 the line also carries "configs": the other BASELINE configs (3, 4(i), 4(ii), 5)
timed on the device with their outputs checked (--no-extra-configs skips them).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "raw-deflate level 6 throughput (uncompressed MiB/s consumed), 1 GiB enwik-style input, CRC-32 on device"
HBM_PEAK = 8000.0   # GB/s (MI355X_MICROARCH.md)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib", type=int, default=1024, help="shard size per GPU in MiB (BASELINE: 1024)")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--mode", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=384)
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--stub", action="store_true", help="CPU-only plumbing test: gloo, ranks sleep instead of compressing")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch(args):
    """--gpus N without a launcher: start N ranks of this script under torch.distributed.run and relay rank 0's line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SZL_BENCH_CHILD"] = "1"
    return subprocess.call(cmd, env=env)


def roofline(alg_bytes, ms):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    return round(ach, 2), round(ach / HBM_PEAK, 5)


def extra_configs(eng, torch, dev, hip_stream, d_in, n, d_out, out_len, level):
    """The other BASELINE configs on one GPU, inputs resident in HBM, device time from the engine's HIP events; every output is
    checked (round trip on the device, CRC-32 against zlib, oracle bytes on samples)."""
    import zlib
    import numpy as np
    import oracle_ffi as O
    from sharpziplib_amd import _lib, corpus
    from sharpziplib_amd.batch import Engine
    out = {}

    def alloc(nbytes):
        return torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)

    def inflate_table(streams, out_lens):
        ist = (_lib.Stream * len(streams))()
        oo = 0
        for i, (s, cap) in enumerate(zip(streams, out_lens)):
            ist[i].in_off, ist[i].in_len, ist[i].out_off, ist[i].out_cap = s.out_off, s.out_len, oo, cap
            oo += (cap + 3) & ~3
        return ist, oo

    # ---- config 4(i): InflaterInputStream over ONE 1 GiB member — the stream the timed region just produced
    if n >= (64 << 20):
        ist, oo = inflate_table([type("S", (), {"out_off": 0, "out_len": out_len})()], [n])
        back = alloc(oo)
        eng.inflate_device(d_out.data_ptr(), back.data_ptr(), ist, flags=_lib.F_NOWRAP | _lib.F_CRC32, hip_stream=hip_stream)   # (first call: allocations)
        eng.inflate_device(d_out.data_ptr(), back.data_ptr(), ist, flags=_lib.F_NOWRAP | _lib.F_CRC32, hip_stream=hip_stream)
        ms = eng.timing()["inflate_ms"]
        ok = ist[0].status == 0 and int(ist[0].out_len) == n and int(ist[0].in_consumed) == out_len and bool(torch.equal(back[:n], d_in[:n]))
        assert ok, "config 4(i): the device Inflater does not return the corpus from the device Deflater's member"
        ach, frac = roofline(n + out_len, ms)
        out["4i_inflate_one_%dMiB_member" % (n >> 20)] = {"device_ms": round(ms, 2), "mib_s": round(n / 2 ** 20 / (ms * 1e-3), 1), "achieved_gb_s": ach,
                                                          "roofline_frac": frac, "checked": "bytes == corpus on device, in_consumed exact, CRC-32"}
        del back
    # ---- config 3: ZipOutputStream entries — 50000 x 64 KiB text entries, level 6 + CRC-32, one call (and back through the Inflater)
    n3, esz = 50000, 65536
    host3 = corpus.generate("enwik", 0x21B0, 0, n3 * esz)
    d3 = alloc(n3 * esz)
    d3[:n3 * esz].copy_(torch.from_numpy(host3))
    st3, _, ot3 = Engine.layout([esz] * n3)
    o3 = alloc(ot3)
    flags = _lib.F_NOWRAP | _lib.F_CRC32
    eng.deflate_device(d3.data_ptr(), o3.data_ptr(), st3, level=level, flags=flags, hip_stream=hip_stream)
    eng.deflate_device(d3.data_ptr(), o3.data_ptr(), st3, level=level, flags=flags, hip_stream=hip_stream)
    tm = eng.timing()
    comp3 = sum(int(s.out_len) for s in st3)
    for i in (0, 1, n3 // 2, n3 - 1):
        s = st3[i]
        got = o3[s.out_off:s.out_off + s.out_len].cpu().numpy().tobytes()
        assert got == O.deflate(host3[i * esz:(i + 1) * esz], level) and s.crc32 == zlib.crc32(host3[i * esz:(i + 1) * esz].tobytes()), "config 3: entry %d" % i
    ach, frac = roofline(n3 * esz + comp3, tm["total_ms"])
    out["3_zip_%d_x_64KiB_deflate" % n3] = {"device_ms": round(tm["total_ms"], 2), "mib_s": round(n3 * esz / 2 ** 20 / (tm["total_ms"] * 1e-3), 1),
                                            "ratio": round(comp3 / (n3 * esz), 4), "achieved_gb_s": ach, "roofline_frac": frac,
                                            "checked": "4 entries == oracle bytes + CRC-32; all entries inflated back below"}
    ist, oo = inflate_table(st3, [esz] * n3)
    b3 = alloc(oo)
    eng.inflate_device(o3.data_ptr(), b3.data_ptr(), ist, flags=_lib.F_NOWRAP, hip_stream=hip_stream)
    eng.inflate_device(o3.data_ptr(), b3.data_ptr(), ist, flags=_lib.F_NOWRAP, hip_stream=hip_stream)
    ms = eng.timing()["inflate_ms"]
    assert all(s.status == 0 for s in ist) and bool(torch.equal(b3[:n3 * esz], d3[:n3 * esz])), "config 3: inflate round trip"
    ach, frac = roofline(n3 * esz + comp3, ms)
    out["3_zip_%d_x_64KiB_inflate" % n3] = {"device_ms": round(ms, 2), "mib_s": round(n3 * esz / 2 ** 20 / (ms * 1e-3), 1), "achieved_gb_s": ach,
                                            "roofline_frac": frac, "checked": "every entry == input on device"}
    del o3, b3
    # ---- config 4(ii): a .gz of many members — 2 GiB as 512 x 4 MiB members (the first 2 GiB of the same bytes), inflate only
    m4, msz = 512, 4 << 20
    st4, _, ot4 = Engine.layout([msz] * m4)
    o4 = alloc(ot4)
    eng.deflate_device(d3.data_ptr(), o4.data_ptr(), st4, level=level, flags=_lib.F_NOWRAP, hip_stream=hip_stream)
    comp4 = sum(int(s.out_len) for s in st4)
    ist, oo = inflate_table(st4, [msz] * m4)
    b4 = alloc(oo)
    eng.inflate_device(o4.data_ptr(), b4.data_ptr(), ist, flags=_lib.F_NOWRAP | _lib.F_CRC32, hip_stream=hip_stream)
    eng.inflate_device(o4.data_ptr(), b4.data_ptr(), ist, flags=_lib.F_NOWRAP | _lib.F_CRC32, hip_stream=hip_stream)
    ms = eng.timing()["inflate_ms"]
    assert all(s.status == 0 and int(s.in_consumed) == int(t.out_len) for s, t in zip(ist, st4)) and bool(torch.equal(b4[:m4 * msz], d3[:m4 * msz])), "config 4(ii)"
    ach, frac = roofline(m4 * msz + comp4, ms)
    out["4ii_inflate_%d_x_4MiB_members" % m4] = {"device_ms": round(ms, 2), "mib_s": round(m4 * msz / 2 ** 20 / (ms * 1e-3), 1), "achieved_gb_s": ach,
                                                 "roofline_frac": frac, "checked": "every member == input on device, in_consumed exact"}
    del o4, b4, d3, host3
    # ---- config 5: Deflater level 9 on repetitive logs, 1 GiB
    n5 = 1 << 30
    host5 = corpus.generate("logs", 0x106, 0, n5)
    d5 = alloc(n5)
    d5[:n5].copy_(torch.from_numpy(host5))
    st5, _, ot5 = Engine.layout([n5])
    o5 = alloc(ot5)
    eng.deflate_device(d5.data_ptr(), o5.data_ptr(), st5, level=9, flags=flags, hip_stream=hip_stream)
    eng.deflate_device(d5.data_ptr(), o5.data_ptr(), st5, level=9, flags=flags, hip_stream=hip_stream)
    tm = eng.timing()
    c5 = int(st5[0].out_len)
    assert st5[0].crc32 == zlib.crc32(host5.tobytes()), "config 5: CRC-32"
    sample = 32 << 20
    ps, _, pot = Engine.layout([sample])
    po = alloc(pot)
    eng.deflate_device(d5.data_ptr(), po.data_ptr(), ps, level=9, flags=_lib.F_NOWRAP, hip_stream=hip_stream)
    assert po[:int(ps[0].out_len)].cpu().numpy().tobytes() == O.deflate(host5[:sample], 9), "config 5: first 32 MiB as a stream != oracle"
    ist, oo = inflate_table(st5, [n5])
    b5 = alloc(oo)
    eng.inflate_device(o5.data_ptr(), b5.data_ptr(), ist, flags=_lib.F_NOWRAP, hip_stream=hip_stream)
    assert ist[0].status == 0 and bool(torch.equal(b5[:n5], d5[:n5])), "config 5: round trip"
    ach, frac = roofline(n5 + c5, tm["total_ms"])
    out["5_level9_logs_1GiB_deflate"] = {"device_ms": round(tm["total_ms"], 2), "mib_s": round(n5 / 2 ** 20 / (tm["total_ms"] * 1e-3), 1),
                                         "ratio": round(c5 / n5, 4), "achieved_gb_s": ach, "roofline_frac": frac,
                                         "stage_ms": {k: round(v, 2) for k, v in tm.items() if k.endswith("_ms")},
                                         "checked": "CRC-32 == zlib, device round trip == input, first 32 MiB as a stream == oracle bytes"}
    return out


def run_stub(args, rank, world, dist):
    """Plumbing only (CPU): the same barriers, MAX reduction, size gather and JSON line as the real run."""
    from sharpziplib_amd import shard
    n = args.mib << 20
    lo, hi = shard.shard_bytes(world * n, rank, world)
    assert hi - lo == n
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))
    if dist is not None:
        dist.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, dist)
    sizes = shard.gather_sizes([n // 3 + rank], dist)
    if rank == 0:
        assert len(sizes) == world
        line = {"metric": METRIC, "value": round(world * n * args.steps / elapsed / 2 ** 20, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u8", "data": "stub", "config": {"workload": "stub (no device work)"}}
        print(json.dumps(line), flush=True)


def run_strong(args, rank, world, dist):
    """ONE stream over all devices of the node (one process): only rank 0 works when started under a launcher."""
    import hashlib
    import numpy as np
    from sharpziplib_amd import _lib, corpus
    from sharpziplib_amd.batch import deflate_multi
    if rank == 0:
        n = args.mib << 20
        host = corpus.generate("enwik", 0xE9, 0, n)
        ndev = int(_lib.lib().szl_device_count())
        assert ndev >= args.gpus, "--gpus %d but the library sees %d device(s)" % (args.gpus, ndev)
        devices = list(range(args.gpus))
        for _ in range(args.warmup):
            deflate_multi([host], devices, level=args.level, crc32=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            (r,) = deflate_multi([host], devices, level=args.level, crc32=True)
        elapsed = time.perf_counter() - t0
        assert r.status == 0
        gpath = os.path.join(ROOT, "tests", "golden", "headline_golden.json")
        checked = "not the golden workload"
        if args.mib == 1024 and args.level == 6 and os.path.exists(gpath):
            g = json.load(open(gpath))["cases"]["cfg2_enwik_1g_l6"]
            assert len(r.data) == g["out_len"] and hashlib.sha256(r.data).hexdigest() == g["out_sha256"] and r.crc32 == g["crc32"], \
                "the stream compressed by %d devices differs from the oracle's (golden sha256)" % args.gpus
            checked = "sha256 == oracle golden"
        line = {"metric": METRIC, "value": round(n * args.steps / elapsed / 2 ** 20, 1), "unit": "MiB/s", "n_gpus": args.gpus, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "u8", "data": "synthetic", "ratio": round(len(r.data) / n, 5),
                "config": {"workload": "configs[1] as ONE %d MiB stream over %d device(s) (szl_deflate_batch_multi_host: position-range units, "
                                       "tokens gathered on device 0); HOST buffers in and out: the time includes the PCIe copies" % (args.mib, args.gpus),
                           "level": args.level, "parallelism": "one-stream x%d" % args.gpus, "input_resident": False},
                "parity": [checked]}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()


def main():
    args = parse_args()
    launched = "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1 and args.mode == "weak":
        sys.exit(relaunch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if launched and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (the line must report the GPUs that ran)" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.stub else "nccl")
    try:
        if args.stub:
            return run_stub(args, rank, world, dist)
        if args.mode == "strong":
            return run_strong(args, rank, world, dist)
        run_weak(args, rank, local_rank, world, dist)
    finally:
        if dist is not None:
            dist.destroy_process_group()


def run_weak(args, rank, local_rank, world, dist):
    import numpy as np
    import torch
    from sharpziplib_amd import _lib, corpus, shard
    from sharpziplib_amd.batch import Engine

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
        assert len(all_sizes) == world == args.gpus, "the line must report the GPUs that ran"
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
        tpath = next((t for t in ([os.path.join(ROOT, "profiles", r, "traffic_pmc.json") for r in ("r03", "r02")] +
                                  [os.path.join(ROOT, "profiles", "r01", "g_traffic_pmc.json")]) if os.path.exists(t)), "")
        if args.mib == 1024 and args.level == 6 and tpath:
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units) of this same command, per launch of k_match;
            # FETCH_SIZE doubled for wide coalesced streaming reads on gfx950 (MI355X_MICROARCH.md §HBM)
            rec = json.load(open(tpath))
            k = next((v for kk, v in rec.items() if "k_match4" in kk or "k_match<" in kk), None)   # the search proper, not the pilot (k_match_lazy)
            if k:
                traffic = int((2 * k["fetch"] + k["write"]) * 1024 / max(1, k.get("dispatches", 1)))
        line = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "ratio": round(sum(s[0] for s in all_sizes) / (world * n), 5), "compressed_mib_s": round(value * ratio, 1),
            "config": {"workload": "configs[1]: GZip-style raw Deflater level %d + CRC-32 on one %d MiB enwik-style stream per GPU "
                                   "(seed 0xE9, shard = rank), bit-identical to the reference Deflater" % (args.level, args.mib),
                       "level": args.level, "shard_mib": args.mib, "parallelism": "stream-per-gpu x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "k_match4", "achieved": round(achieved, 2), "peak": HBM_PEAK, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 5), "traffic": traffic,
                         "traffic_source": (os.path.relpath(tpath, ROOT) + " (rocprofv3 PMC pass of this command; not measured in this run)") if traffic else None,
                         "kernel_ms": round(k_ms, 3), "algorithmic_bytes": int(alg_bytes),
                         "whole_pass_frac": round(alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK, 5)},
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
            del p_out
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
        if world == 1 and not args.no_extra_configs:
            try:
                line["configs"] = extra_configs(eng, torch, dev, hip_stream, d_in, n, d_out, out_len, args.level)
            except AssertionError:
                raise
            except Exception as e:                      # (e.g. not enough memory on a shared box: the headline stands on its own)
                line["configs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        line["parity_checked_bytes"] = parity["checked_bytes"]
        line["parity"] = parity["how"]
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
