#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: compressed MiB/s + ratio, raw-deflate level 6, 1 GiB input.

A "step" is one pass of the hot path (raw deflate L6 + CRC-32, all on the device) over one 1 GiB
shard of the seeded enwik-style corpus that is already resident in HBM.

  --gpus N (default mode "weak"): one process per GPU, every rank compresses its own 1 GiB shard of the corpus, no data-path
      collective (the path shards by independent streams) -> weak scaling.  Started under torch.distributed.run the script
      is one rank; started plainly with N > 1 it launches its own N ranks (127.0.0.1 rendezvous) and relays their line.
  --mode strong: ONE 1 GiB stream over N devices in one process (szl_deflate_stream_multi_device: position-range units taken
      dynamically, tokens gathered on device 0) -> "scaling": "strong".  The stream is uploaded to every device before the timed
      region: no PCIe in the step.
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
    ap.add_argument("--cpu-sample-mib", type=int, default=320)
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--quick-configs", action="store_true", help="the other configs at the round-3 sizes (2 GiB / 25000 entries / 1 GiB) instead of BASELINE.json's")
    ap.add_argument("--stub", action="store_true", help="CPU-only plumbing test: gloo, ranks sleep instead of compressing")
    ap.add_argument("--workload", choices=("headline", "cfg3", "cfg5"), default="headline",
                    help="headline: configs[1], one 1 GiB stream per GPU (the metric).  cfg3: configs[2], --entries x 64 KiB zip entries sharded by "
                         "contiguous groups over the ranks (strong scaling; sizes all-gathered for the archive's layout).  cfg5: configs[4], "
                         "--total-mib of logs at level 9 as 64 MiB streams that start on the ranks in a skewed split and change owner in an "
                         "all-to-all (shard.rebalance over RCCL) before they are compressed where they land (strong scaling)")
    ap.add_argument("--entries", type=int, default=100000, help="cfg3: zip entries in all (BASELINE: 100000)")
    ap.add_argument("--total-mib", type=int, default=4096, help="cfg5: MiB of logs in all (BASELINE: 4096)")
    ap.add_argument("--piece-mib", type=int, default=64, help="cfg5: MiB per stream")
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


def gen_parallel(kind, seed, n, threads=None):
    """the seeded corpus in parallel slices (the generator is position-addressable and releases the GIL)"""
    import threading
    import numpy as np
    from sharpziplib_amd import corpus
    threads = threads or max(1, min(32, len(os.sched_getaffinity(0))))
    out = np.empty(n, dtype=np.uint8)
    piece = 64 << 20
    offs = list(range(0, n, piece))
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                if not offs:
                    return
                o = offs.pop()
            m = min(piece, n - o)
            out[o:o + m] = corpus.generate(kind, seed, o, m)
    th = [threading.Thread(target=work) for _ in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out


def zlib_member_parallel(data, chunk=16 << 20, level=6):
    """ONE raw-deflate member produced by zlib (not by this library): chunks compressed on host threads, each closed by a sync flush
    (the last one finished) — the block structure of a foreign encoder: zlib's block sizes and trees, empty stored blocks between"""
    import threading
    import zlib
    n = len(data)
    offs = list(range(0, n, chunk))
    parts = [None] * len(offs)
    lock = threading.Lock()
    todo = list(range(len(offs)))

    def work():
        while True:
            with lock:
                if not todo:
                    return
                k = todo.pop()
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            o = offs[k]
            b = co.compress(data[o:o + chunk].tobytes())
            b += co.flush(zlib.Z_FINISH if k == len(offs) - 1 else zlib.Z_SYNC_FLUSH)
            parts[k] = b
    th = [threading.Thread(target=work) for _ in range(max(1, min(32, len(os.sched_getaffinity(0)))))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return b"".join(parts)


def extra_configs(eng, torch, dev, hip_stream, d_in, n, d_out, out_len, level, quick=False):
    """The other BASELINE configs on one GPU AT THEIR STATED SIZES (quick: the round-3 sizes), inputs resident in HBM, device time
    from the engine's HIP events; every output is checked (round trip on the device, CRC-32 against zlib, oracle bytes on samples,
    the oracle's frozen sha256 for config 5).  Each entry stands alone: one that fails says so and the others still run."""
    import hashlib
    import zlib
    import numpy as np
    import oracle_ffi as O
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    out = {}
    GiB = 1 << 30
    flags = _lib.F_NOWRAP | _lib.F_CRC32

    def alloc(nbytes):
        return torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)

    def upload(host):
        d = alloc(host.size)
        d[:host.size].copy_(torch.from_numpy(host))
        return d

    def inflate_table(streams, out_lens):
        ist = (_lib.Stream * len(streams))()
        oo = 0
        for i, (s, cap) in enumerate(zip(streams, out_lens)):
            ist[i].in_off, ist[i].in_len, ist[i].out_off, ist[i].out_cap = s.out_off, s.out_len, oo, cap
            oo += (cap + 3) & ~3
        return ist, oo

    def entry(name, nbytes, comp, ms, checked, **more):
        ach, frac = roofline(nbytes + comp, ms)
        e = {"device_ms": round(ms, 2), "mib_s": round(nbytes / 2 ** 20 / (ms * 1e-3), 1), "achieved_gb_s": ach, "roofline_frac": frac, "checked": checked}
        e.update(more)
        out[name] = e

    t_begin = time.perf_counter()
    budget_s = float(os.environ.get("SZL_BENCH_CONFIGS_BUDGET_S", "300"))   # the other configs never hold the headline up for longer than this

    def guarded(name, fn):
        t = time.perf_counter()
        if t - t_begin > budget_s:
            out[name] = {"skipped": "time budget for the other configs (%.0f s) used up" % budget_s}
            return
        try:
            fn()
        except Exception as e:                    # (an entry that cannot run — memory on a shared box — or fails its check says so; the headline stands on its own)
            out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        finally:
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
        out.setdefault("_wall_s", {})[name] = round(time.perf_counter() - t, 1)

    # ---- config 4(i): InflaterInputStream over ONE member — the 1 GiB stream the timed region just produced ...
    def c4i_1g():
        ist, oo = inflate_table([type("S", (), {"out_off": 0, "out_len": out_len})()], [n])
        back = alloc(oo)
        for _ in range(2):                              # (first call: allocations)
            eng.inflate_device(d_out.data_ptr(), back.data_ptr(), ist, flags=flags, hip_stream=hip_stream)
        ms = eng.timing()["inflate_ms"]
        assert ist[0].status == 0 and int(ist[0].out_len) == n and int(ist[0].in_consumed) == out_len and bool(torch.equal(back[:n], d_in[:n])), "device Inflater != corpus"
        entry("4i_inflate_one_%dMiB_member" % (n >> 20), n, out_len, ms, "bytes == corpus on device, in_consumed exact, CRC-32")
    if n >= (64 << 20):
        guarded("4i_inflate_one_%dMiB_member" % (n >> 20), c4i_1g)

    # ---- the CPU beside the inflate configs: the oracle's Inflater (kind "port": C restatement of C/Inflater.cs), one thread, on a bounded
    # sample of the same stream — the first 96 MiB of the corpus as one member (what one core of the box decodes while the device decodes 1 GiB)
    def cpu_inflate():
        m = min(n, 96 << 20)
        sample = d_in[:m].cpu().numpy()
        e2 = Engine()
        try:
            comp = e2.deflate([sample], level=level)[0].data
        finally:
            e2.close()
        t = time.perf_counter()
        got, back, _ = O.inflate(np.frombuffer(comp, dtype=np.uint8), max_out=m + 16)
        dt = time.perf_counter() - t
        assert got == m and zlib.crc32(back) == zlib.crc32(sample), "oracle Inflater"
        out["cpu_baseline_inflate"] = {"value": round(m / 2 ** 20 / dt, 1), "unit": "MiB/s of output", "cores": 1, "kind": "port",
                                       "sample": "one raw-deflate member of the corpus' first %d MiB (level %d), oracle Inflater, one thread" % (m >> 20, level)}
    guarded("cpu_baseline_inflate", cpu_inflate)

    big_n = (2 if quick else 8) * GiB
    host_big = gen_parallel("enwik", 0x21B0, big_n)
    d_big = upload(host_big)

    # ---- ... and ONE 8 GiB member (the stated size): the device Deflater's (window pipeline), back through the chunk-parallel decoder
    def c4i_big():
        e2 = Engine()
        try:
            st, _, ot = Engine.layout([big_n])
            o = alloc(ot)
            for _ in range(2):                                     # (the first call allocates the window's work space inside its events)
                e2.deflate_device(d_big.data_ptr(), o.data_ptr(), st, level=level, flags=flags, hip_stream=hip_stream)
            dms = e2.timing()["total_ms"]
            clen = int(st[0].out_len)
            ist, oo = inflate_table(st, [big_n])
            back = alloc(oo)
            for _ in range(2):
                e2.inflate_device(o.data_ptr(), back.data_ptr(), ist, flags=flags, hip_stream=hip_stream)
            ms = e2.timing()["inflate_ms"]
            assert ist[0].status == 0 and int(ist[0].out_len) == big_n and int(ist[0].in_consumed) == clen and ist[0].crc32 == st[0].crc32, "status / sizes / CRC-32"
            assert bool(torch.equal(back[:big_n], d_big[:big_n])), "8 GiB member: bytes differ"
            entry("4i_inflate_one_%dGiB_member" % (big_n // GiB), big_n, clen, ms, "bytes == input on device, in_consumed exact, CRC-32 == the Deflater's",
                  deflate_ms_window_pipeline=round(dms, 1), workspace_gib=round(e2._L.szl_engine_debug_workspace(e2._h) / GiB, 2))
        finally:
            e2.close()
    guarded("4i_inflate_one_%dGiB_member" % (big_n // GiB), c4i_big)

    # ---- config 4(ii): a .gz of many members — 8 GiB as 2048 x 4 MiB members, inflate only (deflated here in four calls)
    def c4ii():
        msz = 4 << 20
        m4 = big_n // msz
        e2 = Engine()
        try:
            st4, _, ot4 = Engine.layout([msz] * m4)
            o4 = alloc(ot4)
            per = 512
            for a in range(0, m4, per):
                sub = (_lib.Stream * per)()
                for k in range(per):
                    for f in ("in_off", "in_len", "out_off", "out_cap"):
                        setattr(sub[k], f, getattr(st4[a + k], f))
                e2.deflate_device(d_big.data_ptr(), o4.data_ptr(), sub, level=level, flags=_lib.F_NOWRAP, hip_stream=hip_stream)
                for k in range(per):
                    st4[a + k].out_len = sub[k].out_len
            comp4 = sum(int(s.out_len) for s in st4)
            ist, oo = inflate_table(st4, [msz] * m4)
            b4 = alloc(oo)
            for _ in range(2):
                e2.inflate_device(o4.data_ptr(), b4.data_ptr(), ist, flags=flags, hip_stream=hip_stream)
            ms = e2.timing()["inflate_ms"]
            assert all(s.status == 0 and int(s.in_consumed) == int(t.out_len) for s, t in zip(ist, st4)) and bool(torch.equal(b4[:m4 * msz], d_big[:m4 * msz])), "members differ"
            entry("4ii_inflate_%d_x_4MiB_members" % m4, m4 * msz, comp4, ms, "every member == input on device, in_consumed exact")
        finally:
            e2.close()
    guarded("4ii_inflate_members", c4ii)

    # ---- config 3: ZipOutputStream entries — 100000 x 64 KiB text entries, level 6 + CRC-32, one call (and back through the Inflater)
    def c3():
        esz = 65536
        n3 = min(25000 if quick else 100000, big_n // esz)
        e2 = Engine()
        try:
            st3, _, ot3 = Engine.layout([esz] * n3)
            o3 = alloc(ot3)
            for _ in range(2):
                e2.deflate_device(d_big.data_ptr(), o3.data_ptr(), st3, level=level, flags=flags, hip_stream=hip_stream)
            tm = e2.timing()
            comp3 = sum(int(s.out_len) for s in st3)
            for i in (0, 1, n3 // 2, n3 - 1):
                s = st3[i]
                got = o3[s.out_off:s.out_off + s.out_len].cpu().numpy().tobytes()
                assert got == O.deflate(host_big[i * esz:(i + 1) * esz], level) and s.crc32 == zlib.crc32(host_big[i * esz:(i + 1) * esz].tobytes()), "entry %d" % i
            checked = "4 entries == oracle bytes + CRC-32; all entries inflated back below"
            gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "headline_golden.json")))["cases"].get("cfg3_100000x64k_enwik_l6")
            if gold and n3 * esz == gold["n"] and level == gold["level"] and esz == gold["entry"]:
                # EVERY entry against the oracle's frozen digests: sha256 over the entries' compressed bytes in order, and over their CRC-32s
                ho = o3.cpu().numpy()
                h, hc = hashlib.sha256(), hashlib.sha256()
                for s in st3:
                    h.update(ho[s.out_off:s.out_off + s.out_len])
                    hc.update(int(s.crc32).to_bytes(4, "little"))
                assert comp3 == gold["out_len"] and h.hexdigest() == gold["out_sha256"] and hc.hexdigest() == gold["crc_sha256"], "config 3: the entries differ from the oracle's frozen digest"
                del ho
                checked = "ALL %d entries: sha256 of their compressed bytes and of their CRC-32s == the oracle's frozen digests; all inflated back below" % n3
            entry("3_zip_%d_x_64KiB_deflate" % n3, n3 * esz, comp3, tm["total_ms"], checked,
                  ratio=round(comp3 / (n3 * esz), 4), stage_ms={k: round(v, 2) for k, v in tm.items() if k.endswith("_ms")})
            # the same entries ONE AT A TIME through the streaming object, as the unchanged ZipOutputStream drives it: PutNextEntry ->
            # deflater_.Reset() + SetLevel (S/Zip/ZipOutputStream.cs:494-495), Write -> SetInput + Deflate() until IsNeedingInput,
            # CloseEntry -> Finish() + the Deflate() loop (CS/DeflaterOutputStream.cs:100-118); wall clock per entry, bytes == the batch call's
            from sharpziplib_amd.deflater import Deflater
            n3s = min(2000, n3)
            ho3 = o3[:int(st3[n3s - 1].out_off + st3[n3s - 1].out_len)].cpu().numpy()
            dz = Deflater(level, True)
            buf = np.zeros(1 << 17, np.uint8)
            for timed in (False, True):
                t_e = time.perf_counter()
                for i in range(n3s if timed else 50):
                    dz.Reset(); dz.SetLevel(level)
                    dz.SetInput(host_big[i * esz:(i + 1) * esz])
                    k = dz.Deflate(buf)
                    dz.Finish()
                    while not dz.IsFinished:
                        k += dz.Deflate(buf[k:])
                    if timed and (i % 16 == 0 or i == n3s - 1):
                        s = st3[i]
                        assert buf[:k].tobytes() == ho3[s.out_off:s.out_off + s.out_len].tobytes(), "per-entry path: entry %d differs from the batch call's" % i
                dt_e = time.perf_counter() - t_e
            del dz
            out["3s_ZipOutputStream_per_entry_%d" % n3s] = {"ms_per_entry": round(dt_e / n3s * 1e3, 3), "mib_s": round(n3s * esz / 2 ** 20 / dt_e, 1),
                                                          "checked": "every 16th entry == the batch call's bytes (which are checked against the oracle's frozen digests)",
                                                          "note": "one Reset / SetInput / Finish per 64 KiB entry: ~20 dependent kernels per entry; wall clock incl. the Python mirror's calls"}
            ist, oo = inflate_table(st3, [esz] * n3)
            b3 = alloc(oo)
            for _ in range(2):
                e2.inflate_device(o3.data_ptr(), b3.data_ptr(), ist, flags=_lib.F_NOWRAP, hip_stream=hip_stream)
            ms = e2.timing()["inflate_ms"]
            assert all(s.status == 0 for s in ist) and bool(torch.equal(b3[:n3 * esz], d_big[:n3 * esz])), "inflate round trip"
            entry("3_zip_%d_x_64KiB_inflate" % n3, n3 * esz, comp3, ms, "every entry == input on device")
        finally:
            e2.close()
    guarded("3_zip_entries", c3)

    # ---- a member this library did not produce: 1 GiB through zlib (sync-flushed 16 MiB pieces), default knobs, chunk-parallel decoder
    def cz():
        zn = GiB
        comp = np.frombuffer(zlib_member_parallel(host_big[:zn]), dtype=np.uint8)
        dz = upload(comp)
        back = alloc(zn)
        ist = (_lib.Stream * 1)()
        ist[0].in_off, ist[0].in_len, ist[0].out_off, ist[0].out_cap = 0, comp.size, 0, zn
        e2 = Engine()
        try:
            for _ in range(2):
                e2.inflate_device(dz.data_ptr(), back.data_ptr(), ist, flags=flags, hip_stream=hip_stream)
            ms = e2.timing()["inflate_ms"]
            assert ist[0].status == 0 and int(ist[0].out_len) == zn and int(ist[0].in_consumed) == comp.size and bool(torch.equal(back[:zn], d_big[:zn])), "zlib member"
            entry("4z_inflate_one_1GiB_member_made_by_zlib", zn, comp.size, ms, "bytes == input on device, in_consumed exact (zlib level 6, 64 sync-flushed pieces)",
                  par_jobs=int(e2._L.szl_engine_debug_par_jobs(e2._h)))
        finally:
            e2.close()
    guarded("4z_inflate_zlib_member", cz)
    del d_big, host_big
    torch.cuda.empty_cache()

    # ---- config 5: Deflater level 9 on repetitive logs, 4 GiB (the window pipeline), sha256 against the oracle's frozen output
    def c5():
        n5 = (1 if quick else 4) * GiB
        host5 = gen_parallel("logs", 0x106, n5)
        d5 = upload(host5)
        e2 = Engine()
        try:
            st5, _, ot5 = Engine.layout([n5])
            o5 = alloc(ot5)
            for _ in range(2):
                e2.deflate_device(d5.data_ptr(), o5.data_ptr(), st5, level=9, flags=flags, hip_stream=hip_stream)
            tm = e2.timing()
            c5n = int(st5[0].out_len)
            checked = []
            gpath = os.path.join(ROOT, "tests", "golden", "headline_golden.json")
            g = json.load(open(gpath))["cases"].get("cfg5_logs_4g_l9") if os.path.exists(gpath) else None
            if g and n5 == g["n"]:
                got = hashlib.sha256(o5[:c5n].cpu().numpy().tobytes()).hexdigest()
                assert c5n == g["out_len"] and got == g["out_sha256"] and int(st5[0].crc32) == g["crc32"], "4 GiB level 9: device output != the oracle's (golden sha256)"
                checked.append("sha256 + CRC-32 == oracle golden (tests/golden, 4 GiB)")
            else:
                sample = 32 << 20
                ps, _, pot = Engine.layout([sample])
                po = alloc(pot)
                e2.deflate_device(d5.data_ptr(), po.data_ptr(), ps, level=9, flags=_lib.F_NOWRAP, hip_stream=hip_stream)
                assert po[:int(ps[0].out_len)].cpu().numpy().tobytes() == O.deflate(host5[:sample], 9), "first 32 MiB as a stream != oracle"
                checked.append("first 32 MiB as a stream == oracle bytes")
            ist, oo = inflate_table(st5, [n5])
            b5 = alloc(oo)
            e2.inflate_device(o5.data_ptr(), b5.data_ptr(), ist, flags=_lib.F_NOWRAP, hip_stream=hip_stream)
            ims = e2.timing()["inflate_ms"]
            assert ist[0].status == 0 and bool(torch.equal(b5[:n5], d5[:n5])), "round trip"
            checked.append("device round trip == input")
            # the same stream through DeflaterOutputStream (host mirror of CS/DeflaterOutputStream.cs over the streaming Deflater): Write() in
            # 16 MiB pieces, Finish(); parts of it are parsed while the pieces still arrive.  Wall clock, host buffers both ways.
            if g and n5 == g["n"]:
                from sharpziplib_amd.deflater import Deflater
                from sharpziplib_amd.streams import DeflaterOutputStream

                class Sink5:
                    def __init__(self):
                        self.h, self.n = hashlib.sha256(), 0

                    def writable(self):
                        return True

                    def write(self, b):
                        self.h.update(b); self.n += len(b)

                    def flush(self):
                        pass

                    def close(self):
                        pass
                sink = Sink5()
                t_s = time.perf_counter()
                dos = DeflaterOutputStream(sink, Deflater(9, True), 1 << 20)
                for o in range(0, n5, 16 << 20):
                    dos.Write(host5[o:o + (16 << 20)])
                t_w = time.perf_counter()
                dos.Finish()
                t_f = time.perf_counter()
                assert sink.n == g["out_len"] and sink.h.hexdigest() == g["out_sha256"], "DeflaterOutputStream over the 4 GiB stream != the oracle's frozen bytes"
                out["5s_DeflaterOutputStream_%dGiB_level9_wall" % (n5 // GiB)] = {"wall_ms": round((t_f - t_s) * 1e3, 1), "writes_ms": round((t_w - t_s) * 1e3, 1), "finish_ms": round((t_f - t_w) * 1e3, 1),
                                                                               "mib_s": round(n5 / 2 ** 20 / (t_f - t_s), 1), "write_bytes": 16 << 20,
                                                                               "checked": "sha256 of everything written to the base stream == oracle golden (the sink hashes inside the timed region)"}
                del dos
            entry("5_level9_logs_%dGiB_deflate" % (n5 // GiB), n5, c5n, tm["total_ms"], "; ".join(checked), ratio=round(c5n / n5, 4),
                  stage_ms={k: round(v, 2) for k, v in tm.items() if k.endswith("_ms")}, workspace_gib=round(e2._L.szl_engine_debug_workspace(e2._h) / GiB, 2),
                  inflate_back_ms=round(ims, 1))
        finally:
            e2.close()
    guarded("5_level9_logs", c5)

    # ---- config 4 through the classes it names: GZipInputStream / InflaterInputStream (host mirrors of S/GZip/GzipInputStream.cs and
    # CS/InflaterInputStream.cs over the streaming C ABI) as a caller constructs them.  Wall clock of Read() loops over a 256 MiB member,
    # host buffers both ways.  `default_ctor`: GZipInputStream(stream) — 4096 — over the device-aware InflaterInputBuffer (16 MiB read-ahead,
    # pinned; INTEGRATION.md file 3); `reference_sizes`: the unmodified buffer class (readAhead=0), one wavefront per 64 KiB piece.
    def cstream():
        import io
        from sharpziplib_amd.gzipstream import GZipInputStream, member_footer, member_header
        from sharpziplib_amd.inflater import Inflater
        from sharpziplib_amd.streams import InflaterInputStream
        m = 256 << 20
        plain = gen_parallel("enwik", 6, m)
        crc_plain = zlib.crc32(plain.tobytes())
        e2 = Engine()
        try:
            comp, comp_small = [r.data for r in e2.deflate([plain, plain[:8 << 20]], level=level)]
        finally:
            e2.close()
        gz = member_header(0) + comp + member_footer(crc_plain, m)

        def read_all(make, n_out, check):
            st = make()
            buf = np.zeros(4 << 20, np.uint8)
            got, crc = 0, 0
            t = time.perf_counter()
            while True:
                k = st.Read(buf, 0, buf.size)
                if k <= 0:
                    break
                got += k
                if check:
                    crc = zlib.crc32(buf[:k], crc)
            dt = time.perf_counter() - t
            assert got == n_out, "read path: %d of %d bytes" % (got, n_out)
            if check:
                assert crc == zlib.crc32(plain[:n_out].tobytes()), "read path: bytes differ"
            pieces = int(_lib.lib().szl_inflater_debug_bulk_calls(st.inf._h))
            st.IsStreamOwner = False
            st.Dispose()
            return n_out / 2 ** 20 / dt, pieces
        rates = {}
        r, _ = read_all(lambda: InflaterInputStream(io.BytesIO(comp_small), Inflater(True), 65536, readAhead=0), 8 << 20, False)
        rates["reference_sizes_64_KiB_buffer_mib_s"] = round(r, 1)    # (one wavefront: an 8 MiB member is enough to see it)
        for key, make in (("default_ctor", lambda: GZipInputStream(io.BytesIO(gz))),
                          ("hint_64_MiB", lambda: GZipInputStream(io.BytesIO(gz), 64 << 20)),
                          ("InflaterInputStream_default_ctor", lambda: InflaterInputStream(io.BytesIO(comp), Inflater(True)))):
            read_all(make, m, True)                                   # checked, untimed (also the first call's allocations)
            best = max(read_all(make, m, False) for _ in range(2))
            rates[key + "_mib_s"] = round(best[0], 1)
            rates[key + "_parallel_pieces"] = best[1]
        out["4s_GZipInputStream_256MiB_member_by_constructor"] = dict(rates, checked="length; zlib.crc32 of every byte read (untimed pass); the member's CRC-32 / ISIZE trailer against the device CRC-32 (timed passes)",
                                                                     note="wall clock incl. every host copy of the adapter path; Python mirrors of the reference's classes (DESIGN.md §5)")
    guarded("4s_InflaterInputStream", cstream)

    # ---- config 2 through the class it names: GZipOutputStream (host mirror of S/GZip/GzipOutputStream.cs over the streaming Deflater) on
    # the headline's own 1 GiB, host buffers both ways, wall clock from the first Write to the last byte in the sink.  The Deflater keeps
    # what is written in pinned memory and uploads it while the caller is still writing; the CRC-32 runs on the device.  Two call patterns:
    # 16 MiB writes (what the C ABI sustains), and the reference's 4 KiB habit on a 64 MiB sample (the Python mirror's per-call cost).
    def cgzip():
        from sharpziplib_amd.gzipstream import GZipOutputStream
        host = d_in[:n].cpu().numpy()

        class Sink:                                                   # what a file is to the stream: one copy per write, out of the lent buffer
            store = bytearray(b"\x01") * (n // 2 + (1 << 20))      # one buffer for all runs, its pages touched (a fresh one: a page fault per 4 KiB)

            def __init__(self):
                self.buf, self.n = Sink.store, 0

            def writable(self):
                return True

            def write(self, b):
                k = len(b)
                memoryview(self.buf)[self.n:self.n + k] = b; self.n += k        # (one memcpy; a bytearray slice assignment copies twice)

            def flush(self):
                pass

            def close(self):
                pass

        def run(data, piece, bufsize):
            sink = Sink()
            t = time.perf_counter()
            g = GZipOutputStream(sink, bufsize)
            g.SetLevel(level); g.ModifiedTime = 0
            for o in range(0, data.size, piece):
                g.Write(data[o:o + piece])
            g.Finish()
            return time.perf_counter() - t, sink
        run(host[:64 << 20], 16 << 20, 16 << 20)                      # (first call: allocations)
        dt, sink = min((run(host, 16 << 20, 16 << 20) for _ in range(2)), key=lambda r: r[0])
        gz = bytes(memoryview(sink.buf)[:sink.n])
        body = gz[10:-8]
        crc = int.from_bytes(gz[-8:-4], "little")
        assert gz[:4] == b"\x1f\x8b\x08\x00" and int.from_bytes(gz[-4:], "little") == (n & 0xFFFFFFFF), "gzip framing"
        checked = "trailer CRC-32 == zlib.crc32 of the input, ISIZE"
        assert crc == zlib.crc32(host), "device CRC-32 of the written bytes"
        gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "headline_golden.json")))["cases"].get("cfg2_enwik_1g_l6")
        if gold and n == gold["n"] and level == gold["level"]:
            assert len(body) == gold["out_len"] and hashlib.sha256(body).hexdigest() == gold["out_sha256"], "GZipOutputStream body != the oracle's frozen bytes"
            checked += "; deflate body sha256 == oracle golden"
        else:
            assert zlib.decompress(body, -15) == host.tobytes(), "GZipOutputStream body does not inflate to the input"
            checked += "; body inflates to the input (zlib)"
        dt4, sink4 = run(host[:64 << 20], 4096, 4096)
        out["2s_GZipOutputStream_%dMiB_wall" % (n >> 20)] = {"wall_ms": round(dt * 1e3, 1), "mib_s": round(n / 2 ** 20 / dt, 1), "write_bytes": 16 << 20, "checked": checked,
                                                           "reference_habit_4KiB_writes_64MiB_sample_mib_s": round(64 / dt4, 1),
                                                           "note": "wall clock incl. every host copy; Python mirror of the reference's class (the 4 KiB figure is Python call overhead: 3 calls through ctypes per write)"}
    guarded("2s_GZipOutputStream", cgzip)
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
    """ONE stream over all devices of the node (one process): only rank 0 works when started under a launcher.  The stream is RESIDENT
    on every device before the timed region (szl_deflate_stream_multi_device): no PCIe in the step."""
    import hashlib
    import torch
    from sharpziplib_amd import _lib, corpus
    from sharpziplib_amd.batch import Engine, deflate_stream_multi_device
    if rank == 0:
        n = args.mib << 20
        host = corpus.generate("enwik", 0xE9, 0, n)
        ndev = int(_lib.lib().szl_device_count())
        assert ndev >= args.gpus, "--gpus %d but the library sees %d device(s)" % (args.gpus, ndev)
        devices = list(range(args.gpus))
        d_ins = []
        for g in devices:
            t = torch.empty(n + 64, dtype=torch.uint8, device=torch.device("cuda", g))
            t[:n].copy_(torch.from_numpy(host))
            d_ins.append(t)
        streams, _, out_total = Engine.layout([n])
        d_out = torch.empty(out_total + 64, dtype=torch.uint8, device=torch.device("cuda", devices[0]))
        flags = _lib.F_NOWRAP | _lib.F_CRC32

        def step():
            deflate_stream_multi_device([t.data_ptr() for t in d_ins], d_out.data_ptr(), devices, streams, level=args.level, flags=flags)
        for _ in range(args.warmup):
            step()
        for g in devices:
            torch.cuda.synchronize(g)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        for g in devices:
            torch.cuda.synchronize(g)
        elapsed = time.perf_counter() - t0
        assert streams[0].status == 0
        out_len = int(streams[0].out_len)
        comp = d_out[:out_len].cpu().numpy().tobytes()
        gpath = os.path.join(ROOT, "tests", "golden", "headline_golden.json")
        checked = "not the golden workload"
        if args.mib == 1024 and args.level == 6 and os.path.exists(gpath):
            g = json.load(open(gpath))["cases"]["cfg2_enwik_1g_l6"]
            assert out_len == g["out_len"] and hashlib.sha256(comp).hexdigest() == g["out_sha256"] and int(streams[0].crc32) == g["crc32"], \
                "the stream compressed by %d devices differs from the oracle's (golden sha256)" % args.gpus
            checked = "sha256 == oracle golden"
        line = {"metric": METRIC, "value": round(n * args.steps / elapsed / 2 ** 20, 1), "unit": "MiB/s", "n_gpus": args.gpus, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "u8", "data": "synthetic", "ratio": round(out_len / n, 5),
                "config": {"workload": "configs[1] as ONE %d MiB stream over %d device(s) (szl_deflate_stream_multi_device: position-range units, "
                                       "tokens gathered on device 0); the stream is resident on every device before the timed region" % (args.mib, args.gpus),
                           "level": args.level, "parallelism": "one-stream x%d" % args.gpus, "input_resident": True},
                "parity": [checked]}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()


# ---- the multi-rank shapes of BASELINE configs[2] and configs[4] -------------------------------------------------------------------------
# Same launch contract, same JSON schema (n_gpus, roofline, parity) as the headline; the total work is fixed, so "scaling" is "strong".
# --stub runs the whole host side — sharding, the size all-gather, the archive layout, the rebalance all-to-all (gloo), the MAX
# reduction — with a stand-in for the device call (tests/test_bench_launcher.py, world 2).

def _device_setup(local_rank):
    import torch
    from sharpziplib_amd import _lib
    torch.cuda.set_device(local_rank)
    _lib.check(_lib.lib().szl_set_device(local_rank), "szl_set_device")
    dev = torch.device("cuda", local_rank)
    return torch, dev, torch.cuda.current_stream(dev).cuda_stream


def _timed(step, steps, warmup, dist, sync):
    """the contract's timed region: warm-up, barrier + synchronize on both sides, MAX over ranks"""
    from sharpziplib_amd import shard
    for _ in range(warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    return shard.max_over_ranks(time.perf_counter() - t0, dist)


def _gather_bytes_in_order(dist, rank, world, pieces, feed):
    """every rank's compressed streams to rank 0, in global order, for ONE sha256 over all of them (outside the timed region).
    pieces: [(global index, uint8 tensor)] of this rank, ascending.  feed(global index, bytes) runs on rank 0."""
    import torch
    meta = [None] * world
    if dist is not None:
        dist.all_gather_object(meta, [(g, int(t.numel())) for g, t in pieces])
    else:
        meta = [[(g, int(t.numel())) for g, t in pieces]]
    if rank == 0:
        order = sorted((g, r, k) for r in range(world) for k, (g, _) in enumerate(meta[r]))
        mine = dict((g, t) for g, t in pieces)
        bufs = {}
        for r in range(1, world):                               # one receive per peer: its streams back to back
            tot = sum(n for _, n in meta[r])
            buf = torch.empty(tot, dtype=torch.uint8, device=pieces[0][1].device if pieces else "cpu")
            if tot:
                dist.recv(buf, src=r)
            offs, pos = {}, 0
            for g, n in meta[r]:
                offs[g] = (pos, n)
                pos += n
            bufs[r] = (buf.cpu().numpy(), offs)
        for g, r, _ in order:
            if r == 0:
                feed(g, mine[g].cpu().numpy())
            else:
                b, offs = bufs[r]
                o, n = offs[g]
                feed(g, b[o:o + n])
    elif pieces:
        import torch
        dist.send(torch.cat([t for _, t in pieces]), dst=0)


def run_cfg3(args, rank, local_rank, world, dist):
    """configs[2]: N entries of 64 KiB (ZipOutputStream's independent streams), contiguous groups per rank, level 6 + CRC-32 in one
    device call per rank; sizes all-gathered -> every entry's offset in the one archive the host would write (S/Zip/ZipOutputStream.cs:
    885-908 lays local headers, payloads and the central directory out from exactly these)."""
    import hashlib
    import numpy as np
    from sharpziplib_amd import shard
    esz = 65536
    n3 = args.entries
    lo, hi = shard.shard_range(n3, rank, world)
    cnt = hi - lo
    if args.stub:
        sizes = [esz // 3 + (g % 7) for g in range(lo, hi)]
        elapsed = _timed(lambda: time.sleep(0.001 * (1 + rank)), args.steps, args.warmup, dist, lambda: None)
        k_ms, ratio, stage, parity = 0.0, 1 / 3, {}, ["stub"]
    else:
        import zlib
        import oracle_ffi as O
        from sharpziplib_amd import _lib
        from sharpziplib_amd.batch import Engine
        torch, dev, hip_stream = _device_setup(local_rank)
        host = gen_parallel("enwik", 0x21B0, cnt * esz) if lo == 0 else None
        if host is None:
            from sharpziplib_amd import corpus
            host = np.concatenate([corpus.generate("enwik", 0x21B0, o, min(64 << 20, hi * esz - o)) for o in range(lo * esz, hi * esz, 64 << 20)])
        d_in = torch.empty(cnt * esz + 64, dtype=torch.uint8, device=dev)
        d_in[:cnt * esz].copy_(torch.from_numpy(host))
        eng = Engine()
        st3, _, ot3 = Engine.layout([esz] * cnt)
        d_out = torch.empty(ot3 + 64, dtype=torch.uint8, device=dev)
        flags = _lib.F_NOWRAP | _lib.F_CRC32
        tms = []

        def step():
            eng.deflate_device(d_in.data_ptr(), d_out.data_ptr(), st3, level=args.level, flags=flags, hip_stream=hip_stream)
            tms.append(eng.timing())
        elapsed = _timed(step, args.steps, args.warmup, dist, lambda: torch.cuda.synchronize(dev))
        tms = tms[-args.steps:]
        k_ms = sum(t["match_ms"] for t in tms) / len(tms)
        stage = {k: round(sum(t[k] for t in tms) / len(tms), 3) for k in tms[0] if k.endswith("_ms")}
        sizes = [int(s.out_len) for s in st3]
        assert all(s.status == 0 for s in st3)
        for i in sorted({0, cnt // 2, cnt - 1}):                # this rank's own spot check against the oracle run here
            s = st3[i]
            got = d_out[s.out_off:s.out_off + s.out_len].cpu().numpy().tobytes()
            assert got == O.deflate(host[i * esz:(i + 1) * esz], args.level) and s.crc32 == zlib.crc32(host[i * esz:(i + 1) * esz].tobytes()), "rank %d entry %d" % (rank, lo + i)
        parity = ["every rank: 3 of its entries == oracle bytes + CRC-32"]
    all_sizes = shard.gather_sizes(sizes, dist)
    offs, total = shard.member_offsets(all_sizes)               # where every entry's payload lands in the joint archive
    crcs = shard.gather_sizes([0] * cnt if args.stub else [int(s.crc32) for s in st3], dist)
    if not args.stub:
        h = hashlib.sha256()
        comp_ordered = [(lo + i, d_out[s.out_off:s.out_off + s.out_len]) for i, s in enumerate(st3)]
        # rank 0 hashes ALL entries in archive order (peers send their payloads back to back: ~2.5 GB in all, outside the timed region)
        packed = [(lo, torch.cat([t for _, t in comp_ordered]))] if comp_ordered else []
        _gather_bytes_in_order(dist, rank, world, packed, lambda g, b: h.update(b))
    if rank == 0:
        assert sum(len(r) for r in all_sizes) == n3 and offs[-1][-1] + all_sizes[-1][-1] == total
        comp3 = total
        ratio = comp3 / (n3 * esz)
        if not args.stub:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "headline_golden.json")))["cases"].get("cfg3_100000x64k_enwik_l6")
            if gold and n3 * esz == gold["n"] and args.level == gold["level"]:
                hc = hashlib.sha256()
                for row in crcs:
                    for c in row:
                        hc.update(int(c).to_bytes(4, "little"))
                assert comp3 == gold["out_len"] and h.hexdigest() == gold["out_sha256"] and hc.hexdigest() == gold["crc_sha256"], \
                    "config 3 over %d rank(s): the entries differ from the oracle's frozen digests" % world
                parity.append("ALL %d entries, gathered in archive order from %d rank(s): sha256 of their compressed bytes and of their CRC-32s == the oracle's frozen digests" % (n3, world))
        value = n3 * esz * args.steps / elapsed / 2 ** 20
        line = {"metric": "zip entries: raw deflate level %d + CRC-32 of %d x 64 KiB independent streams (uncompressed MiB/s consumed, whole job)" % (args.level, n3),
                "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "u8", "data": "stub" if args.stub else "synthetic", "ratio": round(ratio, 5),
                "config": {"workload": "configs[2]: ZipOutputStream-style entries, %d x 64 KiB of enwik-style text (seed 0x21B0), contiguous groups of entries per GPU, "
                                       "one device call per rank; the ranks exchange only the entries' compressed sizes (all-gather) for the archive's layout; "
                                       "bit-identical to the oracle (C restatement of the reference Deflater)" % n3,
                           "level": args.level, "entries": n3, "parallelism": "entries-per-gpu x%d" % world, "archive_payload_bytes": int(total)},
                "roofline": {"bound": "hbm", "kernel": "k_match9", "achieved": None, "peak": HBM_PEAK, "unit": "GB/s", "frac": None, "traffic": None,
                             "kernel_ms": round(k_ms, 3), "note": "rank 0's stage B per step"},
                "stage_ms": stage, "parity": parity}
        if k_ms > 0:
            alg = (all_sizes[0] and (len(all_sizes[0]) * esz + sum(all_sizes[0]))) or 0   # rank 0's launch: its entries read once, their output written once
            line["roofline"]["achieved"], line["roofline"]["frac"] = roofline(alg, k_ms)
            line["roofline"]["algorithmic_bytes"] = int(alg)
        print(json.dumps(line), flush=True)


def run_cfg5(args, rank, local_rank, world, dist):
    """configs[4]: level 9 on repetitive logs, "with RCCL shard rebalance".  The corpus is cut into streams of --piece-mib ("shard =
    stream": each its own Deflater lifetime, bit-exact per stream); they START on the ranks in a skewed split (rank r holds r + 1
    shares), every step moves them to an even split of cost in ONE all-to-all of bytes (shard.rebalance: all_to_all_single over nccl =
    RCCL over xGMI) and compresses them where they land.  The step is the exchange plus the compression."""
    import hashlib
    import numpy as np
    import torch
    from sharpziplib_amd import shard
    psz = args.piece_mib << 20
    npieces = max(1, (args.total_mib << 20) // psz)
    # the skewed start: cumulative shares 1 : 2 : ... : world
    tri = world * (world + 1) // 2
    bounds = [0]
    for r in range(world):
        bounds.append(min(npieces, round(npieces * sum(range(1, r + 2)) / tri)))
    bounds[-1] = npieces
    lo, hi = bounds[rank], bounds[rank + 1]
    if args.stub:
        dev = "cpu"
        small = 4096
        arena = torch.zeros((hi - lo) * small, dtype=torch.uint8)
        for i in range(hi - lo):
            arena[i * small:(i + 1) * small] = (lo + i) & 0xFF
        views = [arena[i * small:(i + 1) * small] for i in range(hi - lo)]
        sync = lambda: None   # noqa: E731
        hip_stream = 0
    else:
        import zlib  # noqa: F401
        from sharpziplib_amd import _lib, corpus
        from sharpziplib_amd.batch import Engine
        _, dev, hip_stream = _device_setup(local_rank)
        arena = torch.empty((hi - lo) * psz + 64, dtype=torch.uint8, device=dev)
        for i in range(hi - lo):
            arena[i * psz:(i + 1) * psz].copy_(torch.from_numpy(corpus.generate("logs", 0x106, (lo + i) * psz, psz)))
        views = [arena[i * psz:(i + 1) * psz] for i in range(hi - lo)]
        sync = lambda: torch.cuda.synchronize(dev)   # noqa: E731
        eng = Engine()
        flags = _lib.F_NOWRAP | _lib.F_CRC32
    state = {}
    tms, xms = [], []

    def step():
        t = time.perf_counter()
        own = shard.rebalance(views, [float(v.numel()) for v in views], dist, dev)     # [(global piece, tensor)] in global order
        sync()
        xms.append((time.perf_counter() - t) * 1e3)
        if args.stub:
            time.sleep(0.0005 * len(own))
            state.update(own=[(lo + g if dist is None else g, t_) for g, t_ in own], sizes=[int(t_.numel()) // 3 for _, t_ in own], crcs=[0] * len(own), out=None)
            return
        if not own:
            state.update(own=[], sizes=[], crcs=[], out=None)
            return
        # the streams that landed here lie in one buffer (the all-to-all's receive buffer, or the start arena on one rank)
        base = min(t_.data_ptr() for _, t_ in own)
        st5, _, ot5 = Engine.layout([int(t_.numel()) for _, t_ in own])
        for s, (_, t_) in zip(st5, own):
            s.in_off = t_.data_ptr() - base
        if state.get("cap", 0) < ot5:
            state["d_out"] = torch.empty(ot5 + 64, dtype=torch.uint8, device=dev); state["cap"] = ot5
        eng.deflate_device(base, state["d_out"].data_ptr(), st5, level=9, flags=flags, hip_stream=hip_stream)
        tms.append(eng.timing())
        state.update(own=[(lo + g if dist is None else g, t_) for g, t_ in own], st=st5, sizes=[int(s.out_len) for s in st5], crcs=[int(s.crc32) for s in st5])
    elapsed = _timed(step, args.steps, args.warmup, dist, sync)
    own = state["own"]
    rows = shard.gather_sizes([(g, n, c) for (g, _), n, c in zip(own, state["sizes"], state["crcs"])], dist)
    h = hashlib.sha256()
    if not args.stub:
        assert all(s.status == 0 for s in state.get("st", []))
        d_out = state.get("d_out")
        comp = [(g, d_out[s.out_off:s.out_off + s.out_len]) for (g, _), s in zip(own, state.get("st", []))]
        _gather_bytes_in_order(dist, rank, world, comp, lambda g, b: h.update(b))
    if rank == 0:
        flat = sorted(x for row in rows for x in row)
        assert [g for g, _, _ in flat] == list(range(npieces)), "every stream compressed exactly once"
        after = [len(row) for row in rows]
        total_in, total_out = npieces * (4096 if args.stub else psz), sum(n for _, n, _ in flat)
        parity = ["stub"] if args.stub else []
        if not args.stub:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "headline_golden.json")))["cases"].get("cfg5_64x64m_logs_l9")
            if gold and total_in == gold["n"] and psz == gold["entry"]:
                hc = hashlib.sha256()
                for _, _, c in flat:
                    hc.update(int(c).to_bytes(4, "little"))
                assert total_out == gold["out_len"] and h.hexdigest() == gold["out_sha256"] and hc.hexdigest() == gold["crc_sha256"], \
                    "config 5 over %d rank(s): the streams differ from the oracle's frozen digests" % world
                parity.append("ALL %d streams, gathered in order from %d rank(s) after the rebalance: sha256 of their compressed bytes and of their CRC-32s == the oracle's frozen digests" % (npieces, world))
            else:
                parity.append("not the golden workload (sizes only)")
        k_ms = (sum(t["match_ms"] for t in tms[-args.steps:]) / max(1, len(tms[-args.steps:]))) if tms else 0.0
        line = {"metric": "Deflater level 9 on %d MiB of repetitive logs as %d streams, shard rebalance + compression (uncompressed MiB/s consumed, whole job)" % (total_in >> 20, npieces),
                "value": round(total_in * args.steps / elapsed / 2 ** 20, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "u8", "data": "stub" if args.stub else "synthetic", "ratio": round(total_out / total_in, 5),
                "config": {"workload": "configs[4]: Deflater level 9 (max_chain 4096) on %d MiB of log data (seed 0x106) as %d streams of %d MiB; skewed start "
                                       "(rank r holds r + 1 shares), ONE all-to-all of bytes per step to an even split of cost (shard.rebalance over %s), compressed "
                                       "where they land; bit-identical to the oracle (C restatement of the reference Deflater) per stream"
                                       % (total_in >> 20, npieces, args.piece_mib, "gloo" if args.stub else "nccl = RCCL"),
                           "level": 9, "parallelism": "streams-per-gpu x%d after rebalance" % world,
                           "streams_per_rank_before": [bounds[r + 1] - bounds[r] for r in range(world)], "streams_per_rank_after": after},
                "rebalance_ms_rank0": round(sum(xms[-args.steps:]) / max(1, len(xms[-args.steps:])), 3),
                "roofline": {"bound": "hbm", "kernel": "k_match9", "achieved": None, "peak": HBM_PEAK, "unit": "GB/s", "frac": None, "traffic": None,
                             "kernel_ms": round(k_ms, 3), "note": "rank 0's stage B per step"},
                "parity": parity}
        if k_ms > 0 and rows[0]:
            alg = len(rows[0]) * psz + sum(n for _, n, _ in rows[0])
            line["roofline"]["achieved"], line["roofline"]["frac"] = roofline(alg, k_ms)
            line["roofline"]["algorithmic_bytes"] = int(alg)
        print(json.dumps(line), flush=True)


def main():
    args = parse_args()
    launched = "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1 and (args.mode == "weak" or args.workload != "headline"):
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
        if args.workload == "cfg3":
            return run_cfg3(args, rank, local_rank, world, dist)
        if args.workload == "cfg5":
            return run_cfg5(args, rank, local_rank, world, dist)
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
        # roofline of the dominant kernel (k_match9, stage B): algorithmic bytes per launch = input read once +
        # output written once = n*(1+ratio), divided by the kernel's mean duration measured with
        # HIP events on the launch stream inside the timed region.
        k_ms = sum(kern_ms) / len(kern_ms)
        alg_bytes = n * (1.0 + ratio)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        # HBM traffic cannot be counted from inside the process: it comes from the committed rocprofv3 --pmc passes of this
        # same command (tools/gpu_traffic.sh -> profiles/rNN/*traffic_pmc.json); `traffic_source` says which file
        tpath = next((t for t in ([os.path.join(ROOT, "profiles", r, "traffic_pmc.json") for r in ("r06", "r05", "r04", "r03", "r02")] +
                                  [os.path.join(ROOT, "profiles", "r01", "g_traffic_pmc.json")]) if os.path.exists(t)), "")
        if args.mib == 1024 and args.level == 6 and tpath:
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units) of this same command, per launch of k_match;
            # FETCH_SIZE doubled for wide coalesced streaming reads on gfx950 (MI355X_MICROARCH.md §HBM)
            rec = json.load(open(tpath))
            k = next((v for kk, v in rec.items() if "k_match9" in kk), None)   # counters of THIS kernel only (profiles of earlier rounds hold k_match4's)
            if k:
                traffic = int((2 * k["fetch"] + k["write"]) * 1024 / max(1, k.get("dispatches", 1)))
        line = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "ratio": round(sum(s[0] for s in all_sizes) / (world * n), 5), "compressed_mib_s": round(value * ratio, 1),
            "config": {"workload": "configs[1]: GZip-style raw Deflater level %d + CRC-32 on one %d MiB enwik-style stream per GPU "
                                   "(seed 0xE9, shard = rank), bit-identical to the oracle (C restatement of the reference Deflater; the reference itself cannot run here)" % (args.level, args.mib),
                       "level": args.level, "shard_mib": args.mib, "parallelism": "stream-per-gpu x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "k_match9", "achieved": round(achieved, 2), "peak": HBM_PEAK, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 5), "traffic": traffic,
                         "traffic_source": (os.path.relpath(tpath, ROOT) + " (rocprofv3 PMC pass of this command; not measured in this run)") if traffic else None,
                         "kernel_ms": round(k_ms, 3), "algorithmic_bytes": int(alg_bytes),
                         "whole_pass_frac": round(alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK, 5)},
            "stage_ms": {k: round(v, 3) for k, v in stage.items()},
        }
        if not args.no_cpu_baseline and world == 1:   # (rank 0 at N = 1 only: the contract)
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
            # all host cores this process may use: independent oracle Deflaters on disjoint 16 MiB slices, pthreads inside oracle/
            # (szo_deflate_slices_mt) — what a host-side "shard = stream" run over the same corpus could reach on this box; the
            # reference itself is single-threaded per Deflater.  `cores` = the threads that ran = the CPUs the scheduler gives this
            # process (affinity mask and cgroup quota), not the machine's core count (`host_cores`).
            import ctypes
            usable = len(os.sched_getaffinity(0))
            quota = None
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    quota = float(q) / float(per)
            except Exception:
                try:
                    q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    if q > 0:
                        quota = q / per
                except Exception:
                    pass
            nthr = max(1, min(usable, int(quota + 0.5) if quota else usable, n // (16 << 20)))
            nsl = max(nthr, min(n // (16 << 20), 2 * nthr))
            OL = O.lib()
            OL.szo_deflate_slices_mt.restype = ctypes.c_int64
            OL.szo_deflate_slices_mt.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            lens = np.zeros(nsl, dtype=np.uint64)
            t2 = time.perf_counter()
            tot = OL.szo_deflate_slices_mt(host.ctypes.data, 16 << 20, nsl, args.level, nthr, lens.ctypes.data)
            dt2 = time.perf_counter() - t2
            assert tot > 0
            line["cpu_baseline_all_cores"] = {"value": round(nsl * 16 / dt2, 1), "unit": "MiB/s", "cores": nthr, "kind": "port",
                                              "host_cores": os.cpu_count(), "usable_cpus": usable, "cpu_quota": quota,
                                              "sample": "%d independent 16 MiB slices of the same shard, one oracle Deflater per pthread (%d threads)" % (nsl, nthr)}
        if world == 1 and not args.no_extra_configs:
            try:
                line["configs"] = extra_configs(eng, torch, dev, hip_stream, d_in, n, d_out, out_len, args.level, quick=args.quick_configs)
            except Exception as e:                      # (e.g. not enough memory on a shared box: the headline stands on its own)
                line["configs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        line["parity_checked_bytes"] = parity["checked_bytes"]
        line["parity"] = parity["how"]
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
