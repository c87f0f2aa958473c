"""gfxsim.fuzz — random programs / batches / damaged streams on the interpreted device against the oracle (no GPU needed).

    python tools/gfxsim/fuzz.py deflater <first seed> <count> [--big]     random Deflater programs (tests/test_gpu_setlevel.py's driver:
                                                                            SetInput pieces, Flush, SetLevel / SetStrategy across all three
                                                                            functions with bytes pending); --big: 70-140 KB per program, so
                                                                            the 64 KiB window slides and the history tail is carried
    python tools/gfxsim/fuzz.py batch <seed> <rounds>                      random batches: sizes (boundaries favoured), classes, levels 0-9,
                                                                            strategies, framing, checksums
    python tools/gfxsim/fuzz.py inflate <seed> <rounds>                    bit flips, truncations, header-region flips, quirk code sets
                                                                            (tests/test_gpu_inflate_fuzz.py's generators), batch + streaming object
    python tools/gfxsim/fuzz.py par <seed> <rounds>                        members of text, stored runs and deflate payload — whole, damaged,
                                                                            truncated — through the chunk-parallel Inflater against the
                                                                            one-wavefront decoder

One process per invocation (several in parallel use several cores).  Prints one line per mismatch and a summary; exit code 1 on any.
Test infrastructure only.  Round 4's campaign: profiles/r04/gfxsim_fuzz_campaign.log.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np                                   # noqa: E402


def fuzz_deflater(base, count, big):
    import test_gpu_setlevel as TS
    from gfxsim import harness
    chunks = (1, 3, 100, 261, 262, 263, 700, 5000, 20000, 40000, 70000) if big else (1, 2, 3, 50, 100, 261, 262, 263, 264, 500, 700, 1500, 3000)
    totals = [70000, 100000, 140000] if big else [600, 1500, 4000, 9000]
    ok = bad = 0
    for seed in range(base, base + count):
        rng = np.random.default_rng(seed)
        kind = int(rng.integers(0, 6))
        total = int(rng.choice(totals))
        kw = [dict(levels=[5, 6, 7, 8, 9]), dict(levels=[5, 6, 9], strategies=(0, 1, 2)), dict(levels=[1, 2, 3, 4], strategies=(0, 2)),
              dict(levels=[1, 2, 3, 4, 5, 6, 7, 9], flush_p=0.4, cross_kind_at_flush=True), dict(levels=[1, 2, 3, 4, 5, 6, 7, 9], cross_p=0.7),
              dict(levels=[0, 1, 3, 4, 5, 6, 9], cross_p=0.8, flush_p=float(rng.choice([0.02, 0.25, 0.5])))][kind]
        levels = kw.pop("levels")
        try:
            TS._run(levels, seed, total=total, chunk_sizes=chunks, **kw)
            ok += 1
        except AssertionError as ex:
            bad += 1
            print("MISMATCH seed %d kind %d total %d: %s" % (seed, kind, total, str(ex)[:300]), flush=True)
        except Exception as ex:
            bad += 1
            print("ERROR seed %d kind %d total %d: %s | %s" % (seed, kind, total, str(ex)[:300], harness.errors()[-1:]), flush=True)
    return ok + bad, bad


def fuzz_batch(seed, rounds):
    import oracle_ffi as O
    from gfxsim import harness
    from sharpziplib_amd import corpus as C
    from sharpziplib_amd.batch import Engine
    e = Engine()
    gens = [lambda s, n: C.generate("dickens", s, 0, n), lambda s, n: C.generate("logs", s, 0, n), lambda s, n: C.generate("enwik", s, 0, n),
            lambda s, n: C.random_bytes(n, seed=s), lambda s, n: C.zeros(n), lambda s, n: C.four_symbol(n), lambda s, n: C.period10(n),
            lambda s, n: C.mixed(n, seed=s)]
    edge = [0, 1, 2, 3, 4, 5, 258, 259, 260, 261, 262, 263, 264, 520, 1023, 1024, 1025]
    tot = bad = 0
    for rd in range(rounds):
        rng = np.random.default_rng(seed * 1000 + rd)
        bufs = []
        for i in range(int(rng.integers(1, 6))):
            n = int(rng.choice(edge)) if rng.random() < 0.4 else int(rng.integers(0, 4000))
            g = gens[int(rng.integers(0, len(gens)))]
            bufs.append(g(int(rng.integers(1, 1 << 20)), n) if n else np.zeros(0, np.uint8))
        lv = int(rng.integers(0, 10)); st = int(rng.choice([0, 0, 0, 1, 2])); nowrap = bool(rng.integers(0, 2))
        try:
            res = e.deflate(bufs, level=lv, strategy=st, nowrap=nowrap, crc32=True, adler32=True)
            for i, (b, r) in enumerate(zip(bufs, res)):
                tot += 1
                ref = O.deflate(b, lv, nowrap=nowrap, strategy=st)
                if not (r.status == 0 and r.data == ref and r.crc32 == O.crc32(b) and r.adler32 == O.adler32(b)):
                    bad += 1
                    print("MISMATCH seed %d round %d stream %d: n %d level %d strategy %d nowrap %s (%d vs %d bytes)" % (
                        seed, rd, i, b.size, lv, st, nowrap, len(r.data), len(ref)), flush=True)
        except Exception as ex:
            bad += 1
            print("ERROR seed %d round %d: %s %s" % (seed, rd, str(ex)[:200], harness.errors()[-1:]), flush=True)
    return tot, bad


def fuzz_inflate(seed, rounds):
    import oracle_ffi as O
    import corrupt_streams as CS
    import test_gpu_inflate_fuzz as TF
    from sharpziplib_amd import corpus as C
    from sharpziplib_amd.batch import Engine
    e = Engine()
    TF.CAP = 8192
    tot = bad = 0
    for rd in range(rounds):
        rng = np.random.default_rng(seed * 1000 + rd)
        valid = []
        for kind in ("dickens", "logs", "enwik"):
            d = C.generate(kind, int(rng.integers(1, 1 << 20)), 0, int(rng.integers(200, 3000)))
            lv = int(rng.integers(1, 10))
            valid.append(("%s_L%d" % (kind, lv), O.deflate(d, lv, flush=bool(rng.integers(0, 2)))))
        # runs of stored blocks — the decoder follows them inside its copy loop (round 5): zlib's level 0 cut into short blocks by sync
        # flushes (empty stored blocks in between), and incompressible bytes through the reference's own level 0 / level 6
        import zlib
        co = zlib.compressobj(0, zlib.DEFLATED, -15)
        parts = []
        for _ in range(int(rng.integers(3, 9))):
            parts.append(co.compress(C.generate("enwik", int(rng.integers(1, 1 << 20)), 0, int(rng.integers(100, 1500))).tobytes()))
            parts.append(co.flush(zlib.Z_SYNC_FLUSH if rng.integers(0, 2) else zlib.Z_FULL_FLUSH))
        parts.append(co.flush())
        valid.append(("zlib0_flushes", b"".join(parts)))
        rb = rng.integers(0, 256, int(rng.integers(1000, 7000)), dtype=np.uint8)
        valid.append(("random_L0", O.deflate(rb, 0)))
        valid.append(("random_L6", O.deflate(np.concatenate([rb, C.generate("logs", 5, 0, 700), rb[::-1]]), 6)))
        cases = CS.mutations(valid, rng, n_flip=10, n_trunc=3)
        for name, s in valid:
            b = np.frombuffer(s, np.uint8)
            for k in range(10):
                pos = int(rng.integers(0, min(b.size, 60) * 8))
                m = b.copy(); m[pos >> 3] ^= 1 << (pos & 7)
                cases.append(("%s_hdrflip@%d" % (name, pos), m.tobytes()))
        cases += CS.quirk_set_streams(rng, 40)
        fails = TF._run_batch(e, cases) + TF._run_streaming(cases[::7])
        tot += len(cases) + len(cases[::7])
        for f in fails:
            bad += 1
            print("MISMATCH seed %d round %d: %s" % (seed, rd, f), flush=True)
    return tot, bad


def fuzz_par(seed, rounds):
    """members of 150-300 KB made of text in small blocks, runs of stored blocks and deflate data as payload — whole, damaged and truncated —
    through the chunk-parallel Inflater (16 KiB chunks) against the one-wavefront decoder: status, bytes consumed, bytes out"""
    import zlib
    from sharpziplib_amd import _lib
    from sharpziplib_amd import corpus as C
    from sharpziplib_amd.batch import Engine
    L = _lib.lib()
    e = Engine()
    tot = bad = 0
    for rd in range(rounds):
        rng = np.random.default_rng(seed * 1000 + rd)
        co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15, 4)       # memLevel 4: a block every 1024 tokens
        parts, plain = [], []
        while sum(len(x) for x in parts) < int(rng.integers(150000, 280000)):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                d = C.generate(("enwik", "logs", "dickens")[int(rng.integers(0, 3))], int(rng.integers(1, 1 << 20)), 0, int(rng.integers(20000, 90000))).tobytes()
            elif kind == 1:
                d = rng.integers(0, 256, int(rng.integers(10000, 80000)), dtype=np.uint8).tobytes()      # stored blocks
            elif kind == 2:
                ci = zlib.compressobj(6, zlib.DEFLATED, -15, 4)                                          # deflate data as payload
                t = C.generate("enwik", int(rng.integers(1, 1 << 20)), 0, int(rng.integers(40000, 120000))).tobytes()
                d = ci.compress(t) + ci.flush()
            else:
                d = bytes(int(rng.integers(1000, 30000)))                                                 # zeros
            plain.append(d); parts.append(co.compress(d))
            if rng.integers(0, 3) == 0:
                parts.append(co.flush(zlib.Z_SYNC_FLUSH if rng.integers(0, 2) else zlib.Z_FULL_FLUSH))
        parts.append(co.flush())
        m, data = b"".join(parts), b"".join(plain)
        cases = [("whole", m, len(data)), ("short room", m, len(data) - int(rng.integers(1, 50000)))]
        for k in range(3):
            b = bytearray(m); pos = int(rng.integers(0, len(m) * 8)); b[pos >> 3] ^= 1 << (pos & 7)
            cases.append(("flip@%d" % pos, bytes(b), len(data)))
        cases.append(("cut", m[:int(rng.integers(len(m) // 3, len(m)))], len(data)))
        for name, st, cap in cases:
            L.szl_debug_set(b"SZL_INF_CHUNK_KIB", 16); L.szl_debug_set(b"SZL_INF_PAR_MIN_KIB", 64)
            (rp, up), = e.inflate([st], [cap], crc32=True)
            jobs = int(L.szl_engine_debug_par_jobs(e._h))
            L.szl_debug_set(b"SZL_INF_PAR_MIN_KIB", 1 << 22)
            (rs, us), = e.inflate([st], [cap], crc32=True)
            tot += 1
            same = (rp.status, up, rp.data, rp.crc32) == (rs.status, us, rs.data, rs.crc32)
            if name == "whole":
                same = same and rp.status == 0 and rp.data == data
            if not same:
                bad += 1
                print("MISMATCH seed %d round %d %s: parallel (%d, %d, %d bytes, %d jobs) one wavefront (%d, %d, %d bytes)" % (
                    seed, rd, name, rp.status, up, len(rp.data), jobs, rs.status, us, len(rs.data)), flush=True)
    L.szl_debug_set(b"SZL_INF_CHUNK_KIB", -(2 ** 31)); L.szl_debug_set(b"SZL_INF_PAR_MIN_KIB", -(2 ** 31))
    return tot, bad


def main(argv):
    if len(argv) < 3:
        print(__doc__)
        return 2
    from gfxsim import harness
    harness.use(fast_probe=True)
    t0 = time.time()
    what, a, b = argv[0], int(argv[1]), int(argv[2])
    if what == "deflater":
        tot, bad = fuzz_deflater(a, b, "--big" in argv)
    elif what == "batch":
        tot, bad = fuzz_batch(a, b)
    elif what == "inflate":
        tot, bad = fuzz_inflate(a, b)
    elif what == "par":
        tot, bad = fuzz_par(a, b)
    else:
        print(__doc__)
        return 2
    print("done %s %d %d: %d cases, %d mismatches, %.0f s" % (what, a, b, tot, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
