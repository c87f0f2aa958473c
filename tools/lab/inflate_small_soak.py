"""A randomised soak of the Inflater on SMALL members: batches of 1-3000 members of 0 bytes .. 3 MiB (clustered around the sizes where the
call's form changes: one wavefront per member, chunked members), every data class, made by this library at levels 0-9 / every strategy or by
zlib at levels 1-9 (with sync flushes), raw and zlib-framed; every member must come back equal to its input with in_consumed exact.
python tools/lab/inflate_small_soak.py [seconds=240] [seed=1] [foreign]   (foreign: zlib only, random level x memLevel x wbits x strategy, members of up to 40 MiB)"""
import sys, os, time, zlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
FOREIGN = len(sys.argv) > 3 and sys.argv[3] == "foreign"


def data(n):
    k = int(rng.integers(0, 7)); s = int(rng.integers(1, 1 << 20))
    if n == 0: return np.zeros(0, np.uint8)
    if k == 0: return C.generate("enwik", s, 0, n)
    if k == 1: return C.generate("logs", s, 0, n)
    if k == 2: return C.generate("dickens", s, 0, n)
    if k == 3: return C.random_bytes(n, seed=s)
    if k == 4: return C.mixed(n, seed=s)
    if k == 5: return C.zeros(n)
    return C.period10(n)


def size(big):
    k = int(rng.integers(0, 5))
    if k == 0: return int(rng.integers(0, 3000))
    if k == 1: return int(rng.integers(3000, 200000))
    if k == 2: return int(rng.choice([65536, 131072, 262144, 524288, 1 << 20])) + int(rng.integers(-5, 6))
    return int(rng.integers(200000, ((40 << 20) if FOREIGN and rng.random() < 0.3 else (3 << 20)) if big else 400000))


t0 = time.time(); calls = 0; members = 0; total = 0; bad = 0
while time.time() - t0 < budget:
    k = int(rng.choice([1, 2, 5, 17, 100, 700, 3000]))
    nowrap = bool(rng.integers(0, 2))
    pool = [data(size(k <= 17)) for _ in range(min(k, 24))]
    by_zlib = FOREIGN or rng.random() < 0.35
    if by_zlib:
        comp = []
        for d in pool:
            wb = int(rng.integers(9, 16)) if FOREIGN else 15       # ("foreign": every shape zlib can be asked for — blocks of 127 symbols at
            co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -wb if nowrap else wb,   # memLevel 1, static-only, RLE, Huffman-only)
                                  int(rng.integers(1, 10)) if FOREIGN else 8, int(rng.integers(0, 5)) if FOREIGN else 0)
            out = b""; step = int(rng.integers(1 << 12, 1 << 20))
            for o in range(0, max(1, d.size), step):
                out += co.compress(d[o:o + step].tobytes())
                if rng.random() < 0.3: out += co.flush(zlib.Z_SYNC_FLUSH)
            comp.append(out + co.flush())
    else:
        comp = [r.data for r in eng.deflate(pool, level=int(rng.integers(0, 10)), strategy=int(rng.choice([0, 0, 1, 2])), nowrap=nowrap)]
    idx = rng.integers(0, len(pool), k)
    outs = eng.inflate([comp[i] for i in idx], [pool[i].size for i in idx], nowrap=nowrap)
    for i, (r, cons) in zip(idx, outs):
        if not (r.status == 0 and r.data == pool[i].tobytes() and int(cons) == len(comp[i])):
            bad += 1
            print("MISMATCH call %d (%d members, %s, nowrap %s): member of %d bytes (%d compressed): status %d, %d bytes out, consumed %d"
                  % (calls, k, "zlib" if by_zlib else "library", nowrap, pool[i].size, len(comp[i]), r.status, len(r.data), int(cons)), flush=True)
            np.save(os.path.join(R, "gpurun_out", "inflate_small_mismatch_%d.npy" % bad), np.frombuffer(comp[i], np.uint8))
            if bad > 5: break
        total += pool[i].size; members += 1
    calls += 1
print("inflate small-member soak: %d calls, %d members, %.1f GiB of output, %s, %.0f s" % (calls, members, total / 2**30, "all equal to their input, in_consumed exact" if not bad else "%d MISMATCHES" % bad, time.time() - t0), flush=True)
