"""A soak of SMALL calls against the oracle: stage C picks its range length by the call's size and data class (64 positions up to 256 KiB,
128 up to 1 MiB, the long ranges beyond), so calls of 1 byte .. 2.5 MiB — clustered around those limits — of every data class, alone and
in batches of a few (or, with "many", of 50-1500 entries of up to 300 KiB: the ZIP shape), levels 5-9, every strategy, raw and zlib.
python tools/lab/small_call_soak.py [seconds=240] [seed=1] [many | all]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
EDGES = [1 << 16, 256 << 10, 1 << 20, 2 << 20]
MANY = len(sys.argv) > 3 and sys.argv[3] == "many"
ALL = len(sys.argv) > 3 and sys.argv[3] == "all"          # levels 0-9 (DeflateStored, DeflateFast: one wavefront per stream) on calls of up to 300 KiB


def size():
    k = int(rng.integers(0, 3 if MANY or ALL else 4))
    if MANY and k == 1: return int(rng.choice([1 << 16, 1 << 15, 32506, 4096])) + int(rng.integers(-3, 4))
    if k == 0: return int(rng.integers(1, 5000))
    if k == 1: return max(1, int(rng.choice(EDGES)) + int(rng.integers(-300, 300)))
    if k == 2: return int(rng.integers(5000, 300 << 10))
    return int(rng.integers(256 << 10, 2560 << 10))


def data(n):
    k = int(rng.integers(0, 8)); s = int(rng.integers(1, 1 << 20))
    if k == 0: return C.generate("enwik", s, 0, n)
    if k == 1: return C.generate("logs", s, 0, n)
    if k == 2: return C.generate("dickens", s, 0, n)
    if k == 3: return C.random_bytes(n, seed=s)
    if k == 4: return C.mixed(n, seed=s)
    if k == 5: return C.zeros(n)
    if k == 6: return C.period10(n)
    a = C.generate("logs", s, 0, n).copy(); h = n // 2; a[h:] = C.random_bytes(n - h, seed=s)   # long matches, then none
    return a


t0 = time.time(); calls = 0; streams = 0; total = 0; bad = 0
while time.time() - t0 < budget:
    k = int(rng.choice([50, 300, 1500])) if MANY else int(rng.choice([1, 1, 1, 2, 3, 7]))
    bufs = [data(size()) for _ in range(k)]
    level = int(rng.integers(0 if ALL else 5, 10)); strategy = int(rng.choice([0, 0, 0, 1, 2])); nowrap = bool(rng.integers(0, 2))
    got = eng.deflate(bufs, level=level, strategy=strategy, nowrap=nowrap)
    for b, g in zip(bufs, got):
        want = O.deflate(b, level=level, nowrap=nowrap, strategy=strategy)
        want = want[0] if isinstance(want, tuple) else want
        if not (g.status == 0 and g.data == bytes(want)):
            bad += 1
            w = np.frombuffer(bytes(want), np.uint8); q = np.frombuffer(g.data, np.uint8); m = min(w.size, q.size)
            d = np.flatnonzero(w[:m] != q[:m])
            print("MISMATCH call %d (batch of %d) size %d level %d strategy %d nowrap %s: status %d, %d bytes against the oracle's %d, first difference at byte %s"
                  % (calls, k, b.size, level, strategy, nowrap, g.status, q.size, w.size, d[0] if d.size else "-"), flush=True)
            np.save(os.path.join(R, "gpurun_out", "small_call_mismatch_%d.npy" % bad), b)
            for lv, sg in ((level, strategy), (level, 0), (6, strategy), (9, strategy)):
                a1 = eng.deflate([b], level=lv, strategy=sg, nowrap=nowrap)[0]
                print("   alone, level %d strategy %d: %s" % (lv, sg, "equal" if a1.data == O.deflate(b, level=lv, nowrap=nowrap, strategy=sg) else "DIFFERENT"), flush=True)
        total += b.size; streams += 1
    calls += 1
print("small-call soak: %d calls, %d streams, %.1f MiB, levels %s x strategies x raw/zlib, %s, %.0f s" % (calls, streams, total / 2**20, "0-9" if ALL else "5-9", "all equal to the oracle" if not bad else "%d MISMATCHES" % bad, time.time() - t0), flush=True)
