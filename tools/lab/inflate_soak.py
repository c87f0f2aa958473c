"""A randomised soak of the chunk-parallel Inflater on the device: members of mixed data classes and sizes, made by this library at random
levels (0-9) or by zlib (with sync flushes), alone and in batches, inflated and compared with their input.
python tools/lab/inflate_soak.py [seconds=240] [seed=1]"""
import sys, os, time, zlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); rounds = 0; total = 0; par = 0


def piece(n):
    k = int(rng.integers(0, 7))
    if k == 0: return C.generate("enwik", int(rng.integers(1, 1 << 20)), 0, n)
    if k == 1: return C.generate("logs", int(rng.integers(1, 1 << 20)), 0, n)
    if k == 2: return C.generate("dickens", int(rng.integers(1, 1 << 20)), 0, n)
    if k == 3: return C.random_bytes(n, seed=int(rng.integers(1, 1 << 20)))
    if k == 4: return C.mixed(n, seed=int(rng.integers(1, 1 << 20)))
    if k == 5: return C.zeros(n)
    return C.period10(n)


def member():
    n = int(rng.choice([1 << 20, 3 << 20, 9 << 20, 24 << 20, 70 << 20]) * rng.uniform(0.5, 1.3))
    parts = []
    while sum(p.size for p in parts) < n:
        parts.append(piece(int(rng.integers(200000, max(200001, n // 2 + 200001)))))
    d = np.concatenate(parts)[:n]
    if rng.random() < 0.3:
        co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15)
        out = b""
        step = int(rng.integers(1 << 18, 1 << 22))
        for o in range(0, d.size, step):
            out += co.compress(d[o:o + step].tobytes())
            if rng.random() < 0.3: out += co.flush(zlib.Z_SYNC_FLUSH)
        comp = out + co.flush()
    else:
        comp = eng.deflate([d], level=int(rng.integers(0, 10)))[0].data
    return d, comp


while time.time() - t0 < budget:
    k = int(rng.choice([1, 1, 2, 5, 12]))
    ms = [member() for _ in range(k)]
    outs = eng.inflate([m[1] for m in ms], [m[0].size for m in ms])
    for (d, comp), (r, cons) in zip(ms, outs):
        assert r.status == 0 and r.data == d.tobytes() and int(cons) == len(comp), (rounds, d.size, len(comp), r.status)
        total += d.size
    par += int(L.szl_engine_debug_par_jobs(eng._h) > 0)
    rounds += 1
print("inflate soak: %d calls (%d through the chunk-parallel path), %.1f GiB of output, all equal to their input, %.0f s" % (rounds, par, total / 2**30, time.time() - t0), flush=True)
