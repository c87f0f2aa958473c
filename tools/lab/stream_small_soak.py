"""A soak of the streaming Deflater on SMALL streams (one ZIP entry at a time is this shape): 1 byte - 600 KiB of every data class written in
1-4 pieces, sometimes flushed between them, levels 0-9 x strategies x framing, one object Reset() from entry to entry, against the
oracle's Deflater driven by the same calls (tests/oracle_ffi.py stream_deflate).   python tools/lab/stream_small_soak.py [seconds=200] [seed=1]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C
from sharpziplib_amd.deflater import Deflater
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KINDS = ("enwik", "logs", "dickens")


def data(n):
    k = int(rng.integers(0, 7)); s = int(rng.integers(1, 1 << 20))
    if k < 3: return C.generate(KINDS[k], s, 0, n)
    if k == 3: return C.random_bytes(n, seed=s)
    if k == 4: return C.mixed(n, seed=s)
    if k == 5: return C.zeros(n)
    return C.period10(n)


objs = {}
buf = np.zeros(1 << 20, np.uint8)
t0 = time.time(); n_streams = 0; total = 0; bad = 0
while time.time() - t0 < budget:
    n = int(rng.choice([int(rng.integers(1, 4000)), int(rng.integers(4000, 70000)), 65536, int(rng.integers(70000, 600000))]))
    d = data(n)
    level = int(rng.integers(0, 10)) if rng.random() < 0.25 else int(rng.integers(5, 10))
    strategy = int(rng.choice([0, 0, 1, 2])); nowrap = bool(rng.integers(0, 2))
    chunk = max(1, n // int(rng.integers(1, 5))); flush_every = chunk if rng.random() < 0.3 else None
    want = O.stream_deflate(d, level=level, nowrap=nowrap, chunk=chunk, flush_every=flush_every, strategy=strategy)[0]
    key = nowrap
    if key not in objs: objs[key] = Deflater(level, nowrap)
    z = objs[key]; z.Reset(); z.SetLevel(level); z.SetStrategy(strategy)
    out = bytearray(); pos = since = 0

    def drain():
        while True:
            k = z.Deflate(buf)
            if k <= 0: break
            out.extend(buf[:k].tobytes())
    while pos < n:
        c = d[pos:pos + chunk]; z.SetInput(c); drain(); pos += c.size; since += c.size
        if flush_every and since >= flush_every and pos < n:
            z.Flush(); drain(); since = 0
    z.Finish(); drain()
    if not (z.IsFinished and bytes(out) == want):
        bad += 1
        print("MISMATCH stream %d: %d bytes level %d strategy %d nowrap %s chunk %d flush_every %s: %d bytes against %d"
              % (n_streams, n, level, strategy, nowrap, chunk, flush_every, len(out), len(want)), flush=True)
        np.save(os.path.join(R, "gpurun_out", "stream_small_mismatch_%d.npy" % bad), d)
        if bad > 3: break
    n_streams += 1; total += n
print("streaming small-entry soak: %d streams, %.1f MiB, %s, %.0f s" % (n_streams, total / 2**20, "all equal to the oracle" if not bad else "%d MISMATCHES" % bad, time.time() - t0), flush=True)
