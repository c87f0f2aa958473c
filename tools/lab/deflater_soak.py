"""A soak of the streaming Deflater at the product's own part / piece sizes (no knobs): streams of 40-220 MiB written in random pieces,
sometimes flushed in the middle, drained through Deflate() with random buffer sizes or through DeflateView(), compared with the oracle's
object driven by the same calls.  python tools/lab/deflater_soak.py [cases=6] [seed=1]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.deflater import Deflater
L = _lib.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for case in range(cases):
    n = int(rng.integers(40 << 20, 220 << 20))
    kind = ["enwik", "logs", "dickens", "mixed"][int(rng.integers(0, 4))]
    data = C.mixed(n, seed=case + 3) if kind == "mixed" else C.generate(kind, 100 + case, 0, n)
    level = int(rng.choice([5, 6, 6, 7, 9]))
    nowrap = bool(rng.integers(0, 2))
    flush_at = int(rng.integers(n // 4, 3 * n // 4)) if rng.random() < 0.5 else -1
    t0 = time.time()
    o = O.Deflater(level, nowrap); ref = bytearray()
    def odrain():
        while True:
            x = o.deflate(1 << 22)
            if not x: break
            ref.extend(x)
    d = Deflater(level, nowrap); got = bytearray(); views = rng.random() < 0.5
    def ddrain():
        if views:
            while True:
                v = d.DeflateView()
                if v is None: break
                got.extend(bytes(v))
        else:
            buf = np.zeros(int(rng.choice([4096, 1 << 16, 1 << 20, 1 << 24])), np.uint8)
            while True:
                k = d.Deflate(buf)
                if k <= 0: break
                got.extend(buf[:k].tobytes())
    pos = 0; flushed = False
    while pos < n:
        w = int(min(n - pos, rng.integers(1 << 20, 32 << 20)))
        if flush_at >= 0 and not flushed and pos + w > flush_at: w = flush_at - pos if flush_at > pos else w
        c = data[pos:pos + w]
        o.set_input(c); odrain()
        d.SetInput(c); ddrain()
        pos += w
        if flush_at >= 0 and not flushed and pos >= flush_at:
            o.flush(); odrain(); d.Flush(); ddrain(); flushed = True
    o.finish()
    while not o.finished: ref.extend(o.deflate(1 << 22))
    d.Finish()
    while not d.IsFinished:
        before = len(got); ddrain()
        assert len(got) > before or d.IsFinished
    ok = bytes(got) == bytes(ref)
    print("case %d: %s %d MiB level %d %s%s %s: %d bytes, parts %d, %s (%.0f s)" % (case, kind, n >> 20, level, "raw" if nowrap else "zlib", ", Flush() at %d MiB" % (flush_at >> 20) if flush_at >= 0 else "",
          "views" if views else "Deflate()", len(got), L.szl_deflater_debug_pipe_parts(d._h), "== oracle" if ok else "DIFFERENT", time.time() - t0), flush=True)
    assert ok and d.TotalOut == len(ref) and d.TotalIn == n
    del d
print("deflater soak: all equal to the oracle", flush=True)
