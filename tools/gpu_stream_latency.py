"""Streaming-API latency (GPU box): the ZipOutputStream pattern — one Deflater.Reset() + SetInput + Finish() + Deflate() drain
per small entry (S/Zip/ZipOutputStream.cs:494,582-640) — against the same entries as ONE batched call.
Usage: python tools/gpu_stream_latency.py [--entries 2000] [--kib 64] [--level 6]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sharpziplib_amd import corpus, _lib
from sharpziplib_amd.deflater import Deflater
from sharpziplib_amd.batch import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--entries", type=int, default=2000)
ap.add_argument("--kib", type=int, default=64)
ap.add_argument("--level", type=int, default=6)
ap.add_argument("--inflate-mib", type=int, default=256)
a = ap.parse_args()
n = a.kib << 10
data = corpus.generate("enwik", 5, 0, n * a.entries)
d = Deflater(a.level, True)
buf = np.zeros(n + 4096, np.uint8)
outs = []
def one(i):
    d.Reset()
    d.SetInput(data[i * n:(i + 1) * n])
    d.Finish()
    tot = 0
    while not d.IsFinished:
        k = d.Deflate(buf)
        tot += k
    return tot
for i in range(20):
    one(i)                                    # warm-up (allocations, code objects)
t0 = time.perf_counter()
tot = 0
for i in range(a.entries):
    tot += one(i)
t1 = time.perf_counter()
ms = (t1 - t0) * 1e3 / a.entries
print("streaming object: %d entries of %d KiB, level %d: %.3f ms per entry = %.1f MiB/s (ratio %.3f)" % (
    a.entries, a.kib, a.level, ms, a.kib / 1024.0 / (ms / 1e3), tot / (n * a.entries)), flush=True)
eng = Engine()
bufs = [data[i * n:(i + 1) * n] for i in range(a.entries)]
eng.deflate(bufs[:64], level=a.level)
t0 = time.perf_counter()
res = eng.deflate(bufs, level=a.level)
t1 = time.perf_counter()
print("one batched call (host buffers in and out): %.3f ms per entry = %.1f MiB/s" % (
    (t1 - t0) * 1e3 / a.entries, a.entries * a.kib / 1024.0 / (t1 - t0)), flush=True)
assert sum(len(r.data) for r in res) == tot


# ---- the unchanged-host read path: InflaterInputStream (CS/InflaterInputStream.cs:115,486,658) — Fill() gives the Inflater its
# input 4 KiB at a time (the default buffer) or 64 KiB (a caller who passes bufferSize), Read() asks for 4 KiB of output at a time
import io
from sharpziplib_amd.inflater import Inflater
from sharpziplib_amd.streams import InflaterInputStream
m = a.inflate_mib << 20
plain = corpus.generate("enwik", 6, 0, m)
comp = Engine().deflate([plain], level=6)[0].data
import hashlib
want = hashlib.sha256(plain.tobytes()).hexdigest()
L = _lib.lib()
for bufsz, readsz in ((4096, 4096), (65536, 65536), (1 << 20, 1 << 20), (16 << 20, 4 << 20), (64 << 20, 4 << 20)):
    best = 0.0
    for rep in range(2):
        src = io.BytesIO(comp)
        inf = Inflater(True)
        st = InflaterInputStream(src, inf, bufsz)
        out = np.zeros(readsz, np.uint8)
        h = hashlib.sha256() if rep == 0 else None
        t0 = time.perf_counter()
        got = 0
        while True:
            k = st.Read(out, 0, readsz)
            if k <= 0:
                break
            got += k
            if h:
                h.update(out[:k].tobytes())
        dt = time.perf_counter() - t0
        assert got == m and inf.RemainingInput == 0 and inf.TotalIn == len(comp)
        if h:
            assert h.hexdigest() == want
        else:
            best = m / 2 ** 20 / dt
    print("InflaterInputStream, %8d B input buffer, Read(%7d): %8.1f MiB/s of output (%d MiB member; %d pieces through the chunk-parallel decoder)" % (
        bufsz, readsz, best, a.inflate_mib, L.szl_inflater_debug_bulk_calls(inf._h)), flush=True)
