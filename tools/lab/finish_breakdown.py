"""Where Finish() of the streaming Deflater spends its wall time for one long stream: a fresh object per stream (what GZipOutputStream
does, S/GZip/GzipOutputStream.cs:87) against one object reused through Reset() (python tools/gpu_lab.py finish_breakdown [MiB])"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import corpus as C
from sharpziplib_amd.deflater import Deflater
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
data = C.generate("enwik", 0xE9, 0, mib << 20)
piece = 16 << 20


def one(d):
    t0 = time.perf_counter()
    for o in range(0, data.size, piece):
        d.SetInput(data[o:o + piece]); d.DeflateView()
    t1 = time.perf_counter()
    d.Finish()
    v = d.DeflateView()
    t2 = time.perf_counter()
    n = len(v)
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, n


for rep in range(4):
    d = Deflater(6, True)
    w, f, n = one(d)
    print(f"fresh object  #{rep}: writes {w:6.1f} ms  Finish+view {f:6.1f} ms  ({n} bytes)", flush=True)
    del d
d = Deflater(6, True)
for rep in range(4):
    w, f, n = one(d)
    print(f"reused object #{rep}: writes {w:6.1f} ms  Finish+view {f:6.1f} ms", flush=True)
    d.Reset()
