"""Logs at level 9 through DeflaterOutputStream in 16 MiB writes, parts parsed while they arrive (the bench's 5s entry on its own):
python tools/lab/stream_logs9.py [MiB=4096] [library file in csrc/]   — writes / Finish() ms, twice"""
import hashlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from sharpziplib_amd import _lib, corpus
if len(sys.argv) > 2:
    _lib._lib = _lib._load(os.path.join(_lib.CSRC, sys.argv[2]))
from sharpziplib_amd.deflater import Deflater
from sharpziplib_amd.streams import DeflaterOutputStream
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = mb << 20
host = corpus.generate("logs", 0xE9, 0, n)


class Sink:
    def __init__(self): self.h, self.n = hashlib.sha256(), 0
    def writable(self): return True
    def write(self, b): self.h.update(b); self.n += len(b)
    def flush(self): pass
    def close(self): pass


for rep in range(2):
    sink = Sink()
    t_s = time.perf_counter()
    dos = DeflaterOutputStream(sink, Deflater(9, True), 1 << 20)
    for o in range(0, n, 16 << 20):
        dos.Write(host[o:o + (16 << 20)])
    t_w = time.perf_counter()
    dos.Finish()
    t_f = time.perf_counter()
    print("%s run %d: writes %.1f ms  Finish %.1f ms  wall %.1f ms  out %d sha %s" % (sys.argv[2] if len(sys.argv) > 2 else "libszl_amd.so", rep, (t_w - t_s) * 1e3, (t_f - t_w) * 1e3, (t_f - t_s) * 1e3, sink.n, sink.h.hexdigest()[:12]), flush=True)
    del dos
