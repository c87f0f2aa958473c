"""The drop-in write path (round 5): GZipOutputStream over the streaming Deflater, wall clock by write size.

    python tools/gpu_lab.py write_path [--mib 1024]

--mib MiB of text through the Python mirror of GZipOutputStream (S/GZip/GzipOutputStream.cs) into a sink that keeps what it is given; per
line: Write() size, buffer size of the stream, wall time of the Write loop and of Finish(), MiB/s.  Every configuration's output is checked
(zlib inflates it back to the input; the trailer's CRC-32 is the device's)."""
import argparse
import os
import sys
import time
import zlib

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np                                          # noqa: E402

from sharpziplib_amd import corpus                            # noqa: E402
from sharpziplib_amd.gzipstream import GZipOutputStream       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=int, default=1024)
a = ap.parse_args()
n = a.mib << 20
plain = corpus.generate("enwik", 0xE9, 0, n)
want_crc = zlib.crc32(plain)


class Sink:
    """What a file is to the stream: every write is one copy out of the buffer the caller lends for the duration of the call."""

    store = None                                          # one buffer for all runs, touched once (a fresh one costs a page fault per 4 KiB: ~100 ms per run)

    def __init__(self, cap=(plain.size // 2 + (1 << 20))):
        if Sink.store is None or len(Sink.store) < cap:
            Sink.store = bytearray(b"\x01") * cap
        self.buf, self.n = Sink.store, 0

    def writable(self):
        return True

    def write(self, b):
        k = len(b)
        memoryview(self.buf)[self.n:self.n + k] = b; self.n += k        # (one memcpy; a bytearray slice assignment copies twice)

    def getvalue(self):
        return bytes(memoryview(self.buf)[:self.n])

    def flush(self):
        pass

    def close(self):
        pass


def run(piece, bufsize, device_crc=True, data=plain):
    sink = Sink()
    t0 = time.perf_counter()
    g = GZipOutputStream(sink, bufsize, deviceCrc=device_crc)
    g.SetLevel(6); g.ModifiedTime = 0
    for o in range(0, data.size, piece):
        g.Write(data[o:o + piece])
    t1 = time.perf_counter()
    g.Finish()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, sink


for piece, bufsize, dc in ((16 << 20, 16 << 20, True), (64 << 20, 16 << 20, True), (1 << 20, 1 << 20, True), (16 << 20, 16 << 20, False), (16 << 20, 4096, True)):
    w, f, sink = run(piece, bufsize, dc)
    gz = sink.getvalue()
    assert int.from_bytes(gz[-8:-4], "little") == want_crc and zlib.decompress(gz[10:-8], -15) == plain.tobytes(), (piece, bufsize)
    best = min((run(piece, bufsize, dc)[:2] for _ in range(2)), key=lambda r: r[0] + r[1])
    print("Write(%5d KiB) buffer %5d KiB %s | writes %7.1f ms  Finish %7.1f ms | %8.1f MiB/s" % (
        piece >> 10, bufsize >> 10, "device CRC" if dc else "host CRC  ", best[0] * 1e3, best[1] * 1e3, a.mib / (best[0] + best[1])), flush=True)
best = min((run(16 << 20, 16 << 20, True)[:2] for _ in range(2)), key=lambda r: r[0] + r[1])
print("Write(16384 KiB) buffer 16384 KiB device CRC | writes %7.1f ms  Finish %7.1f ms | %8.1f MiB/s   (the first line again, everything warm)" % (best[0] * 1e3, best[1] * 1e3, a.mib / (best[0] + best[1])), flush=True)
w, f, sink = run(4096, 4096, True, plain[:64 << 20])
print("Write(    4 KiB) buffer     4 KiB device CRC | 64 MiB sample: %8.1f MiB/s (the Python mirror's per-call cost)" % (64 / (w + f)), flush=True)
