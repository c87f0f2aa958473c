"""The drop-in read path, by constructor (round 5): GZipInputStream / InflaterInputStream over the device-aware InflaterInputBuffer.

    python tools/gpu_lab.py read_path [--mib 512] [--read-mib 4]

One gzip member of --mib MiB of text read through the Python mirrors of the reference's classes; per line: the constructor, MiB/s of
output (wall clock of the Read() loop, best of 3 after a checked run), pieces through the chunk-parallel decoder and where the object's
time went (szl_inflater_debug_times).  Every configuration's bytes are checked once (zlib.crc32 of everything read)."""
import argparse
import ctypes
import io
import os
import sys
import time
import zlib

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np                                          # noqa: E402

from sharpziplib_amd import _lib, corpus                      # noqa: E402
from sharpziplib_amd.gzipstream import GZipInputStream, write_members   # noqa: E402
from sharpziplib_amd.inflater import Inflater                 # noqa: E402
from sharpziplib_amd.streams import InflaterInputStream       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=int, default=512)
ap.add_argument("--read-mib", type=int, default=4)
ap.add_argument("--debug", action="store_true", help="one more pass of the default constructor with SZL_DEBUG=1 (the passes of every parallel piece on stderr)")
a = ap.parse_args()
L = _lib.lib()
n = a.mib << 20
plain = corpus.generate("enwik", 0xE9, 0, n)
want = zlib.crc32(plain.tobytes())
(gz,) = write_members([plain], level=6)
raw = gz[10:-8]
print("member: %d MiB of text -> %d bytes" % (a.mib, len(gz)), flush=True)
buf = np.zeros(a.read_mib << 20, np.uint8)
LAST = {}
NAMES = ("SetInput", "upload", "decode", "download", "checksums", "steps", "hand-out", "1-wave")


def run(label, make, check):
    st = make()
    crc, got = 0, 0
    t0 = time.perf_counter()
    while True:
        k = st.Read(buf, 0, buf.size)
        if k <= 0:
            break
        got += k
        if check:
            crc = zlib.crc32(buf[:k], crc)
    dt = time.perf_counter() - t0
    assert got == n, (label, got)
    if check:
        assert crc == want, label
    tm = (ctypes.c_double * 8)()
    L.szl_inflater_debug_times(st.inf._h, tm)
    pieces = L.szl_inflater_debug_bulk_calls(st.inf._h)
    key = id(st.inf)                                          # (a pooled Inflater accumulates: report the difference)
    t0s, p0 = LAST.get(key, ([0.0] * 8, 0)) if st.inf is pool else ([0.0] * 8, 0)
    LAST[key] = (list(tm), pieces)
    st.IsStreamOwner = False
    st.Dispose()
    return dt, pieces - p0, [x - y for x, y in zip(tm, t0s)]


from sharpziplib_amd.streams import InflaterInputBuffer       # noqa: E402


def ahead(mib):
    def make():
        st = GZipInputStream(io.BytesIO(gz), readAhead=mib << 20)
        return st
    return make


pool = Inflater(True)                                          # (InflaterPool: the reference rents and returns its Inflaters, S/Core/InflaterPool.cs:21-62)


def pooled():
    pool.Reset()
    return InflaterInputStream(io.BytesIO(raw), pool)


class KeepInf(InflaterInputStream):
    def Dispose(self):                                         # the pooled Inflater lives on
        inf, self.inf = self.inf, None
        super().Dispose()
        self.inf = inf
        inf.DetachInput()


for label, make in (
        ("GZipInputStream(stream)                       [default: 4096]", lambda: GZipInputStream(io.BytesIO(gz))),
        ("GZipInputStream(stream)  read-ahead 32 MiB", ahead(32)),
        ("GZipInputStream(stream)  read-ahead 64 MiB", ahead(64)),
        ("GZipInputStream(stream, 64 MiB)", lambda: GZipInputStream(io.BytesIO(gz), 64 << 20)),
        ("GZipInputStream(stream) host CRC-32           [deviceCrc=False]", lambda: GZipInputStream(io.BytesIO(gz), deviceCrc=False)),
        ("InflaterInputStream(stream, Inflater(true))   [default: 4096]", lambda: InflaterInputStream(io.BytesIO(raw), Inflater(True))),
        ("InflaterInputStream(stream, pooled Inflater)  [default: 4096]", lambda: KeepInf(io.BytesIO(raw), (pool.Reset(), pool)[1])),
        ("InflaterInputStream(stream, inf, 64 MiB)", lambda: InflaterInputStream(io.BytesIO(raw), Inflater(True), 64 << 20)),
        ("InflaterInputStream(stream, pooled, 64 MiB)", lambda: KeepInf(io.BytesIO(raw), (pool.Reset(), pool)[1], 64 << 20)),
        ("InflaterInputStream(stream, inf, 16 MiB) reference sizes, pageable", lambda: InflaterInputStream(io.BytesIO(raw), Inflater(True), 16 << 20, readAhead=0)),
):
    run(label, make, True)
    best = None
    for rep in range(3):
        r = run(label, make, False)
        if best is None or r[0] < best[0]:
            best = r
    dt, pieces, tm = best
    print("%-72s %8.1f MiB/s  %2d pieces | ms: total %6.1f  %s" % (label, a.mib / dt, pieces, dt * 1e3, "  ".join("%s %.1f" % (nm, v) for nm, v in zip(NAMES, tm))), flush=True)
if a.debug:
    L.szl_debug_set(b"SZL_DEBUG", 1)
    run("debug", lambda: GZipInputStream(io.BytesIO(gz)), False)
    L.szl_debug_set(b"SZL_DEBUG", 0)
