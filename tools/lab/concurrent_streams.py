"""Several streaming objects at once (round 5: each runs on a HIP stream of its own): N host threads, each a default-constructed
GZipInputStream over its own 256 MiB member — aggregate MiB/s of output for N = 1, 2, 4, 8.

    python tools/gpu_lab.py concurrent_streams [--mib 256]
"""
import argparse
import io
import os
import sys
import threading
import time
import zlib

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np                                          # noqa: E402

from sharpziplib_amd import corpus                            # noqa: E402
from sharpziplib_amd.gzipstream import GZipInputStream, write_members   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=int, default=256)
a = ap.parse_args()
n = a.mib << 20
plain = corpus.generate("enwik", 0xE9, 0, n)
(gz,) = write_members([plain], level=6)
want = zlib.crc32(plain)


def one(check, out, k):
    st = GZipInputStream(io.BytesIO(gz))
    buf = np.zeros(4 << 20, np.uint8)
    crc, got = 0, 0
    while True:
        r = st.Read(buf, 0, buf.size)
        if r <= 0:
            break
        got += r
        if check:
            crc = zlib.crc32(buf[:r], crc)
    st.Dispose()
    out[k] = (got, crc)


for threads in (1, 2, 4, 8):
    res = {}
    th = [threading.Thread(target=one, args=(True, res, k)) for k in range(threads)]
    [t.start() for t in th]; [t.join() for t in th]
    assert all(res[k] == (n, want) for k in range(threads)), res
    best = 1e9
    for rep in range(2):
        res = {}
        th = [threading.Thread(target=one, args=(False, res, k)) for k in range(threads)]
        t0 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        best = min(best, time.perf_counter() - t0)
    print("%d thread(s), each GZipInputStream(stream) over a %d MiB member: %8.1f MiB/s in all (%6.1f ms)" % (threads, a.mib, threads * a.mib / best, best * 1e3), flush=True)
