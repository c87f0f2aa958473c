"""One stream over several engines (GPU box): which parts ran, how many were re-run, timing against one engine.
Usage: python tools/gpu_multi_stream.py [MiB=512] [engines=2] [level=6]   (SZL_DEBUG=1 prints the parts)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine, deflate_multi
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 2
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ndev = int(_lib.lib().szl_device_count())
devices = [i % ndev for i in range(ne)]
for kind in ("enwik", "logs", "zeros"):
    data = np.zeros(mib << 20, np.uint8) if kind == "zeros" else C.generate(kind, 0xE9, 0, mib << 20)
    eng = Engine()
    eng.deflate([data[:1 << 20]], level=level)
    t = time.perf_counter(); (one,) = eng.deflate([data], level=level, crc32=True); t1 = time.perf_counter() - t
    eng.close()
    deflate_multi([data[:(70 << 20) * ne]], devices, level=level)          # warm the engines of the slots
    t = time.perf_counter(); (r,) = deflate_multi([data], devices, level=level, crc32=True); t2 = time.perf_counter() - t
    print("%s %d MiB level %d: one engine %.0f ms wall, %d engines on devices %s %.0f ms wall; identical %s" % (
        kind, mib, level, t1 * 1e3, ne, devices, t2 * 1e3, hashlib.sha256(one.data).digest() == hashlib.sha256(r.data).digest() and one.crc32 == r.crc32), flush=True)
