"""Stage C of ONE small stream per call by data class: the ranges of 64 positions a call of up to 1 MiB gets (round 6) against 128 / 256
(laboratory library, SZL_RANGE_LEN): python tools/gpu_lab.py small_call_ranges"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from sharpziplib_amd import _lib, corpus
_lib._lib = _lib.lab_lib()
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
for kind in ("enwik", "logs", "dickens"):
    for kib in (16, 64, 256, 1024):
        d = corpus.generate(kind, 7, 0, kib << 10)
        line = []
        for rl in (0, 128, 256):
            L.szl_debug_set(b"SZL_RANGE_LEN", rl if rl else -2147483648)
            best = None
            for rep in range(8):
                eng.deflate([d], level=6); tm = eng.timing()
                if best is None or tm["total_ms"] < best["total_ms"]: best = tm
            line.append("%s: C %.3f total %.3f (unmerged %d)" % ("default" if not rl else "ranges of %d" % rl, best["parse_ms"], best["total_ms"], best["ranges_unmerged"]))
        print("%-8s %5d KiB | %s" % (kind, kib, " | ".join(line)), flush=True)
L.szl_debug_set(b"SZL_RANGE_LEN", -2147483648)
