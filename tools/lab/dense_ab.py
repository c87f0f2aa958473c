"""SZL_INF_DENSE + SZL_INF_SLOTS_PER_CU = 10 (the symbol pass at 3 wavefronts per SIMD) against the default on every shape that decides it.

    python tools/gpu_lab.py dense_ab [--quick]

Round 4 measured one 256 MiB member 21.5 -> 18.3 ms and 64 x 4 MiB members 19.5 -> 58.7 ms (jobs overran regions sized per chunk); with
regions sized by span (round 5's default) the second number has to be taken again — and 2048 x 4 MiB, 8192 x 64 KiB, 512 x 1 MiB with it.
Every output is compared with the input."""
import argparse
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np                                    # noqa: E402

from sharpziplib_amd import _lib, corpus                # noqa: E402
from sharpziplib_amd.batch import Engine                # noqa: E402

FORGET = -2147483648
ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
a = ap.parse_args()
_lib._lib = _lib.lab_lib()          # (round 5: the dense build lives in the laboratory library only)
L = _lib.lib()
eng = Engine()
plain = corpus.generate("enwik", 0xE9, 0, 1 << 30)
shapes = [("one 1 GiB member", [plain]), ("64 x 4 MiB", [plain[i << 22:(i + 1) << 22] for i in range(64)]),
          ("64 x 1 MiB", [plain[i << 20:(i + 1) << 20] for i in range(64)]), ("512 x 1 MiB", [plain[i << 20:(i + 1) << 20] for i in range(512)]),
          ("256 x 4 MiB", [plain[i << 22:(i + 1) << 22] for i in range(256)]), ("8192 x 64 KiB", [plain[i << 16:(i + 1) << 16] for i in range(8192)])]
if a.quick:
    shapes = shapes[:3]
for name, parts in shapes:
    comps = [r.data for r in eng.deflate(parts, level=6)]
    sizes = [p.size for p in parts]
    line = []
    for slots, dense in ((8, 0), (10, 1), (8, 0), (10, 1)):
        L.szl_debug_set(b"SZL_INF_SLOTS_PER_CU", slots); L.szl_debug_set(b"SZL_INF_DENSE", dense)
        best = 1e9
        for rep in range(3):
            out = eng.inflate(comps, sizes)
            best = min(best, eng.timing()["inflate_ms"])
        assert all(o[0].status == 0 and o[0].data == p.tobytes() for o, p in zip(out, parts)), (name, slots, dense)
        line.append("%s %8.2f ms" % ("dense+10" if dense else "default ", best))
    print("%-18s | %s | %d chunk jobs" % (name, " | ".join(line), L.szl_engine_debug_par_jobs(eng._h)), flush=True)
L.szl_debug_set(b"SZL_INF_SLOTS_PER_CU", FORGET); L.szl_debug_set(b"SZL_INF_DENSE", FORGET)
eng.close()
