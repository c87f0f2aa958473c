"""One laboratory knob off / on, alternating calls in ONE process (box-to-box differences cancel): stage times of one stream.
python tools/gpu_lab.py knob_ab SZL_SPEC_WB 0 1 [MiB=1024] [kind=enwik] [level=6]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from sharpziplib_amd import _lib, corpus
_lib._lib = _lib.lab_lib()
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
name = sys.argv[1].encode(); vals = [int(sys.argv[2]), int(sys.argv[3])]
mb = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
kind = sys.argv[5] if len(sys.argv) > 5 else "enwik"
lv = int(sys.argv[6]) if len(sys.argv) > 6 else 6
d = corpus.generate(kind, 0xE9, 0, mb << 20)
best = [dict(), dict()]; ref = None
for rep in range(6):
    for k in (0, 1):
        L.szl_debug_set(name, vals[k])
        r = eng.deflate([d], level=lv)[0]; tm = eng.timing()
        if ref is None: ref = r.data
        assert r.data == ref
        for n in ("total_ms", "links_ms", "match_ms", "parse_ms", "blocks_ms", "encode_ms"):
            best[k][n] = min(best[k].get(n, 1e9), tm[n])
L.szl_debug_set(name, -2147483648)
for k in (0, 1):
    print("%s=%d: total %.2f  A %.2f B %.2f C %.2f D %.2f E %.2f ms (best of 6, %d MiB %s level %d)" % (sys.argv[1], vals[k], best[k]["total_ms"], best[k]["links_ms"], best[k]["match_ms"], best[k]["parse_ms"], best[k]["blocks_ms"], best[k]["encode_ms"], mb, kind, lv))
