"""The stage C parity fix of round 6 at EVERY range length: inputs built so that every fifth range's path holds range_len + 2 tokens
(tests/test_gpu_deflate.py::_lazy_runs_at_range_ends), at the call sizes that select ranges of 512, 1024, 2048 and 4096 positions
(128 MiB - 1 GiB in one stream), against the oracle.   python tools/lab/range_overflow_sizes.py [max_mib=1024]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import hashlib
import numpy as np
import oracle_ffi as O
from test_gpu_deflate import _lazy_runs_at_range_ends
from sharpziplib_amd.batch import Engine
eng = Engine()
top = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for mib, rl in ((128, 512), (256, 1024), (512, 2048), (1024, 4096)):
    if mib > top: break
    d = _lazy_runs_at_range_ends(mib << 20, rl, 5, seed=rl)
    for lv, sg in ((7, 1),) if mib >= 512 else ((7, 1), (6, 0)):
        t0 = time.time(); want = O.deflate(d, lv, strategy=sg); t1 = time.time()
        got = eng.deflate([d], level=lv, strategy=sg)[0]; tm = eng.timing()
        print("%4d MiB, ranges of %4d, level %d strategy %d: %s (%d bytes; oracle %.0f s; device %.1f ms, stage C %.2f, unmerged ranges %d)"
              % (mib, rl, lv, sg, "EQUAL" if got.data == want else "DIFFERENT", len(want), t1 - t0, tm["total_ms"], tm["parse_ms"], tm["ranges_unmerged"]), flush=True)
