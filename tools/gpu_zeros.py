import sys, ctypes, time, zlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
eng = Engine()
for name, data in (('zeros', C.zeros(64 << 20)), ('p10', C.period10(64 << 20)), ('acgt', C.four_symbol(64 << 20))):
    for rep in range(2):
        t = time.time(); r = eng.deflate([data], level=6)[0]; dt = time.time() - t
    tm = eng.timing()
    t = time.time(); ref = O.deflate(data, 6); do = time.time() - t
    print(f"{name}: 64MiB gpu={tm['total_ms']:.1f}ms (A {tm['links_ms']:.1f} B {tm['match_ms']:.1f} C {tm['parse_ms']:.1f} D {tm['blocks_ms']:.1f}) unmerged={tm['ranges_unmerged']} eq={r.data == ref} oracle={do*1e3:.0f}ms", flush=True)
