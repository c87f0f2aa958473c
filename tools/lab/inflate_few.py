"""A few members of about 1 MiB through inflate with the pass times (python tools/gpu_inflate_few.py N KiB)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
nm, msz = int(sys.argv[1]), int(sys.argv[2]) << 10
d = C.generate('enwik', 0xE9, 0, nm * msz)
parts = [d[i * msz:(i + 1) * msz] for i in range(nm)]
comps = [r.data for r in eng.deflate(parts, level=6)]
out = eng.inflate(comps, [msz] * nm)
L.szl_debug_set(b"SZL_DEBUG", 1)
out = eng.inflate(comps, [msz] * nm)
L.szl_debug_set(b"SZL_DEBUG", 0)
out = eng.inflate(comps, [msz] * nm); km = eng.timing()['inflate_ms']
print(f"{nm} x {msz >> 10} KiB: {km:.1f} ms -> {nm * msz / 2**30 / (km / 1e3):.2f} GiB/s ok={all(o[0].data == p.tobytes() for o, p in zip(out, parts))}")
