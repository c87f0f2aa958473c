"""N members of M MiB through the Inflater's batch call with the passes' laps (SZL_DEBUG): python tools/lab/inflate_members_dbg.py [N=512] [MiB=4]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 512
msz = (int(sys.argv[2]) if len(sys.argv) > 2 else 4) << 20
d = C.generate('enwik', 0xE9, 0, nm * msz)
parts = [d[i * msz:(i + 1) * msz] for i in range(nm)]
comps = [r.data for r in eng.deflate(parts, level=6)]
for rep in range(3):
    L.szl_debug_set(b"SZL_DEBUG", 1 if rep == 1 else 0)
    out = eng.inflate(comps, [msz] * nm); km = eng.timing()['inflate_ms']
ok = all(o[0].data == p.tobytes() for o, p in zip(out, parts))
print(f"{nm} x {msz >> 20} MiB members: {km:.1f} ms -> {nm * msz / 2**30 / (km / 1e3):.2f} GiB/s ok={ok}", flush=True)
