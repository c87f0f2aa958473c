"""Members of a few hundred KiB: one wavefront each vs the chunk-parallel form (SZL_INF_PAR_MIN_KIB sweep)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib()
eng = Engine()
for kind in ('enwik', 'logs'):
    for nm, msz in ((64, 1 << 20), (512, 1 << 20), (1024, 512 << 10), (256, 256 << 10), (1024, 256 << 10)):
        d = C.generate(kind, 0xE9, 0, nm * msz)
        parts = [d[i * msz:(i + 1) * msz] for i in range(nm)]
        comps = [r.data for r in eng.deflate(parts, level=6)]
        row = []
        for kib in (512, 256, 128, 64):
            L.szl_debug_set(b"SZL_INF_PAR_MIN_KIB", kib)
            for rep in range(2):
                out = eng.inflate(comps, [msz] * nm); km = eng.timing()['inflate_ms']
            ok = all(o[0].data == p.tobytes() for o, p in zip(out, parts))
            row.append(f"{kib}:{km:.1f}ms{'' if ok else '(BAD)'}")
        print(f"{kind} {nm} x {msz >> 10} KiB (comp {len(comps[0]) >> 10} KiB): " + "  ".join(row), flush=True)
