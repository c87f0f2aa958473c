"""One inflate call for counter collection: N x S KiB members (python tools/gpu_inflate_one.py N S_KiB [kind])."""
import sys
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
nm, msz = int(sys.argv[1]), int(sys.argv[2]) << 10
kind = sys.argv[3] if len(sys.argv) > 3 else 'enwik'
eng = Engine()
d = C.generate(kind, 0xE9, 0, nm * msz)
parts = [d[i * msz:(i + 1) * msz] for i in range(nm)]
comps = [r.data for r in eng.deflate(parts, level=6)]
out = eng.inflate(comps, [msz] * nm)
print("inflate_ms", eng.timing()['inflate_ms'], all(o[0].data == p.tobytes() for o, p in zip(out, parts)))
