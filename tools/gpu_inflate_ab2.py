"""timing-only A/B of library builds on 8192 x 64 KiB inflate (knock-out builds produce wrong bytes on purpose)"""
import sys, subprocess, os
if not sys.argv[1].endswith(".run"):
    for so in sys.argv[1:]:
        subprocess.call([sys.executable, __file__, so + ".run"])
    sys.exit(0)
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import _lib
so = sys.argv[1][:-4]
ref = _lib.SO
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
nm, msz = (int(os.environ.get("NM", 8192)), int(os.environ.get("MSZ", 65536)))
d = C.generate('enwik', 0xE9, 0, nm * msz)
parts = [d[i * msz:(i + 1) * msz] for i in range(nm)]
_lib.SO = os.path.join(_lib.CSRC, so)
eng = Engine()
comps = [r.data for r in eng.deflate(parts, level=6)]
for rep in range(2):
    out = eng.inflate(comps, [msz] * nm); km = eng.timing()['inflate_ms']
print(f"{so}: {km:.1f} ms", flush=True)
