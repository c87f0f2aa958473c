"""A/B of library builds on one long member (python tools/gpu_inflate_ab_big.py libA.so libB.so ...)"""
import sys, subprocess, os
if not sys.argv[1].endswith(".run"):
    for so in sys.argv[1:]:
        subprocess.call([sys.executable, __file__, so + ".run"])
    sys.exit(0)
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import _lib
so = sys.argv[1][:-4]
_lib.SO = os.path.join(_lib.CSRC, so)
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
for kind in ('enwik', 'logs'):
    d = C.generate(kind, 0xE9, 0, 1024 << 20)
    comp = eng.deflate([d], level=6)[0].data
    for rep in range(3):
        (r, cons), = eng.inflate([comp], [d.size]); km = eng.timing()['inflate_ms']
    print(f"{so} {kind} 1 GiB member: {km:.1f} ms ok={r.data == d.tobytes()}", flush=True)
