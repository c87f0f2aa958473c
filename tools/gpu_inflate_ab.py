"""A/B of alternative builds of the library on the inflate workloads: python tools/gpu_inflate_ab.py libA.so [libB.so ...] (one process per library)."""
import sys, subprocess, os
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not sys.argv[1].endswith(".run")):
    for so in sys.argv[1:]:
        subprocess.call([sys.executable, __file__, so + ".run"])
    sys.exit(0)
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from sharpziplib_amd import _lib
so = sys.argv[1][:-4]
_lib.SO = os.path.join(_lib.CSRC, so)
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
print("==", so, flush=True)
for kind, mb in (('enwik', 512), ('logs', 512)):
    d = C.generate(kind, 0xE9, 0, mb << 20)
    comp = eng.deflate([d], level=6)[0].data
    for rep in range(2):
        (r, cons), = eng.inflate([comp], [d.size]); km = eng.timing()['inflate_ms']
    print(f"one member {kind} {mb} MiB: {km:.1f} ms -> {mb/(km/1e3)/1024:.2f} GiB/s ok={r.data == d.tobytes()}", flush=True)
for nm, msz in ((8192, 65536), (512, 1 << 20)):
    d = C.generate('enwik', 0xE9, 0, nm * msz)
    parts = [d[i * msz:(i + 1) * msz] for i in range(nm)]
    comps = [r.data for r in eng.deflate(parts, level=6)]
    kms = []
    for rep in range(4):
        out = eng.inflate(comps, [msz] * nm); kms.append(eng.timing()['inflate_ms'])
    ok = all(o[0].data == p.tobytes() for o, p in zip(out, parts)); km = min(kms[1:])
    print(f"{nm} x {msz>>10} KiB members: {km:.1f} ms -> {nm*msz/2**30/(km/1e3):.2f} GiB/s ok={ok}   (calls: {' '.join('%.1f' % k for k in kms)})", flush=True)
