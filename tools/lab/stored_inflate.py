"""Inflate of incompressible data — members of stored blocks (python tools/gpu_lab.py stored_inflate)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
rnd = np.random.default_rng(1).integers(0, 256, 256 << 20, dtype=np.uint8)
text = C.generate("enwik", 5, 0, 8 << 20)
mixed = np.concatenate([np.concatenate([rnd[i << 22:(i + 1) << 22], text[:1 << 20]]) for i in range(48)])     # 4 MiB random + 1 MiB text, 48 times


def run(name, parts, level=6):
    n = sum(p.size for p in parts)
    comps = [r.data for r in eng.deflate(parts, level=level)]
    for rep in range(2):
        out = eng.inflate(comps, [p.size for p in parts])
        im = eng.timing()["inflate_ms"]
    ok = all(o[0].data == p.tobytes() for o, p in zip(out, parts))
    print("%-52s inflate %8.2f ms = %8.2f ms/GiB ok=%s" % (name, im, im / (n / 2**30), ok), flush=True)


run("1 x 256 MiB random bytes (level 6: stored blocks)", [rnd])
run("1 x 256 MiB random bytes, level 0", [rnd], 0)
run("256 x 1 MiB random bytes", [rnd[i << 20:(i + 1) << 20] for i in range(256)])
run("4096 x 64 KiB random bytes", [rnd[i << 16:(i + 1) << 16] for i in range(4096)])
run("1 x 240 MiB: 4 MiB random + 1 MiB text, 48 times", [mixed])
run("48 x 5 MiB of the same mix", [mixed[i * (5 << 20):(i + 1) * (5 << 20)] for i in range(48)])
