"""Paths the headline does not take, timed next to it (python tools/gpu_lab.py other_paths): batches of mid-size streams, other levels, zlib
framing, incompressible data — device ms per GiB of input, to spot a path that is slower per byte than it has reason to be."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
text = C.generate("enwik", 0xE9, 0, 1 << 30)
logs = C.generate("logs", 0x106, 0, 512 << 20)
rnd = np.random.default_rng(1).integers(0, 256, 256 << 20, dtype=np.uint8)


def run(name, parts, level=6, nowrap=True, inflate=True):
    n = sum(p.size for p in parts)
    for rep in range(2):
        res = eng.deflate(parts, level=level, nowrap=nowrap)
        tm = eng.timing()
    line = "%-44s deflate %7.2f ms = %6.2f ms/GiB (A %.1f B %.1f C %.1f D %.1f E %.1f)" % (
        name, tm["total_ms"], tm["total_ms"] / (n / 2**30), tm["links_ms"], tm["match_ms"], tm["parse_ms"], tm["blocks_ms"], tm["encode_ms"])
    if inflate:
        comps = [r.data for r in res]
        for rep in range(2):
            out = eng.inflate(comps, [p.size for p in parts], nowrap=nowrap)
            im = eng.timing()["inflate_ms"]
        ok = all(o[0].data == p.tobytes() for o, p in zip(out[:4], parts[:4]))
        line += "   inflate %7.2f ms = %6.2f ms/GiB ok=%s" % (im, im / (n / 2**30), ok)
    print(line, flush=True)


def split(d, k):
    return [d[i:i + k] for i in range(0, d.size, k)]


run("1 x 1 GiB text, level 6 (the headline)", [text])
run("64 x 16 MiB text", split(text, 16 << 20))
run("256 x 4 MiB text", split(text, 4 << 20))
run("1024 x 1 MiB text", split(text, 1 << 20))
run("4096 x 256 KiB text", split(text, 256 << 10))
run("1 x 1 GiB text, level 5", [text], level=5)
run("1 x 1 GiB text, level 9", [text], level=9)
run("1 x 1 GiB text, level 6, zlib framing", [text], nowrap=False)
run("1 x 512 MiB logs, level 6", [logs])
run("1 x 256 MiB random bytes, level 6", [rnd])
run("256 x 1 MiB random bytes, level 6", split(rnd, 1 << 20))
run("1 x 256 MiB random bytes, level 0", [rnd], level=0)
run("64 x 4 MiB text, level 1", split(text[:256 << 20], 4 << 20), level=1, inflate=False)
