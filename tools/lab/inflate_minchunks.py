"""SZL_INF_MIN_CHUNKS (chunks a short member is cut into at least: 32) swept over calls of many members (python tools/gpu_lab.py inflate_minchunks)"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from sharpziplib_amd import _lib, corpus                # noqa: E402
from sharpziplib_amd.batch import Engine                # noqa: E402

FORGET = -2147483648
L = _lib.lib()
eng = Engine()
plain = corpus.generate("enwik", 0xE9, 0, 1 << 30)
shapes = [("2048 x 4 MiB", 22, 2048), ("1024 x 4 MiB", 22, 1024), ("256 x 4 MiB", 22, 256), ("128 x 4 MiB", 22, 128), ("512 x 1 MiB", 20, 512),
          ("2048 x 1 MiB", 20, 2048), ("64 x 16 MiB", 24, 64), ("4096 x 512 KiB", 19, 4096)]
sweep = [int(v) for v in sys.argv[1:]] or [32, 16, 8]
for name, sh, n in shapes:
    distinct = min(n, (1 << 30) >> sh)
    parts = [plain[i << sh:(i + 1) << sh] for i in range(distinct)]
    comps = [r.data for r in eng.deflate(parts, level=6)]
    comps = [comps[i % distinct] for i in range(n)]
    sizes = [1 << sh] * n
    line = []
    for mc in sweep + sweep:
        L.szl_debug_set(b"SZL_INF_MIN_CHUNKS", mc)
        best = 1e9
        for rep in range(2):
            out = eng.inflate(comps, sizes)
            best = min(best, eng.timing()["inflate_ms"])
        ok = all(out[i][0].status == 0 for i in range(n)) and all(out[i][0].data == parts[i % distinct].tobytes() for i in range(0, n, max(1, n // 64)))
        assert ok, (name, mc)
        line.append("%2d: %7.2f ms (%d jobs)" % (mc, best, L.szl_engine_debug_par_jobs(eng._h)))
        del out
    print("%-16s | %s" % (name, " | ".join(line)), flush=True)
L.szl_debug_set(b"SZL_INF_MIN_CHUNKS", FORGET)
eng.close()
