"""256 x 4 MiB members: the symbol pass against the chunk size (SZL_INF_CHUNK_KIB) with eight chunks a member at least; pass times of two of them"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from sharpziplib_amd import _lib, corpus
from sharpziplib_amd.batch import Engine
FORGET = -2147483648
L = _lib.lib(); eng = Engine()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
plain = corpus.generate("enwik", 0xE9, 0, 1 << 30)
parts = [plain[(i % 256) << 22:((i % 256) + 1) << 22] for i in range(n)]
comps = [r.data for r in eng.deflate(parts[:256], level=6)]
comps = [comps[i % 256] for i in range(n)]
sizes = [1 << 22] * n
L.szl_debug_set(b"SZL_INF_MIN_CHUNKS", 8)
for kib in (160, 164, 168, 172):
    L.szl_debug_set(b"SZL_INF_CHUNK_KIB", kib)
    best = 1e9
    for rep in range(3):
        out = eng.inflate(comps, sizes); best = min(best, eng.timing()["inflate_ms"])
    print("chunk %3d KiB: %7.2f ms, %d jobs" % (kib, best, L.szl_engine_debug_par_jobs(eng._h)), flush=True)
for kib in (152, 168):
    L.szl_debug_set(b"SZL_INF_CHUNK_KIB", kib)
    L.szl_debug_set(b"SZL_DEBUG", 1)
    print("---- %d KiB" % kib, flush=True); sys.stderr.flush()
    out = eng.inflate(comps, sizes)
    L.szl_debug_set(b"SZL_DEBUG", 0)
L.szl_debug_set(b"SZL_INF_CHUNK_KIB", FORGET); L.szl_debug_set(b"SZL_INF_MIN_CHUNKS", FORGET)
eng.close()
