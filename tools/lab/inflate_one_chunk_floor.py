"""ONE member of 4-128 MiB: the chunk size the rule picks (32 KiB at least) against 16 / 24 KiB (SZL_INF_CHUNK_KIB): python tools/gpu_lab.py inflate_one_chunk_floor"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from sharpziplib_amd import _lib, corpus
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
FORGET = -2147483648
for kind in ("enwik", "logs"):
    plain = corpus.generate(kind, 0x21B0, 0, 128 << 20)
    for mib in (4, 16, 32, 64, 128):
        d = plain[:mib << 20]
        comp = eng.deflate([d], level=6)[0].data
        line = []
        for ck in (0, 24, 16):
            L.szl_debug_set(b"SZL_INF_CHUNK_KIB", ck if ck else FORGET)
            if ck: L.szl_debug_set(b"SZL_INF_MIN_CHUNKS", 8)
            best = 1e9
            for rep in range(4):
                out = eng.inflate([comp], [d.size]); best = min(best, eng.timing()["inflate_ms"])
            assert out[0][0].status == 0 and out[0][0].data == d.tobytes()
            line.append("%s: %6.2f ms (%4d jobs)" % ("rule" if not ck else "%d KiB" % ck, best, L.szl_engine_debug_par_jobs(eng._h)))
        L.szl_debug_set(b"SZL_INF_CHUNK_KIB", FORGET); L.szl_debug_set(b"SZL_INF_MIN_CHUNKS", FORGET)
        print("%-6s %4d MiB (%6d KiB compressed) | %s" % (kind, mib, len(comp) >> 10, " | ".join(line)), flush=True)
