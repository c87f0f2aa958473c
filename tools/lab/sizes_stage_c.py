"""Stage times of ONE stream by size (python tools/gpu_lab.py sizes_stage_c): what the range-length rule of stage C does between 16 MiB and 1 GiB"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import zlib
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
big = C.generate("enwik", 0xE9, 0, 1 << 30)
for mib in (4, 16, 64, 128, 256, 512, 1024):
    d = big[:mib << 20]
    for rep in range(2):
        r = eng.deflate([d], level=6)[0]
        tm = eng.timing()
    print("%5d MiB: total %7.2f ms  A %5.2f  B %6.2f  C %5.2f  D %5.2f  E %5.2f  unmerged %d  ok %s" % (
        mib, tm["total_ms"], tm["links_ms"], tm["match_ms"], tm["parse_ms"], tm["blocks_ms"], tm["encode_ms"], tm["ranges_unmerged"],
        zlib.decompress(bytes(r.data), -15) == d.tobytes() if mib <= 64 else "-"), flush=True)
