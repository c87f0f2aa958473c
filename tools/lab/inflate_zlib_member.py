"""One member made by zlib (level 6, 16 MiB pieces closed by sync flushes — bench.py's 4z entry) through the chunk-parallel decoder, with
the passes' laps: python tools/lab/inflate_zlib_member.py [MiB=1024]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
sys.argv, argv = [sys.argv[0]], sys.argv
import bench
L = _lib.lib(); eng = Engine()
mb = int(argv[1]) if len(argv) > 1 else 1024
d = C.generate('enwik', 0xE9, 0, mb << 20)
comp = np.frombuffer(bench.zlib_member_parallel(d), dtype=np.uint8)
for rep in range(3):
    L.szl_debug_set(b"SZL_DEBUG", 1 if rep == 1 else 0)
    (r, cons), = eng.inflate([comp], [d.size]); km = eng.timing()['inflate_ms']
print(f"zlib member {mb} MiB ({comp.size} compressed): {km:.1f} ms -> {mb / 1024 / (km / 1e3):.2f} GiB/s ok={r.data == d.tobytes()} jobs {L.szl_engine_debug_par_jobs(eng._h)}", flush=True)
