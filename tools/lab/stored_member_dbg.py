"""One member of incompressible data (stored blocks only) through the chunk-parallel Inflater with the pass log (python tools/gpu_lab.py stored_member_dbg)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
rnd = np.random.default_rng(1).integers(0, 256, 64 << 20, dtype=np.uint8)
comp = eng.deflate([rnd], level=6)[0].data
eng.inflate([comp], [rnd.size])
L.szl_debug_set(b"SZL_DEBUG", 1)
(r, c), = eng.inflate([comp], [rnd.size])
L.szl_debug_set(b"SZL_DEBUG", 0)
print("inflate_ms", eng.timing()["inflate_ms"], "ok", r.data == rnd.tobytes(), "compressed", len(comp), "jobs", int(L.szl_engine_debug_par_jobs(eng._h)))
