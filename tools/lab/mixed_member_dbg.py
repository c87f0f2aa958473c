"""One member of 4 MiB incompressible + 1 MiB text stretches through the chunk-parallel Inflater with the pass log (python tools/gpu_lab.py mixed_member_dbg)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
rnd = np.random.default_rng(1).integers(0, 256, 192 << 20, dtype=np.uint8)
text = C.generate("enwik", 5, 0, 8 << 20)
mixed = np.concatenate([np.concatenate([rnd[i << 22:(i + 1) << 22], text[:1 << 20]]) for i in range(48)])
comp = eng.deflate([mixed], level=6)[0].data
eng.inflate([comp], [mixed.size])
L.szl_debug_set(b"SZL_DEBUG", 1)
(r, c), = eng.inflate([comp], [mixed.size])
L.szl_debug_set(b"SZL_DEBUG", 0)
print("inflate_ms", eng.timing()["inflate_ms"], "ok", r.data == mixed.tobytes(), "compressed", len(comp))
