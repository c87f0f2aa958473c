"""A member whose payload is itself deflate data (a .tar.gz of zips / PNGs / docx): the outer encoder stores it, and the inner streams'
block headers are what the block finder finds in it (python tools/gpu_lab.py nested_member [MiB])"""
import sys, os, zlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 192
text = C.generate("enwik", 7, 0, 3 * (mib << 20))
inner = [r.data for r in eng.deflate([text[i:i + (8 << 20)] for i in range(0, text.size, 8 << 20)], level=6)]
payload = np.frombuffer(b"".join(bytes(x) for x in inner), np.uint8)[:mib << 20]
for name, data in (("device level 6", eng.deflate([payload], level=6)[0].data), ("zlib level 6", None)):
    if data is None:
        co = zlib.compressobj(6, zlib.DEFLATED, -15); data = co.compress(payload.tobytes()) + co.flush()
    eng.inflate([data], [payload.size])
    L.szl_debug_set(b"SZL_DEBUG", 1)
    (r, c), = eng.inflate([data], [payload.size])
    L.szl_debug_set(b"SZL_DEBUG", 0)
    print("%s: payload %d MiB of deflate data -> %d bytes; inflate %.2f ms ok=%s jobs %d" % (name, mib, len(data), eng.timing()["inflate_ms"], r.data == payload.tobytes(), int(L.szl_engine_debug_par_jobs(eng._h))), flush=True)
