"""One long member through the chunk-parallel decoder with the pass times (SZL_DEBUG laps): python tools/gpu_inflate_big.py MiB [kind]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib()
eng = Engine()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind = sys.argv[2] if len(sys.argv) > 2 else 'enwik'
d = C.generate(kind, 0xE9, 0, mb << 20)
comp = eng.deflate([d], level=6)[0].data
(r, cons), = eng.inflate([comp], [d.size])
L.szl_debug_set(b"SZL_DEBUG", 1)
(r, cons), = eng.inflate([comp], [d.size]); km = eng.timing()['inflate_ms']
L.szl_debug_set(b"SZL_DEBUG", 0)
(r, cons), = eng.inflate([comp], [d.size]); km = eng.timing()['inflate_ms']
print(f"{kind} {mb} MiB: {km:.1f} ms -> {mb/(km/1e3)/1024:.2f} GiB/s ok={r.data == d.tobytes()}", flush=True)
