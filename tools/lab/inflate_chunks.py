"""One long member through the chunk-parallel decoder at several chunk sizes (SZL_INF_CHUNK_KIB): kernel time, SZL_DEBUG stage times."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib()
eng = Engine()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for kind in ('enwik', 'logs'):
    d = C.generate(kind, 0xE9, 0, mb << 20)
    comp = eng.deflate([d], level=6)[0].data
    for kib in (64, 96, 128, 160, 192, 256, 384):
        L.szl_debug_set(b"SZL_INF_CHUNK_KIB", kib)
        for rep in range(2):
            (r, cons), = eng.inflate([comp], [d.size])
            km = eng.timing()['inflate_ms']
        print(f"{kind} {mb} MiB chunk {kib:3d} KiB: {km:.1f} ms -> {mb/(km/1e3)/1024:.2f} GiB/s ok={r.data == d.tobytes()}", flush=True)
