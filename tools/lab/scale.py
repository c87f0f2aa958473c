import sys, ctypes, time, zlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C, _lib
L = _lib.lib()
eng = L.szl_engine_create()
assert eng, L.szl_last_error()
def run(name, data, level=6, check_oracle=True, reps=2):
    n = data.size
    cap = (int(L.szl_deflate_bound(n)) + 19) & ~3
    out = np.zeros(cap, dtype=np.uint8)
    st = _lib.Stream(0, n, 0, cap, 0, 0, 0, 0, 0, 0)
    for r in range(reps):
        t=time.time()
        rc = L.szl_deflate_batch_host(eng, data.ctypes.data, out.ctypes.data, ctypes.byref(st), 1, level, 0, _lib.F_NOWRAP|_lib.F_CRC32)
        dt=time.time()-t
        if rc != 0:
            print(name, 'FAILED rc', rc, L.szl_last_error()); return
        tm = _lib.Timing(); L.szl_engine_last_timing(eng, ctypes.byref(tm))
        print(f"{name:10s} L{level} n={n>>20}MiB out={st.out_len} ratio={st.out_len/n:.4f} wall={dt*1e3:.0f}ms gpu={tm.total_ms:.1f}ms -> {n/2**20/(tm.total_ms/1e3):.0f} MiB/s [ck {tm.checksum_ms:.1f} A {tm.links_ms:.1f} B {tm.match_ms:.1f} C {tm.parse_ms:.1f} D {tm.blocks_ms:.1f} E {tm.encode_ms:.1f}] tok={tm.tokens} blk={tm.blocks} unmerged={tm.ranges_unmerged} fb={tm.fallback_walks}", flush=True)
    comp = out[:st.out_len].tobytes()
    t=time.time(); back = zlib.decompress(comp, -15); dz=time.time()-t
    print('   zlib roundtrip', back == data.tobytes(), f'({dz:.1f}s)', 'crc', st.crc32 == zlib.crc32(data.tobytes()), flush=True)
    if check_oracle:
        t=time.time(); ref = O.deflate(data, level); do=time.time()-t
        print(f'   oracle eq {ref == comp} (oracle {do:.1f}s = {n/2**20/do:.1f} MiB/s)', flush=True)
sizes = [int(a) for a in sys.argv[1:]] or [64]
for mb in sizes:
    t=time.time(); d = C.generate('enwik', 0xE9, 0, mb<<20); print(f'gen {mb} MiB in {time.time()-t:.1f}s', flush=True)
    run('enwik', d, 6, check_oracle=(mb <= 256))
