"""DeflateFast (levels 1-4) timings: one big stream and many small ones."""
import sys, ctypes, time, zlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib()
eng = Engine()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
d = C.generate('enwik', 0xE9, 0, mb << 20)
for lv in (1, 3):
    t = time.time(); r = eng.deflate([d], level=lv)[0]; dt = time.time() - t
    tm = eng.timing()
    t = time.time(); ref = O.deflate(d, lv); do = time.time() - t
    print(f"single L{lv} {mb}MiB gpu={tm['total_ms']:.0f}ms ({mb/(tm['total_ms']/1e3):.1f} MiB/s) parse={tm['match_ms']:.0f}ms ratio={len(r.data)/d.size:.4f} eq={r.data==ref} oracle={mb/do:.1f} MiB/s", flush=True)
data = C.generate('dickens', 0x21B0, 0, N * 65536)
arr, in_total, out_total = Engine.layout([65536] * N)
hout = np.zeros(out_total + 8, np.uint8)
for lv in (1, 4):
    for rep in range(2):
        _lib.check(L.szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, N, lv, 0, _lib.F_NOWRAP | _lib.F_CRC32), 'batch')
        tm = eng.timing()
    print(f"batch L{lv}: {N} x 64KiB gpu={tm['total_ms']:.1f}ms -> {N*65536/2**20/(tm['total_ms']/1e3):.0f} MiB/s [A {tm['links_ms']:.1f} fast {tm['match_ms']:.1f} D {tm['blocks_ms']:.1f} E {tm['encode_ms']:.1f}] ratio={tm['out_bytes']/tm['in_bytes']:.4f}", flush=True)
    for i in (0, N // 2, N - 1):
        s = arr[i]
        assert hout[s.out_off:s.out_off + s.out_len].tobytes() == O.deflate(data[s.in_off:s.in_off + s.in_len], lv), i
print('spot checks ok')
