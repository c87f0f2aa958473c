"""One-off parity check above 2^31 bytes: GPU raw deflate of a 2.2 GiB enwik-style stream vs the oracle (≈80 s of CPU)."""
import sys, time, hashlib, zlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
n = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else int(2.2 * (1 << 30))
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kind = sys.argv[3] if len(sys.argv) > 3 else 'enwik'
data = C.generate(kind, 0xE9, 0, n)
arr, in_total, out_total = Engine.layout([n])
hout = np.zeros(out_total + 8, np.uint8)
_lib.check(L.szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, 1, level, 0, _lib.F_NOWRAP | _lib.F_CRC32), 'batch')
tm = eng.timing()
print(f"gpu {n/2**30:.2f} GiB L{level} {kind}: {tm['total_ms']:.0f} ms, out {arr[0].out_len}, unmerged={tm['ranges_unmerged']} fb={tm['fallback_walks']}", flush=True)
g = hashlib.sha256(hout[:arr[0].out_len].tobytes()).hexdigest()
t = time.time(); ref = O.deflate(data, level); dt = time.time() - t
print(f"oracle: {dt:.1f}s ({n/2**20/dt:.1f} MiB/s), out {len(ref)}", flush=True)
print('PARITY', g == hashlib.sha256(ref).hexdigest(), flush=True)
