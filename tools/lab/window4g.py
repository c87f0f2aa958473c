"""Config 5 at full size (GPU box): one 4 GiB logs stream, level 9, window pipeline against the monolithic run.
Prints the workspace peak of each and checks that the two outputs are the same bytes (sha256) and inflate to the input's CRC.
Usage: python tools/gpu_window4g.py [GiB=4] [level=9]"""
import hashlib, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
n = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else (4 << 30)
level = int(sys.argv[2]) if len(sys.argv) > 2 else 9
t = time.time(); data = C.generate("logs", 0x106, 0, n); print("gen %.2f GiB in %.1fs" % (n / 2**30, time.time() - t), flush=True)
crc = zlib.crc32(data)
shas = {}
for name, knobs in (("windowed (default: 256 MiB windows)", {"SZL_WINDOW_FROM_KIB": 0}), ("monolithic", {"SZL_WINDOW_KIB": 64 << 20})):
    for k, v in knobs.items():
        L.szl_debug_set(k.encode(), v)
    arr, in_total, out_total = Engine.layout([n])
    hout = np.zeros(out_total + 8, np.uint8)
    t = time.time()
    _lib.check(L.szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, 1, level, 0, _lib.F_NOWRAP | _lib.F_CRC32), "batch")
    dt = time.time() - t
    tm = eng.timing()
    ws = int(L.szl_engine_debug_workspace(eng._h))
    sha = hashlib.sha256(hout[:arr[0].out_len].tobytes()).hexdigest()
    shas[name] = sha
    print("%-38s wall %.2fs gpu %.0f ms (%.0f MiB/s) workspace %.2f GiB = %.2f B per input byte, out %d B, crc ok %s, sha %s" % (
        name, dt, tm["total_ms"], n / 2**20 / (tm["total_ms"] / 1e3), ws / 2**30, ws / n, arr[0].out_len, int(arr[0].crc32) == crc, sha[:16]), flush=True)
    for k in knobs:
        L.szl_debug_set(k.encode(), -2147483648)
print("windowed == monolithic:", len(set(shas.values())) == 1)
(r, consumed), = eng.inflate([hout[:arr[0].out_len]], [n], crc32=True)
print("device inflate of the 4 GiB member: status %d, crc ok %s, %d chunk jobs, %.1f ms" % (
    r.status, r.crc32 == crc, int(L.szl_engine_debug_par_jobs(eng._h)), eng.timing()["inflate_ms"]), flush=True)
