"""PCIe-inclusive rate of the host-buffer entry point (szl_deflate_batch_host) on one 1 GiB stream: wall clock around the call,
with and without the overlapped input copy; output checked against the frozen oracle hash."""
import sys, os, time, hashlib, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sharpziplib_amd import _lib, corpus
from sharpziplib_amd.batch import Engine
L = _lib.lib()
g = json.load(open(os.path.join(ROOT, "tests", "golden", "headline_golden.json")))["cases"]["cfg2_enwik_1g_l6"]
n = 1 << 30
data = corpus.generate("enwik", 0xE9, 0, n)
eng = Engine()
arr, in_total, out_total = Engine.layout([n])
hout = np.zeros(out_total + 8, np.uint8)
for overlap in (1, 0, 1, 0):
    L.szl_debug_set(b"SZL_H2D_OVERLAP", overlap); L.szl_debug_set(b"SZL_WINDOW_FROM_KIB", 0)
    t = time.perf_counter()
    _lib.check(L.szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, 1, 6, 0, _lib.F_NOWRAP | _lib.F_CRC32), "batch")
    dt = time.perf_counter() - t
    tm = eng.timing()
    ok = hashlib.sha256(hout[:arr[0].out_len].tobytes()).hexdigest() == g["out_sha256"] and int(arr[0].crc32) == g["crc32"]
    print("overlap=%d wall %.1f ms (%.2f GiB/s end to end) device total %.1f ms  workspace %.2f GiB  %s" % (
        overlap, dt * 1e3, 1.0 / dt, tm["total_ms"], L.szl_engine_debug_workspace(eng._h) / 2**30, "bit-exact" if ok else "*** DIFFERS ***"), flush=True)
