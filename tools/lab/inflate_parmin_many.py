"""Calls of 513-2048 members of 128-512 KiB of compressed bytes: one wavefront per member (SZL_INF_PAR_MIN_KIB = 512, the rule for more than
512 streams) against the chunked form (128), alternating in one process: python tools/gpu_lab.py inflate_parmin_many"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch
from sharpziplib_amd import _lib, corpus
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
plain = corpus.generate("enwik", 0x21B0, 0, 1 << 30)
for n, kib in ((600, 1024), (1024, 1024), (1536, 1024), (2048, 1024), (700, 512), (1400, 512), (3000, 512), (3000, 1024)):
    msz = kib << 10
    D = min(n, (1 << 30) // msz, 512)
    parts = [plain[i * msz:(i + 1) * msz] for i in range(D)]
    comps = [r.data for r in eng.deflate(parts, level=6)]
    arr = (_lib.Stream * n)()
    io = oo = 0
    for i in range(n):
        b = comps[i % D]
        arr[i].in_off, arr[i].in_len, arr[i].out_off, arr[i].out_cap = io, len(b), oo, msz
        io += (len(b) + 3) & ~3; oo += msz
    hin = np.zeros(io + 8, np.uint8)
    for i, s in enumerate(arr):
        hin[s.in_off:s.in_off + s.in_len] = np.frombuffer(comps[i % D], np.uint8)
    d_in = torch.from_numpy(hin).cuda(); d_out = torch.empty(oo + 8, dtype=torch.uint8, device="cuda")
    best = {}; jobs = {}
    for rep in range(3):
        for v in (512, 128):
            L.szl_debug_set(b"SZL_INF_PAR_MIN_KIB", v)
            eng.inflate_device(d_in.data_ptr(), d_out.data_ptr(), arr, flags=_lib.F_NOWRAP)
            best[v] = min(best.get(v, 1e9), eng.timing()["inflate_ms"]); jobs[v] = L.szl_engine_debug_par_jobs(eng._h)
            assert all(s.status == 0 and s.out_len == msz for s in arr)
    L.szl_debug_set(b"SZL_INF_PAR_MIN_KIB", -2147483648)
    eng.inflate_device(d_in.data_ptr(), d_out.data_ptr(), arr, flags=_lib.F_NOWRAP)
    print("%5d x %4d KiB members (%d KiB compressed): min 512 KiB %7.2f ms (%d jobs) | min 128 KiB %7.2f ms (%d jobs) | default %7.2f ms (%d jobs)"
          % (n, kib, len(comps[0]) >> 10, best[512], jobs[512], best[128], jobs[128], eng.timing()["inflate_ms"], L.szl_engine_debug_par_jobs(eng._h)), flush=True)
    del d_in, d_out
