"""2048 x 4 MiB members (512 distinct ones): one wavefront per member (more than 1024 streams in a call) against the chunk-parallel path
(SZL_INF_PAR_MAX_STREAMS = 4096, laboratory library), alternating in one process: python tools/gpu_lab.py inflate_many_paths [members=2048]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch
from sharpziplib_amd import _lib, corpus
_lib._lib = _lib.lab_lib()
from sharpziplib_amd.batch import Engine
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
msz = (int(sys.argv[2]) << 10) if len(sys.argv) > 2 else 4 << 20          # member size in KiB
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
eng = Engine()
parts = []
for k in range(4):
    plain = corpus.generate("enwik", 0x21B0 + k, 0, 512 << 20)
    parts += [plain[i * msz:(i + 1) * msz] for i in range(min(128, (512 << 20) // msz))]
comps = []
for a in range(0, len(parts), 128):
    comps += [r.data for r in eng.deflate(parts[a:a + 128], level=6)]
D = len(parts)
lens = sorted(len(c) for c in comps)
print("distinct members %d: compressed bytes min %d median %d max %d" % (D, lens[0], lens[D // 2], lens[-1]), flush=True)
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
best = {}
for rep in range(3):
    for v in (1024, hi):
        L.szl_debug_set(b"SZL_INF_PAR_MAX_STREAMS", v)
        eng.inflate_device(d_in.data_ptr(), d_out.data_ptr(), arr, flags=_lib.F_NOWRAP)
        best[v] = min(best.get(v, 1e9), eng.timing()["inflate_ms"])
        assert all(s.status == 0 and s.out_len == msz for s in arr)
        jobs = L.szl_engine_debug_par_jobs(eng._h)
        if rep == 0: print("  SZL_INF_PAR_MAX_STREAMS=%d: %.1f ms, %d chunk jobs" % (v, eng.timing()["inflate_ms"], jobs), flush=True)
L.szl_debug_set(b"SZL_INF_PAR_MAX_STREAMS", -2147483648)
ref = torch.from_numpy(np.concatenate([parts[i % D] for i in range(0, 64)])).cuda()
assert bool(torch.equal(d_out[:64 * msz], ref))
for v in (1024, hi):
    print("%d x %d KiB members, SZL_INF_PAR_MAX_STREAMS=%d (%s): %.1f ms = %.1f GiB/s" % (n, msz >> 10, v, "a wavefront per member" if n > v else "chunk-parallel", best[v], n * msz / 2**30 / (best[v] / 1e3)))
