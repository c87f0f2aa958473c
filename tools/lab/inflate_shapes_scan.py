"""N level-6 members of S KiB of text inflated in one device-resident call, by shape (looking for cliffs in the path rules):
python tools/gpu_lab.py inflate_shapes_scan"""
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
shapes = [(1, 1 << 20), (1, 256 << 10), (1, 64 << 10), (1, 16 << 10), (1, 4 << 10), (1, 1 << 10), (1, 256), (1, 64), (2, 256 << 10), (4, 64 << 10), (8, 32 << 10), (16, 16 << 10), (32, 8 << 10), (64, 4 << 10), (128, 2 << 10),
          (256, 1 << 10), (512, 512), (1024, 256), (2048, 128), (4096, 64), (16384, 64), (3, 100 << 10), (7, 37 << 10), (20, 20 << 10), (100, 2 << 10), (300, 700), (600, 300), (1000, 200), (1600, 300), (2500, 200), (5000, 100), (40, 1 << 10), (10, 4 << 10), (4, 1 << 10), (2, 2 << 10), (16, 256), (4, 256)]
for n, kib in shapes:
    msz = kib << 10
    if n * msz > (1 << 30): continue
    D = min(n, 256)
    parts = [plain[i * msz:(i + 1) * msz] for i in range(D)]
    comps = [r.data for r in eng.deflate(parts, level=6)]
    arr = (_lib.Stream * n)()
    io = oo = 0
    for i in range(n):
        b = comps[i % D]
        arr[i].in_off, arr[i].in_len, arr[i].out_off, arr[i].out_cap = io, len(b), oo, msz
        io += (len(b) + 3) & ~3; oo += (msz + 3) & ~3
    hin = np.zeros(io + 8, np.uint8)
    for i, s in enumerate(arr):
        hin[s.in_off:s.in_off + s.in_len] = np.frombuffer(comps[i % D], np.uint8)
    d_in = torch.from_numpy(hin).cuda(); d_out = torch.empty(oo + 8, dtype=torch.uint8, device="cuda")
    best = 1e9
    for rep in range(3):
        eng.inflate_device(d_in.data_ptr(), d_out.data_ptr(), arr, flags=_lib.F_NOWRAP)
        best = min(best, eng.timing()["inflate_ms"])
    assert all(s.status == 0 and s.out_len == msz for s in arr)
    print("%6d x %7d KiB = %7.1f MiB (members of %5d KiB compressed): %8.2f ms = %5.1f GiB/s, %d chunk jobs" % (n, kib, n * msz / 2**20, len(comps[0]) >> 10, best, n * msz / 2**30 / (best / 1e3), L.szl_engine_debug_par_jobs(eng._h)), flush=True)
    del d_in, d_out
