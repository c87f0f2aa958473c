"""Level-6 deflate of N streams of S KiB in one device-resident call: stage times by shape (looking for cliffs in the engine's sizing rules):
python tools/gpu_lab.py deflate_shapes_scan"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch
from sharpziplib_amd import _lib, corpus
from sharpziplib_amd.batch import Engine
eng = Engine()
plain = corpus.generate("enwik", 0x21B0, 0, 1 << 30)
d_in = torch.from_numpy(plain).cuda()
shapes = [(1, 1 << 20), (1, 256 << 10), (2, 256 << 10), (4, 64 << 10), (1, 64 << 10), (1, 16 << 10), (4, 16 << 10), (16, 16 << 10), (64, 16 << 10), (16, 4 << 10), (64, 4 << 10), (256, 4 << 10), (1024, 1 << 10),
          (4096, 256), (16384, 64), (65536, 16), (3, 100 << 10), (7, 37 << 10), (100, 10 << 10), (1000, 1 << 10), (300, 3 << 10), (1, 8 << 10), (2, 8 << 10), (1, 2 << 10), (2, 2 << 10), (8, 2 << 10), (1, 512), (2, 512), (8, 512), (32, 512)]
for n, kib in shapes:
    size = kib << 10
    if n * size > (1 << 30): continue
    arr, in_total, out_total = Engine.layout([size] * n)
    d_out = torch.empty(out_total + 8, dtype=torch.uint8, device="cuda")
    best = None
    for rep in range(3):
        eng.deflate_device(d_in.data_ptr(), d_out.data_ptr(), arr, level=6, flags=_lib.F_NOWRAP)
        tm = eng.timing()
        if best is None or tm["total_ms"] < best["total_ms"]: best = tm
    assert all(s.status == 0 for s in arr)
    gib = n * size / 2**30
    print("%6d x %7d KiB = %7.1f MiB: %8.2f ms = %5.1f GiB/s   A %.2f B %.2f C %.2f D %.2f E %.2f" % (n, kib, gib * 1024, best["total_ms"], gib / (best["total_ms"] / 1e3),
          best["links_ms"], best["match_ms"], best["parse_ms"], best["blocks_ms"], best["encode_ms"]), flush=True)
    del d_out
