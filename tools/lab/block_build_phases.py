"""Where ONE block's k_block_build spends its time (laboratory library: 100 MHz stamps of workgroup 0): python tools/gpu_lab.py block_build_phases [KiB]"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from sharpziplib_amd import _lib, corpus
_lib._lib = _lib.lab_lib()
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
kib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = corpus.generate('enwik', 7, 0, kib << 10)
for _ in range(5):
    eng.deflate([d], level=6)
st = (ctypes.c_ulonglong * 16)()
L.szl_lab_bb_stamps.argtypes = [ctypes.c_void_p]
assert L.szl_lab_bb_stamps(st) == 0
t = [int(x) for x in st]
us = lambda a, b: (t[b] - t[a]) / 100.0
print("k_block_build, block 0 of a %d KiB call (us): zero + histogram %.1f | lit / dist trees %.1f (lit: leaves %.1f, merges %.1f, BuildLength %.1f) | runs counted %.1f | bl tree %.1f | "
      "sums + decision + bl codes %.1f | header bits %.1f | code tables + copy %.1f | all %.1f"
      % (kib, us(0, 1), us(1, 2), us(1, 8), us(8, 9), us(9, 10), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(0, 7)))
print("D stage of the call: %.1f us" % (eng.timing()["blocks_ms"] * 1e3))
