"""oracle/szl_parallel.c (bench.py's cpu_baseline_all_cores): the pthread helper returns, slice by slice, what the single-threaded
oracle Deflater returns — it only adds threads."""
import ctypes

import numpy as np

import oracle_ffi as O
from sharpziplib_amd import corpus as C


def test_slices_on_threads_equal_the_single_thread_oracle():
    L = O.lib()
    L.szo_deflate_slices_mt.restype = ctypes.c_int64
    L.szo_deflate_slices_mt.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    sl, n = 300000, 9
    d = C.generate("enwik", 31, 0, sl * n)
    for level, threads in ((6, 1), (6, 4), (1, 3), (9, 16)):
        lens = np.zeros(n, dtype=np.uint64)
        tot = L.szo_deflate_slices_mt(d.ctypes.data, sl, n, level, threads, lens.ctypes.data)
        want = [len(O.deflate(d[i * sl:(i + 1) * sl], level)) for i in range(n)]
        assert [int(x) for x in lens] == want and tot == sum(want)
