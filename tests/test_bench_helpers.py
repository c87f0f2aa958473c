"""bench.py's host-side helpers for the other configs (CPU): the corpus generated in parallel slices equals the sequential one, and
the member "made by zlib" on host threads is ONE valid raw-deflate stream of the input."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sharpziplib_amd import corpus as C  # noqa: E402


def test_gen_parallel_equals_the_sequential_generator():
    n = (64 << 20) + 12345                       # more than one 64 MiB piece
    for kind, seed in (("enwik", 0x21B0), ("logs", 0x106)):
        a = bench.gen_parallel(kind, seed, n, threads=3)
        assert a.size == n and np.array_equal(a, C.generate(kind, seed, 0, n))


def test_zlib_member_parallel_is_one_valid_member():
    d = C.generate("enwik", 5, 0, (3 << 20) + 77)
    comp = bench.zlib_member_parallel(d, chunk=1 << 20)
    do = zlib.decompressobj(-15)
    back = do.decompress(comp)
    assert back == d.tobytes() and do.eof and do.unused_data == b""
