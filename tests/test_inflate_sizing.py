"""csrc/szl_inflate_sizing.h — how the chunk-parallel Inflater cuts its members (host arithmetic, compiled here with g++).
The symbol pass runs ceil(jobs / slots) rounds of one chunk's decode.  Properties: the plan is the formula the library
has used since round 2 (checked against its restatement here on random calls); chunks stay within [16 KiB, 256 KiB] in KiB steps,
every chunked member has 8 chunks or more, and whatever the plan, the chunks cover the member."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def S(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("sizing") / "libsizing.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-o", so, os.path.join(HERE, "sizing_harness.cpp")])
    L = ctypes.CDLL(so)
    L.sz_chunk_max.restype = ctypes.c_ulonglong
    L.sz_chunk_max.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong]
    L.sz_plans.restype = ctypes.c_ulonglong
    L.sz_plans.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
    return L


def plans(S, lens, slots, chunk_max=None):
    lens = np.asarray(lens, dtype=np.uint64)
    cm = chunk_max if chunk_max is not None else S.sz_chunk_max(int(lens.sum()), slots)
    cb = np.zeros(lens.size, np.uint64)
    n = np.zeros(lens.size, np.uint32)
    jobs = S.sz_plans(lens.ctypes.data, lens.size, cm, cb.ctypes.data, n.ctypes.data)
    return jobs, cb, n, cm


def formula(lens, chunk_max):
    out = []
    for L in lens:
        cb = min(max((int(L) // 32) & ~1023, 16384), chunk_max)
        out.append((0, 0) if L < 8 * cb else (cb, -(-int(L) // cb)))
    return out


def test_known_calls(S):
    assert plans(S, [1553699] * 64, 2048)[0] == 2112          # the 64 x 4 MiB call of tools/lab/inflate_one.py: 33 chunks each
    j, cb, n, cm = plans(S, [380 << 20], 2048)                # one 1 GiB text member (380 MiB compressed): one round of the slots
    assert 2000 <= j <= 2048 and 180 << 10 <= cb[0] <= 200 << 10


def test_properties_on_random_calls(S):
    rng = np.random.default_rng(4)
    for case in range(3000):
        k = int(rng.integers(1, 200))
        lens = rng.integers(100000, 40 << 20, k) if rng.random() < 0.5 else np.full(k, int(rng.integers(200000, 8 << 20)))
        slots = int(rng.choice([2048, 2560, 1024, 304 * 8]))
        j0, cb0, n0, cm = plans(S, lens, slots)
        assert [(int(a), int(b)) for a, b in zip(cb0, n0)] == formula(lens, cm)
        assert 32 << 10 <= cm <= 256 << 10 and cm % 1024 == 0
        for L, c, m in zip(lens, cb0, n0):
            if c:
                assert 16384 <= c <= (256 << 10) and c % 1024 == 0 and m >= 8 and (m - 1) * c < L <= m * c


def test_a_fixed_chunk_size_is_left_alone(S):
    lens = [3 << 20] * 100
    j0, cb0, n0, _ = plans(S, lens, 2048, chunk_max=16384)
    assert (cb0 == 16384).all() and j0 == 100 * 192
