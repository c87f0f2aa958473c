"""csrc/szl_inflate_sizing.h — how the chunk-parallel Inflater cuts its members (host arithmetic, compiled here with g++).
The symbol pass runs ceil(jobs / slots) rounds of one chunk's decode, so a few stragglers cost a whole round.  Properties: without trimming the plan is the formula the library has always used;
trimming only ever acts on a tail round of less than an eighth of the slots, brings the job count down to whole rounds, keeps chunks
within [16 KiB, 256 KiB] in KiB steps and every member at 8 chunks or more; whatever the plan, the chunks cover the member."""
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
    L.sz_plans.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return L


def plans(S, lens, slots, trim, chunk_max=None):
    lens = np.asarray(lens, dtype=np.uint64)
    cm = chunk_max if chunk_max is not None else S.sz_chunk_max(int(lens.sum()), slots)
    cb = np.zeros(lens.size, np.uint64)
    n = np.zeros(lens.size, np.uint32)
    jobs = S.sz_plans(lens.ctypes.data, lens.size, cm, slots, int(trim), cb.ctypes.data, n.ctypes.data)
    return jobs, cb, n, cm


def old_formula(lens, chunk_max):
    out = []
    for L in lens:
        cb = min(max((int(L) // 32) & ~1023, 16384), chunk_max)
        out.append((0, 0) if L < 8 * cb else (cb, -(-int(L) // cb)))
    return out


def test_a_tail_round_is_trimmed(S):
    lens = [1553699] * 66                                    # 66 members of 4 MiB of text: 33 chunks each = 2178 jobs for 2048 slots
    jobs, cb, n, cm = plans(S, lens, 2048, False)
    assert jobs > 2048 and jobs - 2048 < 2048 // 8
    jobs2, cb2, n2, _ = plans(S, lens, 2048, True)
    assert jobs2 <= 2048 and (cb2 >= cb).all() and (cb2 <= cb * 1.2).all()
    lens = [1553699] * 64                                    # the 64 x 4 MiB call of tools/lab/inflate_one.py: 33 chunks each = 2112 jobs, 64 stragglers
    assert plans(S, lens, 2048, False)[0] == 2112 and plans(S, lens, 2048, True)[0] <= 2048
    lens = [1553699] * 62                                    # 2046 jobs for 2048 slots — nothing to trim
    assert plans(S, lens, 2048, True)[0] == plans(S, lens, 2048, False)[0] == 2046


def test_properties_on_random_calls(S):
    rng = np.random.default_rng(4)
    trimmed = 0
    for case in range(3000):
        k = int(rng.integers(1, 200))
        lens = rng.integers(100000, 40 << 20, k) if rng.random() < 0.5 else np.full(k, int(rng.integers(200000, 8 << 20)))
        slots = int(rng.choice([2048, 2560, 1024, 304 * 8]))
        j0, cb0, n0, cm = plans(S, lens, slots, False)
        assert [(int(a), int(b)) for a, b in zip(cb0, n0)] == old_formula(lens, cm)          # trimming off: the formula as it was
        j1, cb1, n1, _ = plans(S, lens, slots, True)
        for L, c, m in zip(lens, cb1, n1):
            if c:
                assert 16384 <= c <= (256 << 10) and c % 1024 == 0 and m >= 8 and (m - 1) * c < L <= m * c
        assert ((cb1 == 0) == (cb0 == 0)).all()                                               # nobody drops out of (or into) the chunked form
        if j1 != j0:
            trimmed += 1
            tail = j0 % slots
            assert j0 > slots and 0 < tail <= slots // 8 and j1 <= j0 - tail and (cb1 >= cb0).all()
            assert -(-j1 // slots) == j0 // slots                                             # one round fewer
        else:
            assert (cb1 == cb0).all()
    assert trimmed > 30


def test_a_fixed_chunk_size_is_left_alone(S):
    lens = [3 << 20] * 100
    j0, cb0, n0, _ = plans(S, lens, 2048, False, chunk_max=16384)
    assert (cb0 == 16384).all() and j0 == 100 * 192
