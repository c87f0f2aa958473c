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
    L.sz_plans_auto.restype = ctypes.c_ulonglong
    L.sz_plans_auto.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.sz_min_chunks.restype = ctypes.c_ulonglong
    L.sz_min_chunks.argtypes = [ctypes.c_ulonglong]
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
        assert 16 << 10 <= cm <= 256 << 10 and cm % 1024 == 0
        for L, c, m in zip(lens, cb0, n0):
            if c:
                assert 16384 <= c <= (256 << 10) and c % 1024 == 0 and m >= 8 and (m - 1) * c < L <= m * c


def test_a_fixed_chunk_size_is_left_alone(S):
    lens = [3 << 20] * 100
    j0, cb0, n0, _ = plans(S, lens, 2048, chunk_max=16384)
    assert (cb0 == 16384).all() and j0 == 100 * 192


def plans_auto(S, lens, slots):
    lens = np.asarray(lens, dtype=np.uint64)
    cb = np.zeros(lens.size, np.uint64)
    n = np.zeros(lens.size, np.uint32)
    cm = ctypes.c_ulonglong(0)
    jobs = S.sz_plans_auto(lens.ctypes.data, lens.size, slots, cb.ctypes.data, n.ctypes.data, ctypes.byref(cm))
    return jobs, cb, n, cm.value


def test_a_call_that_fills_the_slots_is_whole_rounds_of_them(S):
    """Round 6 (profiles/r06/inflate_min_chunks.log): 256 x 4 MiB text members were 8191 jobs of 47 KiB — 3.2 rounds of 2560 slots, 40.2 ms —
    and are one round of ~150 KiB jobs, 34.9 ms.  The rule: a call whose bytes fill the slots at chunk_max cuts every member at chunk_max."""
    assert S.sz_min_chunks(32 << 10) == 32 and S.sz_min_chunks(33 << 10) == 8
    for k, ln, slots in ((256, 1553699, 2560), (1024, 1553699, 2560), (128, 1553699, 2048)):
        j, cb, n, cm = plans_auto(S, [ln] * k, slots)
        rounds = -(-j // slots)
        assert (cb == cm).all() and j <= rounds * slots and j > (rounds - 1) * slots + slots // 2, (k, j, cm)
    j, cb, n, cm = plans_auto(S, [390000] * 512, 2560)          # members that do not hold eight chunks of chunk_max: eight or nine chunks each
    assert cm == 96 << 10 and (cb == 47 << 10).all() and j == 512 * 9
    # a few short members leave the slots empty: the 32 chunks a member of round 5
    j, cb, n, cm = plans_auto(S, [1553699] * 8, 2048)
    assert cm == 16 << 10 and (n >= 32).all()
    # one long member: nothing changes
    assert plans_auto(S, [380 << 20], 2048)[0] == plans(S, [380 << 20], 2048)[0]
    rng = np.random.default_rng(5)
    for case in range(2000):
        k = int(rng.integers(1, 1025))
        lens = rng.integers(100000, 8 << 20, k)
        slots = int(rng.choice([2048, 2560]))
        j, cb, n, cm = plans_auto(S, lens, slots)
        assert 16 << 10 <= cm <= 256 << 10 and cm % 1024 == 0
        for L, c, m in zip(lens, cb, n):
            if c:
                assert 16384 <= c <= cm and c % 1024 == 0 and m >= 8 and (m - 1) * c < L <= m * c
