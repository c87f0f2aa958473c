"""csrc/szl_inflate_reftree.h (the table k_inflate's exact-table mode decodes through) against the oracle's restatement of
C/InflaterHuffmanTree.cs: the same table entry for entry, the same symbol and bit count for every lookup, for complete sets,
incomplete sets and the sets on which the reference's table is not a canonical decoder."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import corrupt_streams as CS
import oracle_ffi as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def rt(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("reftree") / "libreftree.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-o", so, os.path.join(HERE, "reftree_harness.cpp")])
    L = ctypes.CDLL(so)
    L.rt_table.restype = ctypes.c_int
    L.rt_table.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.rt_symbol.restype = ctypes.c_int
    L.rt_symbol.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    L.rt_quirk.restype = ctypes.c_int
    L.rt_quirk.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.rt_sm_script.restype = ctypes.c_int
    L.rt_sm_script.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return L


def _oracle():
    L = O.lib()
    L.szo_iht_table.restype = ctypes.c_int
    L.szo_iht_table.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.szo_iht_symbol.restype = ctypes.c_int
    L.szo_iht_symbol.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    return L


_random_set = CS.random_code_set


def test_tables_and_lookups_equal_the_oracle(rt):
    ORA = _oracle()
    rng = np.random.default_rng(0xA16)
    quirk = built = index_err = 0
    for case in range(3000):
        n = int(rng.choice([286, 286, 30, 30, 257 + int(rng.integers(30)), 1 + int(rng.integers(30)), 19]))
        lens = _random_set(rng, n, ["complete", "drop", "drop", "longer", "longer"][case % 5])
        want = np.zeros(4096, dtype=np.int16)
        got = np.full(4096, 0x7777, dtype=np.int16)
        wsz = ORA.szo_iht_table(lens.ctypes.data, n, want.ctypes.data, want.size)
        gsz = rt.rt_table(lens.ctypes.data, n, got.ctypes.data, 1184)
        if wsz < 0:
            assert wsz == -13 and gsz == -3, (case, wsz, gsz)   # IndexOutOfRange out of the constructor
            index_err += 1
            continue
        assert gsz == wsz, (case, gsz, wsz)
        assert (got[:gsz] == want[:wsz]).all(), case
        built += 1
        quirk += rt.rt_quirk(lens.ctypes.data, n)
        # lookups: random bit patterns, every amount of input left
        pats = rng.integers(0, 1 << 32, size=160, dtype=np.uint64)
        for j, b in enumerate(pats):
            avail = 32 if j < 96 else int(rng.integers(0, 17))
            bits = int(b) if avail >= 32 else int(b) & ((1 << avail) - 1)
            dropped = ctypes.c_int(0)
            ws = ORA.szo_iht_symbol(want.ctypes.data, wsz, bits, avail, ctypes.byref(dropped))
            gs = rt.rt_symbol(got.ctypes.data, bits, avail)
            if ws == -1:
                assert gs == -1, (case, j, avail)
            elif ws <= -100:
                assert gs == -2, (case, j, ws, gs)               # "Encountered invalid codelength 0"
            else:
                assert gs == (ws | (dropped.value << 16)), (case, j, avail, ws, dropped.value, gs)
    assert quirk > 300 and built > 2000      # the damaged sets do produce the quirk class


def test_capacity_bound_holds():
    """512 + codes of 10+ bits + 120 (the header's RT_CAP_*): the worst sets stay inside it"""
    ORA = _oracle()
    worst = 0
    rng = np.random.default_rng(7)
    for case in range(2000):
        n = 286
        lens = np.zeros(n, dtype=np.uint8)
        # spread long codes over as many partial prefixes as possible
        k = int(rng.integers(1, n))
        lens[:k] = rng.integers(10, 16, size=k)
        kraft = int(np.sum(1 << (16 - lens[:k].astype(np.int64))))
        if kraft > 65536:
            continue
        out = np.zeros(8192, dtype=np.int16)
        sz = ORA.szo_iht_table(lens.ctypes.data, n, out.ctypes.data, out.size)
        if sz > 0:
            worst = max(worst, sz)
    assert worst <= 1024


def test_bit_buffer_scripts_equal_the_oracles_streammanipulator(rt):
    """The emulation of StreamManipulator k_inflate_exact decodes with (ExSM: 16-bit loads, the odd first byte, DropBits that does not
    look, PeekBits shifting by a negative count) against the oracle's restatement, on random scripts: peeks of 1-16 bits, drops of what
    was peeked — and, every so often, of MORE than the buffer holds (what GetSymbol does with a garbage table entry) —, byte alignment,
    the counters.  (GetSymbol on a clean buffer is covered above; on a buffer that has underflowed the reference indexes its table
    with whatever the negative count masks out of the buffer — an IndexOutOfRangeException in C#, nothing to compare.)"""
    ORA = _oracle()
    ORA.szo_sm_script.restype = ctypes.c_int
    ORA.szo_sm_script.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(0x5A)
    underflows = 0
    for case in range(400):
        n = int(rng.integers(2, 300))
        buf = rng.integers(0, 256, size=n + 8, dtype=np.uint8)      # (slack behind the input: neither side may read it)
        # a table with the reference's quirks for GetSymbol
        tree = np.zeros(4096, dtype=np.int16)
        while True:
            lens = _random_set(rng, 286, ["drop", "longer"][case % 2])
            if ORA.szo_iht_table(lens.ctypes.data, 286, tree.ctypes.data, tree.size) > 0:
                break
        tsz = ORA.szo_iht_table(lens.ctypes.data, 286, tree.ctypes.data, tree.size)
        ops = []
        bits_known = 0
        for k in range(int(rng.integers(5, 120))):
            r = rng.random()
            if r < 0.35:
                a = int(rng.integers(1, 17)); ops += [0, a]
            elif r < 0.6:
                ops += [1, int(rng.integers(0, 10))]
            elif r < 0.68:
                ops += [1, int(rng.integers(10, 16))]; underflows += 1     # may exceed what the buffer holds
            elif r < 0.73:
                ops += [2, 0]
            elif r < 0.8:
                ops += [3, 0]
            else:
                ops += [4, 0]
        ops = np.array(ops, dtype=np.int32)
        nops = ops.size // 2
        want = np.zeros(nops, dtype=np.int32); got = np.zeros(nops, dtype=np.int32)
        assert ORA.szo_sm_script(buf.ctypes.data, n, ops.ctypes.data, nops, tree.ctypes.data, tsz, want.ctypes.data) == 0
        assert rt.rt_sm_script(buf.ctypes.data, n, ops.ctypes.data, nops, tree.ctypes.data, got.ctypes.data) == 0
        bad = np.flatnonzero(want != got)
        assert bad.size == 0, (case, n, int(bad[0]), ops[2 * bad[0]:2 * bad[0] + 2].tolist(), int(want[bad[0]]), int(got[bad[0]]), ops[:2 * bad[0] + 2].reshape(-1, 2).tolist()[-8:])
    assert underflows > 500
