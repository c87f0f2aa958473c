"""CPU checks of the DeflateFast restatement the device kernel is built on (oracle/szl_model.c):
the reference's head/prev chain == the all-positions chain of stage A filtered by an "inserted" bit
(C/DeflaterEngine.cs:651-739).  The model's token stream must equal the oracle engine's token trace."""
import ctypes

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C


def _fast_model(data, level, strategy=0, fixpoint_range=None):
    L = O.lib()
    d = np.ascontiguousarray(data, dtype=np.uint8)
    n = d.size
    dp = np.concatenate([d, np.zeros(8, np.uint8)])
    P = O.FastParams()
    assert L.szm_fast_level_params(level, ctypes.byref(P)) == 0
    P.strategy = strategy
    se = np.array([n], np.uint64)
    link = np.zeros(n + 8, np.uint16)
    L.szm_links(dp.ctypes.data, n, se.ctypes.data, 1, link.ctypes.data)
    flags = np.zeros(n + 8, np.uint8)
    tok = np.zeros(n + 8, np.uint32)
    if fixpoint_range is None:
        k = L.szm_fast_parse(dp.ctypes.data, 0, n, link.ctypes.data, ctypes.byref(P), flags.ctypes.data, tok.ctypes.data)
        return tok[:k].copy(), flags[:n].copy(), None
    it = ctypes.c_uint64(0)
    k = L.szm_fast_parse_fixpoint(dp.ctypes.data, 0, n, link.ctypes.data, ctypes.byref(P), fixpoint_range, flags.ctypes.data,
                                  tok.ctypes.data, ctypes.byref(it))
    return tok[:k].copy(), flags[:n].copy(), it.value


INPUTS = {
    "dickens": lambda: C.generate("dickens", 7, 0, 300000), "logs": lambda: C.generate("logs", 7, 0, 300000),
    "zeros": lambda: C.zeros(100000), "random": lambda: C.random_bytes(50000), "mixed": lambda: C.mixed(300000, seed=5),
    "slide": lambda: C.generate("enwik", 11, 0, 70000),   # crosses window index 65274 (:680 strict vs FillWindow :371)
}


@pytest.mark.parametrize("name", sorted(INPUTS))
@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_filtered_chain_model_equals_engine_trace(name, level):
    data = INPUTS[name]()
    tok, flags, _ = _fast_model(data, level)
    _, tr = O.deflate(data, level, trace=True)
    assert np.array_equal(tok, tr["tokens"])
    assert 0 < flags.sum() <= data.size


def test_huffman_only_inserts_but_never_matches():
    data = C.generate("dickens", 3, 0, 50000)
    tok, flags, _ = _fast_model(data, 2, strategy=2)
    _, tr = O.deflate(data, 2, strategy=2, trace=True)
    assert np.array_equal(tok, tr["tokens"]) and (tok >> 16).max() == 0
    assert flags[:-2].all()   # every position with 3 bytes of lookahead is inserted (:686)


def test_range_fixpoint_is_exact_but_slow_to_converge():
    """Why the device runs DeflateFast one wavefront per stream: a range-parallel fixpoint over the inserted bits reproduces the
    sequential parse when it converges, but needs many sweeps (each a full pass)."""
    data = C.generate("dickens", 7, 0, 200000)
    seq, fseq, _ = _fast_model(data, 1)
    fix, ffix, iters = _fast_model(data, 1, fixpoint_range=4096)
    assert np.array_equal(seq, fix) and np.array_equal(fseq, ffix)
    assert iters >= 8   # 49 ranges: far from the 2-3 sweeps a parallel formulation would need to pay off


def test_parse_reads_about_a_third_of_the_match_tables_on_text():
    """The fact stage B's on-demand form exploits."""
    L = O.lib()
    data = C.generate("enwik", 0xE9, 0, 1 << 20)
    M = O.Model(data, 6)
    need = np.zeros(M.n + 8, np.uint8)
    k = L.szm_parse_needed(M._dpad.ctypes.data, 0, M.n, M.link.ctypes.data, M.m2.ctypes.data, M.mq.ctypes.data, ctypes.byref(M.P),
                           need.ctypes.data)
    assert k == int(need.sum()) and 0.25 < k / M.n < 0.45


@pytest.mark.parametrize("kind,lo,hi", [("enwik", 0.40, 0.60), ("logs", 0.08, 0.30)])
def test_on_demand_walkers_cover_the_parse_except_at_tile_crossings(kind, lo, hi):
    """Model of k_match_lazy (stride 16, tiles of 16384): the positions the true parse reads are all evaluated by the tile
    walkers except a handful right after the path crosses into a tile (those are what eval_global serves on the device)."""
    L = O.lib()
    data = C.generate(kind, 0xE9, 0, 1 << 20)
    M = O.Model(data, 6)
    args = (M._dpad.ctypes.data, 0, M.n, M.link.ctypes.data, M.m2.ctypes.data, M.mq.ctypes.data, ctypes.byref(M.P))
    need = np.zeros(M.n + 8, np.uint8)
    L.szm_parse_needed(*args, need.ctypes.data)
    ev = np.zeros(M.n + 8, np.uint8)
    k = L.szm_lazy_eval_set(*args, 16384, 16, ev.ctypes.data)
    assert lo < k / M.n < hi
    miss = np.nonzero((need[:M.n] == 1) & (ev[:M.n] == 0))[0]
    ntiles = M.n // 16384
    assert miss.size <= 6 * ntiles                                   # a few per tile boundary at most
    assert np.all(miss % 16384 < 600)                                # and only right behind a tile start (before the paths merge)
