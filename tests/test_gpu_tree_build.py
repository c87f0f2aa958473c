"""Stage D's Huffman build, a wavefront per tree (run with -m gpu; csrc/szl_kernels_block.hip build_tree): the code lengths of
frequency vectors no token stream would produce — ties everywhere, one and two symbols, Fibonacci-like weights whose codes outgrow
maxLength and go through BuildLength's repair (C/DeflaterHuffman.cs:519-571) — against the oracle's Tree.BuildTree (:196-329, :475-579).
Tie-breaking is by heap position, so equal lengths here mean the heap array was the reference's after every sift.  Tolerance 0."""
import ctypes

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import _lib

pytestmark = pytest.mark.gpu

TREES = {"literal": (286, 257, 15), "distance": (30, 1, 15), "codelen": (19, 4, 7)}


def histograms(rng, nsym, count, total_cap):
    out = np.zeros((count, nsym), np.int32)
    for t in range(count):
        mode = t % 8
        k = int(rng.integers(0, nsym + 1)) if mode != 7 else nsym
        if mode == 4:
            k = min(k, int(rng.integers(16, 23)))                # (few symbols, Fibonacci weights: depth ~ k)
        idx = rng.permutation(nsym)[:k]
        if mode == 0:
            f = rng.integers(1, 16000, k)                      # wide range
        elif mode == 1:
            f = rng.integers(1, 4, k)                          # heavy ties
        elif mode == 2:
            f = np.full(k, int(rng.integers(1, 50)))           # all equal
        elif mode == 3:
            f = (16000.0 / (1 + rng.integers(0, 4096, k))).astype(np.int64) + 1     # geometric-ish
        elif mode == 4:                                        # Fibonacci weights: the deepest tree a block can have -> over-long codes
            fib = [1, 1]
            while len(fib) < min(k, 22):
                fib.append(fib[-1] + fib[-2])
            f = np.array((fib * (k // len(fib) + 1))[:k], dtype=np.int64)
            rng.shuffle(f)
        elif mode == 5:
            f = 2 ** rng.integers(0, 13, k)                    # powers of two
        elif mode == 6:
            f = np.sort(rng.integers(1, 300, k))               # sorted, many equal neighbours
        else:
            f = rng.integers(1, 3, k)
        f = np.asarray(f, dtype=np.int64)
        while f.size and f.sum() > total_cap:                  # what 16384 tokens (+ end of block) can add up to
            f = np.maximum(1, f // 2)
        out[t, idx] = f
    return out


@pytest.mark.parametrize("tree", list(TREES))
def test_code_lengths_equal_the_references_on_random_histograms(tree):
    nsym, min_codes, max_len = TREES[tree]
    rng = np.random.default_rng({"literal": 11, "distance": 12, "codelen": 13}[tree])
    count = 10000 if tree != "literal" else 4000
    H = histograms(rng, nsym, count, 16385 if tree != "codelen" else 320)
    if tree == "literal":
        H[:, 256] = np.maximum(H[:, 256], 1)                   # the end-of-block symbol is always counted (:790)
    lens = np.zeros((count, nsym), np.uint8)
    ncodes = np.zeros(count, np.int32)
    L = _lib.lib()
    _lib.check(L.szl_debug_tree_lengths(np.ascontiguousarray(H).ctypes.data, count, nsym, min_codes, max_len, lens.ctypes.data, ncodes.ctypes.data), "tree probe")
    repaired = 0
    for t in range(count):
        want, nc = O.tree_lengths(H[t], min_codes, max_len)
        assert nc == ncodes[t] and (want == lens[t]).all(), "%s tree %d: lengths differ\nfreqs %s\nwant  %s\ngot   %s" % (tree, t, H[t].tolist(), want.tolist(), lens[t].tolist())
        if want.max(initial=0) == max_len:
            repaired += 1
    assert repaired > 20                                       # the over-long-code repair was on the path
