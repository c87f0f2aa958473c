"""CPU test: the level-0 (DeflateStored) block arithmetic replayed by the product library's host code
(szl_debug_stored_layout — no device involved) equals the oracle engine's stored blocks for the same
SetInput chunking (level 0 output depends on the chunk sizes)."""
import ctypes

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import _lib, corpus as C


def product_layout(chunks, flush=False):
    L = _lib.lib()
    arr = np.array(chunks, dtype=np.uint64)
    rows = np.zeros(3 * 4096, np.uint64)
    n = ctypes.c_size_t(0)
    assert L.szl_debug_stored_layout(arr.ctypes.data, arr.size, 1 if flush else 0, rows.ctypes.data, 4096, ctypes.byref(n)) == 0
    return [(int(rows[3 * i]), int(rows[3 * i + 1]), int(rows[3 * i + 2])) for i in range(n.value)]


def oracle_layout(data, chunks, flush=False):
    """Blocks emitted by the oracle Deflater(0) when fed `chunks` with the DeflaterOutputStream call pattern."""
    tok = np.zeros(8, np.uint32)
    blk = (O.BlockInfo * 4096)()
    tr = O.Trace(tok.ctypes.data, 0, 0, ctypes.addressof(blk), 4096, 0)
    d = O.Deflater(0, True)
    d.L.szo_deflater_set_trace(d.h, ctypes.byref(tr))
    pos = 0
    out = bytearray()
    for c in chunks:
        d.set_input(data[pos:pos + c]); pos += c
        while not d.needs_input:
            b = d.deflate(512)
            if not b:
                break
            out += b
    if flush:
        d.flush()
        while True:
            b = d.deflate(512)
            if not b:
                break
            out += b
    d.finish()
    while not d.finished:
        out += d.deflate(512)
    res, off = [], 0
    for i in range(tr.blk_n):
        res.append((off, blk[i].stored_len, blk[i].last))
        off += blk[i].stored_len
    return res, bytes(out)


CHUNKINGS = [[0], [1], [100000], [65531], [65536, 65536], [32505, 32506, 32507, 1], [7] * 50, [4096] * 40, [70000, 3, 70000],
             [262, 261, 65274, 5, 200000], [1 << 20]]


@pytest.mark.parametrize("chunks", CHUNKINGS, ids=lambda c: "x".join(map(str, c[:4])) + ("..." if len(c) > 4 else ""))
@pytest.mark.parametrize("flush", [False, True])
def test_stored_layout_matches_oracle(chunks, flush):
    data = C.random_bytes(sum(chunks) + 1, seed=3)
    want, stream = oracle_layout(data, chunks, flush)
    got = product_layout(chunks, flush)
    assert got == want
    # and the byte stream is exactly header(5 B) + data per block
    expect = bytearray()
    for off, ln, last in want:
        expect += bytes([last, ln & 0xFF, ln >> 8, (~ln) & 0xFF, ((~ln) >> 8) & 0xFF]) + data[off:off + ln].tobytes()
    assert bytes(expect) == stream
