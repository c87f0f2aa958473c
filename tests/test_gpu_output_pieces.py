"""Finish()'s bytes come back from the device in pieces while the caller already takes them (run with -m gpu; csrc/szl_api.hip
szl_deflater::out_pieces): Deflate() / DeflateView() hand out what has arrived and wait for the next piece only when nothing has.  The
product's pieces are 16 MiB; here SZL_OUT_PIECE_KIB makes them 64 KiB so that a few megabytes are dozens of pieces.  Whatever the caller
does while pieces are on their way — small Deflate() buffers, views, Reset(), destroying the object, a second stream on the same object —
the bytes are the oracle's (C/Deflater.cs:427 hands out of PendingBuffer, which holds everything at once: the reference has no such state)."""
import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import _lib, corpus as C

pytestmark = pytest.mark.gpu
FORGET = -2147483648


@pytest.fixture()
def small_pieces():
    L = _lib.lib()
    L.szl_debug_set(b"SZL_OUT_PIECE_KIB", 64)
    yield L
    L.szl_debug_set(b"SZL_OUT_PIECE_KIB", FORGET)


def _finish_all(d, buf):
    out = bytearray()
    d.Finish()
    while not d.IsFinished:
        n = d.Deflate(buf)
        assert n > 0                                   # (0 before IsFinished would mean "needs input", C/Deflater.cs:482)
        out += buf[:n].tobytes()
    return bytes(out)


@pytest.mark.parametrize("bufsize", [512, 4096, 100000, 1 << 22])
@pytest.mark.parametrize("nowrap", [True, False])
def test_deflate_hands_out_the_oracles_bytes_piece_by_piece(small_pieces, bufsize, nowrap):
    from sharpziplib_amd.deflater import Deflater
    data = C.random_bytes(700000, seed=9).tobytes() + C.generate("enwik", 3, 0, 5 << 20).tobytes()   # (poorly compressible front: many pieces)
    data = np.frombuffer(data, np.uint8)
    want, tin, tout = O.stream_deflate(data, 6, nowrap, chunk=data.size, flush_every=None)
    d = Deflater(6, nowrap)
    d.SetInput(data)
    got = _finish_all(d, np.zeros(bufsize, np.uint8))
    assert got == want and d.TotalOut == tout and d.TotalIn == tin
    if not nowrap:
        assert d.Adler == O.adler32(data)


def test_views_arrive_in_order_and_add_up(small_pieces):
    from sharpziplib_amd.deflater import Deflater
    data = C.generate("dickens", 5, 0, 6 << 20)
    want = O.deflate(data, 7)
    d = Deflater(7, True)
    d.SetInput(data)
    d.Finish()
    got, views = bytearray(), 0
    while not d.IsFinished:
        v = d.DeflateView()
        assert v is not None and len(v) > 0
        got += bytes(v); views += 1
    assert bytes(got) == want and d.TotalOut == len(want)
    assert views >= 1


def test_reset_and_reuse_while_pieces_are_on_their_way(small_pieces):
    from sharpziplib_amd.deflater import Deflater
    a, b = C.generate("enwik", 8, 0, 4 << 20), C.generate("logs", 9, 0, 3 << 20)
    d = Deflater(6, True)
    buf = np.zeros(3000, np.uint8)
    d.SetInput(a); d.Finish()
    n = d.Deflate(buf)                                   # the first bytes only: the rest is still arriving
    assert n == buf.size and buf[:n].tobytes() == O.deflate(a, 6)[:n]
    d.Reset()                                            # C/Deflater.cs:204: everything pending is dropped
    d.SetInput(b)
    assert _finish_all(d, np.zeros(1 << 16, np.uint8)) == O.deflate(b, 6)
    d2 = Deflater(6, True)                               # an object destroyed in the middle of its download
    d2.SetInput(a); d2.Finish()
    assert d2.Deflate(buf) == buf.size
    del d2
    d3 = Deflater(9, True)                               # (the pooled engine and its buffers go to the next object)
    d3.SetInput(b)
    assert _finish_all(d3, np.zeros(1 << 20, np.uint8)) == O.deflate(b, 9)


def test_pipelined_parts_and_pieces_together(small_pieces):
    from sharpziplib_amd.deflater import Deflater
    L = small_pieces
    L.szl_debug_set(b"SZL_PIPE_PART_KIB", 512); L.szl_debug_set(b"SZL_UP_SLAB_KIB", 64)
    try:
        data = C.generate("enwik", 21, 0, 7 << 20)
        d = Deflater(6, True)
        d.EnableCrc32()
        for o in range(0, data.size, 300000):
            d.SetInput(data[o:o + 300000])
            assert d.Deflate(np.zeros(16, np.uint8)) == 0
        got = _finish_all(d, np.zeros(50000, np.uint8))
        assert got == O.deflate(data, 6) and d.Crc32 == O.crc32(data)
        assert L.szl_deflater_debug_pipe_parts(d._h) >= 2
    finally:
        L.szl_debug_set(b"SZL_PIPE_PART_KIB", FORGET); L.szl_debug_set(b"SZL_UP_SLAB_KIB", FORGET)
