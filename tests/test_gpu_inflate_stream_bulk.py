"""The streaming Inflater with LONG inputs (run with -m gpu): InflaterInputStream takes its buffer size as a constructor argument
(CS/InflaterInputStream.cs:342-396) and hands the Inflater whatever Fill() read (:486-500, :658-690).  A SetInput of 2 MiB or more
is brought to a block header by the one-wavefront decoder and then decoded by the chunk-parallel path from the carried 32 KiB
window (csrc/szl_api_inflate.hip inflater_bulk) — same bytes, same IsFinished / IsNeedingInput / RemainingInput / TotalIn / Adler
as the oracle's Inflater (C/Inflater.cs:715-776, :862-884; S/GZip/GzipInputStream.cs:318-319 depends on RemainingInput)."""
import io
import zlib

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import _lib
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


def _inflate_all(inf, pieces, read=1 << 20):
    """feed `pieces` one SetInput at a time (only when the object asks for input), drain with Inflate(read)"""
    out = bytearray()
    buf = np.zeros(read, np.uint8)
    it = iter(pieces)
    while True:
        k = inf.Inflate(buf, 0, read)
        out += buf[:k].tobytes()
        if inf.IsFinished:
            break
        if k == 0:
            if inf.IsNeedingInput:
                try:
                    inf.SetInput(next(it))
                except StopIteration:
                    break
            else:
                raise AssertionError("no progress")
    return bytes(out)


def _bulk(inf):
    return int(_lib.lib().szl_inflater_debug_bulk_calls(inf._h))


@pytest.mark.parametrize("kind,level", [("enwik", 6), ("logs", 9), ("dickens", 5)])
def test_whole_member_in_one_setinput(kind, level):
    from sharpziplib_amd.inflater import Inflater
    plain = C.generate(kind, 77, 0, (96 if kind == "logs" else 24) << 20)
    comp = O.deflate(plain, level)
    assert len(comp) > (3 << 20)
    inf = Inflater(True)
    got = _inflate_all(inf, [comp + b"TRAILING-BYTES"])
    assert got == plain.tobytes()
    assert inf.IsFinished and inf.RemainingInput == len(b"TRAILING-BYTES") and inf.TotalIn == len(comp) and inf.TotalOut == plain.size
    assert _bulk(inf) >= 1, "the long input never reached the chunk-parallel decoder"


def test_pieces_of_many_sizes_and_zlib_framing():
    """4 MiB / 16 MiB / odd-sized pieces with the zlib header and the Adler-32 trailer; a zlib-made member (foreign block structure)"""
    from sharpziplib_amd.inflater import Inflater
    plain = C.generate("enwik", 78, 0, 40 << 20)
    for comp, nowrap in ((O.deflate(plain, 6, nowrap=False), False), (zlib.compress(plain.tobytes(), 6), False), (O.deflate(plain, 6), True)):
        for piece in (4 << 20, (16 << 20) + 12345, 3000001):
            inf = Inflater(nowrap)
            got = _inflate_all(inf, [comp[o:o + piece] for o in range(0, len(comp), piece)], read=777777)
            assert got == plain.tobytes() and inf.IsFinished and inf.RemainingInput == 0 and inf.TotalIn == len(comp)
            if not nowrap:
                assert inf.Adler == zlib.adler32(plain.tobytes())
            assert _bulk(inf) >= 1


def test_small_and_large_inputs_mixed_and_no_dynamic_blocks():
    from sharpziplib_amd.inflater import Inflater
    plain = C.generate("logs", 79, 0, 80 << 20)
    comp = O.deflate(plain, 6)
    # 4 KiB pieces first (one wavefront), then the rest at once (parallel), to a stream that continues in 64 KiB pieces
    cut1, cut2 = 300000, len(comp) - 500000
    assert cut2 - cut1 > (3 << 20)
    pieces = [comp[o:min(o + 4096, cut1)] for o in range(0, cut1, 4096)] + [comp[cut1:cut2]] + [comp[o:o + 65536] for o in range(cut2, len(comp), 65536)]
    inf = Inflater(True)
    assert _inflate_all(inf, pieces, read=100000) == plain.tobytes() and inf.IsFinished and inf.TotalIn == len(comp)
    assert _bulk(inf) >= 1
    # stored blocks only (level 0) and random bytes: nothing for the block finder — the ordinary decoder does it all
    rnd = C.random_bytes(6 << 20, seed=9)
    for c, want in ((O.deflate(plain[:8 << 20], 0), plain[:8 << 20].tobytes()), (O.deflate(rnd, 6), rnd.tobytes())):   # (both > 2 MiB)
        inf = Inflater(True)
        got = _inflate_all(inf, [c])
        assert got == want and inf.IsFinished and inf.TotalIn == len(c)


def test_truncated_and_corrupt_long_inputs():
    from sharpziplib_amd.deflater import SharpZipBaseException
    from sharpziplib_amd.inflater import Inflater
    plain = C.generate("enwik", 80, 0, 16 << 20)
    comp = O.deflate(plain, 6)
    # truncated: everything before the cut comes out, then the object asks for input
    cut = len(comp) * 2 // 3
    inf = Inflater(True)
    got = _inflate_all(inf, [comp[:cut]])
    assert not inf.IsFinished and inf.IsNeedingInput and plain.tobytes().startswith(got) and len(got) > plain.size // 2
    n, delivered, cons = O.inflate_probe(comp[:cut], max_out=plain.size + 16)
    assert len(got) == len(delivered)
    # ... and the stream goes on when the rest arrives
    got += _inflate_all(inf, [comp[cut:]])
    assert got == plain.tobytes() and inf.IsFinished and inf.TotalIn == len(comp)
    # corrupt in the middle: the bytes in front of the damage are delivered, then the same exception as the reference's
    bad = bytearray(comp); bad[len(comp) // 2] ^= 0x55
    n, delivered, _ = O.inflate_probe(bytes(bad), max_out=plain.size + 16)
    inf = Inflater(True)
    out = bytearray()
    buf = np.zeros(1 << 20, np.uint8)
    inf.SetInput(bytes(bad))
    err = None
    try:
        while not inf.IsFinished:
            k = inf.Inflate(buf, 0, buf.size)
            out += buf[:k].tobytes()
            if k == 0:
                break
    except SharpZipBaseException as e:
        err = e
    if n < 0:
        assert err is not None or not inf.IsFinished
        assert bytes(out)[:len(delivered)] == delivered[:len(out)]
    else:
        assert bytes(out) == delivered


def test_gzip_and_inflaterinputstream_with_large_buffers():
    from sharpziplib_amd.gzipstream import GZipInputStream, write_members
    from sharpziplib_amd.inflater import Inflater
    from sharpziplib_amd.streams import InflaterInputStream
    plain = C.generate("enwik", 81, 0, 48 << 20)
    comp = O.deflate(plain, 6)
    for bufsz in (1 << 20, 16 << 20, 64 << 20):
        inf = Inflater(True)
        st = InflaterInputStream(io.BytesIO(comp), inf, bufsz, readAhead=0)    # (the reference's sizes: what the constructor says)
        out = np.zeros(4 << 20, np.uint8)
        got = bytearray()
        while True:
            k = st.Read(out, 0, out.size)
            if k <= 0:
                break
            got += out[:k].tobytes()
        assert bytes(got) == plain.tobytes() and inf.TotalIn == len(comp)
        assert (_bulk(inf) >= 1) == (bufsz >= (2 << 20))
    # two gzip members behind each other through GZipInputStream (RemainingInput must be exact at every member's end)
    a, b = plain[:20 << 20], plain[20 << 20:33 << 20]
    gz = b"".join(write_members([a, b], level=6))
    g = GZipInputStream(io.BytesIO(gz), 32 << 20)
    got = bytearray()
    out = np.zeros(4 << 20, np.uint8)
    while True:
        k = g.Read(out, 0, out.size)
        if k <= 0:
            break
        got += out[:k].tobytes()
    assert bytes(got) == a.tobytes() + b.tobytes()
