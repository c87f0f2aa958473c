"""The device-aware read path (run with -m gpu): InflaterInputBuffer / InflaterInputStream as INTEGRATION.md's file 3 replaces them
(sharpziplib_amd/streams.py is the Python form of sharpziplib_amd/dotnet/InflaterInputStream.Device.cs).  A default-constructed
GZipInputStream (4096, S/GZip/GzipInputStream.cs:72) must hand the device Inflater pieces the chunk-parallel decoder can take —
the buffer class reads 16 MiB ahead into pinned memory — while every member of the two classes keeps the reference's meaning
(CS/InflaterInputStream.cs:22-41, 93, 103, 115, 148-270, 276, 342-396, 420, 472, 486, 658): Available / ReadLe* / ReadRawBuffer /
ReadClearTextBuffer still find the container's trailers in that same buffer (S/GZip/GzipInputStream.cs:305-351)."""
import ctypes
import gzip
import io
import zlib

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import _lib
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


def _bulk(inf):
    return int(_lib.lib().szl_inflater_debug_bulk_calls(inf._h))


def _drain(st, read):
    out, buf = bytearray(), np.zeros(read, np.uint8)
    while True:
        k = st.Read(buf, 0, read)
        if k <= 0:
            return bytes(out)
        out += buf[:k].tobytes()


class _NoSeek(io.RawIOBase):
    """a base stream that can only be read, a few KiB at a time (a socket): Fill() must loop until its buffer is full (:120-128)"""

    def __init__(self, data, piece):
        self._b, self._p, self._piece = data, 0, piece

    def readable(self):
        return True

    def seekable(self):
        return False

    def readinto(self, mv):
        k = min(len(mv), self._piece, len(self._b) - self._p)
        mv[:k] = self._b[self._p:self._p + k]
        self._p += k
        return k


@pytest.mark.parametrize("read", [1 << 20, 70001, 4 << 20])
def test_default_constructed_gzip_input_stream_takes_the_parallel_decoder(read):
    from sharpziplib_amd.gzipstream import GZipInputStream, write_members
    a = C.generate("enwik", 91, 0, 40 << 20)
    b = C.generate("logs", 92, 0, 56 << 20)
    c = C.generate("dickens", 93, 0, 3000)
    gz = b"".join(write_members([a, b, np.zeros(0, np.uint8), c], level=6, names=["a", None, "e", "c"])) + b"\0\0not a member"
    g = GZipInputStream(io.BytesIO(gz))                   # the reference's default constructor: size 4096
    assert g.inputBuffer.RawData.size >= (16 << 20) and g.inputBuffer._pin is not None
    got = _drain(g, read)
    assert got == a.tobytes() + b.tobytes() + c.tobytes()
    assert _bulk(g.inf) >= 2                              # both long members went through many wavefronts
    g.Dispose()
    # the same through a base stream that cannot seek and trickles
    g = GZipInputStream(_NoSeek(gz, 100000))
    assert _drain(g, read) == got and _bulk(g.inf) >= 2
    g.Dispose()


@pytest.mark.parametrize("read_ahead_mib", [3, 16])
def test_pieces_end_on_a_block_boundary_and_a_truncated_stream_still_delivers_every_byte(read_ahead_mib):
    """szl_inflater_expect_more: a buffer filled to the brim promises more input, the piece ends on its last block boundary and the
    remainder waits for the next Fill(); at the end of the base stream the promise is taken back and the remainder is decoded — by
    the bytes, a truncated stream behaves as through the reference's classes (every byte it holds, then "Unexpected EOF")."""
    from sharpziplib_amd.deflater import SharpZipBaseException
    from sharpziplib_amd.inflater import Inflater
    from sharpziplib_amd.streams import InflaterInputStream
    plain = C.generate("enwik", 98, 0, 40 << 20)
    comp = O.deflate(plain, 6)
    st = InflaterInputStream(_NoSeek(comp, 1 << 20), Inflater(True), 4096, readAhead=read_ahead_mib << 20)
    assert _drain(st, 1 << 20) == plain.tobytes()
    pieces = _bulk(st.inf)
    assert pieces >= len(comp) // (read_ahead_mib << 20)
    tm = (ctypes.c_double * 8)()
    _lib.lib().szl_inflater_debug_times(st.inf._h, tm)
    st.Dispose()
    cut = comp[:len(comp) - 123457]
    st = InflaterInputStream(_NoSeek(cut, 1 << 20), Inflater(True), 4096, readAhead=read_ahead_mib << 20)
    got, buf = bytearray(), np.zeros(4096, np.uint8)
    with pytest.raises(SharpZipBaseException, match="Unexpected EOF"):
        while True:
            k = st.Read(buf, 0, buf.size)
            assert k > 0
            got += buf[:k].tobytes()
    want = zlib.decompressobj(-15).decompress(cut)          # (the Read() that meets the end throws with the bytes it had gathered, as the reference's does)
    assert len(want) - 4096 < len(got) <= len(want) and bytes(got) == want[:len(got)]


def test_buffer_class_members_keep_their_meaning_on_the_long_buffer():
    from sharpziplib_amd.inflater import Inflater
    from sharpziplib_amd.streams import InflaterInputBuffer, InflaterInputStream, ZipException
    plain = C.generate("enwik", 94, 0, 6 << 20)
    comp = O.deflate(plain, 6)
    tail = bytes(range(1, 40))
    src = b"\x11\x22\x33\x44\x55\x66\x77\x88\x99\xaa\xbb\xcc\xdd\xee\xff" + comp + tail
    ib = InflaterInputBuffer(io.BytesIO(src), 4096)
    assert ib.RawData.size == len(src) + 1 and ib.RawLength == 0 and ib.Available == 0     # a seekable base stream: no more than it holds
    assert ib.ReadLeByte() == 0x11 and ib.ReadLeShort() == 0x3322 and ib.ReadLeInt() == 0x77665544 and ib.ReadLeLong() == 0xFFEEDDCCBBAA9988
    assert ib.RawLength == len(src) and ib.Available == len(src) - 15 and ib.ClearText is ib.RawData and ib.ClearTextLength == ib.RawLength
    inf = Inflater(True)
    ib.SetInflaterInput(inf)                              # everything that is left, in one SetInput, out of the pinned buffer
    assert ib.Available == 0
    out = np.zeros(plain.size + 10, np.uint8)
    got = 0
    while not inf.IsFinished:
        k = inf.Inflate(out, got, min(1 << 20, out.size - got))
        assert k > 0 or inf.IsFinished                    # (the call that learns of the stream's end may have nothing left to hand out)
        got += k
    assert got == plain.size and out[:got].tobytes() == plain.tobytes() and _bulk(inf) >= 1
    assert inf.RemainingInput == len(tail)
    ib.Available += inf.RemainingInput                    # what GZipInputStream.ReadFooter / ZipInputStream do (:318, ZipInputStream.cs:443)
    t = np.zeros(len(tail), np.uint8)
    assert ib.ReadRawBuffer(t[:20]) == 20 and ib.ReadClearTextBuffer(t, 20, len(tail) - 20) == len(tail) - 20 and t.tobytes() == tail
    assert ib.ReadRawBuffer(t, 0, 5) == 0                 # EOF: 0 (:168-172)
    with pytest.raises(ZipException, match="EOF in header"):
        ib.ReadLeByte()
    with pytest.raises(ValueError):
        ib.ReadRawBuffer(t, 0, -1)
    # bufferSize below 1024 is treated as 1024 (:35-38); the reference's sizes on request
    assert InflaterInputBuffer(io.BytesIO(src), 16, readAhead=0).RawData.size == 1024
    assert InflaterInputBuffer(io.BytesIO(src), 5000, readAhead=0).RawData.size == 5000
    # a constructor that asks for MORE than the read-ahead gets it
    big = InflaterInputBuffer(_Endless(), 20 << 20)
    assert big.RawData.size == 20 << 20
    big.Dispose()
    st = InflaterInputStream(io.BytesIO(comp), Inflater(True))
    assert st.Available == 1 and st.CanRead and not st.CanSeek and not st.CanWrite
    assert _drain(st, 1 << 20) == plain.tobytes() and st.Available == 0
    for f in (lambda: st.Length, lambda: st.Seek(0, 0), lambda: st.SetLength(1), lambda: st.Write(b"", 0, 0), lambda: st.WriteByte(1)):
        with pytest.raises(NotImplementedError):
            f()
    with pytest.raises(ValueError):
        st.Skip(0)


class _Endless(io.RawIOBase):
    def readable(self):
        return True

    def seekable(self):
        return False

    def readinto(self, mv):
        return 0


def test_inflater_outlives_the_stream_that_lent_it_a_buffer():
    """InflaterPool: Dispose() of the stream returns the Inflater, the pinned buffer is freed — the object must not keep referring to it"""
    from sharpziplib_amd.inflater import Inflater
    from sharpziplib_amd.streams import InflaterInputStream
    plain = C.generate("enwik", 95, 0, 8 << 20)
    comp = O.deflate(plain, 6)
    inf = Inflater(True)
    st = InflaterInputStream(io.BytesIO(comp), inf)
    buf = np.zeros(100000, np.uint8)
    assert st.Read(buf, 0, buf.size) == buf.size          # the first piece is decoded, most of its output still waits in the object
    st.IsStreamOwner = False
    st.Dispose()
    out = bytearray(buf.tobytes())
    big = np.zeros(1 << 20, np.uint8)
    while not inf.IsFinished:
        k = inf.Inflate(big)
        assert k > 0
        out += big[:k].tobytes()
    assert bytes(out) == plain.tobytes() and inf.TotalIn == len(comp)
    inf.Reset()
    inf.SetInput(comp[:50000])
    assert inf.Inflate(big) > 0


def test_adler_and_crc_of_what_was_handed_out_in_mid_stream():
    from sharpziplib_amd.inflater import Inflater
    plain = C.generate("dickens", 96, 0, 5 << 20)
    comp = O.deflate(plain, 6, nowrap=False)
    inf = Inflater(False)
    inf.EnableCrc32()
    inf.SetInput(comp)
    got = bytearray()
    buf = np.zeros(333333, np.uint8)
    for step in range(40):
        k = inf.Inflate(buf)
        got += buf[:k].tobytes()
        if step in (0, 3, 7):
            assert inf.Adler == zlib.adler32(bytes(got)) and inf.Crc32 == zlib.crc32(bytes(got))   # (Inflater.Adler: adler.Update in Inflate, C/Inflater.cs:752-756)
        if inf.IsFinished:
            break
    assert bytes(got) == plain.tobytes() and inf.Adler == zlib.adler32(plain.tobytes()) and inf.Crc32 == zlib.crc32(plain.tobytes())
    inf.Reset()
    assert inf.Crc32 == 0 and inf.Adler == 1
    with pytest.raises(Exception):
        inf.SetInput(comp[:10])
        inf.EnableCrc32(False)                            # switched only before the first SetInput


@pytest.mark.parametrize("out_mib", [16, 64])
def test_a_hostile_expansion_does_not_turn_into_gigabytes(out_mib):
    """round-4 ADVICE: DEFLATE expands ~1000:1; a SetInput of a few MiB must not allocate its whole expansion.  The piece handed to the
    parallel decoder is cut by the expected output, a piece whose real output is far beyond the bound is refused; the bytes are the same."""
    from sharpziplib_amd.inflater import Inflater
    L = _lib.lib()
    n = 600 << 20
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = b"".join(co.compress(bytes(8 << 20)) for _ in range(n >> 23)) + co.flush()
    text = C.generate("enwik", 97, 0, 4 << 20)
    assert len(comp) < (1 << 20)
    L.szl_debug_set(b"SZL_INF_BULK_OUT_MIB", out_mib)
    L.szl_debug_set(b"SZL_INF_STREAM_BULK_KIB", 256)
    try:
        inf = Inflater(True)
        inf.SetInput(comp)
        buf = np.zeros(8 << 20, np.uint8)
        tot = 0
        while not inf.IsFinished:
            k = inf.Inflate(buf)
            assert (k > 0 or inf.IsFinished) and not buf[:k].any()
            tot += k
        assert tot == n and inf.TotalIn == len(comp)
        # ordinary text under the same bound: cut into pieces, same bytes
        tz = O.deflate(np.concatenate([text] * 12), 6)
        inf = Inflater(True)
        inf.SetInput(tz)
        got = bytearray()
        while not inf.IsFinished:
            k = inf.Inflate(buf)
            assert k > 0 or inf.IsFinished
            got += buf[:k].tobytes()
        assert bytes(got) == text.tobytes() * 12 and _bulk(inf) >= (2 if out_mib == 16 else 1)
    finally:
        L.szl_debug_set(b"SZL_INF_BULK_OUT_MIB", -2147483648)
        L.szl_debug_set(b"SZL_INF_STREAM_BULK_KIB", -2147483648)


def test_python_gzip_members_and_small_reads_through_the_default_classes():
    from sharpziplib_amd.gzipstream import GZipInputStream, GZipException
    a = C.generate("dickens", 1, 0, 3 << 20).tobytes()
    b = C.generate("logs", 2, 0, 90000).tobytes()
    src = gzip.compress(a, 6) + gzip.compress(b"", 9) + gzip.compress(b, 1) + b"garbage"
    g = GZipInputStream(io.BytesIO(src))
    assert _drain(g, 50000) == a + b
    bad = bytearray(src); bad[len(gzip.compress(a, 6)) - 6] ^= 0x10
    with pytest.raises(GZipException, match="crc sum mismatch"):
        _drain(GZipInputStream(io.BytesIO(bytes(bad))), 1 << 20)
    with pytest.raises(EOFError):
        _drain(GZipInputStream(io.BytesIO(src[:len(gzip.compress(a, 6)) - 3])), 1 << 20)       # "EOS reading GZIP footer" (:322)


def test_streams_on_several_host_threads_overlap_and_stay_exact():
    """every streaming object runs on a HIP stream of its own (round 5): four GZipInputStreams and four GZipOutputStreams driven by eight
    host threads at once — a server's shape — must each produce what they produce alone"""
    import threading
    from sharpziplib_amd.gzipstream import GZipInputStream, GZipOutputStream, write_members
    datas = [C.generate(k, 200 + i, 0, (24 + 4 * i) << 20) for i, k in enumerate(("enwik", "logs", "dickens", "enwik"))]
    members = write_members(datas, level=6)
    small = [d[:3 << 20] for d in datas]
    want_small = [O.deflate(d, 6) for d in small]
    results, errors = {}, []

    def reader(i):
        try:
            results[("r", i)] = _drain(GZipInputStream(io.BytesIO(members[i])), 1 << 20)
        except Exception as e:                             # noqa: BLE001
            errors.append(("r", i, repr(e)))

    def writer(i):
        try:
            bio = io.BytesIO()
            g = GZipOutputStream(bio, 1 << 20)
            g.IsStreamOwner = False
            g.ModifiedTime = 0
            for rep in range(3):                           # (three flushes: three device runs per thread, interleaved with the others')
                a = small[i][rep << 20:(rep + 1) << 20]
                g.Write(a)
            g.Finish()
            results[("w", i)] = bio.getvalue()
        except Exception as e:                             # noqa: BLE001
            errors.append(("w", i, repr(e)))
    th = [threading.Thread(target=f, args=(i,)) for i in range(4) for f in (reader, writer)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for i in range(4):
        assert results[("r", i)] == datas[i].tobytes(), i
        assert results[("w", i)][10:-8] == want_small[i], i
