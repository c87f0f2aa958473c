"""The write path that drops in (run with -m gpu): szl_deflater_deflate_view — Deflate() without its copies — and the device-aware
DeflaterOutputStream over it (sharpziplib_amd/streams.py; C#: dotnet/DeflaterOutputStream.Device.cs).  The reference's stream hands the
compressed bytes on buffer_.Length at a time — 512 by default, 4096 under GZipOutputStream (CS/DeflaterOutputStream.cs:26-29, :100-118,
:242-272; S/GZip/GzipOutputStream.cs:72) — i.e. one P/Invoke and one base-stream Write per half kilobyte; the device-aware one writes what
the Deflater has, out of its pinned queue, in one Write.  The bytes and every property must be what the Deflate() loop gives."""
import io
import zlib

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


def _odrain(o, size=4096):
    out = bytearray()
    while True:
        b = o.deflate(size)
        if not b:
            return bytes(out)
        out += b


def _views(d):
    out = bytearray()
    while True:
        v = d.DeflateView()
        if v is None:
            break
        out += v                                            # (copied before the next call on the object)
    return bytes(out)


@pytest.mark.parametrize("level", [0, 1, 4, 6, 9])
@pytest.mark.parametrize("nowrap", [True, False])
def test_view_hands_out_what_deflate_hands_out(level, nowrap):
    from sharpziplib_amd.deflater import Deflater
    rng = np.random.default_rng(level * 2 + nowrap)
    parts = [C.generate("enwik", 5, 0, 70001), C.generate("logs", 6, 0, 33333), rng.integers(0, 256, 5000, dtype=np.uint8), C.generate("enwik", 7, 0, 200000)]
    d, o = Deflater(level, nowrap), O.Deflater(level, nowrap)
    got, ref = bytearray(), bytearray()
    for i, p in enumerate(parts):
        d.SetInput(p); o.set_input(p)
        got += _views(d); ref += _odrain(o)                  # (before a flush the device hands out nothing; the reference may: its full blocks)
        if i == 1:
            d.Flush(); o.flush()
            got += _views(d); ref += _odrain(o)
            assert bytes(got) == bytes(ref)                  # a sync flush: everything so far, byte aligned
            assert d.TotalOut == len(got) and d.IsNeedingInput
    d.Finish(); o.finish()
    got += _views(d); ref += _odrain(o, 1 << 16)
    assert bytes(got) == bytes(ref)
    assert d.IsFinished and d.TotalOut == len(got) == o.total_out and d.TotalIn == sum(p.size for p in parts)
    assert d.DeflateView() is None                           # finished: nothing more, no error
    if not nowrap:
        assert zlib.decompress(bytes(got)) == b"".join(p.tobytes() for p in parts)


def test_view_and_deflate_interleave_and_the_view_stays_readable():
    from sharpziplib_amd.deflater import Deflater
    data = C.generate("enwik", 9, 0, 1 << 20)
    want = O.deflate(data, 6, nowrap=True)
    d = Deflater(6, True)
    d.SetInput(data); d.Finish()
    buf = np.zeros(1000, np.uint8)
    k = d.Deflate(buf)                                       # the first kilobyte through the copying call
    assert k == 1000 and buf.tobytes() == want[:1000] and not d.IsFinished
    v = d.DeflateView()                                      # ... the rest in place
    assert v is not None and len(v) == len(want) - 1000 and d.IsFinished and d.TotalOut == len(want)
    assert d.IsNeedingInput in (True, False) and d.TotalIn == data.size   # (property reads do not disturb the view)
    assert bytes(v) == want[1000:]
    d.Reset()                                                # the queue is reused from here on
    d.SetInput(data[:5000]); d.Finish()
    assert _views(d) == O.deflate(data[:5000], 6, nowrap=True)


def test_preset_dictionary_header_comes_through_the_view():
    from sharpziplib_amd.deflater import Deflater
    dic = C.generate("enwik", 3, 0, 4000)
    data = C.generate("enwik", 3, 2000, 60000)
    d, o = Deflater(6, False), O.Deflater(6, False)
    d.SetDictionary(dic); o.set_dictionary(dic)
    d.SetInput(data); o.set_input(data)
    d.Finish(); o.finish()
    assert _views(d) == _odrain(o, 8192)


class _CountingSink(io.BytesIO):
    def __init__(self):
        super().__init__()
        self.writes = []

    def write(self, b):
        self.writes.append(len(b))
        return super().write(bytes(b))


@pytest.mark.parametrize("bufsize", [512, 4096, 1 << 20])
def test_stream_writes_the_queue_in_one_piece_whatever_its_buffer(bufsize):
    from sharpziplib_amd.deflater import Deflater
    from sharpziplib_amd.streams import DeflaterOutputStream
    data = C.generate("enwik", 11, 0, 3 << 20)
    sink = _CountingSink()
    s = DeflaterOutputStream(sink, Deflater(6, True), bufsize)
    s.IsStreamOwner = False
    for o in range(0, data.size, 100000):
        s.Write(data[o:o + 100000])
    s.Flush()
    n_flush = len(sink.writes)
    s.Write(data[:12345])
    s.Finish()
    want = O.Deflater(6, True)
    ref = bytearray()
    for o in range(0, data.size, 100000):
        want.set_input(data[o:o + 100000])
        ref += _odrain(want, 1 << 20)
    want.flush(); ref += _odrain(want, 1 << 20)
    want.set_input(data[:12345]); want.finish(); ref += _odrain(want, 1 << 20)
    assert sink.getvalue() == bytes(ref)
    assert n_flush == 1 and len(sink.writes) == 2            # one Write per flush, not one per `bufsize` bytes
    assert zlib.decompress(sink.getvalue(), -15) == data.tobytes() + data[:12345].tobytes()


def test_a_crypto_transform_still_sees_every_byte_in_the_streams_own_buffer():
    from sharpziplib_amd.deflater import Deflater
    from sharpziplib_amd.streams import DeflaterOutputStream

    class Xor:
        def __init__(self):
            self.blocks = []

        def TransformBlock(self, src, so, n, dst, do):
            self.blocks.append(n)
            dst[do:do + n] = src[so:so + n] ^ 0x5A
            return n

    data = C.generate("logs", 12, 0, 300000)
    sink = _CountingSink()
    s = DeflaterOutputStream(sink, Deflater(6, True), 4096)
    s.IsStreamOwner = False
    x = Xor()
    s.cryptoTransform_ = x
    s.Write(data); s.Finish()
    plain = O.deflate(data, 6, nowrap=True)
    assert bytes(b ^ 0x5A for b in sink.getvalue()) == plain
    assert max(x.blocks) <= 4096 and sum(x.blocks) == len(plain) and sink.writes == x.blocks   # the reference's block structure (:256)


def test_gzip_output_stream_default_constructor_takes_the_short_way():
    from sharpziplib_amd.gzipstream import GZipOutputStream
    data = C.generate("enwik", 13, 0, 8 << 20)
    sink = _CountingSink()
    g = GZipOutputStream(sink)                               # size 4096, S/GZip/GzipOutputStream.cs:72
    g.IsStreamOwner = False
    g.ModifiedTime = 0
    for o in range(0, data.size, 1 << 20):
        g.Write(data[o:o + (1 << 20)])
    g.Finish()
    gz = sink.getvalue()
    assert zlib.decompress(gz, 31) == data.tobytes()
    assert gz[10:-8] == O.deflate(data, 6, nowrap=True)
    assert len(sink.writes) == 3                             # header, body, trailer — the reference: 10 + ~700 + 8 writes of <= 4096 bytes


@pytest.mark.parametrize("threads", [1, 3, 4, 7])
def test_long_set_input_pieces_arrive_whole(threads):
    """SetInput copies into pinned memory — on several cores when the piece is long (host_copy, SZL_COPY_THREADS): every byte, in order"""
    from sharpziplib_amd import _lib
    from sharpziplib_amd.deflater import Deflater
    L = _lib.lib()
    rng = np.random.default_rng(threads)
    data = np.concatenate([C.generate("enwik", 21, 0, (29 << 20) + 4099), rng.integers(0, 256, (9 << 20) + 1, dtype=np.uint8)])
    L.szl_debug_set(b"SZL_COPY_THREADS", threads)
    try:
        d = Deflater(6, True)
        d.EnableCrc32()
        pos = 0
        for k in ((8 << 20) + 1, (13 << 20) + 4097, 100, (16 << 20) - 5, data.size):
            piece = data[pos:pos + k]
            pos += piece.size
            d.SetInput(piece)
            assert d.DeflateView() is None
            if pos >= data.size:
                break
        d.Finish()
        out = _views(d)
    finally:
        L.szl_debug_set(b"SZL_COPY_THREADS", -(2 ** 31))
    assert zlib.decompress(out, -15) == data.tobytes() and d.Crc32 == zlib.crc32(data.tobytes()) and d.TotalIn == data.size


def test_engines_of_streaming_objects_are_pooled_and_change_hands_cleanly():
    """A streaming object's engine (work space included) goes back to a pool when the object dies and is the next object's: streams of other
    levels, framings, directions and sizes through a sequence of short-lived objects come out as from fresh ones; SZL_ENGINE_POOL=0 and
    szl_multi_release() give the memory back"""
    from sharpziplib_amd import _lib
    from sharpziplib_amd.deflater import Deflater
    from sharpziplib_amd.inflater import Inflater
    L = _lib.lib()
    big = C.generate("enwik", 31, 0, 6 << 20)
    small = C.generate("logs", 32, 0, 70000)
    for rep in range(2):
        for level, nowrap, data in ((6, True, big), (1, False, small), (9, True, small), (0, False, small), (6, False, big[:300000]), (4, True, big[:500000])):
            d = Deflater(level, nowrap)                      # (takes the engine the previous object left)
            d.SetInput(data); d.Finish()
            out = _views(d)
            del d
            assert out == O.deflate(data, level, nowrap=nowrap), (rep, level, nowrap)
            inf = Inflater(nowrap)
            inf.SetInput(np.frombuffer(out, np.uint8))
            back = bytearray()
            buf = np.zeros(1 << 20, np.uint8)
            while not inf.IsFinished:
                k = inf.Inflate(buf)
                assert k > 0 or inf.IsFinished
                back += buf[:k].tobytes()
            del inf
            assert bytes(back) == data.tobytes()
        if rep == 0:
            assert L.szl_multi_release() == 0                # the pool is emptied: the second round starts from nothing
            L.szl_debug_set(b"SZL_ENGINE_POOL", 0)           # ... and keeps nothing
    L.szl_debug_set(b"SZL_ENGINE_POOL", -(2 ** 31))
