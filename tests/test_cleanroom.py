"""De-risking the oracle's common-mode error.

The reference ships no expected compressed bytes and cannot run here, so "bit-exact" is defined by oracle/*.c, a C
restatement written by the same hand as the device code.  tests/cleanroom/sharpzip_py.py is a second restatement written
only from the C# sources (statement by statement, in Python, without looking at the oracle's C).  Two independent readings
of DeflaterEngine.cs / DeflaterHuffman.cs / PendingBuffer.cs / Deflater.cs that agree bit for bit on every input below are
much stronger evidence than either alone: a misreading would have to be made twice, identically, in two languages.
"""
import json
import os

import numpy as np
import pytest

import oracle_ffi as O
from cleanroom import sharpzip_py as S
from golden.make_golden import make_input
from sharpziplib_amd import corpus as C

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "deflate_golden.json")))["cases"]


def test_reference_fixture_and_tiny_vectors():
    # the one deflate payload the reference itself ships (T/Zip/ZipCorruptionHandling.cs:52-54) + SURVEY App. C.8
    assert S.deflate(b"testfile contents\n", 6).hex() == "2b492d2e49cbcc495548cecf2b49cd2b29e60200"
    for data, hexout in [(b"", "0300"), (b"x", "ab0000"), (b"Hello", "f348cdc9c90700"), (b"a" * 32, "4b240000"), (b"abc" * 10, "4b4c4ac68300")]:
        assert S.deflate(data, 6).hex() == hexout
    assert S.deflate(b"Hello", 0).hex() == "010500faff48656c6c6f"
    assert S.deflate(b"Hello", 6, flush=True).hex() == "f248cdc9c9070820c000"   # Write+Flush+Finish, the reference test's own call pattern


@pytest.mark.parametrize("case", GOLD, ids=lambda c: "%s-L%d" % (c["name"], c["level"]))
def test_golden_inputs(case):
    """Every golden input (first 256 KiB: pure Python) at its level: clean-room bytes == oracle bytes."""
    data = make_input(tuple(case["spec"]))[:262144]
    assert S.deflate(data.tobytes(), case["level"]) == O.deflate(data, case["level"])


def _structured(rng, n):
    """A random concatenation of segment kinds that push the encoder into different regimes."""
    parts = []
    total = 0
    while total < n:
        kind = int(rng.integers(0, 8))
        ln = int(rng.integers(1, max(2, min(n - total, 1 + n // 3)) + 1))
        if kind == 0:
            seg = rng.integers(0, 256, ln, dtype=np.uint8)
        elif kind == 1:
            p = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)
            seg = np.resize(p, ln)
        elif kind == 2:
            off = int(rng.integers(0, 1 << 20))
            seg = C.generate("dickens", 7, off, ln)
        elif kind == 3:
            seg = np.zeros(ln, np.uint8)
        elif kind == 4:
            seg = rng.integers(0, 4, ln, dtype=np.uint8) + 65
        elif kind == 5:
            off = int(rng.integers(0, 1 << 20))
            seg = C.generate("logs", 9, off, ln)
        elif kind == 6 and parts:      # copy of earlier material at some distance
            src = np.concatenate(parts)
            s0 = int(rng.integers(0, src.size))
            seg = np.resize(src[s0:s0 + ln], ln)
        else:
            seg = np.full(ln, int(rng.integers(0, 256)), np.uint8)
        parts.append(seg.astype(np.uint8))
        total += ln
    return np.concatenate(parts)[:n] if parts else np.zeros(0, np.uint8)


def test_thousand_structured_random_inputs():
    rng = np.random.default_rng(0x5EED)
    for i in range(1000):
        n = int(rng.integers(0, 6000)) if i % 10 else int(rng.integers(0, 40))
        data = _structured(rng, n)
        level = int(rng.integers(0, 10))
        nowrap = bool(rng.integers(0, 2))
        strategy = int(rng.choice([0, 0, 0, 1, 2]))
        flush = bool(rng.integers(0, 4) == 0)
        got = S.deflate(data.tobytes(), level, nowrap=nowrap, strategy=strategy, flush=flush)
        want = O.deflate(data, level, nowrap=nowrap, strategy=strategy, flush=flush)
        assert got == want, (i, n, level, nowrap, strategy, flush)


@pytest.mark.parametrize("level", [0, 1, 4, 5, 6, 9])
def test_window_slides_and_block_edges(level):
    """Inputs past 64 KiB (SlideWindow, :441-462), with more than 16384 tokens per stream (block cuts), in every
    compression function: stored, fast, slow."""
    rng = np.random.default_rng(level)
    for data in (_structured(rng, 150000), C.generate("enwik", 0xE9, 0, 200000), C.zeros(140000), C.period10(90000),
                 np.concatenate([C.random_bytes(70000, seed=3), C.generate("dickens", 2, 0, 70000)])):
        assert S.deflate(data.tobytes(), level) == O.deflate(data, level)


def test_chunked_input_and_small_output_buffers():
    """The DeflaterOutputStream drive pattern: many SetInput chunks, 512-byte Deflate() calls, Flush() in the middle."""
    rng = np.random.default_rng(11)
    data = _structured(rng, 90000)
    for level in (1, 6):
        want, _, _ = O.stream_deflate(data, level=level, nowrap=True, chunk=4096, out_chunk=512)
        assert S.deflate(data.tobytes(), level, chunk=4096, out_chunk=512) == want
    # level 0 depends on the chunking (DeflateStored, :614-649): same chunks, same bytes
    want, _, _ = O.stream_deflate(data, level=0, nowrap=True, chunk=7000, out_chunk=512)
    assert S.deflate(data.tobytes(), 0, chunk=7000, out_chunk=512) == want


def test_preset_dictionary_and_set_level():
    d = C.generate("dickens", 5, 0, 40000)
    dic = C.generate("dickens", 5, 40000, 9000)
    o = O.Deflater(6, False)
    o.set_dictionary(dic)
    o.set_input(d)
    o.finish()
    want = bytearray()
    while not o.finished:
        want += o.deflate(4096)
    assert S.deflate(d.tobytes(), 6, nowrap=False, dictionary=dic.tobytes()) == bytes(want)
    # SetLevel between inputs (C/DeflaterEngine.cs:304-361): slow -> fast -> stored -> slow
    s = S.Deflater(6, True)
    o = O.Deflater(6, True)
    outs, outo = bytearray(), bytearray()
    buf = bytearray(4096)
    pieces = [d[:15000], d[15000:22000], d[22000:30000], d[30000:]]
    for piece, lv in zip(pieces, (1, 0, 9, 9)):
        s.SetInput(piece.tobytes()); o.set_input(piece)
        while not s.IsNeedingInput:
            k = s.Deflate(buf, 0, len(buf))
            if k <= 0:
                break
            outs += buf[:k]
        while not o.needs_input:
            b = o.deflate(4096)
            if not b:
                break
            outo += b
        s.SetLevel(lv); o.set_level(lv)
    s.Finish(); o.finish()
    while not s.IsFinished:
        k = s.Deflate(buf, 0, len(buf)); outs += buf[:k]
    while not o.finished:
        outo += o.deflate(4096)
    assert bytes(outs) == bytes(outo)
    import zlib
    assert zlib.decompress(bytes(outs), -15) == d.tobytes()


@pytest.mark.parametrize("seed", list(range(1, 17)))
def test_set_level_across_compression_functions_random_patterns(seed):
    """The call patterns tests/test_gpu_setlevel.py checks the device against the oracle on — SetLevel to another compression
    function (stored / fast / slow) with bytes pending, any number of times, SetStrategy, Flush in between — checked here between
    the oracle and the independent Python transliteration (C/DeflaterEngine.cs:304-361 read twice, by two restatements)."""
    rng = np.random.default_rng(1000 + seed)
    data = np.concatenate([C.generate("enwik", seed, 0, 45000), C.generate("logs", seed + 1, 0, 35000)])
    level = int(rng.choice([0, 2, 6]))
    nowrap = seed % 2 == 1
    s, o = S.Deflater(level, nowrap), O.Deflater(level, nowrap)
    outs, outo = bytearray(), bytearray()
    buf = bytearray(int(rng.choice([64, 700, 4096])))

    def drain():
        while True:
            k = s.Deflate(buf, 0, len(buf))
            if k <= 0:
                break
            outs.extend(buf[:k])
        while True:
            b = o.deflate(len(buf))
            if not b:
                break
            outo.extend(b)

    pos = 0
    while pos < data.size:
        n = int(rng.choice([1, 3, 100, 261, 262, 263, 700, 5000, 20000]))
        c = data[pos:pos + n]
        pos += c.size
        s.SetInput(c.tobytes()); o.set_input(c)
        if rng.random() < 0.2:                      # before the engine has seen the chunk
            lv = int(rng.choice([0, 1, 3, 4, 5, 6, 9]))
            s.SetLevel(lv); o.set_level(lv)
        drain()
        r = rng.random()
        if r < 0.5:
            lv = int(rng.choice([0, 1, 3, 4, 5, 6, 9]))
            s.SetLevel(lv); o.set_level(lv)
            if rng.random() < 0.5:
                drain()
        elif r < 0.6:
            st = int(rng.choice([0, 1, 2]))
            s.SetStrategy(st); o.set_strategy(st)
        if rng.random() < 0.15:
            s.Flush(); o.flush()
            drain()
        assert bytes(outs) == bytes(outo), (seed, pos)
    s.Finish(); o.finish()
    while not s.IsFinished:
        k = s.Deflate(buf, 0, len(buf)); outs.extend(buf[:k])
    while not o.finished:
        outo.extend(o.deflate(4096))
    assert bytes(outs) == bytes(outo)
    import zlib
    assert zlib.decompress(bytes(outs), -15 if nowrap else 15) == data.tobytes()


@pytest.mark.parametrize("seed", list(range(21, 61)))
def test_random_api_programs_with_reset_and_dictionary(seed):
    """Whole-object programs — preset dictionary (zlib framing only, C/Deflater.cs:559), Reset() between streams (S/Zip/ZipOutputStream.cs:494),
    SetLevel / SetStrategy / Flush in mid-stream, output buffers of a few bytes to a few KiB — oracle vs the Python transliteration:
    bytes, TotalIn, TotalOut and Adler after every stream."""
    rng = np.random.default_rng(5000 + seed)
    pool = np.concatenate([C.generate("dickens", seed, 0, 30000), C.generate("logs", seed, 0, 30000), C.generate("enwik", seed, 0, 30000)])
    nowrap = bool(rng.integers(0, 2))
    level = int(rng.choice([0, 1, 4, 5, 6, 8, 9]))
    s, o = S.Deflater(level, nowrap), O.Deflater(level, nowrap)
    for stream in range(3):
        outs, outo = bytearray(), bytearray()
        buf = bytearray(int(rng.choice([7, 100, 1000, 5000])))

        def drain():
            while True:
                k = s.Deflate(buf, 0, len(buf))
                if k <= 0:
                    break
                outs.extend(buf[:k])
            while True:
                b = o.deflate(len(buf))
                if not b:
                    break
                outo.extend(b)

        if not nowrap and rng.random() < 0.5:
            a = int(rng.integers(0, pool.size - 9000))
            dic = pool[a:a + int(rng.choice([2, 3, 500, 8000]))]
            s.SetDictionary(dic.tobytes()); o.set_dictionary(dic)
        total = int(rng.choice([0, 1, 500, 20000, 70000]))
        a0 = int(rng.integers(0, pool.size - total)) if total < pool.size else 0
        data = pool[a0:a0 + total]
        pos = 0
        while pos < data.size:
            n = int(rng.choice([1, 2, 3, 260, 262, 5000, 33000]))
            c = data[pos:pos + n]
            pos += c.size
            s.SetInput(c.tobytes()); o.set_input(c)
            drain()
            r = rng.random()
            if r < 0.3:
                lv = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9]))
                s.SetLevel(lv); o.set_level(lv)
            elif r < 0.4:
                st = int(rng.choice([0, 1, 2]))
                s.SetStrategy(st); o.set_strategy(st)
            elif r < 0.5:
                s.Flush(); o.flush()
                drain()
        s.Finish(); o.finish()
        while not s.IsFinished:
            k = s.Deflate(buf, 0, len(buf)); outs.extend(buf[:k])
        while not o.finished:
            outo.extend(o.deflate(len(buf)))
        assert bytes(outs) == bytes(outo), (seed, stream)
        assert s.TotalIn == o.total_in == data.size and s.TotalOut == o.total_out == len(outs)
        if not nowrap:
            assert s.Adler == o.adler
        s.Reset(); o.reset()
