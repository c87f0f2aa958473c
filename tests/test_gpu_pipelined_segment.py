"""The streaming Deflater's first segment parsed while the caller still writes (run with -m gpu; csrc/szl_api.hip szl_deflater::Pipe):
parts of the pending bytes go through stages A-C on a worker thread as they arrive, Flush() / Finish() parse the rest and run stage D
over all tokens.  Levels 5-9 are chunk-independent (SURVEY 0.6, C/DeflaterEngine.cs:104-137), so the bytes must be the oracle's
whatever the part length and the write sizes are — and whatever interrupts the parts (SetLevel, Reset, a failure)."""
import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import _lib, corpus as C

pytestmark = pytest.mark.gpu
FORGET = -2147483648


def _knobs(**kw):
    for k, v in kw.items():
        _lib.lib().szl_debug_set(k.encode(), int(v))


def _drain(d, buf):
    out = bytearray()
    while True:
        n = d.Deflate(buf)
        if n <= 0:
            break
        out += buf[:n].tobytes()
    return bytes(out)


@pytest.mark.parametrize("kind,level,total,write,part_kib", [
    ("enwik", 6, 9 << 20, 700 * 1024, 1024),          # nine parts of 1 MiB, writes that do not line up with them
    ("logs", 9, 6 << 20, 1 << 20, 512),
    ("enwik", 5, 5 << 20, 123457, 256),               # many small parts
    ("mixed", 7, 6 << 20, 2 << 20, 1024),             # stretches of zeros / periodic bytes: ranges that never merge inside a part
    ("enwik", 6, 3 << 20, 1 << 20, 16384),            # never enough bytes for a part (the first is a quarter of SZL_PIPE_PART_KIB): the plain path
])
def test_pipelined_first_segment_equals_the_oracle(kind, level, total, write, part_kib):
    from sharpziplib_amd.deflater import Deflater
    data = C.mixed(total) if kind == "mixed" else C.generate(kind, 77, 0, total)
    want = O.deflate(data, level)
    _knobs(SZL_PIPE_PART_KIB=part_kib, SZL_UP_SLAB_KIB=64)
    try:
        d = Deflater(level, True)
        d.EnableCrc32()
        buf = np.zeros(1 << 20, np.uint8)
        got = bytearray()
        for o in range(0, total, write):
            d.SetInput(data[o:o + write])
            got += _drain(d, buf)
        d.Finish()
        while not d.IsFinished:
            got += _drain(d, buf)
        assert bytes(got) == want
        assert d.Crc32 == O.crc32(data) and d.TotalIn == total and d.TotalOut == len(want)
        parts = _lib.lib().szl_deflater_debug_pipe_parts(d._h)
        assert (parts >= 2) == (total >= 3 * part_kib * 1024), parts      # the parts really ran (or, too short a stream, did not)
    finally:
        _knobs(SZL_PIPE_PART_KIB=FORGET, SZL_UP_SLAB_KIB=FORGET)


def test_flush_after_parts_then_more_input_and_zlib_framing():
    """Flush() closes the pipelined segment with the sync padding; the stream goes on (history from the pipelined bytes) and ends with the Adler-32"""
    from sharpziplib_amd.deflater import Deflater
    a, b = C.generate("enwik", 5, 0, 5 << 20), C.generate("logs", 6, 0, 300000)
    o = O.Deflater(6, False)
    ref = bytearray()
    o.set_input(a); o.flush()
    while True:
        x = o.deflate(1 << 20)
        if not x:
            break
        ref += x
    o.set_input(b); o.finish()
    while not o.finished:
        ref += o.deflate(1 << 20)
    _knobs(SZL_PIPE_PART_KIB=512, SZL_UP_SLAB_KIB=64)
    try:
        d = Deflater(6, False)
        buf = np.zeros(1 << 20, np.uint8)
        got = bytearray()
        for off in range(0, a.size, 1 << 20):
            d.SetInput(a[off:off + (1 << 20)])
            got += _drain(d, buf)
        d.Flush(); got += _drain(d, buf)
        assert _lib.lib().szl_deflater_debug_pipe_parts(d._h) >= 2
        d.SetInput(b); d.Finish()
        while not d.IsFinished:
            got += _drain(d, buf)
        assert bytes(got) == bytes(ref)
    finally:
        _knobs(SZL_PIPE_PART_KIB=FORGET, SZL_UP_SLAB_KIB=FORGET)


def test_setlevel_and_reset_drop_the_parts():
    from sharpziplib_amd.deflater import Deflater
    data = C.generate("enwik", 9, 0, 6 << 20)
    _knobs(SZL_PIPE_PART_KIB=512, SZL_UP_SLAB_KIB=64)
    try:
        d, o = Deflater(6, True), O.Deflater(6, True)
        d.CallerDrains()
        buf = np.zeros(1 << 20, np.uint8)
        got, ref = bytearray(), bytearray()
        for off in range(0, 4 << 20, 1 << 20):
            d.SetInput(data[off:off + (1 << 20)]); o.set_input(data[off:off + (1 << 20)])
            got += _drain(d, buf)
            while True:
                x = o.deflate(1 << 20)
                if not x:
                    break
                ref += x
        d.SetLevel(9); o.set_level(9)                      # parts parsed at level 6 are of no use now: the segment runs in one piece, with the switch inside
        d.SetInput(data[4 << 20:]); o.set_input(data[4 << 20:])
        d.Finish(); o.finish()
        while not d.IsFinished:
            got += _drain(d, buf)
        while not o.finished:
            ref += o.deflate(1 << 20)
        assert bytes(got) == bytes(ref) and _lib.lib().szl_deflater_debug_pipe_parts(d._h) == 0
        d.Reset(); o.reset()                               # ... and the object is as good as new
        for off in range(0, 3 << 20, 1 << 20):
            d.SetInput(data[off:off + (1 << 20)]); o.set_input(data[off:off + (1 << 20)])
            _drain(d, buf)
            while o.deflate(1 << 20):
                pass
        d.Reset(); o.reset()                               # a stream abandoned in the middle of its parts (what its bit buffer held goes into the next one's first byte, C/PendingBuffer.cs:43)
        d.SetInput(data[:2 << 20]); d.Finish(); o.set_input(data[:2 << 20]); o.finish()
        got, ref = bytearray(), bytearray()
        while not d.IsFinished:
            got += _drain(d, buf)
        while not o.finished:
            ref += o.deflate(1 << 20)
        assert bytes(got) == bytes(ref)
    finally:
        _knobs(SZL_PIPE_PART_KIB=FORGET, SZL_UP_SLAB_KIB=FORGET)
