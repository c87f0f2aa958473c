"""The window pipeline of long single streams (Engine::deflate_windowed): stages A-C run window by window on side
arrays sized for one window, the parse of each window starting on the clean iteration the previous one ended on.  The output
must be bit-identical to the oracle's whatever the window length — forced tiny here (64-256 KiB) so that streams of a few MiB
cross dozens of window boundaries, with matches, lazy evaluations, never-merging ranges and stored blocks straddling them.
Reference: the engine handles any length with one 64 KiB window, C/DeflaterEngine.cs:366-400,441-462."""
import numpy as np
import pytest
import zlib

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


@pytest.fixture()
def small_windows():
    from sharpziplib_amd import _lib
    L = _lib.lib()

    def set_kib(k):
        L.szl_debug_set(b"SZL_WINDOW_KIB", k)
        L.szl_debug_set(b"SZL_WINDOW_FROM_KIB", 0)
    yield set_kib
    L.szl_debug_set(b"SZL_WINDOW_KIB", 256 * 1024)
    L.szl_debug_set(b"SZL_WINDOW_FROM_KIB", 2048 * 1024)


CASES = {
    "enwik": lambda: C.generate("enwik", 0xE9, 0, 3 << 20), "dickens": lambda: C.generate("dickens", 0xD1CE, 0, 2 << 20),
    "logs": lambda: C.generate("logs", 0x106, 0, 2 << 20), "zeros": lambda: C.zeros(1500000), "p10": lambda: C.period10(900000),
    "acgt": lambda: C.four_symbol(800000), "random": lambda: C.random_bytes(1 << 20), "mixed": lambda: C.mixed(2500000, seed=5),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("kib,level", [(64, 6), (112, 9), (256, 5)])
def test_windowed_stream_is_bit_exact(small_windows, name, kib, level):
    from sharpziplib_amd.batch import Engine
    small_windows(kib)
    data = CASES[name]()
    eng = Engine()
    try:
        r = eng.deflate([data], level=level, crc32=True, adler32=True)[0]
        ws_small = eng._L.szl_engine_debug_workspace(eng._h)
        assert r.status == 0 and r.data == O.deflate(data, level)
        assert r.crc32 == O.crc32(data) and r.adler32 == O.adler32(data)
        assert zlib.decompress(r.data, -15) == data.tobytes()
    finally:
        eng.close()
    assert ws_small < 16 * data.size          # side arrays follow the window, not the stream (an unwindowed call holds ~19 B per byte)


def test_ranges_whose_paths_hold_more_tokens_than_positions_in_every_window(small_windows):
    """The stage C parity bug of round 6 (tests/test_gpu_deflate.py::test_a_range_whose_path_holds_more_tokens_than_positions) through the
    window pipeline and through the streaming Deflater's parts: a window's ranges (256 positions) start wherever the window in front of it
    ended, so the runs of lazy literals end on every phase of a range in turn."""
    from test_gpu_deflate import _lazy_runs_at_range_ends
    from sharpziplib_amd.batch import Engine
    from sharpziplib_amd.deflater import Deflater
    data = np.concatenate([_lazy_runs_at_range_ends(1 << 20, 256, 5, seed=11), _lazy_runs_at_range_ends(5 << 19, 256, 5, seed=12, drift=1)])
    for kib, level, strategy in ((256, 7, 1), (1024, 6, 0), (336, 9, 1)):
        small_windows(kib)
        eng = Engine()
        try:
            assert eng.deflate([data], level=level, strategy=strategy)[0].data == O.deflate(data, level, strategy=strategy), (kib, level, strategy)
        finally:
            eng.close()
    d = Deflater(7, True)
    d.SetStrategy(1)
    out = bytearray(); buf = np.zeros(1 << 20, np.uint8)
    for o in range(0, data.size, 700001):
        d.SetInput(data[o:o + 700001])
        while not d.IsNeedingInput:
            k = d.Deflate(buf)
            out += buf[:k].tobytes()
            if k == 0: break
    d.Finish()
    while not d.IsFinished:
        k = d.Deflate(buf)
        out += buf[:k].tobytes()
    assert bytes(out) == O.deflate(data, 7, strategy=1)


def test_both_forms_of_stage_b_and_zlib_framing(monkeypatch):
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    L = _lib.lab_lib()                                   # (the on-demand form lives in the laboratory library since round 5)
    monkeypatch.setattr(_lib, "_lib", L)
    L.szl_debug_set(b"SZL_WINDOW_KIB", 96)
    L.szl_debug_set(b"SZL_WINDOW_FROM_KIB", 0)
    data = C.generate("logs", 7, 0, 2500000)
    eng = Engine()
    try:
        for mode in (0, 1):
            eng.debug_match_mode(mode)
            r = eng.deflate([data], level=9, nowrap=False)[0]
            assert r.data == O.deflate(data, 9, nowrap=False), mode
        eng.debug_match_mode(-1)
        r = eng.deflate([data], level=6, sync_flush_before_finish=True)[0]
        assert r.data == O.deflate(data, 6, flush=True)
    finally:
        eng.close()
        L.szl_debug_set(b"SZL_WINDOW_KIB", -2147483648)
        L.szl_debug_set(b"SZL_WINDOW_FROM_KIB", -2147483648)


def test_streaming_deflater_with_history_goes_through_windows(small_windows):
    """A streaming Deflater segment that is longer than the window AND has 64 KiB of history in front of it."""
    from sharpziplib_amd.deflater import Deflater
    small_windows(64)
    data = C.generate("dickens", 33, 0, 1800000)
    d = Deflater(6, True)
    out = bytearray()
    buf = np.zeros(1 << 16, np.uint8)

    def drain():
        while True:
            k = d.Deflate(buf, 0, buf.size)
            if k <= 0:
                break
            out.extend(buf[:k].tobytes())
    d.SetInput(data[:300000]); drain(); d.Flush(); drain()
    d.SetInput(data[300000:]); drain(); d.Finish()
    while not d.IsFinished:
        k = d.Deflate(buf, 0, buf.size)
        out.extend(buf[:k].tobytes())
    od = O.Deflater(6, True)
    want = bytearray()
    od.set_input(data[:300000])
    while not od.needs_input:
        b = od.deflate(65536)
        if not b:
            break
        want += b
    od.flush()
    while True:
        b = od.deflate(65536)
        if not b:
            break
        want += b
    od.set_input(data[300000:])
    while not od.needs_input:
        b = od.deflate(65536)
        if not b:
            break
        want += b
    od.finish()
    while not od.finished:
        want += od.deflate(65536)
    assert bytes(out) == bytes(want)


def test_default_window_bounds_the_workspace_of_a_long_stream():
    """600 MiB at the default window (256 MiB): three windows; the side arrays stay near one window's worth."""
    from sharpziplib_amd.batch import Engine
    data = C.generate("enwik", 0xE9, 0, 600 << 20)
    eng = Engine()
    eng._L.szl_debug_set(b"SZL_WINDOW_FROM_KIB", 0)
    try:
        r = eng.deflate([data], level=6, crc32=True)[0]
        ws = eng._L.szl_engine_debug_workspace(eng._h)
        assert r.status == 0 and r.crc32 == zlib.crc32(data.tobytes())
        assert zlib.decompress(r.data, -15) == data.tobytes()
        ref = O.deflate(data, 6)
        assert r.data == ref
        # ~14 B per byte of ONE window (links 2, match tables 8, speculative tokens 4) + the token stream with its growth slack:
        # 5.6 GiB here and about the same for a stream ten times longer (an unwindowed call would hold 19 B per stream byte = 11 GiB)
        assert ws < (7 << 30)
    finally:
        eng._L.szl_debug_set(b"SZL_WINDOW_FROM_KIB", 2048 * 1024)
        eng.close()
