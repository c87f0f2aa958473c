"""The forms of the full stage-B search must all produce the reference's match tables, i.e. the oracle's bytes:
  * k_match4 with its own tile length (the product; the tile is as long as LDS allows, longer than the tiles of the on-demand form),
and, in the LABORATORY library only (csrc/libszl_amd_lab.so — measured alternatives that lost; the product library does not contain them):
  * chain compression (SZL_MATCH_KERNEL=3: k_links4t + k_match6 — four-byte sub-chains, hop counts charged to max_chain),
  * the ring-fed engine (SZL_MATCH_KERNEL=4: k_match8 — positions and links in a ring, staged chunk by chunk while walks run),
  * the bucket-order search (SZL_MATCH_KERNEL=5: k_match5 — every tile's window sorted by (hash, position), lockstep walks over the
    sorted array, no prev[] links; round 3).
Reference: FindLongestMatch, C/DeflaterEngine.cs:474-612."""
import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu

KNOBS = [b"SZL9_FORM", b"SZL_MATCH_KERNEL", b"SZL_STRIPE_MIN", b"SZL_STRIPE_KIB", b"SZL_TILE_LEN", b"SZL_WINDOW_KIB", b"SZL_WINDOW_FROM_KIB"]


@pytest.fixture()
def knobs(monkeypatch):
    """Every object of this module's tests is created on the laboratory library (swapped in for the product one)."""
    from sharpziplib_amd import _lib
    L = _lib.lab_lib()
    monkeypatch.setattr(_lib, "_lib", L)

    def set_(**kv):
        for k, v in kv.items():
            L.szl_debug_set(k.encode(), int(v))
    yield set_
    for k in KNOBS:
        L.szl_debug_set(k, -2147483648)       # forget


def test_the_product_library_holds_one_form_of_the_full_search():
    """SZL_MATCH_KERNEL is a laboratory switch: the shipped library ignores it (and still produces the reference's bytes)."""
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    L = _lib.lib()
    data = [C.generate("enwik", 0xE9, 0, 600000)]
    want = O.deflate(data[0], 6)
    try:
        for k in (2, 3, 4, 5):
            L.szl_debug_set(b"SZL_MATCH_KERNEL", k)
            eng = Engine()
            try:
                eng.debug_match_mode(0)
                assert eng.deflate(data, level=6)[0].data == want, k
            finally:
                eng.close()
    finally:
        L.szl_debug_set(b"SZL_MATCH_KERNEL", -2147483648)


def _streams():
    rng = np.random.default_rng(23)
    return [C.generate("enwik", 0xE9, 0, 1500000), C.generate("logs", 0x106, 0, 900000), C.generate("dickens", 0xD1CE, 0, 700000),
            C.four_symbol(300000), C.period10(200000), C.zeros(150000), C.random_bytes(120000, seed=3), C.mixed(1200000, seed=5),
            np.repeat(rng.integers(0, 256, 4000).astype(np.uint8), rng.integers(1, 90, 4000)), np.zeros(0, np.uint8), C.random_bytes(5, seed=1)]


@pytest.mark.parametrize("form", ["tiles", "chain", "ring", "ring_long_stripes", "bucket", "bucket_short_tiles"])
@pytest.mark.parametrize("level", [5, 6, 9])
def test_full_search_forms_are_bit_exact(knobs, form, level):
    from sharpziplib_amd.batch import Engine
    if form == "chain":
        knobs(SZL_MATCH_KERNEL=3)                       # (levels above 6 — max_chain > 128 — fall back to k_match4 by design)
    elif form == "ring":
        knobs(SZL_MATCH_KERNEL=4, SZL_STRIPE_MIN=1, SZL_STRIPE_KIB=64)
    elif form == "ring_long_stripes":
        knobs(SZL_MATCH_KERNEL=4, SZL_STRIPE_MIN=1, SZL_STRIPE_KIB=4096)
    elif form == "bucket":
        knobs(SZL_MATCH_KERNEL=5)
    elif form == "bucket_short_tiles":
        knobs(SZL_MATCH_KERNEL=5, SZL_TILE_LEN=12288)     # (several tiles per stream: windows with history, window-base changes inside a tile)
    else:
        knobs(SZL_MATCH_KERNEL=2)
    data = _streams()
    eng = Engine()
    try:
        eng.debug_match_mode(0)                         # the full search, whatever the pilot would choose
        res = eng.deflate(data, level=level)
        for d, r in zip(data, res):
            assert r.status == 0 and r.data == O.deflate(d, level), (form, level, d.size)
    finally:
        eng.close()


@pytest.mark.parametrize("form", ["tiles", "chain", "ring", "bucket"])
def test_full_search_forms_in_the_window_pipeline(knobs, form):
    """a long stream goes through stage B window by window: links of a window start at its first position"""
    from sharpziplib_amd.batch import Engine
    knobs(SZL_WINDOW_KIB=192, SZL_WINDOW_FROM_KIB=0)
    if form == "chain":
        knobs(SZL_MATCH_KERNEL=3)
    elif form == "ring":
        knobs(SZL_MATCH_KERNEL=4, SZL_STRIPE_MIN=1, SZL_STRIPE_KIB=128)
    elif form == "bucket":
        knobs(SZL_MATCH_KERNEL=5)
    data = C.generate("enwik", 5, 0, 2200000)
    eng = Engine()
    try:
        eng.debug_match_mode(0)
        for level in (6, 5):
            r = eng.deflate([data], level=level)[0]
            assert r.status == 0 and r.data == O.deflate(data, level), (form, level)
    finally:
        eng.close()


def test_streaming_deflater_through_the_lab_forms(knobs):
    """segments with history (the streaming object's calls): candidates reach into bytes of earlier calls"""
    import io
    from sharpziplib_amd.deflater import Deflater
    from sharpziplib_amd.streams import DeflaterOutputStream
    data = C.generate("enwik", 11, 0, 700000)
    ref, tin, tout = O.stream_deflate(data, 6, True, chunk=150000, flush_every=None)
    for kernel in (3, 4, 5):
        knobs(SZL_MATCH_KERNEL=kernel, SZL_STRIPE_MIN=1, SZL_STRIPE_KIB=64)
        d = Deflater(6, True)
        ms = io.BytesIO()
        s = DeflaterOutputStream(ms, d, 4096)
        s.IsStreamOwner = False
        for pos in range(0, data.size, 150000):
            c = data[pos:pos + 150000]
            s.Write(c, 0, c.size)
        s.Finish()
        assert ms.getvalue() == ref, kernel
        assert d.TotalIn == tin and d.TotalOut == tout


@pytest.mark.parametrize("text_form", [0, 1, 2])
@pytest.mark.parametrize("level", [5, 6, 8, 9])
def test_both_forms_of_the_k_match9_text_are_bit_exact(knobs, text_form, level):
    """k_match9 has two forms of its instruction text (csrc/szl_match9_asm.h, SZL9_V) and three builds: form 0, form 1, and both with every
    tile choosing by the share of short prev[] hops in its window (2).  Here a launch is made to run each, whatever its data: same bytes."""
    from sharpziplib_amd.batch import Engine
    knobs(SZL9_FORM=text_form)
    data = _streams()
    eng = Engine()
    try:
        eng.debug_match_mode(0)
        res = eng.deflate(data, level=level)
        for d, r in zip(data, res):
            assert r.status == 0 and r.data == O.deflate(d, level), (text_form, level, d.size)
    finally:
        eng.close()


@pytest.mark.parametrize("text_form", [1, 2])
def test_the_product_library_runs_the_forms_the_engine_picks(text_form):
    """In the product the ENGINE picks the form per launch from a sample of the call's prev[] hops (Engine::pick_text_form; calls of 8 MiB or
    more).  SZL_TEXT_FORM stands in for the sample here, so that small streams of every class go through forms 1 and 2 of the shipped kernel."""
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    L = _lib.lib()
    data = _streams()
    try:
        L.szl_debug_set(b"SZL_TEXT_FORM", text_form)
        eng = Engine()
        try:
            for level in (6, 9):
                for d, r in zip(data, eng.deflate(data, level=level)):
                    assert r.status == 0 and r.data == O.deflate(d, level), (text_form, level, d.size)
        finally:
            eng.close()
    finally:
        L.szl_debug_set(b"SZL_TEXT_FORM", -2147483648)


def test_the_engine_picks_form_1_for_logs_and_form_0_for_text():
    from sharpziplib_amd.batch import Engine
    eng = Engine()
    try:
        for kind, want_form in (("logs", 1), ("enwik", 0)):
            d = C.generate(kind, 0x106, 0, 24 << 20)
            r = eng.deflate([d], level=6)[0]
            assert r.status == 0 and r.data == O.deflate(d, 6)
            assert eng._L.szl_engine_debug_text_form(eng._h) == want_form, kind
    finally:
        eng.close()
