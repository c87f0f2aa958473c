"""CPU tests of the boundary: the C-ABI library builds, loads and exports every symbol that
include/szl.h declares (no compute calls — there is no GPU here), and fails loudly without a device."""
import os
import re

import pytest

from sharpziplib_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "szl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(szl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("szl_deflater_create", "szl_deflater_set_input", "szl_deflater_deflate", "szl_deflater_finish",
                 "szl_deflater_flush", "szl_deflater_reset", "szl_deflater_needs_input", "szl_deflater_is_finished",
                 "szl_deflater_total_in", "szl_deflater_total_out", "szl_deflater_adler", "szl_inflater_create",
                 "szl_inflater_inflate", "szl_inflater_remaining_input", "szl_deflate_batch_device", "szl_inflate_batch_device",
                 "szl_crc32", "szl_adler32"):
        assert must in syms


def test_library_builds_and_exports_every_declared_symbol():
    so = _lib.build()
    assert os.path.exists(so)
    import ctypes
    L = ctypes.CDLL(so)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing


def test_product_never_touches_the_oracle():
    """The shipped package must not import/link anything under oracle/ (no CPU fallback)."""
    pkg = os.path.join(ROOT, "sharpziplib_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_ffi" not in txt and "libszl_oracle" not in txt and "szl_oracle.h" not in txt, os.path.join(dp, f)
                assert "szo_" not in txt and "szm_" not in txt, os.path.join(dp, f)


def test_fails_loudly_without_a_device():
    L = _lib.lib()
    if L.szl_device_count() > 0:
        pytest.skip("a GPU is present")
    assert not L.szl_engine_create()
    assert b"gfx950" in L.szl_last_error()
    assert not L.szl_deflater_create(6, 1)
    from sharpziplib_amd.deflater import Deflater, SharpZipBaseException
    with pytest.raises(SharpZipBaseException):
        Deflater(6, True)


def test_argument_errors_match_reference():
    from sharpziplib_amd.deflater import Deflater
    with pytest.raises(ValueError):
        Deflater(10)          # ArgumentOutOfRangeException C/Deflater.cs:184-187
    with pytest.raises(ValueError):
        Deflater(-2)
    assert _lib.lib().szl_deflate_bound(0) >= 16


def test_long_host_copies_on_several_cores_are_exact():
    """SetInput's copy into pinned memory runs on SZL_COPY_THREADS cores when the piece is long (host_copy, szl_engine.hip): every byte, in
    order, whatever the length, the alignment and the number of threads (host code only: no device needed)"""
    import numpy as np
    L = _lib.lib()
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (40 << 20) + 12345, dtype=np.uint8)
    try:
        for threads in (1, 2, 4, 7, 16):
            L.szl_debug_set(b"SZL_COPY_THREADS", threads)
            for n, so, do in ((0, 0, 0), (1, 3, 5), (4095, 1, 2), (4 << 20, 0, 0), ((4 << 20) + 1, 7, 9), ((8 << 20) - 1, 4096, 1), (33554432 + 4097, 11, 3), (src.size - 64, 5, 0)):
                dst = np.full(n + do + 64, 0xA5, np.uint8)
                assert L.szl_debug_host_copy(dst[do:].ctypes.data, src[so:].ctypes.data, n) == 0
                assert np.array_equal(dst[do:do + n], src[so:so + n]) and (dst[:do] == 0xA5).all() and (dst[do + n:] == 0xA5).all(), (threads, n)
    finally:
        L.szl_debug_set(b"SZL_COPY_THREADS", -(2 ** 31))
