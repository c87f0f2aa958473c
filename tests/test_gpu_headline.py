"""GPU parity at BASELINE.json's headline sizes (run with -m gpu on an MI355X).

The small-stream suite (test_gpu_deflate.py) never compares more than 3 MiB with the oracle; the workloads bench.py times
are 1 GiB streams that cross code paths small inputs never reach (65 536 stage-B tiles, 262 144 stage-C ranges, the
pilot's 1/256 tile sample, arena offsets above 2^32).  Here the device output of exactly those workloads is compared
byte for byte with the oracle run on the same seeded input, and with the sha256 frozen in tests/golden/headline_golden.json
(tests/golden/make_headline.py).  Bit-exact: integer/byte work, tolerance 0.
Reference: C/DeflaterEngine.cs:741-855 (DeflateSlow), C/DeflaterHuffman.cs:788-857 (FlushBlock).
"""
import ctypes
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "headline_golden.json")))["cases"]


@pytest.fixture(scope="module")
def eng():
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    assert _lib.lib().szl_device_count() > 0, "no gfx950 device: the HIP path cannot run (and there is no fallback)"
    e = Engine()
    yield e
    e.close()


def _case_input(name):
    c = GOLD[name]
    return c, C.generate(c["kind"], c["seed"], c["offset"], c["n"])


def _one_stream(eng, data, level, flags_crc=True):
    """One stream through szl_deflate_batch_host, exactly the call bench.py's step makes (plus the PCIe copies)."""
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    arr, in_total, out_total = Engine.layout([data.size])
    hout = np.zeros(out_total + 8, np.uint8)
    flags = _lib.F_NOWRAP | (_lib.F_CRC32 if flags_crc else 0)
    _lib.check(eng._L.szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, 1, level, 0, flags), "batch")   # (the library the engine was made by)
    assert arr[0].status == 0
    return hout[:arr[0].out_len], int(arr[0].crc32)


def test_config2_1gib_level6_bit_exact(eng):
    """configs[1] — the stream bench.py times on rank 0: 1 GiB enwik-style text, seed 0xE9, level 6, CRC-32 on device."""
    c, data = _case_input("cfg2_enwik_1g_l6")
    got, crc = _one_stream(eng, data, c["level"])
    assert got.size == c["out_len"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == c["out_sha256"], "device output differs from the frozen oracle output"
    assert crc == c["crc32"] == zlib.crc32(data.tobytes())
    ref = O.deflate(data, c["level"])            # ~40 s of one host core: the oracle itself, not only its frozen hash
    assert got.tobytes() == ref
    tm = eng.timing()
    assert tm["tokens"] > 0 and tm["blocks"] >= tm["tokens"] // 16384


def test_config5_512mib_level9_logs_bit_exact(eng, lab_eng):
    """configs[4] at 1/8 size: level 9 (max_chain 4096) on repetitive logs — the product library's one form of stage B, and the
    on-demand form (laboratory library) on the same stream: results must not depend on the form."""
    c, data = _case_input("cfg5_logs_512m_l9")
    got, crc = _one_stream(eng, data, c["level"])
    assert not eng.debug_match_mode()             # (the product library holds the full search only)
    assert got.size == c["out_len"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == c["out_sha256"]
    assert crc == c["crc32"]
    assert got.tobytes() == O.deflate(data, c["level"])
    lab_eng.debug_match_mode(1)
    got2, _ = _one_stream(lab_eng, data, c["level"])
    assert lab_eng.debug_match_mode()
    assert hashlib.sha256(got2.tobytes()).hexdigest() == c["out_sha256"]


def test_config3_4096_entries_bit_exact(eng):
    """configs[2] at 4096 entries: many small independent streams in one batch (the ZipOutputStream shape), every entry
    compared with the oracle (C/Deflater.cs Reset()+SetInput+Finish per entry, S/Zip/ZipOutputStream.cs:494)."""
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    c, data = _case_input("cfg3_4096x64k_l6")
    entry = c["entry"]
    N = c["n"] // entry
    arr, in_total, out_total = Engine.layout([entry] * N)
    hout = np.zeros(out_total + 8, np.uint8)
    _lib.check(_lib.lib().szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, N, c["level"], 0,
                                                 _lib.F_NOWRAP | _lib.F_CRC32), "batch")
    h, crcs, total = hashlib.sha256(), hashlib.sha256(), 0
    for i in range(N):
        s = arr[i]
        assert s.status == 0
        h.update(hout[s.out_off:s.out_off + s.out_len].tobytes())
        crcs.update(int(s.crc32).to_bytes(4, "little"))
        total += int(s.out_len)
    assert total == c["out_len"] and h.hexdigest() == c["out_sha256"] and crcs.hexdigest() == c["crc_sha256"]
    for i in list(range(0, N, 97)) + [N - 1]:      # direct oracle comparison on a spread of entries
        s = arr[i]
        d = data[i * entry:(i + 1) * entry]
        assert hout[s.out_off:s.out_off + s.out_len].tobytes() == O.deflate(d, c["level"]), i
        assert s.crc32 == O.crc32(d)


def test_arena_offsets_above_4gib(eng):
    """A stream whose input AND output regions start above 2^32 in the arenas (64-bit offsets in every per-position array),
    next to one at offset 0; device-resident call like bench.py's."""
    import hip_ffi as H
    from sharpziplib_amd import _lib
    c, data = _case_input("off4g_enwik_8m_l6")
    small = C.generate("dickens", 77, 0, 300000)
    L = _lib.lib()
    big_in = (1 << 32) + (1 << 28) + 12345            # not even 4-aligned: input offsets are arbitrary
    big_out = (1 << 32) + (1 << 29)
    n = data.size
    cap0 = (int(L.szl_deflate_bound(small.size)) + 3) & ~3
    cap1 = (int(L.szl_deflate_bound(n)) + 3) & ~3
    d_in = H.DevBuf(big_in + n + 64)
    d_out = H.DevBuf(big_out + cap1 + 64)
    try:
        d_in.upload(0, small)
        d_in.upload(big_in, data)
        # canaries around the output regions: the engine may only touch [out_off, out_off + out_cap)
        d_out.fill(cap0, 4096, 0xA5)
        d_out.fill(big_out - 4096, 4096, 0x5A)
        arr = (_lib.Stream * 2)()
        arr[0].in_off, arr[0].in_len, arr[0].out_off, arr[0].out_cap = 0, small.size, 0, cap0
        arr[1].in_off, arr[1].in_len, arr[1].out_off, arr[1].out_cap = big_in, n, big_out, cap1
        H.hip().hipDeviceSynchronize()
        eng.deflate_device(d_in.addr, d_out.addr, arr, level=c["level"], flags=_lib.F_NOWRAP | _lib.F_CRC32)
        assert arr[0].status == 0 and arr[1].status == 0
        got1 = d_out.download(big_out, int(arr[1].out_len)).tobytes()
        assert len(got1) == c["out_len"] and hashlib.sha256(got1).hexdigest() == c["out_sha256"]
        assert int(arr[1].crc32) == c["crc32"]
        got0 = d_out.download(0, int(arr[0].out_len)).tobytes()
        assert got0 == O.deflate(small, c["level"])
        assert bool((d_out.download(cap0, 4096) == 0xA5).all()) and bool((d_out.download(big_out - 4096, 4096) == 0x5A).all()), \
            "the engine wrote outside the streams' output regions"
    finally:
        d_in.free()
        d_out.free()


# ---- the other BASELINE configs at (or near) full size, in the driver-run suite -------------------------------------------------
def test_config1_dickens_64mib_level6_bit_exact(eng):
    """configs[0]: raw Deflater level 6 on 64 MiB of prose — against the oracle itself and its frozen hash."""
    c, data = _case_input("cfg1_dickens_64m_l6")
    got, crc = _one_stream(eng, data, c["level"])
    assert got.size == c["out_len"] and hashlib.sha256(got.tobytes()).hexdigest() == c["out_sha256"] and crc == c["crc32"]
    assert got.tobytes() == O.deflate(data, c["level"])


def test_config5_2gib_level9_logs_through_the_default_window_pipeline(eng):
    """configs[4] at half size: 2 GiB is where the library switches to the window pipeline by itself (SZL_WINDOW_FROM_KIB default;
    side arrays for one 256 MiB window instead of ~20 bytes per input byte) — sha256 of the oracle's output, frozen."""
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    c, data = _case_input("cfg5_logs_2g_l9")
    fresh = Engine()                                   # (its side arrays are what this call needs: the module's engine keeps the 1 GiB one-piece arrays)
    try:
        got, crc = _one_stream(fresh, data, c["level"])
        assert got.size == c["out_len"]
        assert hashlib.sha256(got.tobytes()).hexdigest() == c["out_sha256"], "2 GiB, level 9, window pipeline: output differs from the oracle's"
        assert crc == c["crc32"]
        ws = int(_lib.lib().szl_engine_debug_workspace(fresh._h))
        assert 0 < ws < 6 * data.size, ws             # (the one-piece run would hold ~20 bytes per input byte)
    finally:
        fresh.close()


def _device_roundtrip(eng, data, sizes, level=6, check_consumed=True):
    """deflate `sizes`-long streams cut from data on the device, inflate them back on the device with the library's default knobs,
    compare on the host; returns (deflate device ms, inflate device ms, compressed bytes)"""
    import hip_ffi as H
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    n = int(sum(sizes))
    st, in_total, out_total = Engine.layout(list(sizes))
    d_in, d_out, d_back = H.DevBuf(n + 64), H.DevBuf(out_total + 64), H.DevBuf(n + 4 * len(sizes) + 64)
    try:
        d_in.upload(0, data[:n])
        eng.deflate_device(d_in.addr, d_out.addr, st, level=level, flags=_lib.F_NOWRAP | _lib.F_CRC32)
        t_def = eng.timing()["total_ms"]
        assert all(s.status == 0 for s in st)
        ist = (_lib.Stream * len(sizes))()
        oo = 0
        for i, (s, cap) in enumerate(zip(st, sizes)):
            ist[i].in_off, ist[i].in_len, ist[i].out_off, ist[i].out_cap = s.out_off, s.out_len, oo, cap
            oo += (cap + 3) & ~3
        eng.inflate_device(d_out.addr, d_back.addr, ist, flags=_lib.F_NOWRAP | _lib.F_CRC32)
        t_inf = eng.timing()["inflate_ms"]
        for s, t in zip(ist, st):
            assert s.status == 0 and s.out_len == t.in_len and s.crc32 == t.crc32
            if check_consumed:
                assert s.in_consumed == t.out_len
        oo = 0
        for i, cap in enumerate(sizes):
            got = d_back.download(oo, cap)
            assert np.array_equal(got, data[st[i].in_off:st[i].in_off + cap]), i
            oo += (cap + 3) & ~3
        return t_def, t_inf, sum(int(s.out_len) for s in st)
    finally:
        d_in.free(); d_out.free(); d_back.free()


def test_config4i_inflate_the_1gib_member(eng):
    """configs[3](i): InflaterInputStream over ONE 1 GiB member, default knobs (chunk-parallel decoder): the member is the device
    Deflater's own output for the headline stream (its sha256 is pinned above); bytes == corpus, in_consumed exact, CRC-32."""
    c, data = _case_input("cfg2_enwik_1g_l6")
    t_def, t_inf, comp = _device_roundtrip(eng, data, [data.size], level=6)
    assert comp == c["out_len"]
    assert eng._L.szl_engine_debug_par_jobs(eng._h) > 1      # the member was decoded by many wavefronts, not one


def test_config4ii_inflate_512_members_of_4mib(eng):
    """configs[3](ii): a .gz of many members — 2 GiB as 512 x 4 MiB members in ONE call."""
    data = C.generate("enwik", 0xEA, 0, 512 * (4 << 20))
    _device_roundtrip(eng, data, [4 << 20] * 512, level=6)


def test_config3_50000_entries_round_trip(eng):
    """configs[2] at half its entry count (the other half is the second GPU's shard): 50000 x 64 KiB entries in one call, every
    entry back through the device Inflater; a spread of entries against the oracle."""
    import hip_ffi as H  # noqa: F401
    n3, esz = 50000, 65536
    data = C.generate("enwik", 0x21B0, 0, n3 * esz)
    _device_roundtrip(eng, data, [esz] * n3, level=6)
    from sharpziplib_amd.batch import Engine
    res = eng.deflate([data[i * esz:(i + 1) * esz] for i in (0, 17, 25000, n3 - 1)], level=6, crc32=True)
    for i, r in zip((0, 17, 25000, n3 - 1), res):
        d = data[i * esz:(i + 1) * esz]
        assert r.data == O.deflate(d, 6) and r.crc32 == O.crc32(d)
