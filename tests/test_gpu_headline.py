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
    _lib.check(_lib.lib().szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, 1, level, 0, flags), "batch")
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


def test_config5_512mib_level9_logs_bit_exact(eng):
    """configs[4] at 1/8 size: level 9 (max_chain 4096) on repetitive logs, with the stage-B form the pilot selects."""
    c, data = _case_input("cfg5_logs_512m_l9")
    eng.debug_match_mode(-1)                      # library default: the pilot decides
    got, crc = _one_stream(eng, data, c["level"])
    used_on_demand = eng.debug_match_mode()
    assert got.size == c["out_len"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == c["out_sha256"]
    assert crc == c["crc32"]
    assert got.tobytes() == O.deflate(data, c["level"])
    # and the other form of stage B on the same stream (results must not depend on the form)
    eng.debug_match_mode(0 if used_on_demand else 1)
    try:
        got2, _ = _one_stream(eng, data, c["level"])
    finally:
        eng.debug_match_mode(-1)
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
