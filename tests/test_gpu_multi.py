"""Several devices behind the C ABI (szl_deflate_batch_multi_host / szl_inflate_batch_multi_host).  Device lists
are written as SLOT counts: `_devs(n)` names n distinct ordinals when the box has them and wraps around otherwise — on a box
with ONE MI355X every slot is ordinal 0, and each group / unit still gets its own host thread, engine and staging (the code
path of an 8-GPU node with the ordinals replaced; peer copies degenerate to device-to-device copies).  Results must equal the
single-device call's and the oracle's for every stream."""
import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


def _devs(n):
    from sharpziplib_amd import _lib
    have = max(1, int(_lib.lib().szl_device_count()))
    return [i % have for i in range(n)]


def _bufs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = ("dickens", "logs", "enwik")[i % 3]
        out.append(C.generate(kind, 500 + i, 0, int(rng.integers(0, 200000))))
    return out


@pytest.mark.parametrize("ndev", [1, 2, 3, 8])
def test_deflate_multi_equals_oracle(ndev):
    from sharpziplib_amd.batch import deflate_multi
    devices = _devs(ndev)
    bufs = _bufs(37, 7) + [C.generate("enwik", 9, 0, 3 << 20)]           # one big stream skews the byte balance
    res = deflate_multi(bufs, devices, level=6, crc32=True)
    for b, r in zip(bufs, res):
        assert r.status == 0 and r.data == O.deflate(b, 6) and r.crc32 == O.crc32(b)


def test_fewer_streams_than_devices_and_empty_streams():
    from sharpziplib_amd.batch import deflate_multi
    bufs = [C.generate("dickens", 3, 0, 5000), np.zeros(0, np.uint8)]
    res = deflate_multi(bufs, _devs(8), level=9)
    assert [r.data for r in res] == [O.deflate(b, 9) for b in bufs]


def test_inflate_multi_roundtrip():
    from sharpziplib_amd.batch import deflate_multi, inflate_multi
    bufs = _bufs(50, 11)
    comp = [r.data for r in deflate_multi(bufs, _devs(4), level=6)]
    back = inflate_multi(comp, [b.size for b in bufs], _devs(3), crc32=True)
    for b, c, (r, consumed) in zip(bufs, comp, back):
        assert r.status == 0 and r.data == b.tobytes() and consumed == len(c) and r.crc32 == O.crc32(b)


def test_bad_device_ordinal_is_an_argument_error():
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import deflate_multi
    with pytest.raises(_lib.SzlError):
        deflate_multi([b"abc"], [0, 99])


# ---- ONE stream over several engines (stream_multi_run in csrc/szl_api.hip): exact position-range partition -------------------
def _knob(name, value):
    from sharpziplib_amd import _lib
    _lib.lib().szl_debug_set(name.encode(), value)


@pytest.fixture()
def small_parts():
    """parts of a few MiB and windows of 1 MiB, so that a modest stream already has several windows per part"""
    _knob("SZL_PART_MIN_KIB", 1024); _knob("SZL_WINDOW_KIB", 1024); _knob("SZL_PART_WARM_KIB", 64)
    yield
    for k in ("SZL_PART_MIN_KIB", "SZL_WINDOW_KIB", "SZL_PART_WARM_KIB"):
        _knob(k, -2147483648)


@pytest.mark.parametrize("ndev", [2, 3, 5])
@pytest.mark.parametrize("kind,level", [("enwik", 6), ("logs", 9), ("dickens", 5)])
def test_one_stream_over_several_engines_equals_oracle(small_parts, ndev, kind, level):
    from sharpziplib_amd.batch import deflate_multi
    devices = _devs(ndev)
    data = C.generate(kind, 0x5EED, 0, 20 << 20)
    (r,) = deflate_multi([data], devices, level=level, crc32=True)
    assert r.status == 0 and r.crc32 == O.crc32(data)
    assert r.data == O.deflate(data, level)


def test_one_stream_with_zlib_framing_and_strategies(small_parts):
    import zlib
    from sharpziplib_amd.batch import deflate_multi
    data = C.generate("enwik", 77, 0, 12 << 20)
    (r,) = deflate_multi([data], _devs(3), level=6, nowrap=False)
    assert r.status == 0 and zlib.decompress(r.data) == data.tobytes() and r.adler32 == zlib.adler32(data.tobytes())
    assert r.data[2:-4] == O.deflate(data, 6)
    for strategy in (1, 2):
        (r,) = deflate_multi([data], _devs(2), level=6, strategy=strategy)
        assert r.data == O.deflate(data, 6, strategy=strategy)


def test_one_stream_whose_parses_never_resynchronise(small_parts):
    """long runs of one byte: a warm-up from an assumed clean state does not land on the true parse, the hand-over check fails
    and the part is run again from the previous part's exit — same bytes in the end"""
    from sharpziplib_amd.batch import deflate_multi
    data = np.concatenate([np.zeros(7 << 20, np.uint8), C.generate("logs", 5, 0, 3 << 20), np.full(6 << 20, 0x55, np.uint8),
                           C.period10(2 << 20) if hasattr(C, "period10") else np.zeros(2 << 20, np.uint8)])
    (r,) = deflate_multi([data], _devs(4), level=6, crc32=True)
    assert r.status == 0 and r.data == O.deflate(data, 6)


def test_units_are_taken_dynamically(small_parts):
    """more units than engines (SZL_PART_UNITS): an engine takes the next unit when it is free; the tokens are gathered in unit
    order while later units run — same bytes whatever the number of units, two concurrent callers serialise"""
    import threading
    from sharpziplib_amd.batch import deflate_multi
    data = np.concatenate([C.generate("logs", 3, 0, 9 << 20), C.generate("enwik", 4, 0, 9 << 20), np.zeros(3 << 20, np.uint8),
                           C.generate("dickens", 5, 0, 6 << 20)])     # parts of very different cost per byte
    want = O.deflate(data, 6)
    for units in (1, 3, 6):
        _knob("SZL_PART_UNITS", units)
        try:
            (r,) = deflate_multi([data], _devs(3), level=6, crc32=True)
        finally:
            _knob("SZL_PART_UNITS", -2147483648)
        assert r.status == 0 and r.data == want and r.crc32 == O.crc32(data), units
    got = [None, None]

    def call(i):
        got[i] = deflate_multi([data], _devs(2 + i), level=6)[0].data
    th = [threading.Thread(target=call, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got[0] == want and got[1] == want


def test_one_long_stream_default_knobs():
    """library defaults (256 MiB windows, 256 KiB warm-up): 192 MiB over two engines against the single-engine call"""
    import hashlib
    from sharpziplib_amd.batch import Engine, deflate_multi
    data = C.generate("enwik", 0xE9, 0, 192 << 20)
    (r,) = deflate_multi([data], _devs(2), level=6, crc32=True)
    eng = Engine()
    (one,) = eng.deflate([data], level=6, crc32=True)
    eng.close()
    assert r.status == 0 and r.crc32 == one.crc32
    assert hashlib.sha256(r.data).hexdigest() == hashlib.sha256(one.data).hexdigest()


@pytest.mark.parametrize("nslots", [2, 3])
def test_one_stream_resident_on_every_device(nslots):
    """szl_deflate_stream_multi_device: the stream is uploaded once to every device (here: the same device behind every slot) and
    compressed by position-range units with no host buffer in the call — same bytes as one engine."""
    import hip_ffi as H
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine, deflate_stream_multi_device
    devices = _devs(nslots)
    assert len(set(devices)) == 1 or True      # (raw hipMalloc below allocates on the current device: the one-GPU box of this suite)
    n = 160 << 20
    host = C.generate("enwik", 0xE9, 0, n)
    d_ins = []
    for g in devices:
        b = H.DevBuf(n + 64 + 4096)
        b.upload(4096, host)                   # (the arena holds something in front of the stream: in_off is honoured)
        d_ins.append(b)
    streams, _, out_total = Engine.layout([n])
    streams[0].in_off = 4096
    streams[0].out_off = 512
    d_out = H.DevBuf(out_total + 64 + 512)
    d_out.fill(0, out_total + 64 + 512, 0)
    for level in (6, 9):
        deflate_stream_multi_device([b.addr for b in d_ins], d_out.addr, devices, streams, level=level, flags=_lib.F_NOWRAP | _lib.F_CRC32)
        assert streams[0].status == 0
        got = d_out.download(512, int(streams[0].out_len)).tobytes()
        one = Engine().deflate([host], level=level, crc32=True)[0]
        assert got == one.data and streams[0].crc32 == one.crc32
    for b in d_ins:
        b.free()
    d_out.free()


def test_multi_release_frees_the_pooled_engines_and_calls_work_again():
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import deflate_multi
    bufs = _bufs(12, 21)
    a = deflate_multi(bufs, _devs(3), level=6)
    assert _lib.lib().szl_multi_release() == 0
    b = deflate_multi(bufs, _devs(2), level=6)
    assert [r.data for r in a] == [r.data for r in b] == [O.deflate(x, 6) for x in bufs]
    # incompressible data: a token per byte — the gather buffer of the one-stream path starts at text density and has to grow
    rnd = C.random_bytes(96 << 20, seed=4)
    (r,) = deflate_multi([rnd], _devs(2), level=6, crc32=True)
    assert r.status == 0 and r.crc32 == O.crc32(rnd)
    import zlib
    assert zlib.decompress(r.data, -15) == rnd.tobytes()
