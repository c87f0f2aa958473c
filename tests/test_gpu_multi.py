"""Several devices behind the C ABI (szl_deflate_batch_multi_host / szl_inflate_batch_multi_host).  The GPU box
of the test run has ONE MI355X, so the device list names it several times: every group still gets its own host thread,
engine and staging — the code path of an 8-GPU node with the ordinals replaced.  Results must equal the single-device call's
and the oracle's for every stream."""
import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


def _bufs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = ("dickens", "logs", "enwik")[i % 3]
        out.append(C.generate(kind, 500 + i, 0, int(rng.integers(0, 200000))))
    return out


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 8])
def test_deflate_multi_equals_oracle(devices):
    from sharpziplib_amd.batch import deflate_multi
    bufs = _bufs(37, 7) + [C.generate("enwik", 9, 0, 3 << 20)]           # one big stream skews the byte balance
    res = deflate_multi(bufs, devices, level=6, crc32=True)
    for b, r in zip(bufs, res):
        assert r.status == 0 and r.data == O.deflate(b, 6) and r.crc32 == O.crc32(b)


def test_fewer_streams_than_devices_and_empty_streams():
    from sharpziplib_amd.batch import deflate_multi
    bufs = [C.generate("dickens", 3, 0, 5000), np.zeros(0, np.uint8)]
    res = deflate_multi(bufs, [0] * 8, level=9)
    assert [r.data for r in res] == [O.deflate(b, 9) for b in bufs]


def test_inflate_multi_roundtrip():
    from sharpziplib_amd.batch import deflate_multi, inflate_multi
    bufs = _bufs(50, 11)
    comp = [r.data for r in deflate_multi(bufs, [0, 0, 0, 0], level=6)]
    back = inflate_multi(comp, [b.size for b in bufs], [0, 0, 0], crc32=True)
    for b, c, (r, consumed) in zip(bufs, comp, back):
        assert r.status == 0 and r.data == b.tobytes() and consumed == len(c) and r.crc32 == O.crc32(b)


def test_bad_device_ordinal_is_an_argument_error():
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import deflate_multi
    with pytest.raises(_lib.SzlError):
        deflate_multi([b"abc"], [0, 99])
