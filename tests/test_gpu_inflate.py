"""GPU parity tests of the device Inflater (C/Inflater.cs) through the C ABI: byte-identical output,
exact TotalIn / RemainingInput, and the reference's error fixtures."""
import base64
import io
import struct
import zlib

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C
from test_oracle import BAD_CD_GOOD_CD64, ZERO_CODE_LENGTH, _first_local_entry

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from sharpziplib_amd.batch import Engine
    e = Engine()
    yield e
    e.close()


CLASSES = {
    "dickens": lambda: C.generate("dickens", 0xD1CE, 0, 500000), "enwik": lambda: C.generate("enwik", 0xE9, 0, 700000),
    "logs": lambda: C.generate("logs", 0x106, 0, 500000), "random": lambda: C.random_bytes(150000),
    "zeros": lambda: C.zeros(400000), "acgt": lambda: C.four_symbol(250000), "p10": lambda: C.period10(150000),
    "mixed": lambda: C.mixed(800000), "empty": lambda: np.zeros(0, np.uint8), "one": lambda: np.frombuffer(b"x", np.uint8),
}


@pytest.mark.parametrize("name", sorted(CLASSES))
def test_batch_inflate_oracle_streams(eng, name):
    data = CLASSES[name]()
    bufs, caps = [], []
    for lv in (0, 1, 6, 9):
        bufs.append(O.deflate(data, lv)); caps.append(data.size)
    bufs.append(O.deflate(data, 6, flush=True)); caps.append(data.size)      # sync-flush blocks inside
    co = zlib.compressobj(6, zlib.DEFLATED, -15)                               # foreign (zlib) encoder, T/Zip/PassthroughTests.cs
    bufs.append(co.compress(data.tobytes()) + co.flush()); caps.append(data.size)
    res = eng.inflate(bufs, caps, crc32=True)
    for b, (r, consumed) in zip(bufs, res):
        assert r.status == 0, r.status
        assert r.data == data.tobytes()
        assert consumed == len(b)
        assert r.crc32 == O.crc32(data)
        n, out, cons = O.inflate(b, max_out=data.size + 16)
        assert out == r.data and cons == consumed


def test_trailing_bytes_are_not_consumed(eng):
    data = C.generate("dickens", 3, 0, 50000)
    comp = O.deflate(data, 6)
    (r, consumed), = eng.inflate([comp + b"TRAILERBYTES"], [data.size])
    assert r.status == 0 and r.data == data.tobytes() and consumed == len(comp)


def test_zlib_framing_batch(eng):
    data = C.generate("logs", 5, 0, 200000)
    comp = O.deflate(data, 6, nowrap=False)
    (r, consumed), = eng.inflate([comp], [data.size], nowrap=False)
    assert r.status == 0 and r.data == data.tobytes() and consumed == len(comp) and r.adler32 == O.adler32(data)
    bad = bytearray(comp); bad[-1] ^= 1
    (r, _), = eng.inflate([bytes(bad)], [data.size], nowrap=False)
    assert r.status == -20   # Adler chksum doesn't match
    bad = bytearray(comp); bad[1] ^= 1
    (r, _), = eng.inflate([bytes(bad)], [data.size], nowrap=False)
    assert r.status == -16   # Header checksum illegal


def test_reference_fixtures(eng):
    z = base64.b64decode(BAD_CD_GOOD_CD64)
    payload, crc, _, _ = _first_local_entry(z)
    (r, consumed), = eng.inflate([payload[:20]], [64], crc32=True)
    assert r.data == b"testfile contents\n" and r.crc32 == 0xFCEA8F1C == crc and consumed == 20
    z = base64.b64decode(ZERO_CODE_LENGTH)
    payload, _, csize, _ = _first_local_entry(z)
    (r, _), = eng.inflate([payload[:csize]], [4096])
    assert r.status == -23   # "Encountered invalid codelength 0" (T/Zip/ZipCorruptionHandling.cs:12-50)


def test_error_statuses(eng):
    data = C.generate("dickens", 9, 0, 30000)
    comp = O.deflate(data, 6)
    (r, _), = eng.inflate([comp[:len(comp) // 2]], [data.size])
    assert r.status == -25   # Unexpected EOF
    (r, _), = eng.inflate([comp], [data.size - 100])
    assert r.status == -6    # output too small
    (r, _), = eng.inflate([b"\x07\x00"], [64])
    assert r.status == -21   # Unknown block type
    (r, _), = eng.inflate([b"\x01\x05\x00\xfa\xfeHello"], [64])
    assert r.status == -22   # broken uncompressed block


def test_many_streams(eng):
    datas = [C.generate("dickens", 100 + i, 0, 20000 + 997 * i) for i in range(40)] + [C.random_bytes(5000 + i, seed=i) for i in range(8)]
    bufs = [O.deflate(d, 6) for d in datas]
    res = eng.inflate(bufs, [d.size for d in datas], crc32=True)
    for d, b, (r, consumed) in zip(datas, bufs, res):
        assert r.status == 0 and r.data == d.tobytes() and consumed == len(b)


def test_gpu_deflate_then_gpu_inflate(eng):
    data = C.mixed(1200000, seed=21)
    comp = eng.deflate([data], level=6)[0].data
    (r, consumed), = eng.inflate([comp], [data.size])
    assert r.status == 0 and r.data == data.tobytes() and consumed == len(comp)


# ---- streaming object through the InflaterInputStream mirror (T/Base/InflaterDeflaterTests.cs:23-47)
@pytest.mark.parametrize("zlib_framing", [True, False])
@pytest.mark.parametrize("bufsize", [1024, 4096, 65536])
@pytest.mark.parametrize("read_ahead", [0, None])     # 0: the reference's buffer sizes (4 KiB pieces for one wavefront); None: the device-aware default
def test_inflater_input_stream(zlib_framing, bufsize, read_ahead):
    from sharpziplib_amd.inflater import Inflater
    from sharpziplib_amd.streams import InflaterInputStream
    data = C.mixed(600000, seed=33)
    comp = O.deflate(data, 6, nowrap=not zlib_framing, flush=True)
    inf = Inflater(not zlib_framing)
    s = InflaterInputStream(io.BytesIO(comp + b"XYZ" * 5), inf, bufsize, readAhead=read_ahead)
    buf2 = np.zeros(data.size, np.uint8)
    idx, count = 0, buf2.size
    while True:
        n = s.Read(buf2, idx, count)
        if n <= 0:
            break
        idx += n
        count -= n
    assert idx == data.size and buf2.tobytes() == data.tobytes()
    assert inf.IsFinished and inf.TotalOut == data.size and inf.TotalIn == len(comp)
    if zlib_framing:
        assert inf.Adler == O.adler32(data)


def test_inflater_remaining_input_and_reset():
    from sharpziplib_amd.inflater import Inflater
    data = C.generate("dickens", 3, 0, 50000)
    comp = O.deflate(data, 6)
    inf = Inflater(True)
    for _ in range(2):
        inf.SetInput(comp + b"TRAILERBYTES")
        out = np.zeros(data.size + 1000, np.uint8)
        got = 0
        while not inf.IsFinished:
            n = inf.Inflate(out, got, 777)
            got += n
            if n == 0 and inf.IsNeedingInput:
                break
        assert got == data.size and out[:got].tobytes() == data.tobytes()
        assert inf.IsFinished and inf.RemainingInput == 12 and inf.TotalIn == len(comp) and inf.TotalOut == data.size
        inf.Reset()


def test_inflater_error_is_raised():
    from sharpziplib_amd.deflater import SharpZipBaseException
    from sharpziplib_amd.inflater import Inflater
    z = base64.b64decode(ZERO_CODE_LENGTH)
    payload, _, csize, _ = _first_local_entry(z)
    inf = Inflater(True)
    inf.SetInput(payload[:csize])
    with pytest.raises(SharpZipBaseException):
        out = np.zeros(4096, np.uint8)
        for _ in range(100):
            inf.Inflate(out)
            if inf.IsFinished:
                break
