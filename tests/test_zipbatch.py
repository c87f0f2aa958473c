"""ZIP batch path: container written as ZipOutputStream.PutNextPassthroughEntry + Finish would
(S/Zip/ZipOutputStream.cs:313-346,885-908; S/Zip/ZipFormat.cs:50,251,389), validated with an independent reader (Python's
zipfile), and the read-side twin of ZipFile.GetInputStream (S/Zip/ZipFile.cs:953-994) on archives from both writers.
The container logic is host code (as in the reference); here the payloads come from the oracle, the -m gpu tests below run
the same functions with the device codec."""
import datetime
import io
import struct
import zipfile
import zlib

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C
from sharpziplib_amd import zipbatch as Z


def oracle_deflate(level):
    return lambda bufs: [(O.deflate(b, level), O.crc32(b)) for b in bufs]


def zlib_inflate(payloads, sizes):
    out = []
    for p, n in zip(payloads, sizes):
        zo = zlib.decompressobj(-15)
        d = zo.decompress(p)
        out.append((d, zlib.crc32(d), 0, len(p) - len(zo.unused_data)))
    return out


def _entries(n, seed=1):
    rng = np.random.default_rng(seed)
    ents = []
    for i in range(n):
        ln = int(rng.integers(0, 70000))
        kind = ("dickens", "logs", "enwik")[i % 3]
        ents.append(("dir%d/e%06d.txt" % (i % 7, i), C.generate(kind, 100 + i, 0, ln)))
    ents.append(("unicode/éè中.txt", C.generate("dickens", 5, 0, 1234)))
    ents.append(("empty.bin", np.zeros(0, np.uint8)))
    return ents


def test_archive_is_valid_and_fields_follow_the_reference():
    ents = _entries(40)
    when = datetime.datetime(2024, 2, 29, 13, 37, 58)
    z = Z.write_zip(ents, level=6, when=when, comment=b"made on the device", compress=oracle_deflate(6))
    zf = zipfile.ZipFile(io.BytesIO(z))
    assert zf.testzip() is None and zf.comment == b"made on the device"
    infos = zf.infolist()
    assert [i.filename for i in infos] == [n for n, _ in ents]
    for info, (name, data) in zip(infos, ents):
        assert zf.read(info) == data.tobytes()
        assert info.CRC == zlib.crc32(data.tobytes()) and info.file_size == data.size
        assert info.compress_type == zipfile.ZIP_DEFLATED and info.flag_bits == 0x0800   # UnicodeText only: no descriptor (sizes known)
        assert info.extract_version == 20 and info.create_version == 51 and info.create_system == 0
        assert info.date_time == (2024, 2, 29, 13, 37, 58)
        # the payload is exactly the reference Deflater's bytes for that entry
        sig, ver, flags, method, t, crc, csize, size, nlen, xlen = struct.unpack_from("<IHHHIIIIHH", z, info.header_offset)
        assert sig == Z.LOCSIG and (csize, size, crc) == (info.compress_size, data.size, info.CRC) and xlen == 0
        p0 = info.header_offset + 30 + nlen
        assert z[p0:p0 + csize] == O.deflate(data, 6)


def test_zip64_end_records_from_65535_entries():
    n = 66000
    ents = [("e%05d" % i, np.frombuffer(b"x%d" % i, np.uint8)) for i in range(n)]
    z = Z.write_zip(ents, level=6, when=datetime.datetime(2020, 1, 1), compress=oracle_deflate(6))
    assert z.count(struct.pack("<I", Z.ZIP64_ENDSIG)) >= 1 and z.count(struct.pack("<I", Z.ZIP64_LOCSIG)) >= 1
    zf = zipfile.ZipFile(io.BytesIO(z))
    assert len(zf.infolist()) == n and zf.read("e65999") == b"x65999"
    back = Z.read_zip(z, inflate=zlib_inflate)
    assert len(back) == n and back[12345] == ("e12345", b"x12345")


def test_reader_on_foreign_and_own_archives():
    ents = _entries(25, seed=3)
    bio = io.BytesIO()
    with zipfile.ZipFile(bio, "w", zipfile.ZIP_DEFLATED) as zf:          # foreign writer (zlib encoder), like T/Zip/PassthroughTests.cs
        for name, data in ents:
            zf.writestr(name, data.tobytes())
        zf.writestr("stored.bin", b"raw bytes", compress_type=zipfile.ZIP_STORED)
    back = Z.read_zip(bio.getvalue(), inflate=zlib_inflate)
    assert back[:-1] == [(n, d.tobytes()) for n, d in ents] and back[-1] == ("stored.bin", b"raw bytes")
    own = Z.write_zip(ents, compress=oracle_deflate(9))
    assert Z.read_zip(own, inflate=zlib_inflate) == [(n, d.tobytes()) for n, d in ents]
    bad = bytearray(own); bad[60] ^= 0x55                                   # corrupt a payload byte: CRC / inflate must notice
    with pytest.raises(Exception):
        Z.read_zip(bytes(bad), inflate=zlib_inflate)


@pytest.mark.gpu
def test_device_archive_end_to_end():
    ents = _entries(300, seed=9)
    z = Z.write_zip(ents, level=6, when=datetime.datetime(2025, 6, 1, 8, 0, 0))
    zf = zipfile.ZipFile(io.BytesIO(z))
    assert zf.testzip() is None
    for info, (name, data) in zip(zf.infolist(), ents):
        assert info.filename == name and info.CRC == zlib.crc32(data.tobytes())
        sig, _, _, _, _, _, csize, _, nlen, xlen = struct.unpack_from("<IHHHIIIIHH", z, info.header_offset)
        p0 = info.header_offset + 30 + nlen + xlen
        assert z[p0:p0 + csize] == O.deflate(data, 6), name               # bit-identical to the reference Deflater per entry
    assert z == Z.write_zip(ents, level=6, when=datetime.datetime(2025, 6, 1, 8, 0, 0), compress=oracle_deflate(6))
    back = Z.read_zip(z)                                                    # device batch inflate + device CRC check
    assert back == [(n, d.tobytes()) for n, d in ents]


@pytest.mark.gpu
def test_device_reads_foreign_archive():
    ents = _entries(60, seed=4)
    bio = io.BytesIO()
    with zipfile.ZipFile(bio, "w", zipfile.ZIP_DEFLATED, compresslevel=9) as zf:
        for name, data in ents:
            zf.writestr(name, data.tobytes())
    assert Z.read_zip(bio.getvalue()) == [(n, d.tobytes()) for n, d in ents]
