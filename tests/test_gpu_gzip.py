"""GZip framing around the device codec (run with -m gpu): S/GZip/GzipOutputStream.cs and S/GZip/GzipInputStream.cs mirrored
in sharpziplib_amd/gzipstream.py.  Expected bytes are assembled from the reference's header/footer layout (:315-379) around
the ORACLE's raw deflate stream; Python's gzip module is the independent reader/writer."""
import gzip
import io
import zlib

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


def _hdr(mtime, name=None):
    h = bytes([0x1F, 0x8B, 8, 8 if name else 0]) + int(mtime).to_bytes(4, "little") + bytes([0, 255])
    return h + (name.encode("latin-1") + b"\0" if name else b"")


@pytest.mark.parametrize("level", [1, 6, 9])
def test_output_stream_with_file_name_is_the_references_bytes(level):
    from sharpziplib_amd.gzipstream import GZipOutputStream
    data = C.generate("enwik", 3, 0, 300000)
    bio = io.BytesIO()
    g = GZipOutputStream(bio)
    g.IsStreamOwner = False
    g.SetLevel(level)
    g.FileName = "some/dir/h\xe9llo.txt"                  # CleanFilename keeps what follows the last '/', Latin-1 on the wire
    g.ModifiedTime = 1234567890
    for a in range(0, data.size, 70001):
        g.Write(data[a:a + 70001])
    g.Finish()
    got = bio.getvalue()
    want = _hdr(1234567890, "h\xe9llo.txt") + O.deflate(data, level) + zlib.crc32(data.tobytes()).to_bytes(4, "little") + (data.size).to_bytes(4, "little")
    assert got == want
    assert gzip.decompress(got) == data.tobytes()
    f = gzip.GzipFile(fileobj=io.BytesIO(got)); f.read()
    assert f.mtime == 1234567890


def test_output_stream_empty_member_and_no_name():
    from sharpziplib_amd.gzipstream import GZipOutputStream
    bio = io.BytesIO()
    g = GZipOutputStream(bio); g.IsStreamOwner = False
    g.ModifiedTime = 7
    g.Finish()                                             # header is still written (:263-266)
    assert bio.getvalue() == _hdr(7) + O.deflate(np.zeros(0, np.uint8), 6) + bytes(8)
    assert gzip.decompress(bio.getvalue()) == b""


def _py_member(data, name=None, extra=None, comment=None, hcrc=None, mtime=99):
    flags = (8 if name else 0) | (4 if extra is not None else 0) | (16 if comment else 0) | (2 if hcrc else 0)
    h = bytes([0x1F, 0x8B, 8, flags]) + int(mtime).to_bytes(4, "little") + bytes([2, 3])
    if extra is not None:
        h += len(extra).to_bytes(2, "little") + extra
    if name:
        h += name + b"\0"
    if comment:
        h += comment + b"\0"
    if hcrc == "reference":                                # the reference compares (first << 8 | second) with crc & 0xffff (:283-291)
        c = zlib.crc32(h) & 0xFFFF
        h += bytes([c >> 8, c & 0xFF])
    elif hcrc == "rfc":
        h += (zlib.crc32(h) & 0xFFFF).to_bytes(2, "little")
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    return h + co.compress(data) + co.flush() + zlib.crc32(data).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little")


@pytest.mark.parametrize("read_ahead,device_crc", [(0, False), (0, True), (None, True), (None, False)])
def test_input_stream_header_fields_members_and_garbage(read_ahead, device_crc):
    import functools
    from sharpziplib_amd import gzipstream
    from sharpziplib_amd.gzipstream import GZipException
    # (0, False) is the reference's own arrangement (4 KiB pieces, CRC-32 on the host); (None, True) the device-aware default
    GZipInputStream = functools.partial(gzipstream.GZipInputStream, readAhead=read_ahead, deviceCrc=device_crc)
    a = C.generate("dickens", 1, 0, 200000).tobytes()
    b = C.generate("logs", 2, 0, 90000).tobytes()
    m1 = _py_member(a, name=b"first.txt", extra=b"ab\x04\x00XYZW", comment=b"a comment", hcrc="reference")
    m2 = _py_member(b"", name=b"empty")
    m3 = _py_member(b)
    g = GZipInputStream(io.BytesIO(m1 + m2 + m3 + b"trailing garbage that is not a member"), 1024)
    buf = np.zeros(50000, np.uint8)
    out = bytearray()
    names = []
    while True:
        n = g.Read(buf, 0, buf.size)
        if n <= 0:
            break
        out += buf[:n].tobytes()
        names.append(g.GetFilename())
    assert bytes(out) == a + b
    assert names[0] == "first.txt" and names[-1] is None
    # python's own writer (FNAME set from the file name), read back
    bio = io.BytesIO()
    with gzip.GzipFile(filename="x/y/name.bin", mode="wb", fileobj=bio, mtime=5) as f:
        f.write(a)
    g = GZipInputStream(io.BytesIO(bio.getvalue()))
    assert g.read_all() == a and g.GetFilename() == "name.bin"
    # errors of ReadHeader / ReadFooter
    bad = bytearray(m3); bad[-5] ^= 1                      # CRC-32 of the trailer
    with pytest.raises(GZipException):
        GZipInputStream(io.BytesIO(bytes(bad))).read_all()
    bad = bytearray(m3); bad[-1] ^= 1                      # ISIZE
    with pytest.raises(GZipException):
        GZipInputStream(io.BytesIO(bytes(bad))).read_all()
    with pytest.raises(GZipException):
        GZipInputStream(io.BytesIO(b"\x1f\x8c" + m3[2:])).read_all()
    rfc = _py_member(a[:1000], name=b"n", hcrc="rfc")      # RFC byte order of the header CRC: the reference rejects it unless both bytes agree
    c = zlib.crc32(rfc[:rfc.index(b"n\0") + 2]) & 0xFFFF
    if (c >> 8) != (c & 0xFF):
        with pytest.raises(GZipException):
            GZipInputStream(io.BytesIO(rfc)).read_all()


def test_batch_members_with_names_and_batch_reader():
    from sharpziplib_amd.gzipstream import write_members, read_members
    from sharpziplib_amd.batch import Engine
    eng = Engine()
    datas = [C.generate(("enwik", "logs", "dickens")[i % 3], 40 + i, 0, 1000 + 7919 * i) for i in range(40)] + [np.zeros(0, np.uint8)]
    names = ["d%d/file%d.txt" % (i, i) if i % 4 else None for i in range(len(datas))]
    members = write_members(datas, level=6, names=names, mtimes=[i for i in range(len(datas))], engine=eng)
    for i, (m, d) in enumerate(zip(members, datas)):
        assert gzip.decompress(m) == d.tobytes()
        assert m[:10] == _hdr(i, "x" if names[i] else None)[:10]
        assert m == _hdr(i, ("file%d.txt" % i) if names[i] else None) + O.deflate(d, 6) + zlib.crc32(d.tobytes()).to_bytes(4, "little") + d.size.to_bytes(4, "little")
    back = read_members(members, engine=eng)
    for (got, name), d, nm in zip(back, datas, names):
        assert got == d.tobytes() and name == (nm.split("/")[-1] if nm else None)
    eng.close()


def test_long_member_takes_the_chunk_parallel_decoder():
    from sharpziplib_amd import _lib
    from sharpziplib_amd.gzipstream import write_members, read_members
    from sharpziplib_amd.batch import Engine
    eng = Engine()
    d = C.generate("enwik", 77, 0, 24 << 20)
    (m,) = write_members([d], level=6, names=["big.xml"], engine=eng)
    ((got, name),) = read_members([m], engine=eng)
    assert got == d.tobytes() and name == "big.xml"
    assert int(_lib.lib().szl_engine_debug_par_jobs(eng._h)) >= 16
    eng.close()
