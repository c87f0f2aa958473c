"""ZIP batch path: a whole archive's entries are compressed / decompressed as ONE device batch, and the
container is written / read exactly as the reference's managed code would.

Writer = what `ZipOutputStream` produces when every entry goes through `PutNextPassthroughEntry` (S/Zip/ZipOutputStream.cs:313-346:
the caller supplies Crc, Size, CompressedSize and the already-deflated bytes) followed by `Finish()` (:885-908):
    local header        ZipFormat.WriteLocalHeader          S/Zip/ZipFormat.cs:50-160  (headerInfoAvailable = true, no descriptor)
    deflated payload    the device's raw-deflate bytes, bit-identical to Deflater(level, true) per entry (:494-495 Reset+SetLevel)
    central directory   ZipFormat.WriteEndEntry             S/Zip/ZipFormat.cs:389-514
    end records         WriteEndOfCentralDirectory :251-319 (+ WriteZip64EndOfCentralDirectory :208-236 from 65535 entries / 4 GiB)
Reader = the read-side twin of `ZipFile.GetInputStream` (S/Zip/ZipFile.cs:953-994): central directory -> (offset, csize, size,
crc) per entry -> all payloads inflated in one batch, CRC-32 of every entry checked on the device.

S/ = /root/reference/src/ICSharpCode.SharpZipLib/.  The (de)compression callables default to the device engine; tests may
inject others (the container logic itself is host code, like in the reference).
"""
import datetime
import struct

import numpy as np

LOCSIG, CENSIG, ENDSIG = 0x04034B50, 0x02014B50, 0x06054B50        # S/Zip/ZipConstants.cs:381,443 and EndOfCentralDirectorySignature
ZIP64_ENDSIG, ZIP64_LOCSIG, DESCSIG = 0x06064B50, 0x07064B50, 0x08074B50
VERSION_MADE_BY, VERSION_ZIP64 = 51, 45                              # ZipConstants.cs:267,298
FLAG_UNICODE, FLAG_DESCRIPTOR = 0x0800, 0x0008
DEFLATED, STORED = 8, 0


def dos_time(dt):
    """ZipEntry.DosTime (S/Zip/ZipEntry.cs:578-620)."""
    year, month, day, hour, minute, second = dt.year, dt.month, dt.day, dt.hour, dt.minute, dt.second
    if year < 1980:
        year, month, day, hour, minute, second = 1980, 1, 1, 0, 0, 0
    elif year > 2107:
        year, month, day, hour, minute, second = 2107, 12, 31, 23, 59, 59
    return ((year - 1980) & 0x7F) << 25 | month << 21 | day << 16 | hour << 11 | minute << 5 | second >> 1


class Entry:
    """The fields of ZipEntry this path needs."""
    __slots__ = ("name", "size", "csize", "crc", "offset", "time", "method", "flags", "comment")

    def __init__(self, name, size=-1, csize=-1, crc=-1, time=0, method=DEFLATED, flags=FLAG_UNICODE, comment=""):
        self.name, self.size, self.csize, self.crc, self.time, self.method, self.flags, self.comment = name, size, csize, crc, time, method, flags, comment
        self.offset = 0

    @property
    def central_requires_zip64(self):   # ZipEntry.CentralHeaderRequiresZip64
        return self.size >= 0xFFFFFFFF or self.csize >= 0xFFFFFFFF or self.offset >= 0xFFFFFFFF

    @property
    def local_requires_zip64(self):     # ZipEntry.LocalHeaderRequiresZip64 (no forced Zip64, sizes known)
        return self.size >= 0xFFFFFFFF or self.csize >= 0xFFFFFFFF

    def version(self, central):         # ZipEntry.Version :485-512
        if (self.central_requires_zip64 if central else self.local_requires_zip64) or self.central_requires_zip64:
            return VERSION_ZIP64
        return 20 if self.method == DEFLATED else 10


def _local_header(e):                   # ZipFormat.WriteLocalHeader :50-160, headerInfoAvailable
    name = e.name.encode("utf-8" if e.flags & FLAG_UNICODE else "cp437")
    if len(name) > 0xFFFF:
        raise ValueError("Entry name too long.")
    extra = b""
    if e.local_requires_zip64:
        extra = struct.pack("<HHQQ", 1, 16, e.size, e.csize)
        sizes = struct.pack("<ii", -1, -1)
    else:
        sizes = struct.pack("<II", e.csize, e.size)
    return struct.pack("<IHHHI", LOCSIG, e.version(False), e.flags, e.method, e.time) + struct.pack("<I", e.crc & 0xFFFFFFFF) + sizes + \
        struct.pack("<HH", len(name), len(extra)) + name + extra


def _central_header(e):                 # ZipFormat.WriteEndEntry :389-514
    name = e.name.encode("utf-8" if e.flags & FLAG_UNICODE else "cp437")
    extra = b""
    if e.central_requires_zip64:
        body = b""
        if e.size >= 0xFFFFFFFF:
            body += struct.pack("<Q", e.size)
        if e.csize >= 0xFFFFFFFF:
            body += struct.pack("<Q", e.csize)
        if e.offset >= 0xFFFFFFFF:
            body += struct.pack("<Q", e.offset)
        extra = struct.pack("<HH", 1, len(body)) + body
    comment = e.comment.encode("utf-8") if e.comment else b""
    csize = 0xFFFFFFFF if e.csize >= 0xFFFFFFFF else e.csize
    size = 0xFFFFFFFF if e.size >= 0xFFFFFFFF else e.size
    off = 0xFFFFFFFF if e.offset >= 0xFFFFFFFF else e.offset
    ext_attr = 16 if e.name.endswith("/") else 0
    return struct.pack("<IHHHHIIII", CENSIG, VERSION_MADE_BY, e.version(True), e.flags, e.method, e.time, e.crc & 0xFFFFFFFF, csize, size) + \
        struct.pack("<HHHHHII", len(name), len(extra), len(comment), 0, 0, ext_attr, off) + name + extra + comment


def _end_records(n, size_entries, start, comment=b""):   # WriteEndOfCentralDirectory :251-319
    out = b""
    if n >= 0xFFFF or start >= 0xFFFFFFFF or size_entries >= 0xFFFFFFFF:
        out += struct.pack("<IQHHIIQQQQ", ZIP64_ENDSIG, 44, VERSION_MADE_BY, VERSION_ZIP64, 0, 0, n, n, size_entries, start)   # :208-236
        out += struct.pack("<IIQI", ZIP64_LOCSIG, 0, start + size_entries, 1)
    n16 = 0xFFFF if n >= 0xFFFF else n
    out += struct.pack("<IHHHH", ENDSIG, 0, 0, n16, n16)
    out += struct.pack("<I", 0xFFFFFFFF if size_entries >= 0xFFFFFFFF else size_entries)
    out += struct.pack("<I", 0xFFFFFFFF if start >= 0xFFFFFFFF else start)
    if len(comment) > 0xFFFF:
        raise ValueError("Comment length is larger than 64K")
    return out + struct.pack("<H", len(comment)) + comment


def device_deflate(level=6):
    """Default compressor: every entry = one stream of one szl_deflate_batch call (raw deflate + CRC-32 on the device)."""
    def run(buffers):
        from .batch import Engine
        eng = Engine()
        try:
            res = eng.deflate(buffers, level=level, crc32=True)
        finally:
            eng.close()
        return [(r.data, r.crc32) for r in res]
    return run


def device_inflate():
    def run(payloads, sizes):
        from .batch import Engine
        eng = Engine()
        try:
            res = eng.inflate(payloads, sizes, crc32=True)
        finally:
            eng.close()
        return [(r.data, r.crc32, r.status, consumed) for r, consumed in res]
    return run


def write_zip(entries, level=6, when=None, comment=b"", compress=None):
    """entries: [(name, bytes-like)].  Returns the archive bytes."""
    compress = compress or device_deflate(level)
    when = when or datetime.datetime.now()
    t = dos_time(when)
    bufs = [np.frombuffer(d, np.uint8) if not isinstance(d, np.ndarray) else d for _, d in entries]
    packed = compress(bufs)                                       # ONE device batch for the whole archive
    out = bytearray()
    metas = []
    for (name, _), buf, (payload, crc) in zip(entries, bufs, packed):
        e = Entry(name, size=int(buf.size), csize=len(payload), crc=int(crc), time=t)
        e.offset = len(out)                                       # entry.Offset = offset (S/Zip/ZipOutputStream.cs:455)
        out += _local_header(e)
        out += payload
        metas.append(e)
    start = len(out)
    for e in metas:                                               # Finish(): central directory (:885-908)
        out += _central_header(e)
    out += _end_records(len(metas), len(out) - start, start, comment)
    return bytes(out)


def read_central_directory(buf):
    """[(Entry)] from the end records + central directory (ZipFile.ReadEntries, S/Zip/ZipFile.cs:3449-3680, the part this path needs)."""
    b = bytes(buf)
    pos = b.rfind(struct.pack("<I", ENDSIG))
    if pos < 0:
        raise ValueError("Cannot find central directory")
    _, _, _, n, _, size_entries, start, clen = struct.unpack_from("<IHHHHIIH", b, pos)
    if n == 0xFFFF or start == 0xFFFFFFFF or size_entries == 0xFFFFFFFF:
        lpos = b.rfind(struct.pack("<I", ZIP64_LOCSIG), 0, pos)
        if lpos < 0:
            raise ValueError("Cannot find Zip64 locator")
        _, _, epos, _ = struct.unpack_from("<IIQI", b, lpos)
        sig, _, _, _, _, _, n, _, size_entries, start = struct.unpack_from("<IQHHIIQQQQ", b, epos)
        if sig != ZIP64_ENDSIG:
            raise ValueError("Invalid Zip64 Central directory signature")
    out = []
    p = start
    for _ in range(n):
        sig, made, ver, flags, method, time, crc, csize, size, nlen, xlen, clen2, _, _, _, off = struct.unpack_from("<IHHHHIIIIHHHHHII", b, p)
        if sig != CENSIG:
            raise ValueError("Wrong Central Directory signature")
        name = b[p + 46:p + 46 + nlen].decode("utf-8" if flags & FLAG_UNICODE else "cp437")
        extra = b[p + 46 + nlen:p + 46 + nlen + xlen]
        q = 0
        while q + 4 <= len(extra):                                 # Zip64 extended information (tag 1)
            tag, ln = struct.unpack_from("<HH", extra, q)
            if tag == 1:
                r = q + 4
                if size == 0xFFFFFFFF:
                    size, = struct.unpack_from("<Q", extra, r); r += 8
                if csize == 0xFFFFFFFF:
                    csize, = struct.unpack_from("<Q", extra, r); r += 8
                if off == 0xFFFFFFFF:
                    off, = struct.unpack_from("<Q", extra, r); r += 8
            q += 4 + ln
        e = Entry(name, size=size, csize=csize, crc=crc, time=time, method=method, flags=flags)
        e.offset = off
        out.append(e)
        p += 46 + nlen + xlen + clen2
    return out


def read_zip(buf, inflate=None):
    """[(name, bytes)] — every Deflated entry of the archive inflated in ONE device batch, Stored ones copied; CRC-32 checked."""
    inflate = inflate or device_inflate()
    b = bytes(buf)
    ents = read_central_directory(b)
    payloads, sizes, idx = [], [], []
    out = [None] * len(ents)
    for i, e in enumerate(ents):
        sig, _, _, _, _, _, _, _, nlen, xlen = struct.unpack_from("<IHHHIIIIHH", b, e.offset)   # ZipFile.LocateEntry :1180-1400 (offset of the data)
        if sig != LOCSIG:
            raise ValueError("Wrong local header signature")
        data0 = e.offset + 30 + nlen + xlen
        raw = b[data0:data0 + e.csize]
        if e.method == STORED:
            out[i] = (e.name, raw)
        elif e.method == DEFLATED:
            payloads.append(raw); sizes.append(e.size); idx.append(i)
        else:
            raise NotImplementedError("Compression method not supported")
    if payloads:
        for i, (data, crc, status, consumed) in zip(idx, inflate(payloads, sizes)):
            e = ents[i]
            if status != 0:
                raise ValueError("entry %r: inflate status %d" % (e.name, status))
            if len(data) != e.size or (crc & 0xFFFFFFFF) != (e.crc & 0xFFFFFFFF):   # ZipInputStream checks :709-751
                raise ValueError("entry %r: size / CRC mismatch" % e.name)
            out[i] = (e.name, data)
    return out
