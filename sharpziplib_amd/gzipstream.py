"""Host-side mirrors of GZipOutputStream / GZipInputStream (S/GZip/GzipOutputStream.cs, S/GZip/GzipInputStream.cs).

RFC 1952 framing stays on the host exactly where the reference has it — a dozen header bytes, an optional file name, an
8-byte trailer — around the codec and the CRC-32, which run on the device through the C ABI (raw Deflater / Inflater and
szl_crc32).  `write_members` / `read_members` are the batch forms (config 4(ii)): many members through ONE
szl_deflate_batch_host / szl_inflate_batch_host call; a single long member inflates through the chunk-parallel decoder.
"""
import ctypes
import time

import numpy as np

from . import _lib
from .deflater import Deflater, SharpZipBaseException
from .streams import DeflaterOutputStream, InflaterInputStream

ID1, ID2, CM_DEFLATE = 0x1F, 0x8B, 8                     # S/GZip/GZipConstants.cs:14-29
FTEXT, FHCRC, FEXTRA, FNAME, FCOMMENT = 1, 2, 4, 8, 16   # GZipFlags :42-73


class GZipException(SharpZipBaseException):
    pass


def _crc32(value, data):
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    out = ctypes.c_uint32(0)
    _lib.check(_lib.lib().szl_crc32(value, a.ctypes.data if a.size else None, a.size, ctypes.byref(out)), "szl_crc32")
    return int(out.value)


def clean_filename(path):                                 # CleanFilename :381-382
    return path[path.rfind("/") + 1:]


def member_header(mod_time, file_name=None):
    """GetHeader :341-379: ID1 ID2 CM FLG MTIME XFL=0 OS=255 [FNAME 0]."""
    flags = FNAME if file_name is not None else 0
    t = int(mod_time) & 0xFFFFFFFF
    h = bytes([ID1, ID2, CM_DEFLATE, flags, t & 0xFF, (t >> 8) & 0xFF, (t >> 16) & 0xFF, (t >> 24) & 0xFF, 0, 255])
    if file_name is not None:
        h += file_name.encode("latin-1") + b"\0"          # GZipConstants.Encoding = Latin1
    return h


def member_footer(crc, total_in):                         # GetFooter :315-339
    return int(crc & 0xFFFFFFFF).to_bytes(4, "little") + int(total_in & 0xFFFFFFFF).to_bytes(4, "little")


class GZipOutputStream(DeflaterOutputStream):
    """new GZipOutputStream(stream[, size]); FileName / ModifiedTime / SetLevel; Write; Finish — the header goes out with the
    first Write or at Finish (:263-271, :384-392), the CRC runs over what was written (:203-214)."""

    def __init__(self, baseOutputStream, size=4096, deviceCrc=True):
        super().__init__(baseOutputStream, Deflater(Deflater.DEFAULT_COMPRESSION, True), size)
        self._state = "Header"
        self._crc = 0
        # deviceCrc (default): the CRC-32 of what is written is kept by the Deflater on the device beside the compression
        # (include/szl.h szl_deflater_crc32) instead of a pass over every Write on the host (:210); False: the reference's arrangement
        self._device_crc = bool(deviceCrc)
        if self._device_crc:
            self.deflater_.EnableCrc32()
        self._flags = 0
        self._file_name = None
        self.ModifiedTime = None                          # seconds since the epoch; None: now (:343)

    @property
    def FileName(self):
        return self._file_name

    @FileName.setter
    def FileName(self, value):                            # :98-112
        self._file_name = clean_filename(value) if value is not None else None
        self._flags = (self._flags | FNAME) if self._file_name else (self._flags & ~FNAME)
        if not self._file_name:
            self._file_name = None

    def SetLevel(self, level):                            # :130-140
        if level < Deflater.NO_COMPRESSION or level > Deflater.BEST_COMPRESSION:
            raise ValueError("level")
        self.deflater_.SetLevel(level)

    def GetLevel(self):
        return self.deflater_.GetLevel()

    def _write_header(self):                              # :384-392
        if self._state == "Header":
            self._state = "Footer"
            mt = int(time.time()) if self.ModifiedTime is None else self.ModifiedTime
            self.baseOutputStream_.write(member_header(mt, self._file_name))

    def Write(self, buffer, offset=0, count=None):        # :203-214
        if self._state == "Header":
            self._write_header()
        if self._state != "Footer":
            raise RuntimeError("Write not permitted in current state")
        a = np.frombuffer(buffer, dtype=np.uint8) if not isinstance(buffer, np.ndarray) else buffer
        count = a.size - offset if count is None else count
        if not self._device_crc:
            self._crc = _crc32(self._crc, a[offset:offset + count])
        super().Write(a, offset, count)

    def Finish(self):                                     # :259-281
        if self._state == "Header":
            self._write_header()                          # "If no data has been written a header should be added"
        if self._state == "Footer":
            self._state = "Finished"
            super().Finish()
            self.baseOutputStream_.write(member_footer(self.deflater_.Crc32 if self._device_crc else self._crc, self.deflater_.TotalIn))

    def Dispose(self):                                    # :170-188
        if not self.isClosed_:
            self.isClosed_ = True
            try:
                self.Finish()
            finally:
                if self._state != "Closed":
                    self._state = "Closed"
                    if self.IsStreamOwner:
                        self.baseOutputStream_.close()

    Close = Dispose


class GZipInputStream(InflaterInputStream):
    """new GZipInputStream(stream[, size]); Read; GetFilename — members may be concatenated, trailing garbage after a
    complete member ends the stream quietly (:107-153).  Over the device-aware InflaterInputStream / InflaterInputBuffer
    (streams.py): the default constructor reads 16 MiB ahead, header and trailer are parsed out of that same buffer with the
    buffer class's own ReadLeByte / ReadClearTextBuffer (:203-291, :305-351).  `deviceCrc` (default): the CRC-32 of what was read
    is kept by the Inflater on the device beside the decode (include/szl.h szl_inflater_crc32) instead of a host pass over every
    buffer returned (:141); `deviceCrc=False` is the reference's own arrangement."""

    def __init__(self, baseInputStream, size=4096, readAhead=None, deviceCrc=True):
        from .inflater import Inflater
        super().__init__(baseInputStream, Inflater(True), size, readAhead)
        self._read_header = False
        self._completed_last_block = False
        self._crc = 0
        self._file_name = None
        self._device_crc = bool(deviceCrc)
        if self._device_crc:
            self.inf.EnableCrc32()

    def GetFilename(self):
        return self._file_name

    def _le_byte(self):                                   # inputBuffer.ReadLeByte with the message ReadHeader gives (:203)
        ib = self.inputBuffer
        if ib.Available <= 0:
            ib.Fill()
            if ib.Available <= 0:
                raise EOFError("EOS reading GZIP header")
        return ib.ReadLeByte()

    def _header(self):                                    # ReadHeader :170-303
        ib = self.inputBuffer
        if ib.Available <= 0:
            ib.Fill()
            if ib.Available <= 0:
                return False                              # no header: EOF
        seen = bytearray()

        def rd():
            b = self._le_byte(); seen.append(b); return b
        if rd() != ID1:
            raise GZipException("Error GZIP header, first magic byte doesn't match")
        if rd() != ID2:
            raise GZipException("Error GZIP header,  second magic byte doesn't match")
        if rd() != CM_DEFLATE:
            raise GZipException("Error GZIP header, data not in deflate format")
        flags = rd()
        if flags & 0xE0:
            raise GZipException("Reserved flag bits in GZIP header != 0")
        for _ in range(6):
            rd()
        if flags & FEXTRA:
            n = rd() | (rd() << 8)
            for _ in range(n):
                rd()
        if flags & FNAME:
            name = bytearray()
            while True:
                b = rd()
                if b <= 0:
                    break
                if len(name) < 1024:
                    name.append(b)
            self._file_name = name.decode("latin-1")
        else:
            self._file_name = None
        if flags & FCOMMENT:
            while rd() > 0:
                pass
        if flags & FHCRC:
            hcrc = _crc32(0, bytes(seen))
            v = (self._le_byte() << 8) | self._le_byte()   # the reference reads the two bytes high first (:283-291)
            if v != (hcrc & 0xFFFF):
                raise GZipException("Header CRC value mismatch")
        self._crc = 0
        self._read_header = True
        return True

    def _footer(self):                                    # ReadFooter :305-351
        total_out = self.inf.TotalOut & 0xFFFFFFFF
        ours = self.inf.Crc32 if self._device_crc else self._crc
        self.inputBuffer.Available += self.inf.RemainingInput
        self.inf.Reset()
        f = np.zeros(8, np.uint8)
        need = 8
        while need > 0:                                   # :317-327
            k = self.inputBuffer.ReadClearTextBuffer(f, 8 - need, need)
            if k <= 0:
                raise EOFError("EOS reading GZIP footer")
            need -= k
        f = f.tobytes()
        crc = int.from_bytes(f[0:4], "little")
        if crc != ours:
            raise GZipException('GZIP crc sum mismatch, theirs "%08x" and ours "%08x"' % (crc, ours))
        if total_out != int.from_bytes(f[4:8], "little"):
            raise GZipException("Number of bytes mismatch in footer")
        self._read_header = False
        self._completed_last_block = True

    def Read(self, buffer, offset, count):                # :107-153
        while True:
            if not self._read_header:
                try:
                    if not self._header():
                        return 0
                except (GZipException, EOFError):
                    if self._completed_last_block:
                        return 0                          # trailing garbage after a complete member
                    raise
            n = super().Read(buffer, offset, count)
            if n > 0 and not self._device_crc:
                self._crc = _crc32(self._crc, np.frombuffer(buffer, dtype=np.uint8)[offset:offset + n])
            if self.inf.IsFinished:
                self._footer()
            if n > 0 or count == 0:
                return n

    def read_all(self, chunk=1 << 20):
        out, buf = bytearray(), np.zeros(chunk, np.uint8)
        while True:
            n = self.Read(buf, 0, chunk)
            if n <= 0:
                return bytes(out)
            out += buf[:n].tobytes()


# ---------------------------------------------------------------------------------------------
# batch forms
def write_members(datas, level=6, names=None, mtimes=None, engine=None):
    """One gzip member per buffer, compressed in ONE device call (raw deflate + CRC-32 on device), framed on the host like
    GZipOutputStream would (with FNAME when `names[i]` is given).  Returns the list of members."""
    from .batch import Engine
    eng = engine or Engine()
    res = eng.deflate(datas, level=level, nowrap=True, crc32=True)
    out = []
    for i, r in enumerate(res):
        if r.status:
            raise SharpZipBaseException("member %d: status %d" % (i, r.status))
        n = len(datas[i]) if not isinstance(datas[i], np.ndarray) else datas[i].size
        nm = clean_filename(names[i]) if names and names[i] else None
        out.append(member_header(mtimes[i] if mtimes else 0, nm or None) + r.data + member_footer(r.crc32, n))
    if engine is None:
        eng.close()
    return out


def parse_member_header(buf, pos=0):
    """ReadHeader on a buffer: returns (offset of the deflate data, file name or None)."""
    b = memoryview(buf)
    if len(b) - pos < 10:
        raise EOFError("EOS reading GZIP header")
    if b[pos] != ID1:
        raise GZipException("Error GZIP header, first magic byte doesn't match")
    if b[pos + 1] != ID2:
        raise GZipException("Error GZIP header,  second magic byte doesn't match")
    if b[pos + 2] != CM_DEFLATE:
        raise GZipException("Error GZIP header, data not in deflate format")
    flags = b[pos + 3]
    if flags & 0xE0:
        raise GZipException("Reserved flag bits in GZIP header != 0")
    p = pos + 10
    name = None

    def need(k):                                   # the streaming reader raises EOFError on a truncated header (ReadHeader :203-291)
        if len(b) - p < k:
            raise EOFError("EOS reading GZIP header")

    def cstring():                                 # a zero-terminated field: searched 64 KiB at a time, never by copying the member
        q = p
        while True:
            chunk = bytes(b[q:q + 65536])
            if not chunk:
                raise EOFError("EOS reading GZIP header")
            k = chunk.find(b"\0")
            if k >= 0:
                return q + k
            q += len(chunk)
    if flags & FEXTRA:
        need(2)
        xlen = b[p] | (b[p + 1] << 8)
        p += 2
        need(xlen)
        p += xlen
    if flags & FNAME:
        e = cstring()
        name = bytes(b[p:min(e, p + 1024)]).decode("latin-1")
        p = e + 1
    if flags & FCOMMENT:
        p = cstring() + 1
    if flags & FHCRC:
        need(2)
        v = (b[p] << 8) | b[p + 1]
        if v != (_crc32(0, bytes(b[pos:p])) & 0xFFFF):
            raise GZipException("Header CRC value mismatch")
        p += 2
    return p, name


def read_members(members, sizes=None, engine=None):
    """Reader fast path for independent members (one per buffer, e.g. the parts of a multi-member archive whose offsets are
    known): headers on the host, ALL members through one szl_inflate_batch_host call with CRC-32 on device, trailers checked
    like ReadFooter.  `sizes[i]` bounds member i's output; default: ISIZE of its trailer (valid below 4 GiB), clamped to what
    its compressed bytes can expand to.
    Returns [(data, file name)]."""
    from .batch import Engine
    eng = engine or Engine()
    bodies, names, caps = [], [], []
    for i, m in enumerate(members):
        start, name = parse_member_header(m)
        bodies.append(np.frombuffer(m, dtype=np.uint8)[start:])
        names.append(name)
        # default bound: the trailer's ISIZE — untrusted, so never more than DEFLATE can expand the member's bytes to (1032:1)
        caps.append(sizes[i] if sizes else min(int.from_bytes(bytes(m[-4:]), "little"), 1032 * (len(m) - start) + 1024))
    res = eng.inflate(bodies, caps, nowrap=True, crc32=True)
    out = []
    for i, ((r, consumed), body) in enumerate(zip(res, bodies)):
        if r.status:
            raise SharpZipBaseException("member %d: status %d" % (i, r.status))
        f = bytes(body[consumed:consumed + 8])
        if len(f) < 8:
            raise EOFError("EOS reading GZIP footer")
        if int.from_bytes(f[0:4], "little") != r.crc32:
            raise GZipException('GZIP crc sum mismatch, theirs "%08x" and ours "%08x"' % (int.from_bytes(f[0:4], "little"), r.crc32))
        if int.from_bytes(f[4:8], "little") != (len(r.data) & 0xFFFFFFFF):
            raise GZipException("Number of bytes mismatch in footer")
        out.append((r.data, names[i]))
    if engine is None:
        eng.close()
    return out
