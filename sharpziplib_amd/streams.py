"""Host-side mirrors of DeflaterOutputStream / InflaterInputStream
(CS/DeflaterOutputStream.cs, CS/InflaterInputStream.cs): the managed stream adapters of the
reference stay on the host and only talk to Deflater / Inflater, exactly as in the reference —
so these are line-for-line *behavioural* mirrors (same loops, same error messages), used by the
parity tests to drive the C ABI with the reference's call patterns."""
import ctypes
import io

import numpy as np

from . import _lib
from .deflater import Deflater, SharpZipBaseException


class ZipException(SharpZipBaseException):                      # src/ICSharpCode.SharpZipLib/Zip/ZipException.cs
    pass


class DeflaterOutputStream:
    def __init__(self, baseOutputStream, deflater=None, bufferSize=512):
        if baseOutputStream is None:
            raise ValueError("baseOutputStream")
        if not baseOutputStream.writable():
            raise ValueError("Must support writing")          # :75-78
        if bufferSize < 512:
            raise ValueError("bufferSize")                     # :80-83
        self.baseOutputStream_ = baseOutputStream
        self.buffer_ = np.zeros(bufferSize, dtype=np.uint8)
        self.deflater_ = deflater if deflater is not None else Deflater()
        if hasattr(self.deflater_, "CallerDrains"):
            self.deflater_.CallerDrains(True)                  # Deflate() below runs until IsNeedingInput (:242-272): say so (include/szl.h)
        self.IsStreamOwner = True
        self.isClosed_ = False
        # ICryptoTransform of the reference (:200): any object with TransformBlock(in, inOff, count, out, outOff) -> count.
        # The hook stays on the host exactly where the reference has it: AFTER the codec, on each block of compressed bytes,
        # in the order they are written.
        self.cryptoTransform_ = None

    def EncryptBlock(self, buffer, offset, length):            # :227-231
        if self.cryptoTransform_ is None:
            return
        self.cryptoTransform_.TransformBlock(buffer, 0, length, buffer, 0)

    def _hand_out(self):
        """One turn of the reference's loops (Finish :104-113, DeflateSyncOrAsync :247-266): the next compressed bytes go to the base stream;
        False where Deflate() returned 0.  Device-aware (INTEGRATION.md file 4, dotnet/DeflaterOutputStream.Device.cs): without a crypto
        transform the bytes are written straight out of the Deflater's pinned queue — all it has, in one Write — instead of being copied
        into buffer_ (512 bytes by default, :26-29) and written from there; with one they pass through buffer_ as in the reference,
        because TransformBlock works in place (:227-231)."""
        view = getattr(self.deflater_, "DeflateView", None)
        if self.cryptoTransform_ is None and view is not None:
            v = view()
            if v is None:
                return False
            self.baseOutputStream_.write(v)                    # (valid until the next call on the Deflater: a stream copies what it is given)
            return True
        n = self.deflater_.Deflate(self.buffer_, 0, self.buffer_.size)
        if n <= 0:
            return False
        self.EncryptBlock(self.buffer_, 0, n)                  # :256
        self.baseOutputStream_.write(self.buffer_[:n].tobytes())
        return True

    def _deflate(self, flushing=False):                        # DeflateSyncOrAsync :242-272
        while flushing or not self.deflater_.IsNeedingInput:
            if not self._hand_out():
                break
        if not self.deflater_.IsNeedingInput:
            raise SharpZipBaseException("DeflaterOutputStream can't deflate all input?")

    def Write(self, buffer, offset=0, count=None):            # :506
        self.deflater_.SetInput(buffer, offset, count)
        self._deflate()

    def WriteByte(self, value):                                # :487
        self.Write(bytes([value & 0xFF]), 0, 1)

    def Flush(self):                                           # :388
        self.deflater_.Flush()
        self._deflate(True)
        self.baseOutputStream_.flush()

    def Finish(self):                                          # :100
        self.deflater_.Finish()
        while not self.deflater_.IsFinished:
            if not self._hand_out():
                break
        if not self.deflater_.IsFinished:
            raise SharpZipBaseException("Can't deflate all input?")
        self.baseOutputStream_.flush()
        self.cryptoTransform_ = None                           # :122-130 (disposed after the last block)

    def Dispose(self):                                         # :412
        if not self.isClosed_:
            self.isClosed_ = True
            try:
                self.Finish()
            finally:
                if self.IsStreamOwner:
                    self.baseOutputStream_.close()

    Close = Dispose

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.Dispose()


class _PinnedArray:
    """A numpy uint8 array over pinned host memory (szl_host_alloc): the base stream reads straight into it and the device reads it by
    DMA; a SetInput out of it costs no host copy (include/szl.h)."""

    def __init__(self, size):
        self._L = _lib.lib()
        self._p = self._L.szl_host_alloc(size)
        if not self._p:
            raise SharpZipBaseException(self._L.szl_last_error().decode())
        carr = (ctypes.c_uint8 * size).from_address(self._p)
        carr._szl_owner = self                                  # arrays made of `carr` keep the allocation alive
        self.array = np.ctypeslib.as_array(carr)

    def free(self):
        if self._p:
            self.array = None
            self._L.szl_host_free(self._p)
            self._p = None

    def __del__(self):
        self.free()


class InflaterInputBuffer:
    """CS/InflaterInputStream.cs:14-330, device-aware (INTEGRATION.md file 3; sharpziplib_amd/dotnet/InflaterInputStream.Device.cs is
    the C# form).  Every member of the reference's class, with one change: the buffer behind RawData is `ReadAheadBytes` long (16 MiB)
    unless the constructor asked for more, and lives in pinned host memory.  `bufferSize` stays what the reference says it is — a
    lower bound (:35-38) — so GZipInputStream(stream) (4096, S/GZip/GzipInputStream.cs:72) and ZipInputStream hand the device Inflater
    pieces the chunk-parallel decoder can take instead of 4 KiB for one wavefront.  `readAhead=0` gives the reference's sizes."""

    ReadAheadBytes = 16 << 20
    ReadAheadLongBytes = 64 << 20     # a base stream that can tell it holds eight read-aheads or more gets this much (a piece of 64 MiB
    #                                   costs the device little more than one of 16: profiles/r05/read_path.log, 2.8 -> 5.2 GiB/s)

    def __init__(self, stream, bufferSize=4096, readAhead=None):
        self.inputStream = stream
        if bufferSize < 1024:
            bufferSize = 1024                                   # :35-38
        ahead = self.ReadAheadBytes if readAhead is None else readAhead
        size = max(bufferSize, ahead)
        if size > bufferSize:                                   # no more than the base stream still holds, where it can tell
            try:
                if stream.seekable():
                    pos = stream.tell()
                    left = stream.seek(0, io.SEEK_END) - pos
                    stream.seek(pos)
                    if readAhead is None and left >= 8 * ahead:
                        size = max(size, self.ReadAheadLongBytes)
                    size = max(bufferSize, min(size, left + 1))
            except (AttributeError, OSError, ValueError):
                pass
        self._pin = _PinnedArray(size) if size >= (64 << 10) else None
        self.rawData = self._pin.array if self._pin else np.zeros(size, dtype=np.uint8)
        self.rawLength = 0
        self.available = 0
        self.clearText = self.rawData                           # :41: same array until a transform is set
        self.clearTextLength = 0
        self.cryptoTransform = None
        self.internalClearText = None
        self._pin_clear = None

    def Dispose(self):
        for p in (self._pin, self._pin_clear):
            if p is not None:
                p.free()
        self._pin = self._pin_clear = None
        self.rawData = self.clearText = self.internalClearText = None

    # :47-89
    RawLength = property(lambda self: self.rawLength)
    RawData = property(lambda self: self.rawData)
    ClearTextLength = property(lambda self: self.clearTextLength)
    ClearText = property(lambda self: self.clearText)

    def SetCryptoTransform(self, value):                        # CryptoTransform setter :276-305: decrypt BEFORE the codec sees the bytes
        self.cryptoTransform = value
        if value is not None:
            if self.clearText is self.rawData:
                if self.internalClearText is None:
                    if self._pin is not None:
                        self._pin_clear = _PinnedArray(self.rawData.size)
                        self.internalClearText = self._pin_clear.array
                    else:
                        self.internalClearText = np.zeros(self.rawData.size, dtype=np.uint8)
                self.clearText = self.internalClearText
            self.clearTextLength = self.rawLength
            if self.available > 0:
                value.TransformBlock(self.rawData, self.rawLength - self.available, self.available, self.clearText, self.rawLength - self.available)
        else:
            self.clearText = self.rawData
            self.clearTextLength = self.rawLength

    CryptoTransform = property(None, SetCryptoTransform)

    @property
    def Available(self):
        return self.available

    @Available.setter
    def Available(self, v):
        self.available = v

    def SetInflaterInput(self, inflater):                       # :103
        if self.available > 0:
            inflater.SetInput(self.clearText, self.clearTextLength - self.available, self.available)
            self.available = 0
            # device-aware: a buffer that Fill() filled to the brim promises more input (include/szl.h szl_inflater_expect_more)
            if hasattr(inflater, "ExpectMoreInput"):
                inflater.ExpectMoreInput(self.rawLength == self.rawData.size)

    def Fill(self):                                             # :115
        self.rawLength = 0
        toRead = self.rawData.size
        readinto = getattr(self.inputStream, "readinto", None)
        mv = memoryview(self.rawData)
        while toRead > 0:
            if readinto is not None:
                k = readinto(mv[self.rawLength:self.rawLength + toRead])
                if not k:
                    break
            else:
                b = self.inputStream.read(toRead)
                if not b:
                    break
                k = len(b)
                self.rawData[self.rawLength:self.rawLength + k] = np.frombuffer(b, dtype=np.uint8)
            self.rawLength += k
            toRead -= k
        if self.cryptoTransform is not None:                    # :131-138
            self.clearTextLength = self.cryptoTransform.TransformBlock(self.rawData, 0, self.rawLength, self.clearText, 0)
        else:
            self.clearTextLength = self.rawLength
        self.available = self.clearTextLength

    def _read_buffer(self, src_of, src_len_of, outBuffer, offset, length):
        if length < 0:
            raise ValueError("length")
        out = np.frombuffer(outBuffer, dtype=np.uint8) if not isinstance(outBuffer, np.ndarray) else outBuffer
        cur, left = offset, length
        while left > 0:
            if self.available <= 0:
                self.Fill()
                if self.available <= 0:
                    return 0
            k = min(left, self.available)
            s0 = src_len_of() - self.available
            out[cur:cur + k] = src_of()[s0:s0 + k]
            cur += k
            left -= k
            self.available -= k
        return length

    def ReadRawBuffer(self, outBuffer, offset=0, length=None):  # :148-186
        return self._read_buffer(lambda: self.rawData, lambda: self.rawLength, outBuffer, offset, len(outBuffer) if length is None else length)

    def ReadClearTextBuffer(self, outBuffer, offset, length):   # :195-226
        return self._read_buffer(lambda: self.clearText, lambda: self.clearTextLength, outBuffer, offset, length)

    def ReadLeByte(self):                                       # :232-245
        if self.available <= 0:
            self.Fill()
            if self.available <= 0:
                raise ZipException("EOF in header")
        b = int(self.rawData[self.rawLength - self.available])
        self.available -= 1
        return b

    def ReadLeShort(self):                                      # :251
        return self.ReadLeByte() | (self.ReadLeByte() << 8)

    def ReadLeInt(self):                                        # :260
        return self.ReadLeShort() | (self.ReadLeShort() << 16)

    def ReadLeLong(self):                                       # :269
        return (self.ReadLeInt() & 0xFFFFFFFF) | (self.ReadLeInt() << 32)


class InflaterInputStream:
    """CS/InflaterInputStream.cs:332-700 over the device-aware buffer class above (same members, same loops, same messages)."""

    def __init__(self, baseInputStream, inflater=None, bufferSize=4096, readAhead=None):
        from .inflater import Inflater
        if baseInputStream is None:
            raise ValueError("baseInputStream")
        if bufferSize <= 0:
            raise ValueError("bufferSize")
        self.baseInputStream = baseInputStream
        self.inf = inflater if inflater is not None else Inflater()
        self.inputBuffer = InflaterInputBuffer(baseInputStream, bufferSize, readAhead)
        self.IsStreamOwner = True
        self.isClosed = False
        self.csize = 0

    def Skip(self, count):                                      # :420-463
        if count <= 0:
            raise ValueError("count")
        if self.baseInputStream.seekable():
            self.baseInputStream.seek(count, io.SEEK_CUR)
            return count
        length = min(2048, count)
        toSkip = count
        while toSkip > 0:
            b = self.baseInputStream.read(min(length, toSkip))
            if not b:
                break
            toSkip -= len(b)
        return count - toSkip

    def StopDecrypting(self):                                   # :468
        self.inputBuffer.CryptoTransform = None

    @property
    def Available(self):                                        # :472
        return 0 if self.inf.IsFinished else 1

    CanRead = property(lambda self: self.baseInputStream.readable())    # :507-545
    CanSeek = property(lambda self: False)
    CanWrite = property(lambda self: False)

    @property
    def Length(self):
        raise NotImplementedError("InflaterInputStream Length is not supported")

    @property
    def Position(self):
        return self.baseInputStream.tell()

    def Flush(self):
        self.baseInputStream.flush()

    def Seek(self, offset, origin):
        raise NotImplementedError("Seek not supported")

    def SetLength(self, value):
        raise NotImplementedError("InflaterInputStream SetLength not supported")

    def Write(self, buffer, offset, count):
        raise NotImplementedError("InflaterInputStream Write not supported")

    def WriteByte(self, value):
        raise NotImplementedError("InflaterInputStream WriteByte not supported")

    def Fill(self):                                             # :486
        if self.inputBuffer.Available <= 0:
            self.inputBuffer.Fill()
            if self.inputBuffer.Available <= 0:
                # device-aware: the base stream has ended — the promise of more input is taken back; if a remainder of the last piece
                # was waiting for it, Inflate() decodes it now and Read() goes on (a truncated stream delivers every byte it holds
                # before "Unexpected EOF", as the reference's does)
                if hasattr(self.inf, "ExpectMoreInput") and self.inf.ExpectMoreInput(False):
                    return
                raise SharpZipBaseException("Unexpected EOF")
        self.inputBuffer.SetInflaterInput(self.inf)

    def Read(self, buffer, offset, count):                      # :658
        if self.inf.IsNeedingDictionary:
            raise SharpZipBaseException("Need a dictionary")
        remaining = count
        while True:
            n = self.inf.Inflate(buffer, offset, remaining)
            offset += n
            remaining -= n
            if remaining == 0 or self.inf.IsFinished:
                break
            if self.inf.IsNeedingInput:
                self.Fill()
            elif n == 0:
                raise ZipException("Invalid input data")
        return count - remaining

    def Dispose(self):                                          # :622-640
        if not self.isClosed:
            self.isClosed = True
            if self.IsStreamOwner:
                self.baseInputStream.close()
        if self.inf is not None and hasattr(self.inf, "DetachInput"):
            self.inf.DetachInput()                              # (the Inflater may live on — InflaterPool — the pinned buffer does not)
        self.inputBuffer.Dispose()
        self.inf = None

    Close = Dispose
