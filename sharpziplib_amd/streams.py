"""Host-side mirrors of DeflaterOutputStream / InflaterInputStream
(CS/DeflaterOutputStream.cs, CS/InflaterInputStream.cs): the managed stream adapters of the
reference stay on the host and only talk to Deflater / Inflater, exactly as in the reference —
so these are line-for-line *behavioural* mirrors (same loops, same error messages), used by the
parity tests to drive the C ABI with the reference's call patterns."""
import io

import numpy as np

from .deflater import Deflater, SharpZipBaseException


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

    def _deflate(self, flushing=False):                        # DeflateSyncOrAsync :242-272
        while flushing or not self.deflater_.IsNeedingInput:
            n = self.deflater_.Deflate(self.buffer_, 0, self.buffer_.size)
            if n <= 0:
                break
            self.EncryptBlock(self.buffer_, 0, n)              # :256
            self.baseOutputStream_.write(self.buffer_[:n].tobytes())
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
            n = self.deflater_.Deflate(self.buffer_, 0, self.buffer_.size)
            if n <= 0:
                break
            self.EncryptBlock(self.buffer_, 0, n)              # :111
            self.baseOutputStream_.write(self.buffer_[:n].tobytes())
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


class InflaterInputBuffer:
    """CS/InflaterInputStream.cs:16-330 (the parts the codec path uses)."""

    def __init__(self, stream, bufferSize=4096):
        self.inputStream = stream
        if bufferSize < 1024:
            bufferSize = 1024                                   # :35-38
        self.rawData = np.zeros(bufferSize, dtype=np.uint8)
        self.rawLength = 0
        self.available = 0
        self.clearText = self.rawData                           # :41: same array until a transform is set
        self.clearTextLength = 0
        self.cryptoTransform = None
        self.internalClearText = None

    def SetCryptoTransform(self, value):                        # CryptoTransform setter :276-305: decrypt BEFORE the codec sees the bytes
        self.cryptoTransform = value
        if value is not None:
            if self.clearText is self.rawData:
                if self.internalClearText is None:
                    self.internalClearText = np.zeros(self.rawData.size, dtype=np.uint8)
                self.clearText = self.internalClearText
            self.clearTextLength = self.rawLength
            if self.available > 0:
                value.TransformBlock(self.rawData, self.rawLength - self.available, self.available, self.clearText, self.rawLength - self.available)
        else:
            self.clearText = self.rawData
            self.clearTextLength = self.rawLength

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

    def Fill(self):                                             # :115
        self.rawLength = 0
        toRead = self.rawData.size
        while toRead > 0:
            b = self.inputStream.read(toRead)
            if not b:
                break
            self.rawData[self.rawLength:self.rawLength + len(b)] = np.frombuffer(b, dtype=np.uint8)
            self.rawLength += len(b)
            toRead -= len(b)
        if self.cryptoTransform is not None:                    # :131-138
            self.clearTextLength = self.cryptoTransform.TransformBlock(self.rawData, 0, self.rawLength, self.clearText, 0)
        else:
            self.clearTextLength = self.rawLength
        self.available = self.clearTextLength


class InflaterInputStream:
    def __init__(self, baseInputStream, inflater=None, bufferSize=4096):
        from .inflater import Inflater
        if baseInputStream is None:
            raise ValueError("baseInputStream")
        if bufferSize <= 0:
            raise ValueError("bufferSize")
        self.baseInputStream = baseInputStream
        self.inf = inflater if inflater is not None else Inflater()
        self.inputBuffer = InflaterInputBuffer(baseInputStream, bufferSize)
        self.IsStreamOwner = True

    @property
    def Available(self):                                        # :472
        return 0 if self.inf.IsFinished else 1

    def Fill(self):                                             # :486
        if self.inputBuffer.Available <= 0:
            self.inputBuffer.Fill()
            if self.inputBuffer.Available <= 0:
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
                raise SharpZipBaseException("Invalid input data")
        return count - remaining

    def Dispose(self):
        if self.IsStreamOwner:
            self.baseInputStream.close()
