"""Host-side mirror of ICSharpCode.SharpZipLib.Zip.Compression.Deflater (C/Deflater.cs) over the
C ABI — same member names, argument meaning and error behaviour, so the parity tests read like the
reference's own (T/Base/InflaterDeflaterTests.cs).  All compression happens in libszl_amd.so on
the device; exceptions mirror the reference's types:
    ArgumentOutOfRangeException -> ValueError, InvalidOperationException -> InvalidOperation,
    SharpZipBaseException -> SharpZipBaseException.
"""
import ctypes
import numpy as np

from . import _lib


class SharpZipBaseException(Exception):
    pass


class InvalidOperation(RuntimeError):
    pass


class NotSupportedOnDevice(SharpZipBaseException):
    pass


def _raise(status, where):
    L = _lib.lib()
    detail = L.szl_last_error().decode()
    msg = "%s: %s%s" % (where, L.szl_strerror(status).decode(), (" — " + detail) if detail else "")
    if status == -1:
        raise ValueError(msg)
    if status == -2:
        raise InvalidOperation(msg)
    if status == -5:
        raise NotSupportedOnDevice(msg)
    if status in (-27, -28):
        raise IndexError(msg)  # IndexOutOfRangeException out of InflaterHuffmanTree.BuildTree (over-subscribed code lengths)
    raise SharpZipBaseException(msg)


class DeflateStrategy:
    Default, Filtered, HuffmanOnly = 0, 1, 2


class Deflater:
    BEST_COMPRESSION, BEST_SPEED, DEFAULT_COMPRESSION, NO_COMPRESSION, DEFLATED = 9, 1, -1, 0, 8  # C/Deflater.cs:62-83

    def __init__(self, level=DEFAULT_COMPRESSION, noZlibHeaderOrFooter=False):
        self._L = _lib.lib()
        if level != -1 and not (0 <= level <= 9):
            raise ValueError("level")  # ArgumentOutOfRangeException C/Deflater.cs:184-187
        self._h = self._L.szl_deflater_create(level, 1 if noZlibHeaderOrFooter else 0)
        if not self._h:
            msg = self._L.szl_last_error().decode()
            if "not on the device path" in msg:
                raise NotSupportedOnDevice(msg)
            raise SharpZipBaseException(msg)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.szl_deflater_destroy(h)
            self._h = None

    def Reset(self):
        s = self._L.szl_deflater_reset(self._h)
        if s < 0:
            _raise(s, "Reset")

    def SetInput(self, buffer, offset=0, count=None):
        a = np.frombuffer(buffer, dtype=np.uint8) if not isinstance(buffer, np.ndarray) else buffer
        count = a.size - offset if count is None else count
        if offset < 0 or count < 0 or offset + count > a.size:
            raise ValueError("count")  # C/DeflaterEngine.cs:148-176
        a = np.ascontiguousarray(a[offset:offset + count])
        s = self._L.szl_deflater_set_input(self._h, a.ctypes.data, count)
        if s < 0:
            _raise(s, "SetInput")

    def SetLevel(self, level):
        s = self._L.szl_deflater_set_level(self._h, level)
        if s < 0:
            _raise(s, "SetLevel")

    def GetLevel(self):
        return self._L.szl_deflater_get_level(self._h)

    def SetStrategy(self, strategy):
        s = self._L.szl_deflater_set_strategy(self._h, strategy)
        if s < 0:
            _raise(s, "SetStrategy")

    def SetDictionary(self, dictionary, index=0, count=None):
        a = np.ascontiguousarray(np.frombuffer(dictionary, dtype=np.uint8))
        count = a.size - index if count is None else count
        if index < 0 or count < 0 or index + count > a.size:
            raise ValueError("count")  # the reference reads buffer[offset..offset+length) (C/DeflaterEngine.cs:198-229)
        s = self._L.szl_deflater_set_dictionary(self._h, a[index:].ctypes.data if count else None, count)
        if s < 0:
            _raise(s, "SetDictionary")

    def Flush(self):
        self._L.szl_deflater_flush(self._h)

    def Finish(self):
        self._L.szl_deflater_finish(self._h)

    def Deflate(self, output, offset=0, length=None):
        """output: writable numpy uint8 array / bytearray. Returns number of bytes written."""
        a = np.frombuffer(output, dtype=np.uint8) if not isinstance(output, np.ndarray) else output
        length = a.size - offset if length is None else length
        if offset < 0 or length < 0 or offset + length > a.size:
            raise IndexError("output[offset..offset+length) does not fit the array")  # C/Deflater.cs:427 indexes the array itself
        if length == 0:
            n = self._L.szl_deflater_deflate(self._h, None, 0)
        else:
            n = self._L.szl_deflater_deflate(self._h, a[offset:].ctypes.data, length)
        if n < 0:
            _raise(n, "Deflate")
        return n

    def DeflateView(self):
        """Everything the next Deflate() calls would hand out, in place (szl_deflater_deflate_view): a memoryview over the object's pinned
        output queue, valid until the next call on the object; None where Deflate() would return 0.  What the device-aware
        DeflaterOutputStream writes to its base stream instead of buffer_.Length bytes at a time (CS/DeflaterOutputStream.cs:242-272)."""
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        s = self._L.szl_deflater_deflate_view(self._h, ctypes.byref(p), ctypes.byref(n))
        if s < 0:
            _raise(s, "Deflate")
        if not n.value:
            return None
        return memoryview((ctypes.c_uint8 * n.value).from_address(p.value)).cast("B")

    @property
    def IsFinished(self):
        return bool(self._L.szl_deflater_is_finished(self._h))

    @property
    def IsNeedingInput(self):
        return bool(self._L.szl_deflater_needs_input(self._h))

    @property
    def TotalIn(self):
        return self._L.szl_deflater_total_in(self._h)

    @property
    def TotalOut(self):
        return self._L.szl_deflater_total_out(self._h)

    @property
    def Adler(self):
        return self._L.szl_deflater_adler(self._h)

    def CallerDrains(self, on=True):
        """(include/szl.h szl_deflater_caller_drains) the caller takes all Deflate() offers before it changes a parameter, as
        DeflaterOutputStream does (CS/DeflaterOutputStream.cs:242-272); without this a SetLevel / SetStrategy with 16 KiB or more
        pending is refused (NotSupportedOnDevice) rather than answered with bytes that depend on an assumption."""
        s = self._L.szl_deflater_caller_drains(self._h, 1 if on else 0)
        if s < 0:
            _raise(s, "CallerDrains")

    # device-side CRC-32 of the input (include/szl.h: what GZipOutputStream / ZipOutputStream keep on the CPU over every Write)
    def EnableCrc32(self, on=True):
        s = self._L.szl_deflater_enable_crc32(self._h, 1 if on else 0)
        if s < 0:
            _raise(s, "EnableCrc32")

    @property
    def Crc32(self):
        return self._L.szl_deflater_crc32(self._h)
