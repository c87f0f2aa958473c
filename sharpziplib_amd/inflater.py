"""Host-side mirror of ICSharpCode.SharpZipLib.Zip.Compression.Inflater (C/Inflater.cs) over the C ABI."""
import numpy as np

from . import _lib
from .deflater import InvalidOperation, SharpZipBaseException, _raise


class Inflater:
    def __init__(self, noHeader=False):
        self._L = _lib.lib()
        self.noHeader = bool(noHeader)
        self._h = self._L.szl_inflater_create(1 if noHeader else 0)
        if not self._h:
            raise SharpZipBaseException(self._L.szl_last_error().decode())

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.szl_inflater_destroy(h)
            self._h = None

    def Reset(self):
        self._L.szl_inflater_reset(self._h)

    def SetInput(self, buffer, index=0, count=None):
        a = np.frombuffer(buffer, dtype=np.uint8) if not isinstance(buffer, np.ndarray) else buffer
        count = a.size - index if count is None else count
        if index < 0 or count < 0 or index + count > a.size:
            raise ValueError("count")
        a = np.ascontiguousarray(a[index:index + count])
        s = self._L.szl_inflater_set_input(self._h, a.ctypes.data, count)
        if s < 0:
            _raise(s, "SetInput")
        self._input = a            # (a long piece out of a pinned buffer is read in place, include/szl.h: the array lives as long as it is referred to)

    def ExpectMoreInput(self, more):
        """the stream shim's hint (include/szl.h szl_inflater_expect_more); returns True if a deferred remainder was released"""
        return self._L.szl_inflater_expect_more(self._h, 1 if more else 0) == 1

    def DetachInput(self):
        self._L.szl_inflater_detach_input(self._h)
        self._input = None

    def SetDictionary(self, buffer, index=0, count=None):
        a = np.ascontiguousarray(np.frombuffer(buffer, dtype=np.uint8))
        count = a.size - index if count is None else count
        if index < 0 or count < 0 or index + count > a.size:
            raise ValueError("count")  # C/Inflater.cs:565-571
        s = self._L.szl_inflater_set_dictionary(self._h, a[index:].ctypes.data if count else None, count)
        if s < 0:
            _raise(s, "SetDictionary")

    def Inflate(self, buffer, offset=0, count=None):
        a = np.frombuffer(buffer, dtype=np.uint8) if not isinstance(buffer, np.ndarray) else buffer
        count = a.size - offset if count is None else count
        if count < 0:
            raise ValueError("count cannot be negative")
        if offset < 0:
            raise ValueError("offset cannot be negative")
        if offset + count > a.size:
            raise ValueError("count exceeds buffer bounds")
        n = self._L.szl_inflater_inflate(self._h, a[offset:].ctypes.data if count else None, count)
        if n < 0:
            _raise(n, "Inflate")
        return n

    @property
    def IsNeedingInput(self):
        return bool(self._L.szl_inflater_needs_input(self._h))

    @property
    def IsNeedingDictionary(self):
        return bool(self._L.szl_inflater_needs_dictionary(self._h))

    @property
    def IsFinished(self):
        return bool(self._L.szl_inflater_is_finished(self._h))

    @property
    def RemainingInput(self):
        return self._L.szl_inflater_remaining_input(self._h)

    @property
    def TotalIn(self):
        return self._L.szl_inflater_total_in(self._h)

    @property
    def TotalOut(self):
        return self._L.szl_inflater_total_out(self._h)

    @property
    def Adler(self):
        return self._L.szl_inflater_adler(self._h)

    # device-side CRC-32 of the bytes handed out (include/szl.h: what GZipInputStream / ZipInputStream keep on the CPU)
    def EnableCrc32(self, on=True):
        s = self._L.szl_inflater_enable_crc32(self._h, 1 if on else 0)
        if s < 0:
            _raise(s, "EnableCrc32")

    @property
    def Crc32(self):
        return self._L.szl_inflater_crc32(self._h)
