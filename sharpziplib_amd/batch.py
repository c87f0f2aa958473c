"""Batch (device-resident) interface over szl_deflate_batch_* — the fast path used by bench.py,
by the ZIP passthrough feed (S/Zip/ZipOutputStream.cs:313 PutNextPassthroughEntry) and by tests.

Each stream is compressed exactly like `new Deflater(level, nowrap)`; SetInput(all); Finish()
(C/Deflater.cs) — bit-identical output, CRC-32/Adler-32 of the input computed on the device.
"""
import ctypes
from collections import namedtuple

import numpy as np

from . import _lib

Result = namedtuple("Result", "data crc32 adler32 status")


class Engine:
    """Owns an szl_engine (device workspace). Not thread-safe, like the reference codec objects."""

    def __init__(self):
        self._L = _lib.lib()
        self._h = self._L.szl_engine_create()
        if not self._h:
            raise RuntimeError("szl_engine_create failed: %s" % self._L.szl_last_error().decode())

    def close(self):
        if self._h:
            self._L.szl_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def bound(n):
        return int(_lib.lib().szl_deflate_bound(n))

    @staticmethod
    def layout(lengths, nowrap=True, extra=0):
        """Stream table for inputs packed back to back (4-byte aligned output regions)."""
        L = _lib.lib()
        arr = (_lib.Stream * len(lengths))()
        io = oo = 0
        for i, n in enumerate(lengths):
            cap = (int(L.szl_deflate_bound(n)) + (0 if nowrap else 6) + extra + 3) & ~3
            arr[i].in_off, arr[i].in_len, arr[i].out_off, arr[i].out_cap = io, n, oo, cap
            io += n
            oo += cap
        return arr, io, oo

    def deflate(self, buffers, level=6, strategy=0, nowrap=True, crc32=False, adler32=False, sync_flush_before_finish=False,
                gzip_mtime=None):
        """Compress host buffers (list of bytes/ndarray); returns [Result]."""
        bufs = [np.ascontiguousarray(np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b, dtype=np.uint8) for b in buffers]
        arr, in_total, out_total = self.layout([b.size for b in bufs], nowrap, 18 if gzip_mtime is not None else 0)
        if gzip_mtime is not None:
            for s in arr:
                s.reserved = gzip_mtime
        hin = np.empty(in_total + 8, dtype=np.uint8)
        for s, b in zip(arr, bufs):
            hin[s.in_off:s.in_off + s.in_len] = b
        hout = np.zeros(out_total + 8, dtype=np.uint8)
        flags = (_lib.F_NOWRAP if nowrap else 0) | (_lib.F_CRC32 if crc32 else 0) | (_lib.F_ADLER32 if adler32 else 0) | \
                (_lib.F_SYNC_FLUSH_BEFORE_FINISH if sync_flush_before_finish else 0) | (_lib.F_GZIP if gzip_mtime is not None else 0)
        _lib.check(self._L.szl_deflate_batch_host(self._h, hin.ctypes.data, hout.ctypes.data, arr, len(bufs), level, strategy, flags),
                   "szl_deflate_batch_host")
        return [Result(hout[s.out_off:s.out_off + s.out_len].tobytes(), s.crc32, s.adler32, s.status) for s in arr]

    def deflate_device(self, d_in_ptr, d_out_ptr, streams, level=6, strategy=0, flags=_lib.F_NOWRAP, hip_stream=0):
        """Device-resident call: raw device pointers (e.g. torch tensor .data_ptr()) + a Stream array."""
        _lib.check(self._L.szl_deflate_batch_device(self._h, d_in_ptr, d_out_ptr, streams, len(streams), level, strategy, flags, hip_stream),
                   "szl_deflate_batch_device")

    def inflate_device(self, d_in_ptr, d_out_ptr, streams, flags=_lib.F_NOWRAP, hip_stream=0):
        """Device-resident inflate of independent streams (szl_inflate_batch_device): per stream status / out_len / in_consumed."""
        _lib.check(self._L.szl_inflate_batch_device(self._h, d_in_ptr, d_out_ptr, streams, len(streams), flags, hip_stream),
                   "szl_inflate_batch_device")

    def timing(self):
        t = _lib.Timing()
        self._L.szl_engine_last_timing(self._h, ctypes.byref(t))
        return {n: getattr(t, n) for n, _ in _lib.Timing._fields_}

    def debug_fetch(self, n_positions):
        link = np.zeros(n_positions + 8, np.uint16)
        m2 = np.zeros(n_positions + 8, np.uint32)
        mq = np.zeros(n_positions + 8, np.uint32)
        tok = np.zeros(n_positions + 8, np.uint32)
        nt = ctypes.c_size_t(0)
        _lib.check(self._L.szl_engine_debug_fetch(self._h, link.ctypes.data, m2.ctypes.data, mq.ctypes.data, n_positions,
                                                  tok.ctypes.data, n_positions, ctypes.byref(nt)), "debug_fetch")
        return link[:n_positions], m2[:n_positions], mq[:n_positions], tok[:nt.value]

    def debug_match_mode(self, mode=99):
        """Force the stage-B form (0 full, 1 on demand, 2 pilot, -1 default); returns True if the last call ran on demand."""
        return bool(self._L.szl_engine_debug_match_mode(self._h, mode))

    def debug_blocks(self, cap=1 << 16):
        rows = np.zeros(8 * cap, np.uint64)
        nr = ctypes.c_size_t(0)
        _lib.check(self._L.szl_engine_debug_blocks(self._h, rows.ctypes.data, cap, ctypes.byref(nr)), "debug_blocks")
        names = ("type", "last", "ntokens", "bit_start", "opt_len", "static_len", "stored_len", "hdr_bits")
        return [dict(zip(names, (int(v) for v in rows[8 * i:8 * i + 8]))) for i in range(min(nr.value, cap))]

    def inflate(self, buffers, out_sizes, nowrap=True, crc32=False, adler32=False):
        """Inflate independent streams (host buffers). out_sizes: capacity of each output region.
        Returns [(Result, consumed)]."""
        bufs = [np.ascontiguousarray(np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b, dtype=np.uint8) for b in buffers]
        arr = (_lib.Stream * len(bufs))()
        io = oo = 0
        for i, (b, cap) in enumerate(zip(bufs, out_sizes)):
            arr[i].in_off, arr[i].in_len, arr[i].out_off, arr[i].out_cap = io, b.size, oo, cap
            io += (b.size + 3) & ~3
            oo += (cap + 3) & ~3
        hin = np.zeros(io + 8, dtype=np.uint8)
        for s, b in zip(arr, bufs):
            hin[s.in_off:s.in_off + s.in_len] = b
        hout = np.zeros(oo + 8, dtype=np.uint8)
        flags = (_lib.F_NOWRAP if nowrap else 0) | (_lib.F_CRC32 if crc32 else 0) | (_lib.F_ADLER32 if adler32 else 0)
        _lib.check(self._L.szl_inflate_batch_host(self._h, hin.ctypes.data, hout.ctypes.data, arr, len(bufs), flags), "szl_inflate_batch_host")
        return [(Result(hout[s.out_off:s.out_off + s.out_len].tobytes(), s.crc32, s.adler32, s.status), int(s.in_consumed)) for s in arr]


def deflate_multi(buffers, devices, level=6, strategy=0, nowrap=True, crc32=False, adler32=False):
    """Compress independent host buffers on several devices (szl_deflate_batch_multi_host): contiguous groups of streams, one
    host thread + engine per device, no exchange between the groups.  Returns [Result] in input order."""
    L = _lib.lib()
    bufs = [np.ascontiguousarray(np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b, dtype=np.uint8) for b in buffers]
    arr, in_total, out_total = Engine.layout([b.size for b in bufs], nowrap)
    hin = np.empty(in_total + 8, dtype=np.uint8)
    for s, b in zip(arr, bufs):
        hin[s.in_off:s.in_off + s.in_len] = b
    hout = np.zeros(out_total + 8, dtype=np.uint8)
    flags = (_lib.F_NOWRAP if nowrap else 0) | (_lib.F_CRC32 if crc32 else 0) | (_lib.F_ADLER32 if adler32 else 0)
    devs = (ctypes.c_int * len(devices))(*devices)
    _lib.check(L.szl_deflate_batch_multi_host(devs, len(devices), hin.ctypes.data, hout.ctypes.data, arr, len(bufs), level, strategy, flags),
               "szl_deflate_batch_multi_host")
    return [Result(hout[s.out_off:s.out_off + s.out_len].tobytes(), s.crc32, s.adler32, s.status) for s in arr]


def inflate_multi(buffers, out_sizes, devices, nowrap=True, crc32=False):
    """Inflate independent streams on several devices (szl_inflate_batch_multi_host). Returns [(Result, consumed)]."""
    L = _lib.lib()
    bufs = [np.ascontiguousarray(np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b, dtype=np.uint8) for b in buffers]
    arr = (_lib.Stream * len(bufs))()
    io = oo = 0
    for i, (b, cap) in enumerate(zip(bufs, out_sizes)):
        arr[i].in_off, arr[i].in_len, arr[i].out_off, arr[i].out_cap = io, b.size, oo, cap
        io += (b.size + 3) & ~3
        oo += (cap + 3) & ~3
    hin = np.zeros(io + 8, dtype=np.uint8)
    for s, b in zip(arr, bufs):
        hin[s.in_off:s.in_off + s.in_len] = b
    hout = np.zeros(oo + 8, dtype=np.uint8)
    flags = (_lib.F_NOWRAP if nowrap else 0) | (_lib.F_CRC32 if crc32 else 0)
    devs = (ctypes.c_int * len(devices))(*devices)
    _lib.check(L.szl_inflate_batch_multi_host(devs, len(devices), hin.ctypes.data, hout.ctypes.data, arr, len(bufs), flags),
               "szl_inflate_batch_multi_host")
    return [(Result(hout[s.out_off:s.out_off + s.out_len].tobytes(), s.crc32, s.adler32, s.status), int(s.in_consumed)) for s in arr]


def deflate_stream_multi_device(d_in_ptrs, d_out_ptr, devices, stream, level=6, strategy=0, flags=_lib.F_NOWRAP):
    """ONE stream over several devices, input already resident on each of them (szl_deflate_stream_multi_device): d_in_ptrs[g] = device
    pointer of the input arena on devices[g], d_out_ptr = output arena on devices[0]; `stream` = a one-element Stream array."""
    L = _lib.lib()
    devs = (ctypes.c_int * len(devices))(*devices)
    ptrs = (ctypes.c_void_p * len(devices))(*d_in_ptrs)
    _lib.check(L.szl_deflate_stream_multi_device(devs, len(devices), ptrs, d_out_ptr, stream, level, strategy, flags), "szl_deflate_stream_multi_device")
