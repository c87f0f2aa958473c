"""Minimal ctypes binding of the HIP runtime for tests that need raw device buffers (device-resident C-ABI calls) without
dragging torch into the process.  libszl_amd.so links the same runtime, so the allocations live in the same context."""
import ctypes
import os

_hip = None


def hip():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64.so not found")
        _hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        _hip.hipFree.argtypes = [ctypes.c_void_p]
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        _hip.hipDeviceSynchronize.argtypes = []
    return _hip


H2D, D2H = 1, 2


class DevBuf:
    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        self.n = nbytes
        rc = hip().hipMalloc(ctypes.byref(self.ptr), nbytes)
        if rc != 0:
            raise MemoryError("hipMalloc(%d) failed: %d" % (nbytes, rc))

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def addr(self):
        return self.ptr.value

    def upload(self, off, arr):
        rc = hip().hipMemcpy(ctypes.c_void_p(self.ptr.value + off), arr.ctypes.data, arr.nbytes, H2D)
        assert rc == 0, rc

    def download(self, off, n):
        import numpy as np
        out = np.empty(n, np.uint8)
        rc = hip().hipMemcpy(out.ctypes.data, ctypes.c_void_p(self.ptr.value + off), n, D2H)
        assert rc == 0, rc
        return out

    def fill(self, off, n, value):
        rc = hip().hipMemset(ctypes.c_void_p(self.ptr.value + off), value, n)
        assert rc == 0, rc
