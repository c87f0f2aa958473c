"""Seeded synthetic corpora: workload plumbing for tests and bench.py.

`generate(kind, seed, offset, length)` returns bytes [offset, offset+length) of the infinite
stream (kind, seed); see szl_corpus.c for the text models.  numpy helpers cover the adversarial
parity sets (uniform random, zeros, 4-symbol alphabet, period-10 text).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libszl_corpus.so")
KINDS = {"dickens": 0, "enwik": 1, "logs": 2}
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "szl_corpus.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-pthread", "-o", _SO, src, "-lm"])


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.szc_generate.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_size_t,
                                      ctypes.c_void_p, ctypes.c_int]
        _lib.szc_generate.restype = ctypes.c_int
    return _lib


def generate(kind, seed, offset, length, threads=None):
    """numpy uint8 array with bytes [offset, offset+length) of corpus stream (kind, seed)."""
    lib = _load()
    k = KINDS[kind] if isinstance(kind, str) else int(kind)
    out = np.empty(length, dtype=np.uint8)
    if length:
        rc = lib.szc_generate(k, seed, offset, length, out.ctypes.data, threads or (os.cpu_count() or 1))
        if rc != 0:
            raise ValueError("bad corpus kind %r" % (kind,))
    return out


def _splitmix_bytes(seed, n):
    """Uniform random bytes from splitmix64 (vectorised)."""
    m = (n + 7) // 8
    with np.errstate(over="ignore"):
        i = np.arange(1, m + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.view(np.uint8)[:n].copy()


def random_bytes(n, seed=1):
    return _splitmix_bytes(seed, n)


def zeros(n):
    return np.zeros(n, dtype=np.uint8)


def four_symbol(n, seed=2):
    return (np.frombuffer(b"acgt", dtype=np.uint8))[_splitmix_bytes(seed, n) & 3]


def period10(n):
    return np.resize(np.frombuffer(b"abcdefghij", dtype=np.uint8), n).copy()


def mixed(n, seed=3):
    """text / random / zeros / logs / low-entropy stripes of irregular lengths (exercises every block type)."""
    out = np.empty(n, dtype=np.uint8)
    r = _splitmix_bytes(seed ^ 0xABCDEF, 4096).astype(np.int64)
    pos, i = 0, 0
    while pos < n:
        ln = int(1000 + (r[i % 4096] * 257 + r[(i + 1) % 4096]) % 70000)
        ln = min(ln, n - pos)
        k = i % 6
        if k == 0:
            out[pos:pos + ln] = generate("dickens", seed + i, 0, ln, threads=1)
        elif k == 1:
            out[pos:pos + ln] = random_bytes(ln, seed + i)
        elif k == 2:
            out[pos:pos + ln] = 0
        elif k == 3:
            out[pos:pos + ln] = generate("logs", seed + i, 0, ln, threads=1)
        elif k == 4:
            out[pos:pos + ln] = four_symbol(ln, seed + i)
        else:
            out[pos:pos + ln] = period10(ln)
        pos += ln
        i += 1
    return out
