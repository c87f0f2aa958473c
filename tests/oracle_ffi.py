"""ctypes binding of oracle/libszl_oracle.so — the CPU restatement of the reference (checker only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_DIR, "libszl_oracle.so")
_lib = None


class BlockInfo(ctypes.Structure):
    _fields_ = [("first_token", ctypes.c_int64), ("ntokens", ctypes.c_int32), ("type", ctypes.c_int32),
                ("last", ctypes.c_int32), ("stored_offset", ctypes.c_int32), ("stored_len", ctypes.c_int32),
                ("opt_len", ctypes.c_int32), ("static_len", ctypes.c_int32), ("bit_start", ctypes.c_int64)]


class Trace(ctypes.Structure):
    _fields_ = [("tok", ctypes.c_void_p), ("tok_cap", ctypes.c_size_t), ("tok_n", ctypes.c_size_t),
                ("blk", ctypes.c_void_p), ("blk_cap", ctypes.c_size_t), ("blk_n", ctypes.c_size_t)]


class FastParams(ctypes.Structure):
    _fields_ = [("nice", ctypes.c_int), ("max_chain", ctypes.c_int), ("max_lazy", ctypes.c_int), ("strategy", ctypes.c_int)]


class Params(ctypes.Structure):
    _fields_ = [("good", ctypes.c_int), ("nice", ctypes.c_int), ("max_chain", ctypes.c_int), ("strategy", ctypes.c_int)]


def build(force=False):
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = ctypes.CDLL(_SO)
    vp, sz, i32, i64, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32
    L.szo_crc32.restype = u32; L.szo_crc32.argtypes = [u32, vp, sz]
    L.szo_adler32.restype = u32; L.szo_adler32.argtypes = [u32, vp, sz]
    L.szo_deflate_oneshot.restype = i64
    L.szo_deflate_oneshot.argtypes = [vp, sz, i32, i32, i32, i32, vp, sz, vp]
    L.szo_inflate_oneshot.restype = i64
    L.szo_inflate_oneshot.argtypes = [vp, sz, i32, vp, sz, ctypes.POINTER(sz)]
    L.szo_quirk_sets_seen.restype = i32; L.szo_quirk_sets_seen.argtypes = [i32]
    L.szo_inflate_probe.restype = i64
    L.szo_inflate_probe.argtypes = [vp, sz, i32, vp, sz, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    for name, res, args in [
        ("szo_deflater_new", vp, [i32, i32]), ("szo_deflater_free", None, [vp]), ("szo_deflater_reset", None, [vp]),
        ("szo_deflater_set_level", i32, [vp, i32]), ("szo_deflater_set_strategy", None, [vp, i32]),
        ("szo_deflater_set_dictionary", i32, [vp, vp, i32]), ("szo_deflater_set_input", i32, [vp, vp, i32]),
        ("szo_deflater_flush", None, [vp]), ("szo_deflater_finish", None, [vp]),
        ("szo_deflater_deflate", i32, [vp, vp, i32]), ("szo_deflater_needs_input", i32, [vp]),
        ("szo_deflater_is_finished", i32, [vp]), ("szo_deflater_total_in", i64, [vp]),
        ("szo_deflater_total_out", i64, [vp]), ("szo_deflater_adler", u32, [vp]), ("szo_deflater_set_trace", None, [vp, vp]),
        ("szo_inflater_new", vp, [i32]), ("szo_inflater_free", None, [vp]), ("szo_inflater_reset", None, [vp]),
        ("szo_inflater_set_input", i32, [vp, vp, i32]), ("szo_inflater_set_dictionary", i32, [vp, vp, i32]),
        ("szo_inflater_inflate", i32, [vp, vp, i32]), ("szo_inflater_needs_input", i32, [vp]),
        ("szo_inflater_needs_dictionary", i32, [vp]), ("szo_inflater_is_finished", i32, [vp]),
        ("szo_inflater_remaining_input", i32, [vp]), ("szo_inflater_total_in", i64, [vp]),
        ("szo_inflater_total_out", i64, [vp]), ("szo_inflater_adler", u32, [vp]),
        ("szm_level_params", i32, [i32, vp]), ("szm_links", None, [vp, sz, vp, sz, vp]),
        ("szm_match_tables", None, [vp, sz, sz, vp, vp, vp, vp]),
        ("szm_parse", sz, [vp, sz, sz, vp, vp, vp, vp, vp, vp]),
        ("szm_parse_ranges", sz, [vp, sz, sz, vp, vp, vp, vp, sz, vp, vp]),
        ("szm_block_table", sz, [vp, sz, i32, vp, vp, vp]), ("szm_base_of", i64, [i64]),
        ("szo_dotnet_random_bytes", None, [ctypes.c_int32, vp, sz]),
        ("szm_fast_level_params", i32, [i32, vp]), ("szm_base_of_fast", i64, [i64]),
        ("szm_fast_parse", sz, [vp, sz, sz, vp, vp, vp, vp]),
        ("szm_fast_parse_fixpoint", sz, [vp, sz, sz, vp, vp, sz, vp, vp, vp]),
        ("szm_parse_needed", sz, [vp, sz, sz, vp, vp, vp, vp, vp]),
        ("szm_first_node", sz, [vp, sz, vp, vp, vp, vp, sz, sz]),
        ("szm_links4", None, [vp, sz, vp, vp, vp]), ("szm_match_tables_c4", None, [vp, sz, sz, sz, vp, vp, vp, vp, vp, vp, vp]),
        ("szm_match_tables_k6", None, [vp, sz, sz, sz, vp, vp, vp, vp, vp, vp, vp]),
        ("szm_match_tables_k7", None, [vp, sz, sz, sz, vp, ctypes.c_int, vp, vp, vp, vp]),
        ("szm_lazy_eval_set", sz, [vp, sz, sz, vp, vp, vp, vp, sz, sz, vp]),
    ]:
        f = getattr(L, name); f.restype = res; f.argtypes = args
    _lib = L
    return L


def _buf(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    return a


def dotnet_random_bytes(seed, n):
    """new System.Random(seed).NextBytes(new byte[n]) — T/TestSupport/Utils.cs:79-85 GetDummyBytes."""
    out = np.empty(n, dtype=np.uint8)
    lib().szo_dotnet_random_bytes(seed, out.ctypes.data, n)
    return out


class Deflater:
    """The oracle's Deflater object (C/Deflater.cs member set) for call-pattern tests."""

    def __init__(self, level=-1, nowrap=False):
        self.L = lib()
        self.h = self.L.szo_deflater_new(level, 1 if nowrap else 0)
        if not self.h:
            raise ValueError("level")
        self._keep = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.szo_deflater_free(self.h)
            self.h = None

    def set_input(self, data):
        self._keep = np.ascontiguousarray(_buf(data))
        return self.L.szo_deflater_set_input(self.h, self._keep.ctypes.data, self._keep.size)

    def flush(self): self.L.szo_deflater_flush(self.h)
    def finish(self): self.L.szo_deflater_finish(self.h)
    def reset(self): self.L.szo_deflater_reset(self.h)
    def set_level(self, lv): return self.L.szo_deflater_set_level(self.h, lv)
    def set_strategy(self, s): self.L.szo_deflater_set_strategy(self.h, s)
    def set_dictionary(self, d):
        a = np.ascontiguousarray(_buf(d)); return self.L.szo_deflater_set_dictionary(self.h, a.ctypes.data, a.size)
    @property
    def needs_input(self): return bool(self.L.szo_deflater_needs_input(self.h))
    @property
    def finished(self): return bool(self.L.szo_deflater_is_finished(self.h))
    @property
    def total_in(self): return self.L.szo_deflater_total_in(self.h)
    @property
    def total_out(self): return self.L.szo_deflater_total_out(self.h)
    @property
    def adler(self): return self.L.szo_deflater_adler(self.h)

    def deflate(self, n):
        out = np.empty(max(n, 1), dtype=np.uint8)
        k = self.L.szo_deflater_deflate(self.h, out.ctypes.data, n)
        if k < 0:
            raise RuntimeError(k)
        return out[:k].tobytes()


class Inflater:
    def __init__(self, nowrap=False):
        self.L = lib()
        self.h = self.L.szo_inflater_new(1 if nowrap else 0)
        self._keep = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.szo_inflater_free(self.h)
            self.h = None

    def set_input(self, data):
        self._keep = np.ascontiguousarray(_buf(data))
        return self.L.szo_inflater_set_input(self.h, self._keep.ctypes.data, self._keep.size)

    def inflate(self, n):
        out = np.empty(max(n, 1), dtype=np.uint8)
        k = self.L.szo_inflater_inflate(self.h, out.ctypes.data, n)
        return k, out[:max(k, 0)].tobytes()

    def reset(self): self.L.szo_inflater_reset(self.h)
    @property
    def needs_input(self): return bool(self.L.szo_inflater_needs_input(self.h))
    @property
    def finished(self): return bool(self.L.szo_inflater_is_finished(self.h))
    @property
    def remaining_input(self): return self.L.szo_inflater_remaining_input(self.h)
    @property
    def total_in(self): return self.L.szo_inflater_total_in(self.h)
    @property
    def total_out(self): return self.L.szo_inflater_total_out(self.h)
    @property
    def adler(self): return self.L.szo_inflater_adler(self.h)


def stream_deflate(data, level=6, nowrap=True, chunk=None, flush_every=None, out_chunk=512, strategy=0, final_flush=False):
    """DeflaterOutputStream.Write/Flush/Finish call pattern (CS/DeflaterOutputStream.cs:506,388,100) on the oracle."""
    d = Deflater(level, nowrap)
    d.set_strategy(strategy)
    a = _buf(data)
    out = bytearray()
    chunk = chunk or max(a.size, 1)
    pos, since = 0, 0
    while pos < a.size:
        c = a[pos:pos + chunk]
        d.set_input(c)
        while not d.needs_input:
            b = d.deflate(out_chunk)
            if not b:
                break
            out += b
        pos += c.size
        since += c.size
        if flush_every and since >= flush_every and pos < a.size:
            d.flush()
            while True:
                b = d.deflate(out_chunk)
                if not b:
                    break
                out += b
            since = 0
    if final_flush:
        d.flush()
        while True:
            b = d.deflate(out_chunk)
            if not b:
                break
            out += b
    d.finish()
    while not d.finished:
        b = d.deflate(out_chunk)
        if not b:
            break
        out += b
    assert d.finished
    return bytes(out), d.total_in, d.total_out


def crc32(data, value=0):
    a = _buf(data)
    return lib().szo_crc32(value, a.ctypes.data, a.size)


def adler32(data, value=1):
    a = _buf(data)
    return lib().szo_adler32(value, a.ctypes.data, a.size)


def out_bound(n):
    return n + n // 3 + 4096


def deflate(data, level=6, nowrap=True, strategy=0, flush=False, trace=False):
    """One-shot Write[+Flush]+Finish through the oracle Deflater. Returns bytes (and trace dict)."""
    a = _buf(data)
    cap = out_bound(a.size)
    out = np.empty(cap, dtype=np.uint8)
    tr = None
    if trace:
        tok = np.zeros(a.size + 8, dtype=np.uint32)
        blk = (BlockInfo * (a.size // 8 + 64))()
        tr = Trace(tok.ctypes.data, tok.size, 0, ctypes.addressof(blk), len(blk), 0)
    n = lib().szo_deflate_oneshot(a.ctypes.data, a.size, level, 1 if nowrap else 0, strategy, 1 if flush else 0,
                                  out.ctypes.data, cap, ctypes.byref(tr) if tr is not None else None)
    if n < 0:
        raise RuntimeError("oracle deflate failed: %d" % n)
    res = out[:n].tobytes()
    if trace:
        blocks = [{f: getattr(blk[i], f) for f, _ in BlockInfo._fields_} for i in range(tr.blk_n)]
        return res, {"tokens": tok[:tr.tok_n].copy(), "blocks": blocks}
    return res


def inflate(data, nowrap=True, max_out=None):
    """Returns (status_or_len, bytes, consumed). Negative status = SZO_ERR_* / -102 unexpected EOF."""
    a = _buf(data)
    cap = max_out if max_out is not None else max(1 << 16, a.size * 1100 + 1024)
    out = np.empty(cap, dtype=np.uint8)
    cons = ctypes.c_size_t(0)
    n = lib().szo_inflate_oneshot(a.ctypes.data, a.size, 1 if nowrap else 0, out.ctypes.data, cap, ctypes.byref(cons))
    return n, out[:max(n, 0)].tobytes(), cons.value


def inflate_probe(data, nowrap=True, max_out=1 << 20):
    """(status_or_len, delivered_bytes, consumed): one byte per Inflate() call, so `delivered_bytes` is the longest prefix a
    caller of the reference can have received before the exception that ends a corrupt stream."""
    a = _buf(data)
    out = np.empty(max_out, dtype=np.uint8)
    cons, prod = ctypes.c_size_t(0), ctypes.c_size_t(0)
    lib().szo_quirk_sets_seen(1)
    n = lib().szo_inflate_probe(a.ctypes.data, a.size, 1 if nowrap else 0, out.ctypes.data, max_out, ctypes.byref(cons), ctypes.byref(prod))
    inflate_probe.quirk_sets = lib().szo_quirk_sets_seen(1)
    return n, out[:prod.value].tobytes(), cons.value


class Model:
    """Stage-by-stage CPU model of the parallel decomposition (oracle/szl_model.c)."""

    def __init__(self, data, level=6, strategy=0, seg_ends=None):
        self.L = lib()
        self.d = np.ascontiguousarray(_buf(data))
        self.n = self.d.size
        # +pad so d.ctypes is valid for n == 0
        self._dpad = np.concatenate([self.d, np.zeros(8, np.uint8)])
        self.P = Params()
        if self.L.szm_level_params(level, ctypes.byref(self.P)) != 0:
            raise ValueError("level %d is not a DEFLATE_SLOW level" % level)
        self.P.strategy = strategy
        self.seg_ends = np.array(seg_ends if seg_ends is not None else [self.n], dtype=np.uint64)
        self.link = np.zeros(self.n + 8, dtype=np.uint16)
        self.L.szm_links(self._dpad.ctypes.data, self.n, self.seg_ends.ctypes.data, self.seg_ends.size, self.link.ctypes.data)
        self.m2 = np.zeros(self.n + 8, dtype=np.uint32)
        self.mq = np.zeros(self.n + 8, dtype=np.uint32)
        s = 0
        for e in self.seg_ends:
            self.L.szm_match_tables(self._dpad.ctypes.data, s, int(e), self.link.ctypes.data, ctypes.byref(self.P),
                                    self.m2.ctypes.data, self.mq.ctypes.data)
            s = int(e)

    def parse(self, seg_start=0, seg_end=None, R=None):
        seg_end = self.n if seg_end is None else seg_end
        tok = np.zeros(seg_end - seg_start + 8, dtype=np.uint32)
        stats = np.zeros(4, dtype=np.uint64)
        if R is None:
            k = self.L.szm_parse(self._dpad.ctypes.data, seg_start, seg_end, self.link.ctypes.data, self.m2.ctypes.data,
                                 self.mq.ctypes.data, ctypes.byref(self.P), tok.ctypes.data, stats.ctypes.data)
        else:
            k = self.L.szm_parse_ranges(self._dpad.ctypes.data, seg_start, seg_end, self.link.ctypes.data,
                                        self.m2.ctypes.data, self.mq.ctypes.data, ctypes.byref(self.P), R,
                                        tok.ctypes.data, stats.ctypes.data)
        return tok[:k].copy(), stats

    def match_tables_c4(self, kernel_shape=False):
        """(m2, mq, candidates examined) of the chain-compressed walk (link4 + skip4); must equal self.m2 / self.mq"""
        l4 = np.zeros(self.n + 8, np.uint16); s4 = np.zeros(self.n + 8, np.uint16)
        self.L.szm_links4(self._dpad.ctypes.data, self.n, self.link.ctypes.data, l4.ctypes.data, s4.ctypes.data)
        m2 = np.zeros(self.n + 8, np.uint32); mq = np.zeros(self.n + 8, np.uint32)
        steps = np.zeros(1, np.uint64)
        s = 0
        for e in self.seg_ends:
            (self.L.szm_match_tables_k6 if kernel_shape else self.L.szm_match_tables_c4)(self._dpad.ctypes.data, self.n, s, int(e), self.link.ctypes.data, l4.ctypes.data, s4.ctypes.data,
                                       ctypes.byref(self.P), m2.ctypes.data, mq.ctypes.data, steps.ctypes.data)
            s = int(e)
        return m2, mq, int(steps[0])

    def match_tables_k7(self, dist_cap=32512):
        """(m2, mq, candidates examined) of the device form of the compressed walk (no slow routine); must equal self.m2 / self.mq"""
        m2 = np.zeros(self.n + 8, np.uint32); mq = np.zeros(self.n + 8, np.uint32)
        steps = np.zeros(1, np.uint64)
        s = 0
        for e in self.seg_ends:
            self.L.szm_match_tables_k7(self._dpad.ctypes.data, self.n, s, int(e), self.link.ctypes.data, dist_cap, ctypes.byref(self.P),
                                       m2.ctypes.data, mq.ctypes.data, steps.ctypes.data)
            s = int(e)
        return m2, mq, int(steps[0])

    def first_node(self, start, at_least):
        """first clean iteration >= at_least of the parse that starts, clean, at `start`"""
        return int(self.L.szm_first_node(self._dpad.ctypes.data, self.n, self.link.ctypes.data, self.m2.ctypes.data, self.mq.ctypes.data,
                                         ctypes.byref(self.P), start, at_least))

    def block_table(self, tok, finish=True):
        nb_cap = tok.size // 16384 + 4
        first = np.zeros(nb_cap, np.int64); cnt = np.zeros(nb_cap, np.int32); last = np.zeros(nb_cap, np.int32)
        t = np.concatenate([tok, np.zeros(1, np.uint32)])
        nb = self.L.szm_block_table(t.ctypes.data, tok.size, 1 if finish else 0, first.ctypes.data, cnt.ctypes.data, last.ctypes.data)
        return first[:nb], cnt[:nb], last[:nb]


def tree_lengths(freqs, min_codes, max_len):
    """(code lengths, numCodes) the oracle's Tree.BuildTree + BuildLength give one frequency vector (C/DeflaterHuffman.cs:196-329, :475-579)."""
    f = np.ascontiguousarray(np.asarray(freqs, dtype=np.int16))
    out = np.zeros(f.size, dtype=np.uint8)
    nc = ctypes.c_int(0)
    L = lib()
    L.szo_tree_lengths.restype = ctypes.c_int
    L.szo_tree_lengths.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    assert L.szo_tree_lengths(f.ctypes.data, f.size, min_codes, max_len, out.ctypes.data, ctypes.byref(nc)) == 0
    return out, nc.value
