"""ctypes loader for libszl_amd.so (the C ABI of include/szl.h).  Fails loudly if the HIP
library is missing — there is no CPU fallback anywhere in this package."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO = os.path.join(CSRC, "libszl_amd.so")
SO_LAB = os.path.join(CSRC, "libszl_amd_lab.so")   # the same + the laboratory forms of stage B (csrc/Makefile); tests only
_lib = None
_lab = None


class Stream(ctypes.Structure):
    _fields_ = [("in_off", ctypes.c_uint64), ("in_len", ctypes.c_uint64), ("out_off", ctypes.c_uint64),
                ("out_cap", ctypes.c_uint64), ("out_len", ctypes.c_uint64), ("crc32", ctypes.c_uint32),
                ("adler32", ctypes.c_uint32), ("status", ctypes.c_int32), ("reserved", ctypes.c_uint32), ("in_consumed", ctypes.c_uint64)]


class Timing(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("total_ms", "checksum_ms", "links_ms", "match_ms", "parse_ms", "blocks_ms", "encode_ms")] + \
               [(n, ctypes.c_uint64) for n in ("in_bytes", "out_bytes", "tokens", "blocks", "ranges_unmerged", "fallback_walks")] + \
               [("inflate_ms", ctypes.c_float), ("pilot_ms", ctypes.c_float), ("links_guard_trips", ctypes.c_uint32)]


F_NOWRAP, F_CRC32, F_ADLER32, F_SYNC_FLUSH_BEFORE_FINISH, F_GZIP = 1, 2, 4, 8, 16


def build(force=False):
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC])
    return SO


def lib():
    global _lib
    if _lib is None:
        _lib = _load(SO)
    return _lib


def lab_lib():
    """The laboratory build (measured alternatives of stage B that lost: chain compression, the ring, the bucket-order search).
    tests/test_gpu_stage_b_forms.py swaps it in for the product library to keep those forms bit-exact; nothing else loads it."""
    global _lab
    if _lab is None:
        _lab = _load(SO_LAB)
    return _lab


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError("%s is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "sharpziplib_amd has no CPU fallback" % os.path.basename(path))
    L = ctypes.CDLL(path)
    vp, sz, i32, i64, u32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64
    sig = {
        "szl_strerror": (ctypes.c_char_p, [i32]), "szl_last_error": (ctypes.c_char_p, []),
        "szl_device_count": (i32, []), "szl_set_device": (i32, [i32]),
        "szl_crc32": (i32, [u32, vp, sz, ctypes.POINTER(u32)]), "szl_adler32": (i32, [u32, vp, sz, ctypes.POINTER(u32)]),
        "szl_deflater_create": (vp, [i32, i32]), "szl_deflater_destroy": (None, [vp]), "szl_deflater_reset": (i32, [vp]),
        "szl_deflater_set_level": (i32, [vp, i32]), "szl_deflater_get_level": (i32, [vp]),
        "szl_deflater_set_strategy": (i32, [vp, i32]), "szl_deflater_set_dictionary": (i32, [vp, vp, i32]),
        "szl_deflater_set_input": (i32, [vp, vp, i32]), "szl_deflater_flush": (i32, [vp]), "szl_deflater_finish": (i32, [vp]),
        "szl_deflater_deflate": (i32, [vp, vp, i32]), "szl_deflater_needs_input": (i32, [vp]),
        "szl_deflater_deflate_view": (i32, [vp, ctypes.POINTER(vp), ctypes.POINTER(i64)]),
        "szl_deflater_is_finished": (i32, [vp]), "szl_deflater_total_in": (i64, [vp]), "szl_deflater_total_out": (i64, [vp]),
        "szl_deflater_adler": (u32, [vp]), "szl_deflater_enable_crc32": (i32, [vp, i32]), "szl_deflater_caller_drains": (i32, [vp, i32]), "szl_deflater_debug_pipe_parts": (i32, [vp]), "szl_deflater_crc32": (u32, [vp]),
        "szl_deflate_bound": (u64, [u64]), "szl_engine_create": (vp, []), "szl_engine_destroy": (None, [vp]),
        "szl_deflate_batch_device": (i32, [vp, vp, vp, vp, sz, i32, i32, ctypes.c_uint, vp]),
        "szl_deflate_batch_host": (i32, [vp, vp, vp, vp, sz, i32, i32, ctypes.c_uint]),
        "szl_deflate_batch_multi_host": (i32, [vp, i32, vp, vp, vp, sz, i32, i32, ctypes.c_uint]),
        "szl_deflate_stream_multi_device": (i32, [vp, i32, vp, vp, vp, i32, i32, ctypes.c_uint]),
        "szl_multi_release": (i32, []), "szl_trim": (i32, []),
        "szl_inflate_batch_multi_host": (i32, [vp, i32, vp, vp, vp, sz, ctypes.c_uint]),
        "szl_engine_last_timing": (i32, [vp, vp]),
        "szl_engine_debug_fetch": (i32, [vp, vp, vp, vp, sz, vp, sz, ctypes.POINTER(sz)]),
        "szl_engine_debug_blocks": (i32, [vp, vp, sz, ctypes.POINTER(sz)]),
        "szl_engine_debug_match_mode": (i32, [vp, i32]),
        "szl_debug_set": (i32, [ctypes.c_char_p, i32]), "szl_debug_host_copy": (i32, [vp, vp, sz]),
        "szl_engine_debug_workspace": (u64, [vp]),
        "szl_engine_debug_par_jobs": (ctypes.c_uint32, [vp]),
        "szl_engine_debug_text_form": (ctypes.c_int, [vp]),
        "szl_inflater_debug_bulk_calls": (ctypes.c_uint32, [vp]), "szl_inflater_debug_times": (i32, [vp, vp]),
        "szl_debug_stored_layout": (i32, [vp, sz, i32, vp, sz, ctypes.POINTER(sz)]),
        "szl_debug_tree_lengths": (i32, [vp, i32, i32, i32, i32, vp, vp]),
        "szl_inflater_create": (vp, [i32]), "szl_inflater_destroy": (None, [vp]), "szl_inflater_reset": (i32, [vp]),
        "szl_inflater_set_input": (i32, [vp, vp, i32]), "szl_inflater_set_dictionary": (i32, [vp, vp, i32]),
        "szl_inflater_inflate": (i32, [vp, vp, i32]), "szl_inflater_needs_input": (i32, [vp]),
        "szl_inflater_needs_dictionary": (i32, [vp]), "szl_inflater_is_finished": (i32, [vp]),
        "szl_inflater_remaining_input": (i32, [vp]), "szl_inflater_total_in": (i64, [vp]),
        "szl_inflater_total_out": (i64, [vp]), "szl_inflater_adler": (u32, [vp]),
        "szl_inflater_enable_crc32": (i32, [vp, i32]), "szl_inflater_crc32": (u32, [vp]), "szl_inflater_detach_input": (i32, [vp]), "szl_inflater_expect_more": (i32, [vp, i32]),
        "szl_host_alloc": (vp, [sz]), "szl_host_free": (None, [vp]), "szl_host_register": (i32, [vp, sz]), "szl_host_unregister": (i32, [vp]),
        "szl_inflate_batch_device": (i32, [vp, vp, vp, vp, sz, ctypes.c_uint, vp]),
        "szl_inflate_batch_host": (i32, [vp, vp, vp, vp, sz, ctypes.c_uint]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    return L


EXPORTS = None  # filled lazily by tests: names declared in include/szl.h


class SzlError(RuntimeError):
    def __init__(self, status, where=""):
        L = lib()
        msg = L.szl_strerror(status).decode()
        detail = L.szl_last_error().decode()
        super().__init__("%s: %s (%d)%s" % (where, msg, status, (" — " + detail) if detail else ""))
        self.status = status


def check(status, where=""):
    if status < 0:
        raise SzlError(status, where)
    return status
