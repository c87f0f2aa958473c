"""gfxsim.cpu — a functional wave64 interpreter for the gfx950 instructions hipcc emits for this repository's kernels
(compiler output and the hand-written inline assembly alike).  One Wave object = one wavefront: 64 lanes of VGPRs as numpy
rows, SGPRs as Python ints, exec / vcc / scc, a program counter into a Module (gfxsim.asm).  No timing: s_waitcnt and s_nop
do nothing, memory is coherent at once; what it gives is the *values* the machine code computes, so the product's kernels can
be checked against the oracle in a container without a GPU, plus wave-instruction counts per class.

Semantics that are hardware behaviour rather than ISA text follow what round 4 measured on the device (see tools/wavesim.py):
ds_bpermute reads 0 from inactive source lanes, d16 loads clear the other half (SRAM ECC), one ds_wrxchg serves its lanes in
ascending order.

Test infrastructure only.
"""
import os

import numpy as np

from .asm import AsmError, f32_bits

np.seterr(over="ignore", invalid="ignore", divide="ignore")

U32, U64, I32, I64, U8, U16, I16, F32 = np.uint32, np.uint64, np.int32, np.int64, np.uint8, np.uint16, np.int16, np.float32
M32, M64 = 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF
LANE = np.arange(64, dtype=np.int64)
LANE_U64 = np.arange(64, dtype=U64)
ONE64 = U64(1)
BITS64 = ONE64 << LANE_U64
AR = {n: np.arange(n, dtype=np.int64) for n in (1, 2, 4, 8, 12, 16)}
SHARED_HI, PRIVATE_HI = 0x00020000, 0x00030000      # flat apertures (upper address dword), far from any arena address
CODE_HI = 0x00040000


class SimError(Exception):
    pass


class SimTrap(SimError):
    pass


def mask_to_bool(m):
    return ((U64(m) >> LANE_U64) & ONE64).astype(bool)


def bool_to_mask(b):
    return int(np.bitwise_or.reduce(np.where(b, BITS64, U64(0))))


def sx(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


# ------------------------------------------------------------------------------------------------------------------
# operand access: closures built once per instruction
# ------------------------------------------------------------------------------------------------------------------
def _const32(o):
    if o[0] == "k":
        return o[1] & M32
    if o[0] == "f":
        return f32_bits(o[1])
    return None


def ssrc(o, width=32):
    """SALU read -> Python int"""
    k = o[0]
    if k == "s":
        n = o[1]
        if width == 64:
            return lambda w: w.S[n] | (w.S[n + 1] << 32)
        return lambda w: w.S[n]
    if k == "k":
        v = o[1] & (M64 if width == 64 else M32)
        return lambda w: v
    if k == "f":
        v = f32_bits(o[1])
        return lambda w: v
    if k == "scc":
        return lambda w: w.scc
    if k == "aperture":
        hi = {"shared": SHARED_HI, "private": PRIVATE_HI}.get(o[1])
        if hi is None:
            raise AsmError("aperture limit registers are not modelled")
        v = hi << 32
        return (lambda w: v) if width == 64 else (lambda w: 0)
    if k == "sym":
        name, kind, add = o[1], o[2], o[3]
        def g(w, name=name, kind=kind, add=add):
            return w.rt.reloc(w.mod, name, kind, add, w.pc - 1)
        return g
    if k == "label":
        name = o[1]
        return lambda w: w.rt.absolute(w.mod, name)
    raise AsmError("scalar source not understood: %r" % (o,))


def sdst(o, width=32):
    """SALU write"""
    k = o[0]
    if k == "s":
        n = o[1]
        if n in (126, 127):
            if width == 64:
                return lambda w, v: w.set_exec(v & M64)
            if n == 126:
                return lambda w, v: w.set_exec((w.exec & ~M32) | (v & M32))
            return lambda w, v: w.set_exec((w.exec & M32) | ((v & M32) << 32))
        if width == 64:
            def p(w, v):
                w.S[n] = v & M32
                w.S[n + 1] = (v >> 32) & M32
            return p
        def p(w, v):
            w.S[n] = v & M32
        return p
    if k == "null":
        return lambda w, v: None
    raise AsmError("scalar destination not understood: %r" % (o,))


def vsrc(o, width=32):
    """VALU read -> uint32 (or uint64) numpy array of 64, or a numpy scalar (broadcasts)"""
    k = o[0]
    if k == "mod":
        inner, mods = vsrc(o[1], width), o[2]
        if "sext" in mods:
            return inner          # applied by the SDWA select
        def g(w):
            x = np.asarray(inner(w))
            f = x.view(F32)
            if "abs" in mods:
                f = np.abs(f)
            if "neg" in mods:
                f = -f
            return np.asarray(f, dtype=F32).view(U32)
        return g
    if k == "v":
        n = o[1]
        if width == 64:
            return lambda w: w.V[n].astype(U64) | (w.V[n + 1].astype(U64) << U64(32))
        return lambda w: w.V[n]
    if k == "a":
        n = o[1]
        return lambda w: w.A[n]
    if k == "s":
        n = o[1]
        if width == 64:
            return lambda w: U64(w.S[n] | (w.S[n + 1] << 32))
        return lambda w: U32(w.S[n])
    if k == "k":
        v = U64(o[1] & M64) if width == 64 else U32(o[1] & M32)
        return lambda w: v
    if k == "f":
        v = U32(f32_bits(o[1]))
        return lambda w: v
    if k == "scc":
        return lambda w: U32(w.scc)
    if k in ("aperture", "sym", "label"):
        s = ssrc(o, width)
        return (lambda w: U64(s(w))) if width == 64 else (lambda w: U32(s(w) & M32))
    raise AsmError("vector source not understood: %r" % (o,))


def vdst(o, width=32):
    k = o[0]
    if k in ("v", "a"):
        n = o[1]
        file = "V" if k == "v" else "A"
        if width == 64:
            def p(w, val):
                R = getattr(w, file)
                val = np.asarray(val, dtype=U64)
                lo, hi = (val & U64(M32)).astype(U32), (val >> U64(32)).astype(U32)
                if w.full:
                    R[n][:] = lo
                    R[n + 1][:] = hi
                else:
                    np.copyto(R[n], lo, where=w.em)
                    np.copyto(R[n + 1], hi, where=w.em)
            return p
        def p(w, val):
            R = getattr(w, file)
            if w.full:
                R[n][:] = val
            else:
                np.copyto(R[n], val, where=w.em, casting="unsafe")
        return p
    raise AsmError("vector destination not understood: %r" % (o,))


def _i(x):
    return np.asarray(x).astype(I32)


def _u(x):
    return np.asarray(x).astype(U32)


def _w(x):
    return np.asarray(x).astype(U64)


def _f(x):
    return np.asarray(x, dtype=U32).view(F32)


def _fu(x):
    return np.asarray(x, dtype=F32).view(U32)


def _popcount(x):
    return np.bitwise_count(np.asarray(x)).astype(U32)


def _ffbl(a):
    a = np.asarray(a, dtype=U32)
    low = a & (~a + U32(1))
    r = np.bitwise_count(low - U32(1)).astype(U32)
    return np.where(a == 0, U32(M32), r)


def _ffbh(a):
    a = np.asarray(a, dtype=U32)
    x = a.astype(np.float64)
    lg = np.floor(np.log2(np.maximum(x, 1.0))).astype(U32)
    return np.where(a == 0, U32(M32), U32(31) - lg)


def _bfrev(a):
    a = np.asarray(a, dtype=U32)
    a = ((a >> U32(1)) & U32(0x55555555)) | ((a & U32(0x55555555)) << U32(1))
    a = ((a >> U32(2)) & U32(0x33333333)) | ((a & U32(0x33333333)) << U32(2))
    a = ((a >> U32(4)) & U32(0x0F0F0F0F)) | ((a & U32(0x0F0F0F0F)) << U32(4))
    a = ((a >> U32(8)) & U32(0x00FF00FF)) | ((a & U32(0x00FF00FF)) << U32(8))
    return (a >> U32(16)) | (a << U32(16))


def _mul24u(a, b):
    return ((_w(a) & U64(0xFFFFFF)) * (_w(b) & U64(0xFFFFFF)))


def _s24(a):
    return ((_u(a) << U32(8)).astype(I32) >> I32(8)).astype(I64)


def _perm(s0, s1, sel):
    s0, s1, sel = _w(s0), _w(s1), _u(sel)
    both = (s0 << U64(32)) | s1
    out = np.zeros(np.broadcast(s0, s1, sel).shape, dtype=U32)
    for i in range(4):
        c = (sel >> U32(8 * i)) & U32(0xFF)
        byte = ((both >> (_w(np.minimum(c, 7)) * U64(8))) & U64(0xFF)).astype(U32)
        sign = lambda k: np.where(((both >> U64(16 * k + 15)) & ONE64) != 0, U32(0xFF), U32(0))
        byte = np.where(c == 8, sign(0), byte)
        byte = np.where(c == 9, sign(1), byte)
        byte = np.where(c == 10, sign(2), byte)
        byte = np.where(c == 11, sign(3), byte)
        byte = np.where(c == 12, U32(0), byte)
        byte = np.where(c >= 13, U32(0xFF), byte)
        out = out | (byte << U32(8 * i))
    return out


def _bitop3(a, b, c, tt, mask=M32):
    a, b, c = _u(a), _u(b), _u(c)
    r = np.zeros(np.broadcast(a, b, c).shape, dtype=U32)
    for idx in range(8):
        if (tt >> idx) & 1:
            t = (a if idx & 4 else ~a) & (b if idx & 2 else ~b) & (c if idx & 1 else ~c)
            r = r | t
    return r & U32(mask)


def _cvt_u32_f32(a):
    f = _f(a).astype(np.float64)
    f = np.where(np.isnan(f), 0.0, f)
    f = np.clip(np.trunc(f), 0.0, 4294967295.0)
    return f.astype(U64).astype(U32)


def _cvt_i32_f32(a):
    f = _f(a).astype(np.float64)
    f = np.where(np.isnan(f), 0.0, f)
    f = np.clip(np.trunc(f), -2147483648.0, 2147483647.0)
    return f.astype(I64).astype(I32).astype(U32)


BIN32 = {
    "v_add_u32": lambda a, b: a + b,
    "v_sub_u32": lambda a, b: a - b,
    "v_subrev_u32": lambda a, b: b - a,
    "v_and_b32": lambda a, b: a & b,
    "v_or_b32": lambda a, b: a | b,
    "v_xor_b32": lambda a, b: a ^ b,
    "v_xnor_b32": lambda a, b: ~(a ^ b),
    "v_lshlrev_b32": lambda a, b: b << (a & U32(31)),
    "v_lshrrev_b32": lambda a, b: b >> (a & U32(31)),
    "v_ashrrev_i32": lambda a, b: (_i(b) >> _i(a & U32(31))).astype(U32),
    "v_min_u32": lambda a, b: np.minimum(a, b),
    "v_max_u32": lambda a, b: np.maximum(a, b),
    "v_min_i32": lambda a, b: np.minimum(_i(a), _i(b)).astype(U32),
    "v_max_i32": lambda a, b: np.maximum(_i(a), _i(b)).astype(U32),
    "v_mul_lo_u32": lambda a, b: (_w(a) * _w(b)).astype(U32),
    "v_mul_hi_u32": lambda a, b: ((_w(a) * _w(b)) >> U64(32)).astype(U32),
    "v_mul_hi_i32": lambda a, b: ((_i(a).astype(I64) * _i(b).astype(I64)) >> I64(32)).astype(U32),
    "v_mul_u32_u24": lambda a, b: _mul24u(a, b).astype(U32),
    "v_mul_hi_u32_u24": lambda a, b: (_mul24u(a, b) >> U64(32)).astype(U32),
    "v_mul_i32_i24": lambda a, b: (_s24(a) * _s24(b)).astype(U32),
    "v_bfm_b32": lambda a, b: ((U32(1) << (a & U32(31))) - U32(1)) << (b & U32(31)),
    "v_bcnt_u32_b32": lambda a, b: _popcount(a) + b,
    "v_add_u16": lambda a, b: (a + b) & U32(0xFFFF),
    "v_sub_u16": lambda a, b: (a - b) & U32(0xFFFF),
    "v_subrev_u16": lambda a, b: (b - a) & U32(0xFFFF),
    "v_mul_lo_u16": lambda a, b: (a * b) & U32(0xFFFF),
    "v_lshlrev_b16": lambda a, b: (b << (a & U32(15))) & U32(0xFFFF),
    "v_lshrrev_b16": lambda a, b: (b & U32(0xFFFF)) >> (a & U32(15)),
    "v_ashrrev_i16": lambda a, b: (_u(b).astype(U16).astype(I16) >> (a & U32(15)).astype(I16)).astype(U16).astype(U32),
    "v_max_u16": lambda a, b: np.maximum(a & U32(0xFFFF), b & U32(0xFFFF)),
    "v_min_u16": lambda a, b: np.minimum(a & U32(0xFFFF), b & U32(0xFFFF)),
    "v_max_i16": lambda a, b: np.maximum(_u(a).astype(U16).astype(I16), _u(b).astype(U16).astype(I16)).astype(U16).astype(U32),
    "v_min_i16": lambda a, b: np.minimum(_u(a).astype(U16).astype(I16), _u(b).astype(U16).astype(I16)).astype(U16).astype(U32),
    "v_mul_f32": lambda a, b: _fu(_f(a) * _f(b)),
    "v_add_f32": lambda a, b: _fu(_f(a) + _f(b)),
    "v_sub_f32": lambda a, b: _fu(_f(a) - _f(b)),
    "v_subrev_f32": lambda a, b: _fu(_f(b) - _f(a)),
    "v_max_f32": lambda a, b: _fu(np.maximum(_f(a), _f(b))),
    "v_min_f32": lambda a, b: _fu(np.minimum(_f(a), _f(b))),
}

# integer clamp (VOP3 `clamp`): the result saturates instead of wrapping
CLAMP32 = {
    "v_add_u32": lambda a, b: np.minimum(_w(a) + _w(b), U64(M32)).astype(U32),
    "v_sub_u32": lambda a, b: np.where(a >= b, a - b, U32(0)).astype(U32),
    "v_subrev_u32": lambda a, b: np.where(b >= a, b - a, U32(0)).astype(U32),
    "v_add_i32": lambda a, b: np.clip(_i(a).astype(I64) + _i(b).astype(I64), -2147483648, 2147483647).astype(I32).astype(U32),
    "v_sub_i32": lambda a, b: np.clip(_i(a).astype(I64) - _i(b).astype(I64), -2147483648, 2147483647).astype(I32).astype(U32),
}

UN32 = {
    "v_mov_b32": lambda a: a,
    "v_not_b32": lambda a: ~_u(a),
    "v_bfrev_b32": _bfrev,
    "v_ffbl_b32": _ffbl,
    "v_ffbh_u32": _ffbh,
    "v_cvt_f32_u32": lambda a: _fu(_u(a).astype(F32)),
    "v_cvt_f32_i32": lambda a: _fu(_i(a).astype(F32)),
    "v_cvt_u32_f32": _cvt_u32_f32,
    "v_cvt_i32_f32": _cvt_i32_f32,
    "v_cvt_f32_ubyte0": lambda a: _fu((_u(a) & U32(0xFF)).astype(F32)),
    "v_rcp_f32": lambda a: _fu(F32(1.0) / _f(a)),
    "v_rcp_iflag_f32": lambda a: _fu(F32(1.0) / _f(a)),
    "v_trunc_f32": lambda a: _fu(np.trunc(_f(a))),
    "v_floor_f32": lambda a: _fu(np.floor(_f(a))),
    "v_rndne_f32": lambda a: _fu(np.rint(_f(a))),
    "v_bcnt_u32": lambda a: _popcount(a),
    "v_sext_i32_i8": lambda a: _u(a).astype(U8).astype(np.int8).astype(I32).astype(U32),
    "v_sext_i32_i16": lambda a: _u(a).astype(U16).astype(I16).astype(I32).astype(U32),
}

TRI32 = {
    "v_lshl_add_u32": lambda a, b, c: (a << (b & U32(31))) + c,
    "v_add_lshl_u32": lambda a, b, c: (a + b) << (c & U32(31)),
    "v_lshl_or_b32": lambda a, b, c: (a << (b & U32(31))) | c,
    "v_and_or_b32": lambda a, b, c: (a & b) | c,
    "v_or3_b32": lambda a, b, c: a | b | c,
    "v_add3_u32": lambda a, b, c: a + b + c,
    "v_xad_u32": lambda a, b, c: (a ^ b) + c,
    "v_bfe_u32": lambda a, b, c: (a >> (b & U32(31))) & ((U32(1) << (c & U32(31))) - U32(1)),
    "v_bfe_i32": lambda a, b, c: _bfe_i32(a, b, c),
    "v_bfi_b32": lambda a, b, c: (a & b) | (~_u(a) & c),
    "v_alignbit_b32": lambda a, b, c: ((((_w(a) << U64(32)) | _w(b)) >> _w(c & U32(31))) & U64(M32)).astype(U32),
    "v_alignbyte_b32": lambda a, b, c: ((((_w(a) << U64(32)) | _w(b)) >> (_w(c & U32(3)) * U64(8))) & U64(M32)).astype(U32),
    "v_perm_b32": _perm,
    "v_mad_u32_u24": lambda a, b, c: (_mul24u(a, b) + _w(c)).astype(U32),
    "v_mad_i32_i24": lambda a, b, c: (_s24(a) * _s24(b) + _i(c).astype(I64)).astype(U32),
    "v_min3_u32": lambda a, b, c: np.minimum(np.minimum(a, b), c),
    "v_max3_u32": lambda a, b, c: np.maximum(np.maximum(a, b), c),
    "v_min3_i32": lambda a, b, c: np.minimum(np.minimum(_i(a), _i(b)), _i(c)).astype(U32),
    "v_max3_i32": lambda a, b, c: np.maximum(np.maximum(_i(a), _i(b)), _i(c)).astype(U32),
    "v_med3_i32": lambda a, b, c: np.sort(np.stack(np.broadcast_arrays(_i(a), _i(b), _i(c))), axis=0)[1].astype(U32),
    "v_med3_u32": lambda a, b, c: np.sort(np.stack(np.broadcast_arrays(_u(a), _u(b), _u(c))), axis=0)[1].astype(U32),
    "v_fma_f32": lambda a, b, c: _fu((_f(a).astype(np.float64) * _f(b).astype(np.float64) + _f(c).astype(np.float64)).astype(F32)),
    "v_mad_f32": lambda a, b, c: _fu(_f(a) * _f(b) + _f(c)),
    "v_mad_u16": lambda a, b, c: (a * b + c) & U32(0xFFFF),
}


def _bfe_i32(a, b, c):
    off, wd = _u(b) & U32(31), _u(c) & U32(31)
    x = (_u(a) >> off) & ((U32(1) << wd) - U32(1))
    sign = (wd != 0) & (((x >> np.where(wd == 0, U32(0), wd - U32(1))) & U32(1)) != 0)
    return np.where(sign, x | ~((U32(1) << wd) - U32(1)), x)


CMP = {
    "f": lambda a, b: np.zeros(np.broadcast(a, b).shape, dtype=bool), "lt": lambda a, b: a < b, "eq": lambda a, b: a == b,
    "le": lambda a, b: a <= b, "gt": lambda a, b: a > b, "lg": lambda a, b: a != b, "ne": lambda a, b: a != b, "ge": lambda a, b: a >= b,
    "t": lambda a, b: np.ones(np.broadcast(a, b).shape, dtype=bool), "neq": lambda a, b: ~(a == b), "nlt": lambda a, b: ~(a < b),
    "nle": lambda a, b: ~(a <= b), "ngt": lambda a, b: ~(a > b), "nge": lambda a, b: ~(a >= b), "nlg": lambda a, b: ~(a != b),
    "o": lambda a, b: ~(np.isnan(a) | np.isnan(b)), "u": lambda a, b: np.isnan(a) | np.isnan(b),
}
CMP_TYPE = {
    "u32": (32, lambda x: _u(x)), "i32": (32, lambda x: _i(x)), "u64": (64, lambda x: _w(x)), "i64": (64, lambda x: _w(x).astype(I64)),
    "u16": (32, lambda x: _u(x) & U32(0xFFFF)), "i16": (32, lambda x: _u(x).astype(U16).astype(I16)), "f32": (32, lambda x: _f(x)),
}


def _sel(x, sel, sext):
    if sel == "DWORD":
        return x
    x = _u(x)
    if sel.startswith("BYTE_"):
        r = (x >> U32(8 * int(sel[5]))) & U32(0xFF)
        if sext:
            r = r.astype(U8).astype(np.int8).astype(I32).astype(U32)
        return r
    if sel.startswith("WORD_"):
        r = (x >> U32(16 * int(sel[5]))) & U32(0xFFFF)
        if sext:
            r = r.astype(U16).astype(I16).astype(I32).astype(U32)
        return r
    raise AsmError("sdwa select " + sel)


def _dpp_source(I):
    """-> (src_lane[64] int64, valid[64] bool) for the dpp control of instruction I"""
    m = I.mods
    src = LANE.copy()
    valid = np.ones(64, dtype=bool)
    row, col = LANE // 16, LANE % 16
    if "quad_perm" in m:
        q = m["quad_perm"]
        src = (LANE & ~3) + np.array([q[i & 3] for i in range(64)])
    elif "row_shr" in m:
        n = m["row_shr"]
        src, valid = LANE - n, col >= n
    elif "row_shl" in m:
        n = m["row_shl"]
        src, valid = LANE + n, col + n < 16
    elif "row_ror" in m:
        n = m["row_ror"]
        src = row * 16 + (col - n) % 16
    elif "wave_shr" in m:
        src, valid = LANE - 1, LANE >= 1
    elif "wave_shl" in m:
        src, valid = LANE + 1, LANE < 63
    elif "wave_ror" in m:
        src = (LANE - 1) % 64
    elif "wave_rol" in m:
        src = (LANE + 1) % 64
    elif "row_mirror" in m:
        src = row * 16 + (15 - col)
    elif "row_half_mirror" in m:
        src = (LANE & ~7) + (7 - (LANE & 7))
    elif "row_bcast" in m:
        n = m["row_bcast"]
        if n == 15:
            src, valid = (row - 1) * 16 + 15, row >= 1
        elif n == 31:
            src, valid = np.full(64, 31), row >= 2
        else:
            raise AsmError("row_bcast:%d" % n)
    else:
        raise AsmError("dpp control not understood: %r" % (m,))
    rm, bm = m.get("row_mask", 0xF), m.get("bank_mask", 0xF)
    enabled = (((rm >> row) & 1) != 0) & (((bm >> (col // 4)) & 1) != 0)
    return np.clip(src, 0, 63), valid, enabled


# ------------------------------------------------------------------------------------------------------------------
class Wave:
    def __init__(self, rt, mod, kernel, wg, index):
        self.rt, self.mod, self.kernel, self.wg, self.index = rt, mod, kernel, wg, index
        self.S = [0] * 128
        self.S[126] = self.S[127] = M32       # exec is readable as s[126:127]
        self.V = np.zeros((512, 64), dtype=U32)
        self.A = np.zeros((256, 64), dtype=U32)
        self.scc = 0
        self.exec = M64
        self.em = np.ones(64, dtype=bool)
        self.full = True
        self.lanes = LANE
        self.pc = kernel.entry
        self.done = False
        self.at_barrier = False
        self.lds = wg.lds
        self.mem = rt.mem
        self.scratch = None
        self.steps = 0
        self.count = {}
        self.trace = None

    # vcc lives in S[106:107]
    @property
    def vcc(self):
        return self.S[106] | (self.S[107] << 32)

    def set_exec(self, v):
        self.exec = v
        self.S[126], self.S[127] = v & M32, v >> 32
        if v == M64:
            self.full = True
            self.em = FULL_EM
            self.lanes = LANE
        else:
            self.full = False
            self.em = mask_to_bool(v)
            self.lanes = np.nonzero(self.em)[0]

    def run(self, budget):
        """execute up to `budget` instructions; stops early at s_barrier / s_endpgm"""
        ins = self.mod.ins
        n = 0
        cnt = self.count
        try:
            while n < budget:
                I = ins[self.pc]
                self.pc += 1
                fn = I.fn
                if fn is None:
                    fn = I.fn = decode(self.mod, I, self.pc - 1)
                if self.trace is not None:
                    self.trace(self, I)
                if self.mem.shadow is not None or self.rt.racecheck:
                    self.mem.where = (self.kernel.name, I.line)
                    if self.rt.racecheck:
                        self.mem.agent = (self.wg.number * 64 + self.index, self.wg.epoch)
                fn(self)
                n += 1
                c = I.cls
                cnt[c] = cnt.get(c, 0) + 1
                if self.done or self.at_barrier:
                    break
        except (SimError, AsmError, IndexError, ValueError, OverflowError, KeyError) as e:
            I = ins[self.pc - 1]
            raise SimError("%s line %d (pc %d) `%s`: %s: %s   [wave %d of workgroup %s]" % (self.mod.name, I.line, self.pc - 1, I.op, type(e).__name__, e,
                                                                                      self.index, self.wg.id)) from e
        self.steps += n
        return n


FULL_EM = np.ones(64, dtype=bool)


# ------------------------------------------------------------------------------------------------------------------
# decode: instruction -> closure
# ------------------------------------------------------------------------------------------------------------------
def decode(mod, I, pc):
    b = I.base
    for table in (_decode_salu, _decode_valu, _decode_mem):
        fn = table(mod, I, pc)
        if fn is not None:
            return fn
    raise AsmError("instruction not modelled: %s" % I.op)


def _branch_target(mod, I, pc):
    o = I.ops[0]
    if o[0] == "label":
        return mod.target(pc, o[1])
    if o[0] == "k":          # inline assembly "1f" parses as label; a plain number would be a relative offset
        raise AsmError("numeric branch offsets are not modelled")
    raise AsmError("branch target %r" % (o,))


def _decode_salu(mod, I, pc):
    b, o = I.base, I.ops
    I.cls = "salu"
    if b in ("s_waitcnt", "s_nop", "s_sleep", "s_setprio", "s_waitcnt_vscnt", "s_waitcnt_depctr", "s_inst_prefetch", "s_clause", "s_icache_inv", "s_dcache_wb",
             "s_dcache_inv", "s_setreg_imm32_b32", "s_setreg_b32", "s_ttracedata", "s_incperflevel", "s_decperflevel", "buffer_wbl2", "buffer_inv", "buffer_wbinvl1",
             "buffer_wbinvl1_vol", "s_sethalt"):
        I.cls = "other"
        return lambda w: None
    if b == "s_endpgm":
        I.cls = "other"
        def f(w):
            w.done = True
        return f
    if b == "s_barrier":
        I.cls = "other"
        def f(w):
            w.at_barrier = True
        return f
    if b == "s_trap":
        def f(w):
            raise SimTrap("s_trap %s" % I.aux)
        return f
    if b == "s_branch":
        I.cls = "branch"
        t = _branch_target(mod, I, pc)
        def f(w):
            w.pc = t
        return f
    if b.startswith("s_cbranch_"):
        I.cls = "branch"
        t = _branch_target(mod, I, pc)
        kind = b[len("s_cbranch_"):]
        if kind == "scc0":
            def f(w):
                if not w.scc:
                    w.pc = t
        elif kind == "scc1":
            def f(w):
                if w.scc:
                    w.pc = t
        elif kind == "vccz":
            def f(w):
                if not (w.S[106] | w.S[107]):
                    w.pc = t
        elif kind == "vccnz":
            def f(w):
                if w.S[106] | w.S[107]:
                    w.pc = t
        elif kind == "execz":
            def f(w):
                if not w.exec:
                    w.pc = t
        elif kind == "execnz":
            def f(w):
                if w.exec:
                    w.pc = t
        else:
            return None
        return f
    if b == "s_getpc_b64":
        p = sdst(o[0], 64)
        return lambda w: p(w, w.rt.code_addr(w.mod, w.pc))
    if b == "s_setpc_b64":
        I.cls = "branch"
        g = ssrc(o[0], 64)
        def f(w):
            w.pc = w.rt.code_pc(w.mod, g(w))
        return f
    if b == "s_swappc_b64":
        I.cls = "branch"
        p, g = sdst(o[0], 64), ssrc(o[1], 64)
        def f(w):
            t = g(w)
            p(w, w.rt.code_addr(w.mod, w.pc))
            w.pc = w.rt.code_pc(w.mod, t)
        return f
    if b == "s_memrealtime" or b == "s_memtime":
        I.cls = "smem"
        p = sdst(o[0], 64)
        def f(w):
            w.rt.clock += 100
            p(w, w.rt.clock)
        return f
    if b.startswith("s_load_dword") or b.startswith("s_buffer_load_dword"):
        I.cls = "smem"
        n = {"": 1, "x2": 2, "x3": 3, "x4": 4, "x8": 8, "x16": 16}[b.split("dword")[1]]
        d0 = o[0][1]
        if o[0][0] != "s":
            raise AsmError("s_load destination")
        gb = ssrc(o[1], 64)
        go = ssrc(o[2]) if len(o) > 2 else (lambda w: 0)
        imm = I.mods.get("offset", 0)
        def f(w):
            a = gb(w) + go(w) + imm
            vals = w.mem.read_dwords(a, n)
            for j in range(n):
                d = d0 + j
                if d in (126, 127):
                    raise SimError("s_load into exec")
                w.S[d] = vals[j]
        return f
    if b.startswith("s_store_dword"):
        I.cls = "smem"
        n = {"": 1, "x2": 2, "x4": 4}[b.split("dword")[1]]
        s0 = o[0][1]
        gb, go = ssrc(o[1], 64), ssrc(o[2])
        def f(w):
            a = gb(w) + go(w)
            w.mem.write_dwords(a, [w.S[s0 + j] for j in range(n)])
        return f
    # --- SOP1 / SOP2 / SOPC / SOPK -------------------------------------------------------------------------
    if b in ("s_mov_b32", "s_mov_b64", "s_cmov_b32", "s_cmov_b64"):
        wd = 64 if b.endswith("64") else 32
        p, g = sdst(o[0], wd), ssrc(o[1], wd)
        if b.startswith("s_cmov"):
            def f(w):
                if w.scc:
                    p(w, g(w))
            return f
        return lambda w: p(w, g(w))
    if b == "s_movk_i32":
        p, k = sdst(o[0]), sx(o[1][1], 16) & M32
        return lambda w: p(w, k)
    if b in ("s_addk_i32", "s_mulk_i32"):
        p, g, k = sdst(o[0]), ssrc(o[0]), sx(o[1][1], 16)
        if b == "s_addk_i32":
            def f(w):
                a = sx(g(w), 32)
                r = a + k
                w.scc = int(r > 0x7FFFFFFF or r < -0x80000000)
                p(w, r & M32)
        else:
            def f(w):
                p(w, (sx(g(w), 32) * k) & M32)
        return f
    if b.startswith("s_cmpk_"):
        _, _, cond, ty = b.split("_")
        g = ssrc(o[0])
        k = o[1][1]
        signed = ty == "i32"
        kk = sx(k, 16) if signed else (k & 0xFFFF)
        cf = _PYCMP[cond]
        def f(w):
            a = g(w)
            if signed:
                a = sx(a, 32)
            w.scc = int(cf(a, kk))
        return f
    if b.startswith("s_cmp_"):
        _, _, cond, ty = b.split("_")
        wd = 64 if ty.endswith("64") else 32
        ga, gb_ = ssrc(o[0], wd), ssrc(o[1], wd)
        signed = ty[0] == "i"
        cf = _PYCMP[cond]
        def f(w):
            a, c = ga(w), gb_(w)
            if signed:
                a, c = sx(a, wd), sx(c, wd)
            w.scc = int(cf(a, c))
        return f
    if b in ("s_bitcmp0_b32", "s_bitcmp1_b32", "s_bitcmp0_b64", "s_bitcmp1_b64"):
        wd = 64 if b.endswith("64") else 32
        ga, gb_ = ssrc(o[0], wd), ssrc(o[1])
        want = int(b[8])
        def f(w):
            w.scc = int(((ga(w) >> (gb_(w) & (wd - 1))) & 1) == want)
        return f
    if b in _SOP2:
        wd_d, wd_a, wd_b, fn = _SOP2[b]
        p, ga, gb_ = sdst(o[0], wd_d), ssrc(o[1], wd_a), ssrc(o[2], wd_b)
        def f(w):
            r, scc = fn(ga(w), gb_(w), w.scc)
            p(w, r)
            if scc is not None:
                w.scc = scc
        return f
    if b in _SOP1:
        wd_d, wd_a, fn = _SOP1[b]
        p, ga = sdst(o[0], wd_d), ssrc(o[1], wd_a)
        def f(w):
            r, scc = fn(ga(w))
            p(w, r)
            if scc is not None:
                w.scc = scc
        return f
    if b in ("s_cselect_b32", "s_cselect_b64"):
        wd = 64 if b.endswith("64") else 32
        p, ga, gb_ = sdst(o[0], wd), ssrc(o[1], wd), ssrc(o[2], wd)
        return lambda w: p(w, ga(w) if w.scc else gb_(w))
    if b.endswith("_saveexec_b64"):
        kind = b[2:-len("_saveexec_b64")]
        p, ga = sdst(o[0], 64), ssrc(o[1], 64)
        fn = {"and": lambda s, e: s & e, "or": lambda s, e: s | e, "xor": lambda s, e: s ^ e, "andn2": lambda s, e: s & ~e, "orn2": lambda s, e: s | ~e,
              "nand": lambda s, e: ~(s & e), "nor": lambda s, e: ~(s | e), "xnor": lambda s, e: ~(s ^ e), "andn1": lambda s, e: ~s & e, "orn1": lambda s, e: ~s | e}[kind]
        def f(w):
            s = ga(w)
            old = w.exec
            p(w, old)
            r = fn(s, old) & M64
            w.set_exec(r)
            w.scc = int(r != 0)
        return f
    if b in ("s_bitset1_b32", "s_bitset0_b32", "s_bitset1_b64", "s_bitset0_b64"):
        wd = 64 if b.endswith("64") else 32
        p, gd, ga = sdst(o[0], wd), ssrc(o[0], wd), ssrc(o[1])
        one = b[8] == "1"
        def f(w):
            bit = 1 << (ga(w) & (wd - 1))
            p(w, (gd(w) | bit) if one else (gd(w) & ~bit))
        return f
    return None


_PYCMP = {"eq": lambda a, b: a == b, "lg": lambda a, b: a != b, "ne": lambda a, b: a != b, "gt": lambda a, b: a > b, "ge": lambda a, b: a >= b,
          "lt": lambda a, b: a < b, "le": lambda a, b: a <= b}


def _add32(a, b, c):
    r = a + b
    return r & M32, int(r > M32)


def _addc32(a, b, c):
    r = a + b + c
    return r & M32, int(r > M32)


def _sub32(a, b, c):
    return (a - b) & M32, int(b > a)


def _subb32(a, b, c):
    return (a - b - c) & M32, int(b + c > a)


def _addi32(a, b, c):
    r = sx(a, 32) + sx(b, 32)
    return r & M32, int(r > 0x7FFFFFFF or r < -0x80000000)


def _subi32(a, b, c):
    r = sx(a, 32) - sx(b, 32)
    return r & M32, int(r > 0x7FFFFFFF or r < -0x80000000)


def _logic(fn, mask):
    def g(a, b, c):
        r = fn(a, b) & mask
        return r, int(r != 0)
    return g


def _bfe(a, b, signed, wd):
    off = b & (wd - 1)
    width = (b >> 16) & 0x7F
    if width == 0:
        return 0, 0
    r = (a >> off) & ((1 << width) - 1)
    if signed and width <= wd and (r >> (width - 1)) & 1 and off + width <= wd:
        r |= ((1 << wd) - 1) & ~((1 << width) - 1)
    elif signed and off + width > wd:
        # the field runs past the top: sign comes from the top bit
        r = (sx(a, wd) >> off) & ((1 << wd) - 1)
    r &= (1 << wd) - 1
    return r, int(r != 0)


_SOP2 = {
    "s_add_u32": (32, 32, 32, _add32), "s_addc_u32": (32, 32, 32, _addc32), "s_sub_u32": (32, 32, 32, _sub32), "s_subb_u32": (32, 32, 32, _subb32),
    "s_add_i32": (32, 32, 32, _addi32), "s_sub_i32": (32, 32, 32, _subi32),
    "s_and_b32": (32, 32, 32, _logic(lambda a, b: a & b, M32)), "s_and_b64": (64, 64, 64, _logic(lambda a, b: a & b, M64)),
    "s_or_b32": (32, 32, 32, _logic(lambda a, b: a | b, M32)), "s_or_b64": (64, 64, 64, _logic(lambda a, b: a | b, M64)),
    "s_xor_b32": (32, 32, 32, _logic(lambda a, b: a ^ b, M32)), "s_xor_b64": (64, 64, 64, _logic(lambda a, b: a ^ b, M64)),
    "s_andn2_b32": (32, 32, 32, _logic(lambda a, b: a & ~b, M32)), "s_andn2_b64": (64, 64, 64, _logic(lambda a, b: a & ~b, M64)),
    "s_orn2_b32": (32, 32, 32, _logic(lambda a, b: a | ~b, M32)), "s_orn2_b64": (64, 64, 64, _logic(lambda a, b: a | ~b, M64)),
    "s_nand_b32": (32, 32, 32, _logic(lambda a, b: ~(a & b), M32)), "s_nand_b64": (64, 64, 64, _logic(lambda a, b: ~(a & b), M64)),
    "s_nor_b32": (32, 32, 32, _logic(lambda a, b: ~(a | b), M32)), "s_nor_b64": (64, 64, 64, _logic(lambda a, b: ~(a | b), M64)),
    "s_xnor_b32": (32, 32, 32, _logic(lambda a, b: ~(a ^ b), M32)), "s_xnor_b64": (64, 64, 64, _logic(lambda a, b: ~(a ^ b), M64)),
    "s_lshl_b32": (32, 32, 32, lambda a, b, c: (lambda r: (r, int(r != 0)))((a << (b & 31)) & M32)),
    "s_lshl_b64": (64, 64, 32, lambda a, b, c: (lambda r: (r, int(r != 0)))((a << (b & 63)) & M64)),
    "s_lshr_b32": (32, 32, 32, lambda a, b, c: (lambda r: (r, int(r != 0)))(a >> (b & 31))),
    "s_lshr_b64": (64, 64, 32, lambda a, b, c: (lambda r: (r, int(r != 0)))(a >> (b & 63))),
    "s_ashr_i32": (32, 32, 32, lambda a, b, c: (lambda r: (r, int(r != 0)))((sx(a, 32) >> (b & 31)) & M32)),
    "s_ashr_i64": (64, 64, 32, lambda a, b, c: (lambda r: (r, int(r != 0)))((sx(a, 64) >> (b & 63)) & M64)),
    "s_mul_i32": (32, 32, 32, lambda a, b, c: ((a * b) & M32, None)),
    "s_mul_hi_u32": (32, 32, 32, lambda a, b, c: ((a * b) >> 32, None)),
    "s_mul_hi_i32": (32, 32, 32, lambda a, b, c: (((sx(a, 32) * sx(b, 32)) >> 32) & M32, None)),
    "s_min_u32": (32, 32, 32, lambda a, b, c: (min(a, b), int(a < b))), "s_max_u32": (32, 32, 32, lambda a, b, c: (max(a, b), int(a > b))),
    "s_min_i32": (32, 32, 32, lambda a, b, c: ((a, 1) if sx(a, 32) < sx(b, 32) else (b, 0))),
    "s_max_i32": (32, 32, 32, lambda a, b, c: ((a, 1) if sx(a, 32) > sx(b, 32) else (b, 0))),
    "s_bfm_b32": (32, 32, 32, lambda a, b, c: ((((1 << (a & 31)) - 1) << (b & 31)) & M32, None)),
    "s_bfm_b64": (64, 32, 32, lambda a, b, c: ((((1 << (a & 63)) - 1) << (b & 63)) & M64, None)),
    "s_bfe_u32": (32, 32, 32, lambda a, b, c: _bfe(a, b, False, 32)), "s_bfe_i32": (32, 32, 32, lambda a, b, c: _bfe(a, b, True, 32)),
    "s_bfe_u64": (64, 64, 32, lambda a, b, c: _bfe(a, b, False, 64)), "s_bfe_i64": (64, 64, 32, lambda a, b, c: _bfe(a, b, True, 64)),
    "s_pack_ll_b32_b16": (32, 32, 32, lambda a, b, c: ((a & 0xFFFF) | ((b & 0xFFFF) << 16), None)),
    "s_pack_lh_b32_b16": (32, 32, 32, lambda a, b, c: ((a & 0xFFFF) | (b & 0xFFFF0000), None)),
    "s_pack_hh_b32_b16": (32, 32, 32, lambda a, b, c: ((a >> 16) | (b & 0xFFFF0000), None)),
    "s_lshl1_add_u32": (32, 32, 32, lambda a, b, c: (lambda r: (r & M32, int(r > M32)))((a << 1) + b)),
    "s_lshl2_add_u32": (32, 32, 32, lambda a, b, c: (lambda r: (r & M32, int(r > M32)))((a << 2) + b)),
    "s_lshl3_add_u32": (32, 32, 32, lambda a, b, c: (lambda r: (r & M32, int(r > M32)))((a << 3) + b)),
    "s_lshl4_add_u32": (32, 32, 32, lambda a, b, c: (lambda r: (r & M32, int(r > M32)))((a << 4) + b)),
}


def _ff1(a):
    return ((a & -a).bit_length() - 1) & M32 if a else M32


def _flbit(a, wd):
    return (wd - a.bit_length()) if a else M32


def _brev(a, wd):
    return int(format(a, "0%db" % wd)[::-1], 2)


_SOP1 = {
    "s_not_b32": (32, 32, lambda a: (lambda r: (r, int(r != 0)))(~a & M32)), "s_not_b64": (64, 64, lambda a: (lambda r: (r, int(r != 0)))(~a & M64)),
    "s_brev_b32": (32, 32, lambda a: (_brev(a, 32), None)), "s_brev_b64": (64, 64, lambda a: (_brev(a, 64), None)),
    "s_bcnt1_i32_b32": (32, 32, lambda a: (lambda r: (r, int(r != 0)))(bin(a).count("1"))),
    "s_bcnt1_i32_b64": (32, 64, lambda a: (lambda r: (r, int(r != 0)))(bin(a).count("1"))),
    "s_bcnt0_i32_b32": (32, 32, lambda a: (lambda r: (r, int(r != 0)))(32 - bin(a).count("1"))),
    "s_bcnt0_i32_b64": (32, 64, lambda a: (lambda r: (r, int(r != 0)))(64 - bin(a).count("1"))),
    "s_ff1_i32_b32": (32, 32, lambda a: (_ff1(a), None)), "s_ff1_i32_b64": (32, 64, lambda a: (_ff1(a), None)),
    "s_ff0_i32_b32": (32, 32, lambda a: (_ff1(~a & M32), None)), "s_ff0_i32_b64": (32, 64, lambda a: (_ff1(~a & M64), None)),
    "s_flbit_i32_b32": (32, 32, lambda a: (_flbit(a, 32), None)), "s_flbit_i32_b64": (32, 64, lambda a: (_flbit(a, 64), None)),
    "s_sext_i32_i8": (32, 32, lambda a: (sx(a, 8) & M32, None)), "s_sext_i32_i16": (32, 32, lambda a: (sx(a, 16) & M32, None)),
    "s_abs_i32": (32, 32, lambda a: (lambda r: (r, int(r != 0)))(abs(sx(a, 32)) & M32)),
    "s_wqm_b64": (64, 64, lambda a: (lambda r: (r, int(r != 0)))(sum(0xF << (4 * q) for q in range(16) if (a >> (4 * q)) & 0xF))),
}


# ------------------------------------------------------------------------------------------------------------------
def _mask_dst(o):
    """destination of a lane mask (vcc / sgpr pair / exec)"""
    return sdst(o, 64)


def _carry_src(o):
    g = ssrc(o, 64)
    return lambda w: mask_to_bool(g(w))


def _decode_valu(mod, I, pc):
    b, o, enc, m = I.base, I.ops, I.enc, I.mods
    if not (b.startswith("v_")):
        return None
    I.cls = "valu"
    sdwa = enc == "sdwa"
    dpp = enc == "dpp" or any(k in m for k in ("quad_perm", "row_shr", "row_shl", "row_ror", "row_bcast", "wave_shr", "wave_shl", "row_mirror", "row_half_mirror",
                                                 "wave_ror", "wave_rol"))
    if "clamp" in m and b in CLAMP32:
        p, (ga, gb_) = vdst(o[0]), [vsrc(o[1]), vsrc(o[2])]
        fn = CLAMP32[b]
        if sdwa or dpp:
            raise AsmError("clamp with sdwa / dpp")
        return lambda w: p(w, fn(_u(ga(w)), _u(gb_(w))))
    if "clamp" in m or "omod" in m or "mul" in m or "div" in m:
        raise AsmError("clamp / omod are not modelled")

    def srcs(idx_list, width=32):
        gs = []
        for k, i in enumerate(idx_list):
            g = vsrc(o[i], width)
            if sdwa and k < 2:
                sel = m.get("src%d_sel" % k, "DWORD")
                sext = o[i][0] == "mod" and "sext" in o[i][2]
                if sel != "DWORD" or sext:
                    g = (lambda g, sel, sext: (lambda w: _sel(g(w), sel, sext)))(g, sel, sext)
            gs.append(g)
        return gs

    if sdwa and (m.get("dst_sel", "DWORD") != "DWORD" or m.get("dst_unused", "UNUSED_PAD") not in ("UNUSED_PAD",)):
        raise AsmError("sdwa destination select is not modelled: %r" % (m,))

    def wrap_dpp(fn_write):
        """fn_write(w, permuted_src0_getter) — evaluates the op with src0 read through the dpp permutation, writes enabled lanes"""
        return fn_write

    # ---- compares ----------------------------------------------------------------------------------------------
    if b.startswith("v_cmp_") or b.startswith("v_cmpx_"):
        x = b.startswith("v_cmpx_")
        _, _, cond, ty = b.split("_", 3)
        if ty == "class_f32" or cond == "class":
            raise AsmError("v_cmp_class is not modelled")
        wd, conv = CMP_TYPE[ty]
        cf = CMP[cond]
        if dpp:
            raise AsmError("dpp compare")
        if len(o) == 3:
            pd = _mask_dst(o[0])
            ga, gb_ = srcs([1, 2], wd)
        else:                      # v_cmpx with the implicit destination (exec only)
            pd = None
            ga, gb_ = srcs([0, 1], wd)
        def f(w):
            r = np.broadcast_to(cf(conv(ga(w)), conv(gb_(w))), (64,))
            mk = bool_to_mask(r & w.em)
            if pd is not None:
                pd(w, mk)
            if x:
                w.set_exec(mk)
        return f
    # ---- lane access -------------------------------------------------------------------------------------------
    if b == "v_readlane_b32":
        p, n, gl = sdst(o[0]), o[1][1], ssrc(o[2])
        file = "V" if o[1][0] == "v" else "A"
        return lambda w: p(w, int(getattr(w, file)[n][gl(w) & 63]))
    if b == "v_readfirstlane_b32":
        p = sdst(o[0])
        if o[1][0] in ("v", "a"):
            n, file = o[1][1], ("V" if o[1][0] == "v" else "A")
            def f(w):
                lane = int(w.lanes[0]) if w.exec else 0
                p(w, int(getattr(w, file)[n][lane]))
        else:
            g = ssrc(o[1])
            def f(w):
                p(w, g(w))
        return f
    if b == "v_writelane_b32":
        n, gv, gl = o[0][1], ssrc(o[1]), ssrc(o[2])
        def f(w):
            w.V[n][gl(w) & 63] = gv(w) & M32
        return f
    if b in ("v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_accvgpr_mov_b32"):
        p, g = vdst(o[0]), vsrc(o[1])
        return lambda w: p(w, g(w))
    if b == "v_mov_b64":
        p, g = vdst(o[0], 64), vsrc(o[1], 64)
        return lambda w: p(w, g(w))
    if b in ("v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):
        p, (ga, gb_) = vdst(o[0]), srcs([1, 2])
        lo = b == "v_mbcnt_lo_u32_b32"
        below_lo = np.array([(1 << min(l, 32)) - 1 for l in range(64)], dtype=U32)
        below_hi = np.array([(1 << max(l - 32, 0)) - 1 for l in range(64)], dtype=U32)
        tbl = below_lo if lo else below_hi
        return lambda w: p(w, _popcount(_u(ga(w)) & tbl) + gb_(w))
    if b == "v_cndmask_b32":
        p = vdst(o[0])
        ga, gb_ = srcs([1, 2])
        gm = _carry_src(o[3]) if len(o) > 3 else _carry_src(("s", 106, 2))
        if dpp:
            src_l, valid, enabled = _dpp_source(I)
            raise AsmError("dpp cndmask")
        return lambda w: p(w, np.where(gm(w), gb_(w), ga(w)))
    # ---- carry arithmetic --------------------------------------------------------------------------------------
    if b in ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32"):
        p, pc_ = vdst(o[0]), _mask_dst(o[1])
        ga, gb_ = srcs([2, 3])
        kind = b
        def f(w):
            a, c = _w(ga(w)), _w(gb_(w))
            if kind == "v_subrev_co_u32":
                a, c = c, a
            if kind == "v_add_co_u32":
                r = a + c
                cy = r > U64(M32)
            else:
                r = a - c
                cy = a < c
            cy = np.broadcast_to(cy, (64,)) & w.em
            p(w, (r & U64(M32)).astype(U32))
            pc_(w, bool_to_mask(cy))
        return f
    if b in ("v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32"):
        p, pc_ = vdst(o[0]), _mask_dst(o[1])
        ga, gb_ = srcs([2, 3])
        gc = _carry_src(o[4])
        kind = b
        def f(w):
            a, c = _w(ga(w)), _w(gb_(w))
            ci = gc(w).astype(U64)
            if kind == "v_subbrev_co_u32":
                a, c = c, a
            if kind == "v_addc_co_u32":
                r = a + c + ci
                cy = r > U64(M32)
            else:
                r = a - c - ci
                cy = (c + ci) > a
            cy = np.broadcast_to(cy, (64,)) & w.em
            p(w, (r & U64(M32)).astype(U32))
            pc_(w, bool_to_mask(cy))
        return f
    if b == "v_mad_u64_u32" or b == "v_mad_i64_i32":
        p, pc_ = vdst(o[0], 64), (None if o[1][0] == "null" else _mask_dst(o[1]))
        ga, gb_ = srcs([2, 3])
        gc = vsrc(o[4], 64)
        signed = b == "v_mad_i64_i32"
        def f(w):
            if signed:
                prod = (_i(ga(w)).astype(I64) * _i(gb_(w)).astype(I64)).astype(U64)
            else:
                prod = _w(ga(w)) * _w(gb_(w))
            c = _w(gc(w))
            r = prod + c
            p(w, r)
            if pc_ is not None:
                pc_(w, bool_to_mask(np.broadcast_to(r < c, (64,)) & w.em))
        return f
    if b in ("v_lshl_add_u64",):
        p = vdst(o[0], 64)
        ga, gs_, gc = vsrc(o[1], 64), vsrc(o[2]), vsrc(o[3], 64)
        return lambda w: p(w, (_w(ga(w)) << (_w(gs_(w)) & U64(7))) + _w(gc(w)))
    if b in ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64"):
        p = vdst(o[0], 64)
        gs_, ga = vsrc(o[1]), vsrc(o[2], 64)
        if b == "v_lshlrev_b64":
            return lambda w: p(w, _w(ga(w)) << (_w(gs_(w)) & U64(63)))
        if b == "v_lshrrev_b64":
            return lambda w: p(w, _w(ga(w)) >> (_w(gs_(w)) & U64(63)))
        return lambda w: p(w, (_w(ga(w)).astype(I64) >> (_w(gs_(w)) & U64(63)).astype(I64)).astype(U64))
    if b in ("v_add_u64", "v_sub_u64"):
        p, ga, gb_ = vdst(o[0], 64), vsrc(o[1], 64), vsrc(o[2], 64)
        return (lambda w: p(w, _w(ga(w)) + _w(gb_(w)))) if b == "v_add_u64" else (lambda w: p(w, _w(ga(w)) - _w(gb_(w))))
    if b == "v_bitop3_b32" or b == "v_bitop3_b16":
        p = vdst(o[0])
        ga, gb_, gc = srcs([1, 2, 3])
        tt = m.get("bitop3", 0)
        mask = M32 if b.endswith("32") else 0xFFFF
        if "op_sel" in m and any(m["op_sel"]):
            raise AsmError("op_sel on v_bitop3_b16")
        return lambda w: p(w, _bitop3(ga(w), gb_(w), gc(w), tt, mask))
    if b in ("v_fmamk_f32", "v_fmaak_f32", "v_madmk_f32", "v_madak_f32"):
        p = vdst(o[0])
        g1, g2, g3 = srcs([1, 2, 3])
        if b in ("v_fmamk_f32", "v_madmk_f32"):      # D = S0 * K + S1   (operands: dst, s0, K, s1)
            return lambda w: p(w, TRI32["v_fma_f32"](g1(w), g2(w), g3(w)))
        return lambda w: p(w, TRI32["v_fma_f32"](g1(w), g2(w), g3(w)))
    if b.startswith("v_pk_"):
        p = vdst(o[0])
        gsrc = srcs(list(range(1, len(o))))
        opsel = m.get("op_sel", [0] * 3)
        opsel_hi = m.get("op_sel_hi", [1] * 3)
        fn = _PK[b]
        def f(w):
            xs = [_u(g(w)) for g in gsrc]
            lo = fn(*[(x >> U32(16 * opsel[i])) & U32(0xFFFF) for i, x in enumerate(xs)]) & U32(0xFFFF)
            hi = fn(*[(x >> U32(16 * opsel_hi[i])) & U32(0xFFFF) for i, x in enumerate(xs)]) & U32(0xFFFF)
            p(w, lo | (hi << U32(16)))
        return f
    # ---- table driven ------------------------------------------------------------------------------------------
    if "op_sel" in m and any(m["op_sel"]):
        raise AsmError("op_sel is not modelled for %s" % b)
    fn = BIN32.get(b)
    nsrc = 2
    if fn is None:
        fn = UN32.get(b)
        nsrc = 1
    if fn is None:
        fn = TRI32.get(b)
        nsrc = 3
    if fn is None:
        return None
    if len(o) != nsrc + 1:
        raise AsmError("%s: expected %d sources, got %r" % (b, nsrc, o))
    p = vdst(o[0])
    gs = srcs(list(range(1, nsrc + 1)))
    if dpp:
        src_l, valid, enabled = _dpp_source(I)
        bound = bool(m.get("bound_ctrl", 0)) or ("bound_ctrl" in m)
        n_d = o[0][1]
        g0 = gs[0]
        rest = gs[1:]
        def f(w):
            a = np.broadcast_to(_u(g0(w)), (64,))
            ok = valid & w.em[src_l]
            a = np.where(ok, a[src_l], U32(0))
            wr = enabled & w.em & (ok | bound)
            r = fn(a, *[g(w) for g in rest])
            np.copyto(w.V[n_d], np.broadcast_to(r, (64,)), where=wr, casting="unsafe")
        return f
    if nsrc == 1:
        g0 = gs[0]
        return lambda w: p(w, fn(g0(w)))
    if nsrc == 2:
        g0, g1 = gs
        return lambda w: p(w, fn(g0(w), g1(w)))
    g0, g1, g2 = gs
    return lambda w: p(w, fn(g0(w), g1(w), g2(w)))


_PK = {
    "v_pk_add_u16": lambda a, b: a + b, "v_pk_sub_u16": lambda a, b: a - b, "v_pk_sub_i16": lambda a, b: a - b, "v_pk_add_i16": lambda a, b: a + b,
    "v_pk_lshlrev_b16": lambda a, b: b << (a & U32(15)), "v_pk_lshrrev_b16": lambda a, b: b >> (a & U32(15)),
    "v_pk_mul_lo_u16": lambda a, b: a * b, "v_pk_max_u16": lambda a, b: np.maximum(a, b), "v_pk_min_u16": lambda a, b: np.minimum(a, b),
    "v_pk_mad_u16": lambda a, b, c: a * b + c,
}


# ------------------------------------------------------------------------------------------------------------------
# memory
# ------------------------------------------------------------------------------------------------------------------
_LOADS = {  # suffix -> (bytes, kind)
    "ubyte": (1, "u8"), "sbyte": (1, "i8"), "ushort": (2, "u16"), "sshort": (2, "i16"), "dword": (4, "w"), "dwordx2": (8, "w"), "dwordx3": (12, "w"),
    "dwordx4": (16, "w"), "ubyte_d16": (1, "d16"), "ubyte_d16_hi": (1, "d16hi"), "sbyte_d16": (1, "d16s"), "sbyte_d16_hi": (1, "d16his"),
    "short_d16": (2, "d16"), "short_d16_hi": (2, "d16hi"),
}
_STORES = {"byte": (1, 0), "short": (2, 0), "dword": (4, 0), "dwordx2": (8, 0), "dwordx3": (12, 0), "dwordx4": (16, 0), "byte_d16_hi": (1, 2), "short_d16_hi": (2, 2)}
_DS_READ = {"b32": (4, "w"), "b64": (8, "w"), "b96": (12, "w"), "b128": (16, "w"), "u8": (1, "u8"), "i8": (1, "i8"), "u16": (2, "u16"), "i16": (2, "i16"),
            "u8_d16": (1, "d16"), "u8_d16_hi": (1, "d16hi"), "i8_d16": (1, "d16s"), "i8_d16_hi": (1, "d16his"), "u16_d16": (2, "d16"), "u16_d16_hi": (2, "d16hi")}
_DS_WRITE = {"b32": (4, 0), "b64": (8, 0), "b96": (12, 0), "b128": (16, 0), "b8": (1, 0), "b16": (2, 0), "b8_d16_hi": (1, 2), "b16_d16_hi": (2, 2)}

_ATOMIC = {
    "add": lambda o, d, wd: (o + d), "sub": lambda o, d, wd: (o - d), "and": lambda o, d, wd: o & d, "or": lambda o, d, wd: o | d, "xor": lambda o, d, wd: o ^ d,
    "umin": lambda o, d, wd: min(o, d), "umax": lambda o, d, wd: max(o, d), "smin": lambda o, d, wd: o if sx(o, wd) < sx(d, wd) else d,
    "smax": lambda o, d, wd: o if sx(o, wd) > sx(d, wd) else d, "swap": lambda o, d, wd: d, "inc": lambda o, d, wd: 0 if o >= d else o + 1,
    "dec": lambda o, d, wd: d if (o == 0 or o > d) else o - 1,
}


def _load_to_regs(w, d0, raw, kind, lanes):
    """raw: (k, nbytes) uint8 for the lanes in `lanes`"""
    V = w.V
    if kind == "w":
        words = np.ascontiguousarray(raw).view("<u4")
        for j in range(words.shape[1]):
            V[d0 + j][lanes] = words[:, j]
    elif kind == "u8":
        V[d0][lanes] = raw[:, 0]
    elif kind == "i8":
        V[d0][lanes] = raw[:, 0].astype(np.int8).astype(I32).astype(U32)
    elif kind == "u16":
        V[d0][lanes] = np.ascontiguousarray(raw).view("<u2")[:, 0]
    elif kind == "i16":
        V[d0][lanes] = np.ascontiguousarray(raw).view("<i2")[:, 0].astype(I32).astype(U32)
    else:
        # d16 loads: gfx950 runs with SRAM ECC — the whole register is written, the other half reads 0 afterwards (measured, round 4)
        if raw.shape[1] == 1:
            v = raw[:, 0].astype(U32)
            if kind.endswith("s"):
                v = (raw[:, 0].astype(np.int8).astype(I16).astype(U16)).astype(U32)
        else:
            v = np.ascontiguousarray(raw).view("<u2")[:, 0].astype(U32)
        if "hi" in kind:
            v = v << U32(16)
        V[d0][lanes] = v


def _regs_to_bytes(w, s0, nbytes, shift, lanes, file="V"):
    R = getattr(w, file)
    if nbytes >= 4:
        cols = [R[s0 + j][lanes] for j in range(nbytes // 4)]
        return np.ascontiguousarray(np.stack(cols, axis=1)).view(U8)
    x = R[s0][lanes] >> U32(8 * shift)
    if nbytes == 1:
        return (x & U32(0xFF)).astype(U8)[:, None]
    return np.ascontiguousarray((x & U32(0xFFFF)).astype("<u2")[:, None]).view(U8)


def _flat_addr_getter(o_vaddr, o_saddr, imm):
    """global addressing: (vaddr64, off) or (vaddr32 offset, saddr64)"""
    if o_saddr is not None and o_saddr[0] == "s":
        gs_ = ssrc(o_saddr, 64)
        if o_vaddr[0] == "off":
            def g(w):
                return np.full(64, gs_(w) + imm, dtype=np.int64)
            return g
        n = o_vaddr[1]
        def g(w):
            return w.V[n].astype(np.int64) + (gs_(w) + imm)
        return g
    n = o_vaddr[1]
    if o_vaddr[2] != 2:
        raise AsmError("64-bit address expected")
    def g(w):
        return (w.V[n].astype(np.int64) | (w.V[n + 1].astype(np.int64) << 32)) + imm
    return g


def _decode_mem(mod, I, pc):
    b, o, m = I.base, I.ops, I.mods
    imm = m.get("offset", 0)
    seg = b.split("_")[0]
    if seg in ("global", "flat", "scratch"):
        I.cls = "vmem"
        rest = b[len(seg) + 1:]
        if seg == "scratch":
            return _decode_scratch(mod, I, rest)
        if rest.startswith("load_"):
            nbytes, kind = _LOADS[rest[5:]]
            d0 = o[0][1]
            if o[0][0] != "v":
                raise AsmError("load destination")
            ga = _flat_addr_getter(o[1], o[2] if len(o) > 2 else None, imm)
            flat = seg == "flat"
            if "lds" in m:
                raise AsmError("load ... lds is not modelled")
            def f(w):
                lanes = w.lanes
                if lanes.size == 0:
                    return
                addr = ga(w)[lanes]
                raw = w.rt.read(w, addr, nbytes, flat)
                _load_to_regs(w, d0, raw, kind, lanes)
            return f
        if rest.startswith("store_"):
            nbytes, shift = _STORES[rest[6:]]
            ga = _flat_addr_getter(o[0], o[2] if len(o) > 2 else None, imm)
            s0, file = o[1][1], ("V" if o[1][0] == "v" else "A")
            flat = seg == "flat"
            def f(w):
                lanes = w.lanes
                if lanes.size == 0:
                    return
                addr = ga(w)[lanes]
                w.rt.write(w, addr, _regs_to_bytes(w, s0, nbytes, shift, lanes, file), flat)
            return f
        if rest.startswith("atomic_"):
            name = rest[7:]
            wd = 64 if name.endswith("_x2") else 32
            if wd == 64:
                name = name[:-3]
            ret = ("glc" in m) or ("sc0" in m)
            if ret:
                d0, oa, od, osa = o[0][1], o[1], o[2], (o[3] if len(o) > 3 else None)
            else:
                d0, oa, od, osa = None, o[0], o[1], (o[2] if len(o) > 2 else None)
            ga = _flat_addr_getter(oa, osa, imm)
            s0 = od[1]
            flat = seg == "flat"
            def f(w):
                nw = wd // 32
                for l in w.lanes:
                    a = int(ga(w)[l])
                    data = int(w.V[s0][l]) | ((int(w.V[s0 + 1][l]) << 32) if nw == 2 else 0)
                    if name == "cmpswap":       # data = {cmp (high), src (low)}
                        if nw == 2:
                            src = data
                            cmpv = int(w.V[s0 + 2][l]) | (int(w.V[s0 + 3][l]) << 32)
                        else:
                            src, cmpv = int(w.V[s0][l]), int(w.V[s0 + 1][l])
                        old = w.rt.read_int(w, a, wd // 8, flat)
                        if old == cmpv:
                            w.rt.write_int(w, a, src, wd // 8, flat)
                    else:
                        old = w.rt.read_int(w, a, wd // 8, flat)
                        new = _ATOMIC[name](old, data, wd) & ((1 << wd) - 1)
                        w.rt.write_int(w, a, new, wd // 8, flat)
                    if ret:
                        w.V[d0][l] = old & M32
                        if nw == 2:
                            w.V[d0 + 1][l] = old >> 32
            return f
        return None
    if seg == "ds":
        I.cls = "lds"
        return _decode_ds(mod, I)
    if seg == "buffer":
        raise AsmError("buffer_* memory instructions are not modelled")
    return None


def _decode_scratch(mod, I, rest):
    o, m = I.ops, I.mods
    imm = m.get("offset", 0)
    def addr_getter(ov, os_):
        gs_ = ssrc(os_) if os_[0] == "s" else (lambda w: 0)
        if ov[0] == "off":
            return lambda w: np.full(64, gs_(w) + imm, dtype=np.int64)
        n = ov[1]
        return lambda w: w.V[n].astype(np.int64) + (gs_(w) + imm)
    if rest.startswith("load_"):
        nbytes, kind = _LOADS[rest[5:]]
        d0 = o[0][1]
        ga = addr_getter(o[1], o[2])
        def f(w):
            lanes = w.lanes
            if lanes.size == 0:
                return
            a = ga(w)[lanes]
            sc = w.scratch_mem()
            if a.min() < 0 or a.max() + nbytes > sc.shape[1]:
                raise SimError("scratch read out of range")
            raw = sc[lanes[:, None], a[:, None] + AR[nbytes]]
            _load_to_regs(w, d0, raw, kind, lanes)
        return f
    if rest.startswith("store_"):
        nbytes, shift = _STORES[rest[6:]]
        ga = addr_getter(o[0], o[2])
        s0, file = o[1][1], ("V" if o[1][0] == "v" else "A")
        def f(w):
            lanes = w.lanes
            if lanes.size == 0:
                return
            a = ga(w)[lanes]
            sc = w.scratch_mem()
            if a.min() < 0 or a.max() + nbytes > sc.shape[1]:
                raise SimError("scratch write out of range")
            sc[lanes[:, None], a[:, None] + AR[nbytes]] = _regs_to_bytes(w, s0, nbytes, shift, lanes, file)
        return f
    return None


XCHG_MODE = os.environ.get("GFXSIM_XCHG") or None      # None | "desc" | "flaky:<per mille>"
XCHG_FLAKY = int(XCHG_MODE.split(":")[1]) if XCHG_MODE and XCHG_MODE.startswith("flaky:") else 0
XCHG_AFTER = [int(XCHG_MODE.split(":")[2])] if XCHG_MODE and XCHG_MODE.count(":") == 2 else [0]    # "flaky:<per mille>:<n>": only after n exchanges (the probe passes)
_xchg_rng = np.random.default_rng(int(os.environ.get("GFXSIM_XCHG_SEED", "165")))


class Race:
    """racecheck: per LDS byte, who wrote / read it since the workgroup's last barrier.  The interpreter runs the wavefronts of a
    workgroup one slice after the other, so a race cannot change ITS result — this flags accesses whose order the hardware does not fix:
    two wavefronts touching the same byte between two barriers, one of them writing, not both atomically."""

    def __init__(self, n):
        self.epoch = 1
        self.wr_wave = np.full(n, -1, dtype=np.int16)
        self.wr_epoch = np.zeros(n, dtype=np.int32)
        self.wr_atomic = np.zeros(n, dtype=bool)
        self.rd_wave = np.full(n, -1, dtype=np.int16)        # -2: several wavefronts
        self.rd_epoch = np.zeros(n, dtype=np.int32)

    def access(self, w, idx, kind):
        me, E = w.index, self.epoch
        wcur = self.wr_epoch[idx] == E
        other_w = wcur & (self.wr_wave[idx] != me)
        if kind == "r":
            if other_w.any():
                w.mem._finding("LDS race: read of a byte another wavefront wrote since the last barrier", int(idx[np.nonzero(other_w)[0][0]]))
            rcur = self.rd_epoch[idx] == E
            self.rd_wave[idx] = np.where(rcur & (self.rd_wave[idx] != me), -2, me)
            self.rd_epoch[idx] = E
            return
        rcur = (self.rd_epoch[idx] == E) & (self.rd_wave[idx] != me)
        if kind == "a":
            other_w &= ~self.wr_atomic[idx]
        if other_w.any():
            w.mem._finding("LDS race: write to a byte another wavefront wrote since the last barrier" + (" (atomic vs plain)" if kind == "a" else ""),
                           int(idx[np.nonzero(other_w)[0][0]]))
        if rcur.any():
            w.mem._finding("LDS race: write to a byte another wavefront read since the last barrier", int(idx[np.nonzero(rcur)[0][0]]))
        self.wr_wave[idx] = me
        self.wr_epoch[idx] = E
        self.wr_atomic[idx] = kind == "a"


def _lds_rd(w, a, n):
    """memcheck: LDS is not cleared between workgroups on the device — a read of bytes this workgroup has not written sees leftovers"""
    if w.wg.race is not None and a.size:
        w.wg.race.access(w, (a[:, None] + np.arange(n)).ravel(), "r")
    d = w.wg.lds_def
    if d is not None and a.size:
        ok = d[a[:, None] + np.arange(n)]
        if not ok.all():
            bad = np.nonzero(~ok.all(axis=1))[0][0]
            w.mem._finding("LDS read of bytes this workgroup has not written", int(a[bad]))


def _lds_wr(w, a, n):
    if w.wg.race is not None and a.size:
        w.wg.race.access(w, (a[:, None] + np.arange(n)).ravel(), "w")
    d = w.wg.lds_def
    if d is not None and a.size:
        d[(a[:, None] + np.arange(n)).ravel()] = True


def _lds_check(w, a, n):
    if a.size and (a.min() < 0 or a.max() + n > w.lds.size):
        raise SimError("LDS access out of range: byte %d..%d of %d" % (int(a.min()), int(a.max()) + n, w.lds.size))


def _decode_ds(mod, I):
    b, o, m = I.base, I.ops, I.mods
    imm = m.get("offset", 0)
    name = b[3:]
    if "gds" in m:
        raise AsmError("gds")
    if name.startswith("read2") or name.startswith("write2"):
        st64 = "st64" in name
        el = 8 if name.endswith("b64") else 4
        mul = el * (64 if st64 else 1)
        o0, o1 = m.get("offset0", 0) * mul, m.get("offset1", 0) * mul
        if name.startswith("read2"):
            d0, na = o[0][1], o[1][1]
            def f(w):
                lanes = w.lanes
                if lanes.size == 0:
                    return
                base = w.V[na][lanes].astype(np.int64)
                for k, off in enumerate((o0, o1)):
                    a = (base + off) & M32
                    _lds_check(w, a, el)
                    _lds_rd(w, a, el)
                    raw = w.lds[a[:, None] + AR[el]]
                    _load_to_regs(w, d0 + k * (el // 4), raw, "w", lanes)
            return f
        na, s0, s1 = o[0][1], o[1][1], o[2][1]
        def f(w):
            lanes = w.lanes
            if lanes.size == 0:
                return
            base = w.V[na][lanes].astype(np.int64)
            for off, s in ((o0, s0), (o1, s1)):
                a = (base + off) & M32
                _lds_check(w, a, el)
                _lds_wr(w, a, el)
                w.lds[(a[:, None] + AR[el]).ravel()] = _regs_to_bytes(w, s, el, 0, lanes).ravel()
        return f
    if name.startswith("read_"):
        nbytes, kind = _DS_READ[name[5:]]
        d0, na = o[0][1], o[1][1]
        def f(w):
            lanes = w.lanes
            if lanes.size == 0:
                return
            a = (w.V[na][lanes].astype(np.int64) + imm) & M32
            _lds_check(w, a, nbytes)
            _lds_rd(w, a, nbytes)
            _load_to_regs(w, d0, w.lds[a[:, None] + AR[nbytes]], kind, lanes)
        return f
    if name.startswith("write_"):
        nbytes, shift = _DS_WRITE[name[6:]]
        na, s0 = o[0][1], o[1][1]
        file = "V" if o[1][0] == "v" else "A"
        def f(w):
            lanes = w.lanes
            if lanes.size == 0:
                return
            a = (w.V[na][lanes].astype(np.int64) + imm) & M32
            _lds_check(w, a, nbytes)
            _lds_wr(w, a, nbytes)
            w.lds[(a[:, None] + AR[nbytes]).ravel()] = _regs_to_bytes(w, s0, nbytes, shift, lanes, file).ravel()
        return f
    if name in ("bpermute_b32", "permute_b32"):
        d0, na, ns = o[0][1], o[1][1], o[2][1]
        back = name == "bpermute_b32"
        def f(w):
            idx = ((w.V[na].astype(np.int64) + imm) >> 2) & 63
            data = np.where(w.em, w.V[ns], U32(0))
            if back:
                res = data[idx]
            else:
                res = np.zeros(64, dtype=U32)
                for l in w.lanes:
                    res[idx[l]] = data[l]
            np.copyto(w.V[d0], res, where=w.em)
        return f
    if name == "swizzle_b32":
        d0, ns = o[0][1], o[1][1]
        pat = imm
        def f(w):
            data = np.where(w.em, w.V[ns], U32(0))
            if pat & 0x8000:
                q = [(pat >> (2 * i)) & 3 for i in range(4)]
                src = (LANE & ~3) + np.array([q[i & 3] for i in range(64)])
            else:
                and_m, or_m, xor_m = pat & 31, (pat >> 5) & 31, (pat >> 10) & 31
                l5 = LANE & 31
                src = (LANE & 32) | (((l5 & and_m) | or_m) ^ xor_m)
            np.copyto(w.V[d0], data[src], where=w.em)
        return f
    # atomics: ds_<op>[_rtn]_<type>
    parts = name.split("_")
    rtn = "rtn" in parts
    ty = parts[-1]
    opn = parts[0]
    wd = 64 if ty.endswith("64") else 32
    signed = ty[0] == "i"
    key = {"min": "smin" if signed else "umin", "max": "smax" if signed else "umax", "wrxchg": "swap"}.get(opn, opn)
    if key == "cmpst":
        if rtn:
            d0, na, nc, nd = o[0][1], o[1][1], o[2][1], o[3][1]
        else:
            d0, na, nc, nd = None, o[0][1], o[1][1], o[2][1]
        def f(w):
            for l in w.lanes:
                a = (int(w.V[na][l]) + imm) & M32
                _lds_check(w, np.array([a]), wd // 8)
                if w.wg.race is not None:
                    w.wg.race.access(w, np.arange(a, a + wd // 8), "a")
                old = int.from_bytes(w.lds[a:a + wd // 8].tobytes(), "little")
                cmpv = int(w.V[nc][l]) | ((int(w.V[nc + 1][l]) << 32) if wd == 64 else 0)
                if old == cmpv:
                    new = int(w.V[nd][l]) | ((int(w.V[nd + 1][l]) << 32) if wd == 64 else 0)
                    w.lds[a:a + wd // 8] = np.frombuffer(new.to_bytes(wd // 8, "little"), dtype=U8)
                    if w.wg.lds_def is not None:
                        w.wg.lds_def[a:a + wd // 8] = True
                if rtn:
                    w.V[d0][l] = old & M32
                    if wd == 64:
                        w.V[d0 + 1][l] = old >> 32
        return f
    if key not in _ATOMIC:
        return None
    fn = _ATOMIC[key]
    if rtn:
        d0, na, nd = o[0][1], o[1][1], o[2][1]
    else:
        d0, na, nd = None, o[0][1], o[1][1]
    nb = wd // 8
    swap = key == "swap"
    def f(w):
        lds = w.lds
        lanes = w.lanes              # ascending lane order — what DESIGN 4.1 assumes of the chip for ds_wrxchg and checks there
        if swap and XCHG_MODE is not None:
            # what if the chip served an exchange in another order?  (tests of the product's safety net: probe, per-exchange check, k_links2)
            if XCHG_MODE == "desc":
                lanes = lanes[::-1]
            elif XCHG_AFTER[0] > 0:
                XCHG_AFTER[0] -= 1
            elif _xchg_rng.integers(0, 1000) < XCHG_FLAKY and lanes.size > 1:
                lanes = lanes.copy()
                i = int(_xchg_rng.integers(0, lanes.size - 1))
                lanes[i], lanes[i + 1] = lanes[i + 1], lanes[i]
        for l in lanes:
            a = (int(w.V[na][l]) + imm) & M32
            if a < 0 or a + nb > lds.size:
                raise SimError("LDS atomic out of range: %d" % a)
            if w.wg.race is not None:
                w.wg.race.access(w, np.arange(a, a + nb), "a")
            if w.wg.lds_def is not None:
                d_ = w.wg.lds_def
                if not d_[a:a + nb].all():
                    w.mem._finding("LDS read of bytes this workgroup has not written", a)
                d_[a:a + nb] = True
            old = int.from_bytes(lds[a:a + nb].tobytes(), "little")
            data = int(w.V[nd][l]) | ((int(w.V[nd + 1][l]) << 32) if wd == 64 else 0)
            new = fn(old, data, wd) & ((1 << wd) - 1)
            lds[a:a + nb] = np.frombuffer(new.to_bytes(nb, "little"), dtype=U8)
            if rtn:
                w.V[d0][l] = old & M32
                if wd == 64:
                    w.V[d0 + 1][l] = old >> 32
    return f
