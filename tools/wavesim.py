"""wavesim — a small wave64 interpreter for the gfx950 instruction subset the hand-written stage-B engines use.

Why: there is no GPU in the build container, and the engines of szl_kernels_match2.hip / szl_kernels_match9.hip are inline
assembly.  This interpreter runs the engine's INSTRUCTION TEXT (the preprocessed macro text with its %[name] operands, exactly
what the compiler is handed) on a tile staged the way the kernel stages it, so that
  * the match tables it produces can be diffed against oracle/szl_model.c on the CPU before a GPU minute is spent,
  * s_waitcnt discipline is checked (a VGPR with an LDS read in flight must not be touched before the count says so),
  * VALU / SALU / LDS / branch wave-instructions per position and lane occupancy per phase are counted (the quantities the
    round-3 PMC profile showed the kernel is bound by).
It is a functional model: no timing.  Waves of a workgroup are interleaved round-robin every few hundred instructions and
share the LDS image (the only shared state is the tile counter, taken with ds_add_rtn_u32).

Not part of the product and never shipped; tools/ and tests/ only.
"""
import re
from collections import deque

import numpy as np

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1
LANES = np.arange(64, dtype=np.uint64)


def _mask_to_bool(m):
    return ((np.uint64(m) >> LANES) & np.uint64(1)).astype(bool)


def _bool_to_mask(b):
    return int(np.bitwise_or.reduce(np.where(b, np.uint64(1) << LANES, np.uint64(0))))


class SimError(Exception):
    pass


_cmp_ops = {
    "eq": lambda a, b: a == b, "ne": lambda a, b: a != b, "lg": lambda a, b: a != b, "lt": lambda a, b: a < b, "le": lambda a, b: a <= b,
    "gt": lambda a, b: a > b, "ge": lambda a, b: a >= b,
}


class Program:
    """Parsed instruction text.  Lines: 'label:' or 'op a, b, c mod:val'."""

    def __init__(self, text):
        self.ins = []          # (op, [operands], {mods}, phase)
        self.labels = {}       # name -> [pcs]
        phase = ""
        for raw in text.replace("\\n", "\n").replace("\\t", " ").split("\n"):
            line = raw.strip()
            if not line:
                continue
            if line.startswith(";"):
                m = re.match(r";\s*@phase\s+(\S+)", line)
                if m:
                    phase = m.group(1)
                continue
            line = line.split(";")[0].strip()
            m = re.match(r"^([A-Za-z0-9_.]+):$", line)
            if m:
                self.labels.setdefault(m.group(1), []).append(len(self.ins))
                continue
            parts = line.split(None, 1)
            op = parts[0]
            ops, mods = [], {}
            if len(parts) > 1:
                rest = parts[1]
                toks = [t.strip() for t in rest.split(",")]
                # modifiers ride on the last token, space separated
                last = toks[-1].split()
                toks[-1] = last[0] if last else ""
                for mod in last[1:]:
                    if ":" in mod:
                        k, v = mod.split(":")
                        mods[k] = int(v, 0)
                    else:
                        mods[mod] = 1
                ops = [t for t in toks if t != ""]
            self.ins.append((op, ops, mods, phase))

    def target(self, pc, lab):
        m = re.match(r"^(\d+)([fb])$", lab)
        if m:
            cands = self.labels.get(m.group(1), [])
            if m.group(2) == "f":
                c = [x for x in cands if x > pc]
                if not c:
                    raise SimError("no forward label %s from %d" % (lab, pc))
                return min(c)
            c = [x for x in cands if x <= pc]
            if not c:
                raise SimError("no backward label %s from %d" % (lab, pc))
            return max(c)
        if lab in self.labels:
            return self.labels[lab][0]
        raise SimError("unknown label " + lab)


class Wave:
    def __init__(self, prog, lds, vregs, sregs, globals_, lane_base=0):
        """vregs: dict name -> initial value (int or array); sregs: name -> int; globals_: name(of 64-bit sgpr operand) -> np.uint32 array"""
        self.p = prog
        self.lds = lds
        self.v = {k: (np.full(64, v, dtype=np.uint32) if np.isscalar(v) else np.array(v, dtype=np.uint32)) for k, v in vregs.items()}
        self.s = dict(sregs)
        self.g = globals_
        self.exec = M64
        self.vcc = 0
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.done = False
        self.fifo = deque()          # VGPR names with an LDS read in flight, in issue order
        self.cnt = {}                # (phase, class) -> wave-instructions
        self.lanes = {}              # (phase, class) -> active lanes summed
        self.steps = 0
        self.trace = None
        self.tail_flag = None        # name of an SGPR: instructions issued while it is non-zero are booked under phase "tail"
        self.tail_steps = 0

    # ---- operand access
    def _isv(self, t):
        return t.startswith("%[") and t[2:-1] in self.v

    def _name(self, t):
        return t[2:-1]

    def rv(self, t):
        """read a 32-bit per-lane source operand"""
        if t.startswith("%["):
            n = t[2:-1]
            if n in self.v:
                if n in self.fifo:
                    raise SimError("pc %d: VGPR %s read with its LDS load in flight (missing s_waitcnt)" % (self.pc, n))
                return self.v[n]
            if n in self.s:
                return np.uint32(self.s[n] & M32)
            raise SimError("unknown operand " + t)
        if t == "vcc_lo":
            return np.uint32(self.vcc & M32)
        if t == "vcc_hi":
            return np.uint32(self.vcc >> 32)
        if t == "exec_lo":
            return np.uint32(self.exec & M32)
        if t == "exec_hi":
            return np.uint32(self.exec >> 32)
        if t == "m0":
            return np.uint32(self.m0)
        return np.uint32(int(t, 0) & M32)

    def rs(self, t, bits=32):
        """read a scalar operand"""
        mask = M64 if bits == 64 else M32
        if t.startswith("%["):
            n = t[2:-1]
            if n not in self.s:
                raise SimError("pc %d: scalar read of non-scalar %s" % (self.pc, t))
            return self.s[n] & mask
        if t == "exec":
            return self.exec
        if t == "vcc":
            return self.vcc
        if t in ("exec_lo", "exec_hi", "vcc_lo", "vcc_hi", "m0"):
            return int(self.rv(t))
        if t == "scc":
            return self.scc
        return int(t, 0) & mask

    def ws(self, t, val, bits=32):
        val &= M64 if bits == 64 else M32
        if t == "exec":
            self.exec = val
        elif t == "vcc":
            self.vcc = val
        elif t == "m0":
            self.m0 = val
        elif t == "exec_lo":
            self.exec = (self.exec & ~M32) | val
        elif t == "exec_hi":
            self.exec = (self.exec & M32) | (val << 32)
        elif t == "vcc_lo":
            self.vcc = (self.vcc & ~M32) | val
        elif t == "vcc_hi":
            self.vcc = (self.vcc & M32) | (val << 32)
        elif t.startswith("%["):
            n = t[2:-1]
            if n in self.v:
                raise SimError("pc %d: scalar write to VGPR %s" % (self.pc, n))
            self.s[n] = val
        else:
            raise SimError("bad scalar destination " + t)

    def wv(self, t, res, em):
        n = t[2:-1]
        if n not in self.v:
            raise SimError("pc %d: vector write to non-VGPR %s" % (self.pc, t))
        if n in self.fifo:
            raise SimError("pc %d: VGPR %s written with its LDS load in flight" % (self.pc, n))
        res = np.broadcast_to(np.asarray(res, dtype=np.uint32), (64,))
        self.v[n] = np.where(em, res, self.v[n])

    def _count(self, phase, cls, nl):
        if self.tail_flag and self.s.get(self.tail_flag):
            phase = "tail:" + phase
            self.tail_steps += 1
        k = (phase, cls)
        self.cnt[k] = self.cnt.get(k, 0) + 1
        self.lanes[k] = self.lanes.get(k, 0) + nl

    # ---- one instruction
    def step(self):
        op, o, mods, phase = self.p.ins[self.pc]
        pc = self.pc
        self.pc += 1
        self.steps += 1
        if self.trace is not None:
            self.trace(self, pc, op, o)
        if op.startswith("v_"):
            em = _mask_to_bool(self.exec)
            self._count(phase, "valu", int(em.sum()))
            self._valu(op, o, mods, em)
        elif op.startswith("ds_"):
            em = _mask_to_bool(self.exec)
            self._count(phase, "lds", int(em.sum()))
            self._ds(op, o, mods, em)
        elif op.startswith("global_"):
            em = _mask_to_bool(self.exec)
            self._count(phase, "vmem", int(em.sum()))
            self._global(op, o, mods, em)
        elif op.startswith("s_"):
            self._count(phase, "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu", 0)
            self._salu(op, o, mods, pc)
        else:
            raise SimError("pc %d: unknown instruction %s" % (pc, op))
        if self.pc >= len(self.p.ins):
            self.done = True

    def _valu(self, op, o, mods, em):
        i32 = lambda a: np.asarray(a, dtype=np.uint32).astype(np.int32)   # noqa: E731
        u64 = lambda a: np.asarray(a, dtype=np.uint32).astype(np.uint64)  # noqa: E731
        m = re.match(r"^v_cmp(x?)_(\w+)_([iu])32$", op)
        if m:
            x, c, ty = m.groups()
            a, b = self.rv(o[1]), self.rv(o[2])
            if ty == "i":
                a, b = i32(a), i32(b)
            r = np.broadcast_to(_cmp_ops[c](a, b), (64,)) & em
            mask = _bool_to_mask(r)
            self.ws(o[0], mask, 64)
            if x:
                self.exec = mask
            return
        if op == "v_mov_b32":
            return self.wv(o[0], self.rv(o[1]), em)
        if op == "v_add_u32":
            return self.wv(o[0], (u64(self.rv(o[1])) + u64(self.rv(o[2]))) & M32, em)
        if op == "v_sub_u32":
            return self.wv(o[0], (u64(self.rv(o[1])) - u64(self.rv(o[2]))) & M32, em)
        if op == "v_subrev_u32":
            return self.wv(o[0], (u64(self.rv(o[2])) - u64(self.rv(o[1]))) & M32, em)
        if op == "v_add3_u32":
            return self.wv(o[0], (u64(self.rv(o[1])) + u64(self.rv(o[2])) + u64(self.rv(o[3]))) & M32, em)
        if op == "v_lshl_add_u32":
            return self.wv(o[0], ((u64(self.rv(o[1])) << (u64(self.rv(o[2])) & 31)) + u64(self.rv(o[3]))) & M32, em)
        if op == "v_add_lshl_u32":
            return self.wv(o[0], ((u64(self.rv(o[1])) + u64(self.rv(o[2]))) << (u64(self.rv(o[3])) & 31)) & M32, em)
        if op == "v_lshl_or_b32":
            return self.wv(o[0], ((u64(self.rv(o[1])) << (u64(self.rv(o[2])) & 31)) & M32) | u64(self.rv(o[3])), em)
        if op == "v_and_or_b32":
            return self.wv(o[0], (u64(self.rv(o[1])) & u64(self.rv(o[2]))) | u64(self.rv(o[3])), em)
        if op == "v_or3_b32":
            return self.wv(o[0], u64(self.rv(o[1])) | u64(self.rv(o[2])) | u64(self.rv(o[3])), em)
        if op == "v_and_b32":
            return self.wv(o[0], u64(self.rv(o[1])) & u64(self.rv(o[2])), em)
        if op == "v_or_b32":
            return self.wv(o[0], u64(self.rv(o[1])) | u64(self.rv(o[2])), em)
        if op == "v_xor_b32":
            return self.wv(o[0], u64(self.rv(o[1])) ^ u64(self.rv(o[2])), em)
        if op == "v_lshlrev_b32":
            return self.wv(o[0], (u64(self.rv(o[2])) << (u64(self.rv(o[1])) & 31)) & M32, em)
        if op == "v_lshrrev_b32":
            return self.wv(o[0], u64(self.rv(o[2])) >> (u64(self.rv(o[1])) & 31), em)
        if op == "v_ashrrev_i32":
            return self.wv(o[0], (i32(self.rv(o[2])) >> (i32(self.rv(o[1])) & 31)).astype(np.uint32), em)
        if op == "v_min_u32":
            return self.wv(o[0], np.minimum(u64(self.rv(o[1])), u64(self.rv(o[2]))), em)
        if op == "v_max_u32":
            return self.wv(o[0], np.maximum(u64(self.rv(o[1])), u64(self.rv(o[2]))), em)
        if op == "v_min3_u32":
            return self.wv(o[0], np.minimum(np.minimum(u64(self.rv(o[1])), u64(self.rv(o[2]))), u64(self.rv(o[3]))), em)
        if op == "v_min_i32":
            return self.wv(o[0], np.minimum(i32(self.rv(o[1])), i32(self.rv(o[2]))).astype(np.uint32), em)
        if op == "v_max_i32":
            return self.wv(o[0], np.maximum(i32(self.rv(o[1])), i32(self.rv(o[2]))).astype(np.uint32), em)
        if op == "v_med3_i32":
            a, b, c = i32(self.rv(o[1])), i32(self.rv(o[2])), i32(self.rv(o[3]))
            a, b, c = np.broadcast_to(a, (64,)), np.broadcast_to(b, (64,)), np.broadcast_to(c, (64,))
            return self.wv(o[0], np.sort(np.stack([a, b, c]), axis=0)[1].astype(np.uint32), em)
        if op == "v_ffbl_b32":
            a = np.broadcast_to(u64(self.rv(o[1])), (64,))
            low = a & (~a + np.uint64(1))
            r = np.where(a == 0, np.uint64(0xFFFFFFFF), np.log2(np.maximum(low, 1).astype(np.float64)).astype(np.uint64))
            return self.wv(o[0], r, em)
        if op == "v_alignbyte_b32":      # D = ({S0,S1} >> (8 * S2[1:0])) & 0xffffffff
            hi, lo, sh = u64(self.rv(o[1])), u64(self.rv(o[2])), (u64(self.rv(o[3])) & 3) * 8
            return self.wv(o[0], (((hi << np.uint64(32)) | lo) >> sh) & M32, em)
        if op == "v_alignbit_b32":
            hi, lo, sh = u64(self.rv(o[1])), u64(self.rv(o[2])), (u64(self.rv(o[3])) & 31)
            return self.wv(o[0], (((hi << np.uint64(32)) | lo) >> sh) & M32, em)
        if op == "v_bfe_u32":
            a, off, w = u64(self.rv(o[1])), u64(self.rv(o[2])) & 31, u64(self.rv(o[3])) & 31
            return self.wv(o[0], (a >> off) & ((np.uint64(1) << w) - np.uint64(1)), em)
        if op == "v_cndmask_b32":       # D = sel ? S1 : S0 ; sel operand o[3] (vcc or sgpr pair)
            sel = _mask_to_bool(self.rs(o[3], 64))
            a = np.broadcast_to(self.rv(o[1]), (64,))
            b = np.broadcast_to(self.rv(o[2]), (64,))
            return self.wv(o[0], np.where(sel, b, a), em)
        if op in ("v_subrev_co_u32", "v_sub_co_u32", "v_add_co_u32"):
            a, b = u64(self.rv(o[2])), u64(self.rv(o[3]))
            if op == "v_subrev_co_u32":
                a, b = b, a
            if op == "v_add_co_u32":
                r = a + b
                c = r > M32
            else:
                r = a - b
                c = a < b
            self.wv(o[0], r & M32, em)
            self.ws(o[1], _bool_to_mask(np.broadcast_to(c, (64,)) & em), 64)
            return
        if op == "v_mbcnt_lo_u32_b32":
            msk = int(self.rv(o[1]))
            r = np.array([bin(msk & ((1 << min(l, 32)) - 1)).count("1") for l in range(64)], dtype=np.uint64)
            return self.wv(o[0], (r + u64(self.rv(o[2]))) & M32, em)
        if op == "v_mbcnt_hi_u32_b32":
            msk = int(self.rv(o[1]))
            r = np.array([bin(msk & ((1 << max(l - 32, 0)) - 1)).count("1") for l in range(64)], dtype=np.uint64)
            return self.wv(o[0], (r + u64(self.rv(o[2]))) & M32, em)
        if op == "v_readfirstlane_b32":
            src = self.rv(o[1])
            lane = (self.exec & -self.exec).bit_length() - 1 if self.exec else 0
            return self.ws(o[0], int(np.broadcast_to(src, (64,))[lane]))
        if op == "v_mul_i32_i24":
            s24 = lambda a: ((np.asarray(a, dtype=np.uint32).astype(np.int64) & 0xFFFFFF) ^ 0x800000) - 0x800000   # noqa: E731
            return self.wv(o[0], (s24(self.rv(o[1])) * s24(self.rv(o[2]))) & M32, em)
        if op == "v_mul_u32_u24":
            return self.wv(o[0], ((u64(self.rv(o[1])) & 0xFFFFFF) * (u64(self.rv(o[2])) & 0xFFFFFF)) & M32, em)
        if op == "v_mad_u32_u24":
            return self.wv(o[0], ((u64(self.rv(o[1])) & 0xFFFFFF) * (u64(self.rv(o[2])) & 0xFFFFFF) + u64(self.rv(o[3]))) & M32, em)
        if op == "v_perm_b32":          # D.byte[i] = selector byte i picks from {S0,S1} (S1 = bytes 0-3, S0 = bytes 4-7)
            s0, s1, sel = (np.broadcast_to(u64(self.rv(x)), (64,)) for x in o[1:4])
            comb = (s0 << np.uint64(32)) | s1
            r = np.zeros(64, dtype=np.uint64)
            for i in range(4):
                sb = (sel >> np.uint64(8 * i)) & np.uint64(0xFF)
                byte = np.where(sb < 8, (comb >> ((sb & np.uint64(7)) * np.uint64(8))) & np.uint64(0xFF),
                                np.where(sb == 12, np.uint64(0), np.where(sb >= 13, np.uint64(0xFF), np.uint64(0))))
                r |= byte << np.uint64(8 * i)
            return self.wv(o[0], r, em)
        raise SimError("pc %d: VALU op %s not modelled" % (self.pc - 1, op))

    def _ds(self, op, o, mods, em):
        off = mods.get("offset", 0)
        if op in ("ds_read_u8", "ds_read_u16", "ds_read_b32", "ds_read_i8", "ds_read_u8_d16_hi", "ds_read_u16_d16_hi"):
            addr = (np.broadcast_to(self.rv(o[1]), (64,)).astype(np.int64) + off)
            n = {"ds_read_u8": 1, "ds_read_i8": 1, "ds_read_u16": 2, "ds_read_b32": 4, "ds_read_u8_d16_hi": 1, "ds_read_u16_d16_hi": 2}[op]
            res = np.zeros(64, dtype=np.uint64)
            a = np.where(em, addr, 0)
            if ((a < 0) | (a + n > self.lds.size)).any():
                # out-of-range LDS reads return 0 on the hardware; flag it: the engines never rely on that
                bad = np.where(em & ((addr < 0) | (addr + n > self.lds.size)))[0]
                raise SimError("pc %d: LDS read out of range, lanes %s addr %s" % (self.pc - 1, bad[:4], addr[bad[:4]]))
            if n > 1 and (a % n != 0).any():
                self.cnt[("", "lds_unaligned")] = self.cnt.get(("", "lds_unaligned"), 0) + 1
            for k in range(n):
                res |= self.lds[a + k].astype(np.uint64) << np.uint64(8 * k)
            dn = self._name(o[0])
            # the destination may equal the address register: the read takes its address at issue
            if op.endswith("_d16_hi"):
                # gfx950 runs with SRAM ECC: a d16 load writes the WHOLE register (the other half reads 0 afterwards) — measured:
                # the filter bytes of k_match9 merged this way came out wrong on the device (profiles/r04/c1_lab.log)
                res = res.astype(np.uint32) << np.uint32(16)
            if dn in self.fifo:
                raise SimError("pc %d: VGPR %s is already the target of an LDS read in flight" % (self.pc - 1, dn))
            self.v[dn] = np.where(em, res.astype(np.uint32), self.v[dn])
            self.fifo.append(dn)
            return
        if op == "ds_write_b8":
            addr = (np.broadcast_to(self.rv(o[0]), (64,)).astype(np.int64) + off)
            data = np.broadcast_to(self.rv(o[1]), (64,))
            for l in np.where(em)[0]:
                a = int(addr[l])
                if a < 0 or a >= self.lds.size:
                    raise SimError("pc %d: LDS write out of range" % (self.pc - 1))
                self.lds[a] = int(data[l]) & 0xFF
            self.fifo.append("@write")
            return
        if op == "ds_bpermute_b32":       # D[lane] = S[(addr[lane] >> 2) & 63]; data of inactive source lanes reads as 0
            addr = np.broadcast_to(self.rv(o[1]), (64,)).astype(np.int64) + off
            data = np.where(em, np.broadcast_to(self.rv(o[2]), (64,)), 0).astype(np.uint32)
            res = data[(addr >> 2) & 63]
            dn = self._name(o[0])
            if dn in self.fifo:
                raise SimError("pc %d: VGPR %s is already the target of an LDS read in flight" % (self.pc - 1, dn))
            self.v[dn] = np.where(em, res, self.v[dn])
            self.fifo.append(dn)
            return
        if op == "ds_write_b32":
            addr = (np.broadcast_to(self.rv(o[0]), (64,)).astype(np.int64) + off)
            data = np.broadcast_to(self.rv(o[1]), (64,))
            for l in np.where(em)[0]:
                a = int(addr[l])
                self.lds[a:a + 4] = np.frombuffer(int(data[l]).to_bytes(4, "little"), dtype=np.uint8)
            self.fifo.append("@write")
            return
        if op == "ds_add_rtn_u32":
            addr = (np.broadcast_to(self.rv(o[1]), (64,)).astype(np.int64) + off)
            data = np.broadcast_to(self.rv(o[2]), (64,))
            res = np.zeros(64, dtype=np.uint32)
            for l in np.where(em)[0]:
                a = int(addr[l])
                old = int.from_bytes(self.lds[a:a + 4].tobytes(), "little")
                res[l] = old
                self.lds[a:a + 4] = np.frombuffer(((old + int(data[l])) & M32).to_bytes(4, "little"), dtype=np.uint8)
            dn = self._name(o[0])
            self.v[dn] = np.where(em, res, self.v[dn])
            self.fifo.append(dn)
            return
        if op == "ds_min_u32":             # ds_min_u32 addr, data [offset:N]
            addr = (np.broadcast_to(self.rv(o[0]), (64,)).astype(np.int64) + off)
            data = np.broadcast_to(self.rv(o[1]), (64,))
            for l in np.where(em)[0]:
                a = int(addr[l])
                old = int.from_bytes(self.lds[a:a + 4].tobytes(), "little")
                self.lds[a:a + 4] = np.frombuffer(min(old, int(data[l]) & M32).to_bytes(4, "little"), dtype=np.uint8)
            self.fifo.append("@write")
            return
        raise SimError("pc %d: DS op %s not modelled" % (self.pc - 1, op))

    def _global(self, op, o, mods, em):
        if op == "global_store_dword":          # global_store_dword voff, vdata, sbase [offset:N]
            arr = self.g[self._name(o[2])]
            off = np.broadcast_to(self.rv(o[0]), (64,)).astype(np.int64) + mods.get("offset", 0)
            data = np.broadcast_to(self.rv(o[1]), (64,))
            if (off[em] % 4 != 0).any():
                raise SimError("unaligned global store")
            idx = off[em] // 4
            if ((idx < 0) | (idx >= arr.size)).any():
                raise SimError("pc %d: global store out of range: %s" % (self.pc - 1, idx[(idx < 0) | (idx >= arr.size)][:4]))
            arr[idx] = data[em]
            return
        raise SimError("global op %s not modelled" % op)

    def _salu(self, op, o, mods, pc):
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", " ".join(o) + " " + " ".join(mods))
            txt = " ".join(self.p.ins[pc][1])
            m = re.search(r"lgkmcnt\((\d+)\)", txt)
            if m:
                n = int(m.group(1))
                while len(self.fifo) > n:
                    self.fifo.popleft()
            return
        if op in ("s_nop", "s_sleep", "s_setprio"):
            return
        if op in ("s_memrealtime", "s_memtime"):
            return self.ws(o[0], self.steps, 64)
        if op == "s_mov_b64":
            return self.ws(o[0], self.rs(o[1], 64) if not re.match(r"^-?\d", o[1]) else int(o[1], 0) & M64, 64)
        if op == "s_mov_b32":
            return self.ws(o[0], self.rs(o[1]))
        if op == "s_movk_i32":
            return self.ws(o[0], int(o[1], 0) & M32)
        b64 = {"s_and_b64": lambda a, b: a & b, "s_or_b64": lambda a, b: a | b, "s_xor_b64": lambda a, b: a ^ b,
               "s_andn2_b64": lambda a, b: a & ~b, "s_orn2_b64": lambda a, b: a | ~b, "s_nor_b64": lambda a, b: ~(a | b),
               "s_nand_b64": lambda a, b: ~(a & b)}
        if op in b64:
            r = b64[op](self.rs(o[1], 64), self.rs(o[2], 64)) & M64
            self.ws(o[0], r, 64)
            self.scc = int(r != 0)
            return
        if op == "s_not_b64":
            r = ~self.rs(o[1], 64) & M64
            self.ws(o[0], r, 64)
            self.scc = int(r != 0)
            return
        if op in ("s_and_saveexec_b64", "s_or_saveexec_b64", "s_andn2_saveexec_b64"):
            old = self.exec
            src = self.rs(o[1], 64)
            self.exec = {"s_and_saveexec_b64": src & old, "s_or_saveexec_b64": src | old, "s_andn2_saveexec_b64": src & ~old}[op] & M64
            self.ws(o[0], old, 64)
            self.scc = int(self.exec != 0)
            return
        b32 = {"s_and_b32": lambda a, b: a & b, "s_or_b32": lambda a, b: a | b, "s_xor_b32": lambda a, b: a ^ b, "s_andn2_b32": lambda a, b: a & ~b,
               "s_lshl_b32": lambda a, b: a << (b & 31), "s_lshr_b32": lambda a, b: a >> (b & 31)}
        if op in b32:
            r = b32[op](self.rs(o[1]), self.rs(o[2])) & M32
            self.ws(o[0], r)
            self.scc = int(r != 0)
            return
        if op == "s_bcnt1_i32_b64":
            r = bin(self.rs(o[1], 64)).count("1")
            self.ws(o[0], r)
            self.scc = int(r != 0)
            return
        if op == "s_ff1_i32_b64":
            a = self.rs(o[1], 64)
            return self.ws(o[0], ((a & -a).bit_length() - 1) & M32 if a else M32)
        sgn = lambda a: a - (1 << 32) if a & 0x80000000 else a   # noqa: E731
        if op in ("s_add_u32", "s_add_i32"):
            a, b = self.rs(o[1]), self.rs(o[2])
            r = a + b
            self.scc = int(r > M32) if op == "s_add_u32" else int(not (-(1 << 31) <= sgn(a) + sgn(b) < (1 << 31)))
            return self.ws(o[0], r)
        if op in ("s_sub_u32", "s_sub_i32"):
            a, b = self.rs(o[1]), self.rs(o[2])
            self.scc = int(a < b) if op == "s_sub_u32" else int(not (-(1 << 31) <= sgn(a) - sgn(b) < (1 << 31)))
            return self.ws(o[0], a - b)
        if op == "s_addk_i32":
            return self.ws(o[0], self.rs(o[0]) + int(o[1], 0))
        if op == "s_mul_i32":
            return self.ws(o[0], sgn(self.rs(o[1])) * sgn(self.rs(o[2])))
        if op in ("s_min_i32", "s_max_i32", "s_min_u32", "s_max_u32"):
            a, b = self.rs(o[1]), self.rs(o[2])
            if op.endswith("i32"):
                ka, kb = sgn(a), sgn(b)
            else:
                ka, kb = a, b
            first = (ka <= kb) if "min" in op else (ka >= kb)
            self.scc = int(first)
            return self.ws(o[0], a if first else b)
        m = re.match(r"^s_cmp_(\w+)_([iu])(32|64)$", op)
        if m:
            c, ty, w = m.groups()
            a, b = self.rs(o[0], int(w)), self.rs(o[1], int(w))
            if ty == "i":
                a, b = sgn(a), sgn(b)
            self.scc = int(_cmp_ops[c](a, b))
            return
        if op == "s_cselect_b32":
            return self.ws(o[0], self.rs(o[1]) if self.scc else self.rs(o[2]))
        if op == "s_cselect_b64":
            return self.ws(o[0], self.rs(o[1], 64) if self.scc else self.rs(o[2], 64), 64)
        if op == "s_branch":
            self.pc = self.p.target(pc, o[0])
            return
        cb = {"s_cbranch_scc1": lambda: self.scc, "s_cbranch_scc0": lambda: not self.scc, "s_cbranch_execz": lambda: self.exec == 0,
              "s_cbranch_execnz": lambda: self.exec != 0, "s_cbranch_vccz": lambda: self.vcc == 0, "s_cbranch_vccnz": lambda: self.vcc != 0}
        if op in cb:
            if cb[op]():
                self.pc = self.p.target(pc, o[0])
            return
        if op == "s_endpgm":
            self.done = True
            return
        raise SimError("pc %d: SALU op %s not modelled" % (pc, op))


def run_workgroup(waves, quantum=300, max_steps=200_000_000):
    """round-robin the waves until all have run off the end of the program"""
    total = 0
    live = list(waves)
    while live:
        for w in list(live):
            for _ in range(quantum):
                if w.done:
                    break
                w.step()
            total += quantum
            if w.done:
                live.remove(w)
        if total > max_steps:
            raise SimError("step limit")


def merge_counts(waves):
    cnt, lanes = {}, {}
    for w in waves:
        for k, v in w.cnt.items():
            cnt[k] = cnt.get(k, 0) + v
        for k, v in w.lanes.items():
            lanes[k] = lanes.get(k, 0) + v
    return cnt, lanes
