"""gfxsim.asm — reads the device assembly hipcc emits for gfx950 (`hipcc --cuda-device-only -S`) into a Module:
instructions (decoded once into operand tuples), labels, the kernels' descriptors (.amdhsa_*), their kernarg layout
(.amdgpu_metadata) and the data symbols of the translation unit (.rodata / .data / .bss tables such as the CRC table).

Test infrastructure only (tools/ and tests/); never part of the product.
"""
import re
import struct

import yaml


class AsmError(Exception):
    pass


_SPECIAL = {
    "vcc": ("s", 106, 2), "vcc_lo": ("s", 106, 1), "vcc_hi": ("s", 107, 1),
    "exec": ("s", 126, 2), "exec_lo": ("s", 126, 1), "exec_hi": ("s", 127, 1),
    "m0": ("s", 124, 1), "flat_scratch": ("s", 102, 2), "flat_scratch_lo": ("s", 102, 1), "flat_scratch_hi": ("s", 103, 1),
    "scc": ("scc",), "src_scc": ("scc",), "off": ("off",), "null": ("null",),
    "src_shared_base": ("aperture", "shared"), "src_private_base": ("aperture", "private"),
    "src_shared_limit": ("aperture", "shared_limit"), "src_private_limit": ("aperture", "private_limit"),
}

_FLOAT_RE = re.compile(r"^-?\d+\.\d*(e[-+]?\d+)?$")


def split_top(s, sep):
    """split s at sep where bracket depth is 0"""
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        if depth == 0 and (ch == sep or (sep == " " and ch == "\t")):
            out.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    out.append("".join(cur))
    return [x for x in (t.strip() for t in out) if x != ""]


def parse_int(t):
    t = t.strip()
    neg = t.startswith("-")
    if neg:
        t = t[1:]
    v = int(t, 0)
    return -v if neg else v


def parse_operand(t):
    """-> tuple: ('v', first, count) ('s', first, count) ('a', first, count) ('k', int) ('f', float) ('sym', name, kind, addend)
    ('label', name) ('off',) ('scc',) ('aperture', which); wrapped as ('mod', inner, {'sext','neg','abs'}) when modified"""
    t = t.strip()
    mods = set()
    while True:
        if t.startswith("sext(") and t.endswith(")"):
            mods.add("sext")
            t = t[5:-1]
        elif t.startswith("-|") and t.endswith("|"):
            mods.update(("neg", "abs"))
            t = t[2:-1]
        elif t.startswith("|") and t.endswith("|"):
            mods.add("abs")
            t = t[1:-1]
        elif t.startswith("-") and len(t) > 1 and t[1] in "vs|a":
            mods.add("neg")
            t = t[1:]
        elif t.startswith("abs(") and t.endswith(")"):
            mods.add("abs")
            t = t[4:-1]
        elif t.startswith("neg(") and t.endswith(")"):
            mods.add("neg")
            t = t[4:-1]
        else:
            break
    o = _parse_plain(t)
    if mods:
        return ("mod", o, frozenset(mods))
    return o


def _parse_plain(t):
    if t in _SPECIAL:
        return _SPECIAL[t]
    m = re.match(r"^([vsa])(\d+)$", t)
    if m:
        return (m.group(1), int(m.group(2)), 1)
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", t)
    if m:
        return (m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1)
    m = re.match(r"^([vsa])\[(\d+)\]$", t)
    if m:
        return (m.group(1), int(m.group(2)), 1)
    m = re.match(r"^ttmp(\d+)$", t)
    if m:
        return ("s", 108 + int(m.group(1)), 1)
    if re.match(r"^-?(0x[0-9a-fA-F]+|\d+)$", t):
        return ("k", parse_int(t))
    if _FLOAT_RE.match(t):
        return ("f", float(t))
    m = re.match(r"^([A-Za-z_.$][\w.$]*)@(rel32@lo|rel32@hi|abs32@lo|abs32@hi)([+-]\d+)?$", t)
    if m:
        return ("sym", m.group(1), m.group(2), int(m.group(3) or 0))
    if re.match(r"^[A-Za-z_.$][\w.$]*([+-]\d+)?$", t) or re.match(r"^\d+[fb]$", t):
        return ("label", t)
    raise AsmError("operand not understood: %r" % t)


class Ins:
    __slots__ = ("op", "base", "enc", "ops", "mods", "fn", "line", "cls", "aux")

    def __repr__(self):
        return "%s %s %s" % (self.op, self.ops, self.mods or "")


_ENC_SUFFIX = ("_e32", "_e64", "_sdwa", "_dpp")


def parse_ins(text, lineno):
    parts = text.split(None, 1)
    I = Ins()
    I.op = parts[0]
    I.base, I.enc = I.op, ""
    for sfx in _ENC_SUFFIX:
        if I.op.endswith(sfx):
            I.base, I.enc = I.op[: -len(sfx)], sfx[1:]
            break
    I.ops, I.mods, I.fn, I.line, I.cls, I.aux = [], {}, None, lineno, None, None
    if len(parts) == 1:
        return I
    rest = parts[1].strip()
    if I.base in ("s_waitcnt", "s_nop", "s_sleep", "s_setprio", "s_waitcnt_vscnt", "s_waitcnt_depctr", "s_inst_prefetch", "s_clause", "s_setreg_imm32_b32",
                  "s_setreg_b32", "s_getreg_b32", "s_sethalt", "s_trap", "s_icache_inv", "s_dcache_wb", "s_dcache_inv", "s_ttracedata", "s_sendmsg", "s_incperflevel",
                  "s_decperflevel"):
        I.aux = rest
        return I
    pieces = split_top(rest, ",")
    for i, piece in enumerate(pieces):
        toks = split_top(piece, " ")
        first = True
        for tk in toks:
            if first and not _is_modifier(tk):
                I.ops.append(parse_operand(tk))
                first = False
            else:
                first = False
                if ":" in tk:
                    k, v = tk.split(":", 1)
                    if v.startswith("["):
                        I.mods[k] = [parse_int(x) for x in v[1:-1].split(",")]
                    elif re.match(r"^-?(0x[0-9a-fA-F]+|\d+)$", v):
                        I.mods[k] = parse_int(v)
                    else:
                        I.mods[k] = v
                else:
                    I.mods[tk] = 1
    return I


_MOD_WORDS = {"glc", "slc", "dlc", "sc0", "sc1", "nt", "gds", "clamp", "lds", "offen", "idxen", "tfe", "lwe", "d16", "unorm", "da", "r128", "a16", "nv", "high", "neg_lo", "neg_hi"}


def _is_modifier(tk):
    if tk in _MOD_WORDS:
        return True
    m = re.match(r"^([a-z_0-9]+):", tk)
    return bool(m)


class Kernel:
    def __init__(self, name):
        self.name = name
        self.desc = {}
        self.args = []          # metadata .args
        self.entry = None       # pc
        self.meta = {}


class Module:
    """One translation unit's device assembly."""

    def __init__(self, text, name="<asm>"):
        self.name = name
        self.ins = []
        self.labels = {}        # name -> pc          (text labels)
        self.numlabels = {}     # "1" -> [pcs]        (local numeric labels of inline assembly)
        self.data = {}          # section -> bytearray
        self.datasym = {}       # symbol -> (section, offset)
        self.datareloc = []     # (section, offset, symbol, addend, size)
        self.absolute = {}      # .set name, value
        self.kernels = {}
        self.lds_syms = {}      # name -> (size, align)   (.amdgpu_lds)
        self._parse(text)

    def _parse(self, text):
        section = ".text"
        in_meta = False
        meta_lines = []
        cur_kd = None
        for lineno, raw in enumerate(text.split("\n"), 1):
            if in_meta:
                if raw.strip() == ".end_amdgpu_metadata":
                    in_meta = False
                else:
                    meta_lines.append(raw)
                continue
            line = raw
            st = line.strip()
            if not st:
                continue
            if st.startswith(".ascii") or st.startswith(".asciz") or st.startswith(".string"):
                self._data_string(section, st)
                continue
            if ";" in line:
                line = line.split(";", 1)[0]
                st = line.strip()
                if not st:
                    continue
            if st.startswith("//"):
                continue
            # several statements may share a line in inline assembly ("a\n\tb" is already split by the compiler)
            m = re.match(r"^([A-Za-z_.$][\w.$]*|\d+):\s*(.*)$", st)
            if m and not st.startswith(".amdhsa") and "::" not in st.split()[0]:
                lab, st = m.group(1), m.group(2).strip()
                self._label(section, lab)
                if not st:
                    continue
            if cur_kd is not None:
                if st == ".end_amdhsa_kernel":
                    cur_kd = None
                else:
                    p = st.split()
                    if len(p) == 2:
                        try:
                            cur_kd.desc[p[0][len(".amdhsa_"):]] = parse_int(p[1])
                        except ValueError:
                            cur_kd.desc[p[0][len(".amdhsa_"):]] = p[1]
                continue
            if st.startswith("."):
                d = st.split(None, 1)
                name, arg = d[0], (d[1].strip() if len(d) > 1 else "")
                if name == ".amdgpu_metadata":
                    in_meta = True
                elif name == ".amdhsa_kernel":
                    cur_kd = self.kernels.setdefault(arg, Kernel(arg))
                elif name == ".text":
                    section = ".text"
                elif name in (".data", ".bss", ".rodata"):
                    section = name
                elif name == ".section":
                    section = arg.split(",")[0].strip().strip('"')
                    if section.startswith(".text"):
                        section = ".text"
                elif name == ".amdgpu_lds":
                    a = [x.strip() for x in arg.split(",")]
                    self.lds_syms[a[0]] = (parse_int(a[1]), parse_int(a[2]) if len(a) > 2 else 4)
                elif name == ".set" or name == ".equ":
                    a = [x.strip() for x in arg.split(",", 1)]
                    try:
                        self.absolute[a[0]] = parse_int(a[1])
                    except (ValueError, IndexError):
                        pass
                elif section != ".text":
                    self._data_directive(section, name, arg)
                continue
            if section != ".text":
                raise AsmError("%s:%d: instruction outside .text: %s" % (self.name, lineno, st))
            try:
                self.ins.append(parse_ins(st, lineno))
            except (AsmError, ValueError) as e:
                raise AsmError("%s:%d: %s   [%s]" % (self.name, lineno, e, st))
        if meta_lines:
            md = yaml.safe_load("\n".join(meta_lines))
            for k in md.get("amdhsa.kernels", []):
                kn = self.kernels.setdefault(k[".name"], Kernel(k[".name"]))
                kn.args = k.get(".args", [])
                kn.meta = k
        for kn in self.kernels.values():
            kn.entry = self.labels.get(kn.name)

    def _buf(self, section):
        return self.data.setdefault(section, bytearray())

    def _label(self, section, lab):
        if section == ".text":
            if lab.isdigit():
                self.numlabels.setdefault(lab, []).append(len(self.ins))
            else:
                self.labels[lab] = len(self.ins)
        else:
            self.datasym[lab] = (section, len(self._buf(section)))

    def _data_string(self, section, st):
        kind, rest = st.split(None, 1)
        m = re.match(r'^"(.*)"\s*$', rest.strip())
        if not m:
            raise AsmError("string directive not understood: " + st)
        b = bytes(m.group(1), "latin-1").decode("unicode_escape").encode("latin-1")
        buf = self._buf(section)
        buf += b
        if kind in (".asciz", ".string"):
            buf += b"\0"

    def _data_directive(self, section, name, arg):
        buf = self._buf(section)
        sizes = {".byte": 1, ".short": 2, ".2byte": 2, ".hword": 2, ".long": 4, ".4byte": 4, ".int": 4, ".quad": 8, ".8byte": 8}
        if name in sizes:
            n = sizes[name]
            for tk in arg.split(","):
                tk = tk.strip()
                try:
                    v = parse_int(tk)
                    buf += (v & ((1 << (8 * n)) - 1)).to_bytes(n, "little")
                except ValueError:
                    m = re.match(r"^([A-Za-z_.$][\w.$]*)([+-]\d+)?$", tk)
                    if not m:
                        raise AsmError("data expression not understood: " + tk)
                    self.datareloc.append((section, len(buf), m.group(1), int(m.group(2) or 0), n))
                    buf += b"\0" * n
        elif name in (".zero", ".space", ".skip"):
            a = [x.strip() for x in arg.split(",")]
            buf += bytes([parse_int(a[1]) & 255 if len(a) > 1 else 0]) * parse_int(a[0])
        elif name == ".fill":
            a = [parse_int(x) for x in arg.split(",")]
            rep, size, val = a[0], (a[1] if len(a) > 1 else 1), (a[2] if len(a) > 2 else 0)
            buf += (val & ((1 << (8 * size)) - 1)).to_bytes(size, "little") * rep
        elif name in (".p2align", ".align", ".balign"):
            a = parse_int(arg.split(",")[0])
            al = (1 << a) if name == ".p2align" else a
            while len(buf) % al:
                buf.append(0)
        # everything else (.type, .size, .globl, .protected, .weak, .hidden, .ident, .addrsig ...) carries nothing the model needs

    # ---- label resolution -------------------------------------------------------------------------------------
    def target(self, pc, lab):
        m = re.match(r"^(\d+)([fb])$", lab)
        if m:
            cands = self.numlabels.get(m.group(1), [])
            if m.group(2) == "f":
                c = [x for x in cands if x > pc]
                if not c:
                    raise AsmError("no forward label %s from pc %d" % (lab, pc))
                return min(c)
            c = [x for x in cands if x <= pc]
            if not c:
                raise AsmError("no backward label %s from pc %d" % (lab, pc))
            return max(c)
        if lab in self.labels:
            return self.labels[lab]
        raise AsmError("unknown label " + lab)


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]
