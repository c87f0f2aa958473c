"""Deterministic corpus of corrupted / hand-assembled deflate streams for the differential tests of the Inflater
(device vs oracle): bit flips, truncations, bad stored lengths, illegal length/distance symbols, matches that reach
before the start of the stream, and dynamic headers with over-subscribed or incomplete code-length sets.

Reference behaviour being probed: C/Inflater.cs:283-386,429-552; C/InflaterHuffmanTree.cs:87-169 (no completeness check,
:116-121 commented out), :181-235; C/InflaterDynHeader.cs:42-120; CS/OutputWindow.cs:63-92.
"""
import numpy as np


class BitWriter:
    """LSB-first bit packing as RFC 1951 §3.1.1 (Huffman codes are written MSB-first, i.e. bit-reversed)."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def bits(self, v, k):
        self.acc |= (v & ((1 << k) - 1)) << self.n
        self.n += k
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, k):  # Huffman code c of k bits, most significant code bit first
        r = 0
        for i in range(k):
            r |= ((c >> (k - 1 - i)) & 1) << i
        self.bits(r, k)

    def done(self):
        if self.n:
            self.out.append(self.acc & 0xFF)
            self.acc, self.n = 0, 0
        return bytes(self.out)


def canonical(lengths):
    """RFC 1951 §3.2.2 code assignment (what a sane encoder would have meant); {symbol: (code, len)}."""
    maxl = max(lengths) if len(lengths) else 0
    cnt = [0] * (maxl + 2)
    for l in lengths:
        if l:
            cnt[l] += 1
    code, nxt = 0, [0] * (maxl + 2)
    for l in range(1, maxl + 1):
        code = (code + cnt[l - 1]) << 1 if l > 1 else 0
        nxt[l] = code
    out = {}
    for s, l in enumerate(lengths):
        if l:
            out[s] = (nxt[l] & ((1 << l) - 1), l)   # over-subscribed sets overflow l bits: keep the low bits
            nxt[l] += 1
    return out


STATIC_LL = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
STATIC_D = [5] * 32
_LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
_DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193,
          12289, 16385, 24577]
_DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


def put_tokens(w, tokens, ll, dd):
    """tokens: int literal | ('m', len, dist) | ('sym', litlen symbol) | ('dsym', len, dist symbol, extra) | 'eob'."""
    for t in tokens:
        if t == "eob":
            w.code(*ll[256])
        elif isinstance(t, int):
            w.code(*ll[t])
        elif t[0] == "sym":
            w.code(*ll[t[1]])
        elif t[0] == "m":
            _, ln, dist = t
            ls = max(i for i in range(29) if _LBASE[i] <= ln)
            if ln == 258:
                ls = 28
            w.code(*ll[257 + ls]); w.bits(ln - _LBASE[ls], _LEXT[ls])
            ds = max(i for i in range(30) if _DBASE[i] <= dist)
            w.code(*dd[ds]); w.bits(dist - _DBASE[ds], _DEXT[ds])
        elif t[0] == "dsym":
            _, ln, ds, extra = t
            ls = max(i for i in range(29) if _LBASE[i] <= ln)
            w.code(*ll[257 + ls]); w.bits(ln - _LBASE[ls], _LEXT[ls])
            w.code(*dd[ds]); w.bits(extra, _DEXT[ds] if ds < 30 else 0)


def static_block(tokens, last=True, w=None):
    w = w or BitWriter()
    w.bits(1 if last else 0, 1); w.bits(1, 2)
    put_tokens(w, tokens, canonical(STATIC_LL), canonical(STATIC_D))
    return w


_META_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
# a complete code-length code: 13 symbols of 4 bits + 6 of 5 bits (13/16 + 6/32 = 1)
_META_LENS = {s: 4 for s in range(13)}
_META_LENS.update({s: 5 for s in range(13, 19)})


def dynamic_block(ll_lens, d_lens, tokens, last=True, meta_lens=None, w=None, rle=False):
    """A dynamic block whose code-length sets are exactly `ll_lens` (>= 257 entries) / `d_lens`, however malformed.
    The body uses the canonical assignment of those lengths."""
    w = w or BitWriter()
    ml = dict(_META_LENS if meta_lens is None else meta_lens)
    w.bits(1 if last else 0, 1); w.bits(2, 2)
    w.bits(len(ll_lens) - 257, 5); w.bits(len(d_lens) - 1, 5)
    hclen = 19
    w.bits(hclen - 4, 4)
    for s in _META_ORDER[:hclen]:
        w.bits(ml.get(s, 0), 3)
    mc = canonical([ml.get(s, 0) for s in range(19)])
    seq = list(ll_lens) + list(d_lens)
    i = 0
    while i < len(seq):
        v = seq[i]
        run = 1
        while rle and i + run < len(seq) and seq[i + run] == v:
            run += 1
        if rle and v == 0 and run >= 11 and 18 in mc:
            r = min(run, 138); w.code(*mc[18]); w.bits(r - 11, 7); i += r
        elif rle and v == 0 and run >= 3 and 17 in mc:
            r = min(run, 10); w.code(*mc[17]); w.bits(r - 3, 3); i += r
        else:
            if v in mc:
                w.code(*mc[v])
            i += 1
    put_tokens(w, tokens, canonical(list(ll_lens)), canonical(list(d_lens)))
    return w


def crafted():
    """[(name, stream bytes)] of hand-assembled streams, valid and malformed."""
    out = []
    text = [ord(c) for c in "the quick brown fox jumps over the lazy dog. "]
    # --- valid static / dynamic sanity (the builder itself must be right)
    out.append(("static_valid", static_block(text + [("m", 20, 45), "eob"]).done()))
    # --- a match that reaches before the first output byte: zeros from the fresh window (CS/OutputWindow.cs:22,63-92)
    out.append(("dist_before_start_near", static_block([ord("a"), ("m", 10, 100), ord("b"), "eob"]).done()))
    out.append(("dist_before_start_far", static_block([ord("a")] * 5 + [("m", 258, 32768), ("m", 50, 20000), ord("z"), "eob"]).done()))
    out.append(("dist_before_start_first_token", static_block([("m", 3, 1), ord("q"), "eob"]).done()))
    lots = [int(x) for x in (np.arange(9000) * 7 % 251)]
    out.append(("dist_before_start_after_9k", static_block(lots + [("m", 100, 9001), ("m", 258, 30000), ord("e"), "eob"]).done()))
    # --- illegal symbols
    out.append(("illegal_len_286", static_block(text + [("sym", 286), "eob"]).done()))          # C/Inflater.cs:323-326
    out.append(("illegal_len_287", static_block(text + [("sym", 287), "eob"]).done()))
    out.append(("illegal_dist_30", static_block(text + [("dsym", 5, 30, 0), "eob"]).done()))    # :356-359
    out.append(("illegal_dist_31", static_block(text + [("dsym", 5, 31, 0), "eob"]).done()))
    # --- block header / stored block
    out.append(("block_type_3", bytes([0x07, 0x00])))
    out.append(("stored_bad_nlen", bytes([0x01, 0x05, 0x00, 0xFA, 0xFE]) + b"Hello"))
    out.append(("stored_truncated", bytes([0x01, 0x05, 0x00, 0xFA, 0xFF]) + b"Hel"))
    out.append(("stored_then_static", bytes([0x00, 0x03, 0x00, 0xFC, 0xFF]) + b"abc" + static_block([("m", 6, 3), "eob"]).done()))
    # --- dynamic headers
    ll = [0] * 266
    for c in set(text):
        ll[c] = 6
    for c in range(256, 266):       # end of block + the length symbols for match lengths 3..11
        ll[c] = 6
    used = sum(1 for x in ll if x)
    assert used <= 64
    ll_complete = list(ll)
    # pad to a complete code: remaining 6-bit slots as extra literals
    free = 64 - used
    k = 0
    while free:
        if ll_complete[k] == 0:
            ll_complete[k] = 6; free -= 1
        k += 1
    dd1 = [1, 1]
    out.append(("dyn_valid_complete", dynamic_block(ll_complete, dd1, text + [("m", 9, 2), "eob"]).done()))
    out.append(("dyn_valid_rle", dynamic_block(ll_complete, dd1, text + ["eob"], rle=True).done()))
    out.append(("dyn_incomplete_short", dynamic_block(ll, dd1, text + ["eob"]).done()))                  # incomplete, all codes <= 9 bits
    out.append(("dyn_single_dist_code", dynamic_block(ll_complete, [1], text + [("m", 4, 1), "eob"]).done()))  # zlib-style 1-code distance tree
    # incomplete set, then a bit pattern with no code -> "Encountered invalid codelength 0" (C/InflaterHuffmanTree.cs:191-193)
    w = dynamic_block(ll, dd1, text[:10])
    w.bits(0x3F, 6); w.bits(0x1FF, 9)  # all-ones: beyond the assigned 6-bit codes
    out.append(("dyn_incomplete_unassigned_pattern", w.done()))
    # over-subscribed literal/length set: BuildTree throws at header time (C/InflaterHuffmanTree.cs:133-166)
    over = list(ll_complete); over[255] = 1
    out.append(("dyn_oversubscribed_litlen", dynamic_block(over, dd1, text + ["eob"]).done()))
    out.append(("dyn_oversubscribed_dist", dynamic_block(ll_complete, [1, 1, 1], text + ["eob"]).done()))
    out.append(("dyn_oversubscribed_meta", dynamic_block(ll_complete, dd1, text + ["eob"], meta_lens={s: 3 for s in range(19)}).done()))
    noeob = list(ll_complete); noeob[256] = 0
    out.append(("dyn_no_eob_code", dynamic_block(noeob, dd1, text).done()))                               # C/InflaterDynHeader.cs:113
    out.append(("dyn_too_many_litlen", _raw_dyn_counts(30, 0)))                                            # HLIT 287 > 286 (:50-52)
    out.append(("dyn_too_many_dist", _raw_dyn_counts(0, 31)))
    # repeat-previous as the first symbol (:83) and a repeat that overruns the table (:106)
    w = BitWriter(); w.bits(1, 1); w.bits(2, 2); w.bits(0, 5); w.bits(0, 5); w.bits(15, 4)
    for s in _META_ORDER:
        w.bits(_META_LENS[s], 3)
    mc = canonical([_META_LENS[s] for s in range(19)])
    w.code(*mc[16]); w.bits(0, 2)
    out.append(("dyn_repeat_first", w.done() + b"\x00" * 8))
    w = BitWriter(); w.bits(1, 1); w.bits(2, 2); w.bits(0, 5); w.bits(0, 5); w.bits(15, 4)
    for s in _META_ORDER:
        w.bits(_META_LENS[s], 3)
    w.code(*mc[8])
    for _ in range(3):
        w.code(*mc[18]); w.bits(127, 7)
    out.append(("dyn_repeat_overrun", w.done() + b"\x00" * 8))
    return out


def _raw_dyn_counts(hlit, hdist):
    w = BitWriter(); w.bits(1, 1); w.bits(2, 2); w.bits(hlit, 5); w.bits(hdist, 5); w.bits(15, 4)
    for _ in range(19):
        w.bits(4, 3)
    return w.done() + b"\x00" * 16


def crafted_long_code_incomplete():
    """Incomplete sets that contain codes of 10+ bits: the reference's second-level tables then hold unassigned slots
    (returned as symbol 0 with 0 bits, C/InflaterHuffmanTree.cs:200-203) and codes in the last partial 9-bit prefix
    corrupt the primary table (:153-163).  Kept apart: see DESIGN.md for how the device treats them."""
    out = []
    ll = [0] * 257
    for i in range(40):
        ll[i + 40] = 12
    ll[256] = 3
    text = [45, 50, 60, 70]
    out.append(("dyn_incomplete_long_ok_symbols", dynamic_block(ll, [1, 1], text + ["eob"]).done()))
    w = dynamic_block(ll, [1, 1], text)
    w.bits(0, 3)  # canonical code 000 = the 3-bit EOB ... then garbage
    out.append(("dyn_incomplete_long_eob", w.done()))
    w = dynamic_block(ll, [1, 1], text)
    w.code(0b001000000111, 12)  # inside a second-level table, slot never assigned
    w.bits(0, 16)
    out.append(("dyn_incomplete_long_unassigned_slot", w.done()))
    return out


def random_code_set(rng, n, style):
    """n code lengths that are not over-subscribed: a random complete prefix code over some of the symbols ("complete"), then with
    codes removed ("drop") or made longer ("longer") — incomplete, and with codes of 10+ bits the class on which the reference's
    lookup table is not a canonical decoder (C/InflaterHuffmanTree.cs:153-163,200-203)."""
    leaves = [1, 1]
    while len(leaves) < n and rng.random() < 0.995:
        k = int(rng.integers(len(leaves)))
        if leaves[k] >= 15:
            if all(l >= 15 for l in leaves):
                break
            continue
        l = leaves.pop(k) + 1
        leaves += [l, l]
    lens = np.zeros(n, dtype=np.uint8)
    idx = rng.permutation(n)[:len(leaves)]
    lens[idx] = leaves[:len(idx)]
    if style == "complete":
        return lens
    used = np.flatnonzero(lens)
    drop = max(1, int(len(used) * (0.02 + 0.3 * rng.random())))
    if style == "drop":
        lens[rng.choice(used, size=min(drop, len(used) - 1), replace=False)] = 0
    elif style == "longer":
        for k in rng.choice(used, size=min(drop, len(used)), replace=False):
            lens[k] = min(15, int(lens[k]) + int(rng.integers(1, 4)))
    return lens


def quirk_set_streams(rng, count):
    """[(name, bytes)]: a dynamic block whose literal/length and/or distance set is a damaged random code, followed by RANDOM bits
    (they walk all over the table: unassigned second-level slots, entries a partial prefix left in the primary table, real
    codes) — whole, and cut short at random places (what GetSymbol does with fewer than 9 / fewer than `bitlen` bits left)."""
    out = []
    k = 0
    while len(out) < count:
        k += 1
        hl = 257 + int(rng.integers(0, 30))
        ll = random_code_set(rng, hl, ["drop", "longer", "longer", "complete"][k % 4])
        if ll[256] == 0:
            ll[256] = int(rng.integers(1, 16))
            if int(np.sum(1 << (16 - ll[ll > 0].astype(np.int64)))) > 65536:
                continue
        hd = 1 + int(rng.integers(0, 30))
        dd = random_code_set(rng, hd, ["longer", "drop", "complete", "drop"][(k // 4) % 4]) if hd > 1 else np.array([int(rng.integers(0, 3))], dtype=np.uint8)
        w = dynamic_block([int(x) for x in ll], [int(x) for x in dd], [], last=bool(k & 1))
        body = rng.integers(0, 256, size=int(rng.integers(4, 400)), dtype=np.uint8).tobytes()
        head = w.done()
        s = head + body
        out.append(("quirkset%d" % k, s))
        for t in range(2):
            cut = len(head) + int(rng.integers(0, min(len(body), 40)))
            out.append(("quirkset%d_trunc@%d" % (k, cut), s[:cut]))
    return out[:count]


def mutations(valid, rng, n_flip=6, n_trunc=3):
    """[(name, bytes)]: single-bit flips and truncations of each valid stream."""
    out = []
    for name, s in valid:
        b = np.frombuffer(s, np.uint8)
        for k in range(n_flip):
            pos = int(rng.integers(0, b.size * 8))
            m = b.copy(); m[pos >> 3] ^= 1 << (pos & 7)
            out.append(("%s_flip%d@%d" % (name, k, pos), m.tobytes()))
        for k in range(n_trunc):
            cut = int(rng.integers(0, b.size))
            out.append(("%s_trunc%d@%d" % (name, k, cut), s[:cut]))
    return out
