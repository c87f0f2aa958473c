// szl_inflate_reftree.h — the reference's Inflater lookup table, entry for entry (C/InflaterHuffmanTree.cs:87-169), and its
// GetSymbol (:181-235), for the code sets on which that table differs from every canonical decoder: INCOMPLETE sets that hold
// codes of 10 bits or more.  There the reference
//   * leaves second-level slots no code reaches at 0, which GetSymbol reads as "symbol 0, 0 bits" without complaint (:200-203),
//   * builds no second level for the last, partial 9-bit prefix (`end = code & 0x1ff80` rounds down, :126,:133): the long codes
//     of that prefix find 0 where their pointer should be and are written into the PRIMARY table at `revcode >> 9` (:153-163),
//     over whatever was there, and a later long code that reads such an entry as its pointer indexes out of range (the
//     constructor throws IndexOutOfRangeException).
// Garbage in, garbage out — but the same garbage: k_inflate stops in front of a block with such a set and k_inflate_exact
// (csrc/szl_kernels_inflate_exact.hip: the reference's Inflater step for step, its bit buffer included) decodes it through this table.
// Plain C++ without HIP so that the same text is compiled for the CPU and compared with oracle/szl_inflate_oracle.c entry by entry
// and symbol by symbol (tests/test_reftree.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SZL_RT_FN __host__ __device__ inline
#else
#define SZL_RT_FN static inline
#endif

namespace szl {

enum : int { RT_CAP_LITLEN = 1024, RT_CAP_DIST = 1184 };   // 512 + codes of 10+ bits + 120 covers every set that is not over-subscribed
enum : int { RT_NEED_INPUT = -1, RT_CODELEN_ZERO = -2, RT_INDEX = -3 };

SZL_RT_FN uint32_t rt_bitrev16(uint32_t v) {   // DeflaterHuffman.BitReverse (C/DeflaterHuffman.cs:924-930) for v < 65536
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0F0Fu) << 4) | ((v >> 4) & 0x0F0Fu);
    return ((v & 0x00FFu) << 8) | ((v >> 8) & 0x00FFu);
}

// True for a set BuildTree's table differs on: not over-subscribed, incomplete, with a code of 10+ bits (cnt[l] = codes of length l).
SZL_RT_FN bool rt_is_quirk_set(const uint32_t *cnt) {
    uint32_t kraft = 0, longc = 0;
    for (int l = 1; l <= 15; l++) { kraft += cnt[l] << (16 - l); if (l >= 10) longc += cnt[l]; }
    return kraft < 65536u && longc != 0;
}

// BuildTree (:87-169).  lens[0..n): code lengths; tree[0..cap): the table (int16 entries as the reference's short[]); blc / nxt: 16
// words of work space each (the caller's, so that the device keeps them out of scratch memory).  The set must not be
// over-subscribed (the caller has rejected those: the reference throws out of BitReverse).  Returns the table's size or RT_INDEX.
SZL_RT_FN int rt_build(const uint8_t *lens, int n, int16_t *tree, int cap, uint32_t *blc, uint32_t *nxt) {
    for (int b = 0; b < 16; b++) { blc[b] = 0; nxt[b] = 0; }
    for (int i = 0; i < n; i++) { const int bits = lens[i]; if (bits > 0) blc[bits]++; }
    uint32_t code = 0;
    int treeSize = 512;
    for (int bits = 1; bits <= 15; bits++) {
        nxt[bits] = code;
        code += blc[bits] << (16 - bits);
        if (bits >= 10) {
            const int start = (int)(nxt[bits] & 0x1ff80u), end = (int)(code & 0x1ff80u);
            treeSize += (end - start) >> (16 - bits);
        }
    }
    if (treeSize > cap) return RT_INDEX;                      // (cannot happen for a set that is not over-subscribed)
    for (int i = 0; i < treeSize; i++) tree[i] = 0;
    int treePtr = 512;
    for (int bits = 15; bits >= 10; bits--) {
        const int end = (int)(code & 0x1ff80u);
        code -= blc[bits] << (16 - bits);
        const int start = (int)(code & 0x1ff80u);
        for (int i = start; i < end; i += 1 << 7) {
            tree[rt_bitrev16((uint32_t)i)] = (int16_t)(((-treePtr) * 16) | bits);
            treePtr += 1 << (bits - 9);
        }
    }
    for (int i = 0; i < n; i++) {
        const int bits = lens[i];
        if (bits == 0) continue;
        code = nxt[bits];
        int revcode = (int)rt_bitrev16(code);
        if (bits <= 9) {
            do { tree[revcode] = (int16_t)((i << 4) | bits); revcode += 1 << bits; } while (revcode < 512);
        } else {
            int subTree = tree[revcode & 511];
            const int treeLen = 1 << (subTree & 15);
            subTree = -(subTree >> 4);
            do {
                const int idx = subTree | (revcode >> 9);
                if (idx < 0 || idx >= treeSize) return RT_INDEX;   // IndexOutOfRangeException out of the constructor
                tree[idx] = (int16_t)((i << 4) | bits);
                revcode += 1 << bits;
            } while (revcode < treeLen);
        }
        nxt[bits] = code + (1u << (16 - bits));
    }
    return treeSize;
}

// GetSymbol (:181-235) on the next stream bits (`bits`, LSB first, zero behind the input's end) of which `avail` exist
// (PeekBits(k) succeeds exactly when avail >= k; AvailableBits is what is left when it does not).  Returns symbol | dropped << 16,
// RT_NEED_INPUT or RT_CODELEN_ZERO.  NB the dropped count of an entry out of a partial prefix can exceed `avail` (:194 drops
// without looking): the caller's bit position then runs past the input like the reference's bitsInBuffer_ goes negative.
SZL_RT_FN int rt_get_symbol(const int16_t *tree, uint32_t bits, uint32_t avail) {
    if (avail >= 9) {
        int symbol = tree[bits & 511u];
        const int bitlen = symbol & 15;
        if (symbol >= 0) {
            if (bitlen == 0) return RT_CODELEN_ZERO;
            return (symbol >> 4) | (bitlen << 16);
        }
        const int subtree = -(symbol >> 4);
        if (avail >= (uint32_t)bitlen) {
            symbol = tree[subtree | (int)((bits & ((1u << bitlen) - 1u)) >> 9)];
            return (symbol >> 4) | ((symbol & 15) << 16);
        }
        symbol = tree[subtree | (int)((bits & ((1u << avail) - 1u)) >> 9)];
        if ((uint32_t)(symbol & 15) <= avail) return (symbol >> 4) | ((symbol & 15) << 16);
        return RT_NEED_INPUT;
    }
    const int symbol = tree[bits & ((1u << avail) - 1u)];
    if (symbol >= 0 && (uint32_t)(symbol & 15) <= avail) return (symbol >> 4) | ((symbol & 15) << 16);
    return RT_NEED_INPUT;
}

// ---- the reference's StreamManipulator (CS/StreamManipulator.cs) as k_inflate_exact runs it: a 32-bit buffer filled 16 bits at a
// time, DropBits that does not look (so bitsInBuffer_ can go negative when GetSymbol drops a garbage entry's bit count after peeking
// 9 bits), PeekBits that then shifts what it loads by a negative count (C# masks shift counts to 5 bits).  Plain C++ like the table
// code above: tests/test_reftree.py runs random peek / drop scripts against the oracle's StreamManipulator.
struct ExSM {            // CS/StreamManipulator.cs
    const uint8_t *in; uint64_t we;     // window_, windowEnd_
    uint64_t ws;                        // windowStart_
    uint32_t buffer; int32_t bits;      // buffer_, bitsInBuffer_
    uint32_t lazy;                      // see the header: the 16 bits a peek before the block may already have loaded
    int32_t dirty;                      // > 0: bits still to be dropped before the buffer holds nothing but stream bits again
};
SZL_RT_FN uint32_t ex_load16(ExSM &s) {
    const uint32_t lo = s.ws < s.we ? s.in[s.ws] : 0u, hi = s.ws + 1 < s.we ? s.in[s.ws + 1] : 0u;
    s.ws += 2;
    return lo | (hi << 8);
}
SZL_RT_FN int ex_peek(ExSM &s, int n) {   // :31-48
    if (s.bits < n) {
        if (s.ws >= s.we) return -1;
        const uint32_t v = ex_load16(s);
        s.buffer |= (uint32_t)((int32_t)v << (s.bits & 31));
        s.bits += 16;
        s.lazy = 0;
    }
    return (int)(s.buffer & ((1u << (n & 31)) - 1u));
}
SZL_RT_FN void ex_drop(ExSM &s, int k) {  // :86-90
    if (s.lazy && k > s.bits && s.ws < s.we) { const uint32_t v = ex_load16(s); s.buffer |= v << (s.bits & 31); s.bits += 16; s.lazy = 0; }
    if (k > s.bits) s.dirty = 32 + 16; else if (s.dirty > 0) s.dirty -= k;
    s.buffer >>= (k & 31);
    s.bits -= k;
}
SZL_RT_FN int64_t ex_available_bytes(const ExSM &s) { return (int64_t)s.we - (int64_t)s.ws + (int64_t)(s.bits >> 3); }  // :131

// GetSymbol (C/InflaterHuffmanTree.cs:181-235) on the live bit buffer.  >= 0 symbol, -1 need input, -2 "invalid codelength 0"
SZL_RT_FN int ex_get_symbol(const int16_t *tree, ExSM &s) {
    int lookahead, symbol;
    if ((lookahead = ex_peek(s, 9)) >= 0) {
        symbol = tree[lookahead];
        const int bitlen = symbol & 15;
        if (symbol >= 0) {
            if (bitlen == 0) return -2;
            ex_drop(s, bitlen);
            return symbol >> 4;
        }
        const int subtree = -(symbol >> 4);
        if ((lookahead = ex_peek(s, bitlen)) >= 0) {
            symbol = tree[subtree | (lookahead >> 9)];
            ex_drop(s, symbol & 15);
            return symbol >> 4;
        }
        const int bits = s.bits;
        lookahead = ex_peek(s, bits);
        symbol = tree[subtree | (lookahead >> 9)];
        if ((symbol & 15) <= bits) { ex_drop(s, symbol & 15); return symbol >> 4; }
        return -1;
    }
    const int bits = s.bits;
    lookahead = ex_peek(s, bits);
    symbol = tree[lookahead & 511];
    if (symbol >= 0 && (symbol & 15) <= bits) { ex_drop(s, symbol & 15); return symbol >> 4; }
    return -1;
}


} // namespace szl
