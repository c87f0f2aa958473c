/*
 * szl_inflate_oracle.c — CPU restatement of SharpZipLib's Inflater (decompress side) and of
 * its CRC32 / Adler32.  TEST INFRASTRUCTURE ONLY (see szl_oracle.h).
 *
 * Follows (paths under /root/reference/src/ICSharpCode.SharpZipLib/):
 *   Zip/Compression/Inflater.cs · InflaterHuffmanTree.cs · InflaterDynHeader.cs ·
 *   Zip/Compression/Streams/StreamManipulator.cs · OutputWindow.cs ·
 *   Checksum/Crc32.cs · CrcUtilities.cs · Adler32.cs
 * Pinned by the reference's own known answers: T/Checksum/ChecksumTests.cs (CRC/Adler KATs),
 * T/Zip/ZipCorruptionHandling.cs:12-16 (must fail with "invalid codelength 0"), :52-54.
 */
#include "szl_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ================================================================= Checksum/ */
static uint32_t crc_table[256];
static int crc_ready = 0;
static void crc_init(void) { /* CrcUtilities.cs:25-52 (slice 0 of the table; poly 0xEDB88320 Crc32.cs:50) */
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t r = i;
        for (int k = 0; k < 8; k++) r = (r & 1) ? 0xEDB88320u ^ (r >> 1) : r >> 1;
        crc_table[i] = r;
    }
    crc_ready = 1;
}
uint32_t szo_crc32(uint32_t value, const uint8_t *p, size_t n) { /* Crc32.cs:138-171; Value = ~checkValue :82 */
    if (!crc_ready) crc_init();
    uint32_t c = ~value;
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return ~c;
}
uint32_t szo_adler32(uint32_t value, const uint8_t *p, size_t count) { /* Adler32.cs:134-161 */
    uint32_t s1 = value & 0xFFFF, s2 = value >> 16;
    size_t off = 0;
    while (count > 0) {
        size_t n = 3800;
        if (n > count) n = count;
        count -= n;
        while (n-- > 0) { s1 = s1 + p[off++]; s2 = s2 + s1; }
        s1 %= 65521u; s2 %= 65521u;
    }
    return (s2 << 16) | s1;
}

/* ================================================================= StreamManipulator.cs */
typedef struct {
    const uint8_t *window_;
    int windowStart_, windowEnd_;
    uint32_t buffer_;
    int bitsInBuffer_;
} SM;
static int sm_peek(SM *s, int bitCount) { /* :31 */
    if (s->bitsInBuffer_ < bitCount) {
        if (s->windowStart_ == s->windowEnd_) return -1;
        uint32_t lo = s->window_[s->windowStart_++] & 0xff;
        uint32_t hi = s->window_[s->windowStart_++] & 0xff;
        s->buffer_ |= (lo | (hi << 8)) << s->bitsInBuffer_;
        s->bitsInBuffer_ += 16;
    }
    return (int)(s->buffer_ & ((1u << bitCount) - 1));
}
static void sm_drop(SM *s, int bitCount) { s->buffer_ >>= bitCount; s->bitsInBuffer_ -= bitCount; } /* :86 */
static int sm_available_bytes(const SM *s) { return s->windowEnd_ - s->windowStart_ + (s->bitsInBuffer_ >> 3); } /* :131 */
static void sm_skip_to_byte(SM *s) { s->buffer_ >>= (s->bitsInBuffer_ & 7); s->bitsInBuffer_ &= ~7; } /* :142 */
static int sm_needs_input(const SM *s) { return s->windowStart_ == s->windowEnd_; } /* :152 */
static int sm_copy_bytes(SM *s, uint8_t *output, int offset, int length) { /* :183 */
    int count = 0;
    while (s->bitsInBuffer_ > 0 && length > 0) {
        output[offset++] = (uint8_t)s->buffer_;
        s->buffer_ >>= 8;
        s->bitsInBuffer_ -= 8;
        length--;
        count++;
    }
    if (length == 0) return count;
    int avail = s->windowEnd_ - s->windowStart_;
    if (length > avail) length = avail;
    memcpy(output + offset, s->window_ + s->windowStart_, (size_t)length);
    s->windowStart_ += length;
    if (((s->windowStart_ - s->windowEnd_) & 1) != 0) {
        s->buffer_ = (uint32_t)(s->window_[s->windowStart_++] & 0xff);
        s->bitsInBuffer_ = 8;
    }
    return count + length;
}
static void sm_reset(SM *s) { s->buffer_ = 0; s->windowStart_ = s->windowEnd_ = s->bitsInBuffer_ = 0; } /* :233 */
static int sm_set_input(SM *s, const uint8_t *buffer, int offset, int count) { /* :244 */
    if (count < 0) return SZO_ERR_ARG;
    if (s->windowStart_ < s->windowEnd_) return SZO_ERR_STATE;
    int end = offset + count;
    if ((count & 1) != 0) {
        s->buffer_ |= (uint32_t)((buffer[offset++] & 0xff) << s->bitsInBuffer_);
        s->bitsInBuffer_ += 8;
    }
    s->window_ = buffer;
    s->windowStart_ = offset;
    s->windowEnd_ = end;
    return 0;
}

/* ================================================================= OutputWindow.cs */
enum { OW_SIZE = 1 << 15, OW_MASK = OW_SIZE - 1 };
typedef struct {
    uint8_t window[OW_SIZE];
    int windowEnd, windowFilled;
} OW;
static int ow_write(OW *w, int value) { /* :35 */
    if (w->windowFilled++ == OW_SIZE) return SZO_ERR_WINDOW_FULL;
    w->window[w->windowEnd++] = (uint8_t)value;
    w->windowEnd &= OW_MASK;
    return 0;
}
static int ow_repeat(OW *w, int length, int distance) { /* :63 (+SlowRepeat :45) */
    if ((w->windowFilled += length) > OW_SIZE) return SZO_ERR_WINDOW_FULL;
    int repStart = (w->windowEnd - distance) & OW_MASK;
    int border = OW_SIZE - length;
    if (repStart <= border && w->windowEnd < border) {
        if (length <= distance) {
            memmove(w->window + w->windowEnd, w->window + repStart, (size_t)length);
            w->windowEnd += length;
        } else {
            while (length-- > 0) w->window[w->windowEnd++] = w->window[repStart++];
        }
    } else {
        while (length-- > 0) {
            w->window[w->windowEnd++] = w->window[repStart++];
            w->windowEnd &= OW_MASK;
            repStart &= OW_MASK;
        }
    }
    return 0;
}
static int ow_copy_stored(OW *w, SM *input, int length) { /* :100 */
    int a = OW_SIZE - w->windowFilled, b = sm_available_bytes(input);
    if (length > a) length = a;
    if (length > b) length = b;
    int copied;
    int tailLen = OW_SIZE - w->windowEnd;
    if (length > tailLen) {
        copied = sm_copy_bytes(input, w->window, w->windowEnd, tailLen);
        if (copied == tailLen) copied += sm_copy_bytes(input, w->window, 0, length - tailLen);
    } else {
        copied = sm_copy_bytes(input, w->window, w->windowEnd, length);
    }
    w->windowEnd = (w->windowEnd + copied) & OW_MASK;
    w->windowFilled += copied;
    return copied;
}
static int ow_copy_dict(OW *w, const uint8_t *dict, int offset, int length) { /* :130 */
    if (w->windowFilled > 0) return SZO_ERR_STATE;
    if (length > OW_SIZE) { offset += length - OW_SIZE; length = OW_SIZE; }
    memcpy(w->window, dict + offset, (size_t)length);
    w->windowEnd = length & OW_MASK;
    return 0;
}
static int ow_copy_output(OW *w, uint8_t *output, int offset, int len) { /* :182 */
    int copyEnd = w->windowEnd;
    if (len > w->windowFilled) len = w->windowFilled;
    else copyEnd = (w->windowEnd - w->windowFilled + len) & OW_MASK;
    int copied = len;
    int tailLen = len - copyEnd;
    if (tailLen > 0) {
        memcpy(output + offset, w->window + OW_SIZE - tailLen, (size_t)tailLen);
        offset += tailLen;
        len = copyEnd;
    }
    memcpy(output + offset, w->window + copyEnd - len, (size_t)len);
    w->windowFilled -= copied;
    return copied;
}

/* ================================================================= InflaterHuffmanTree.cs */
static const uint8_t bit4Reverse[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
/* DeflaterHuffman.BitReverse (C/DeflaterHuffman.cs:924-930) returns short; used as index after int promotion.
 * `bit4Reverse[toReverse >> 12]` indexes a 16-entry array: for toReverse >= 65536 (over-subscribed code lengths) the
 * reference throws IndexOutOfRangeException inside BuildTree — reported here as BR_RANGE (never UB). */
#define BR_RANGE 0x40000000
static int bit_reverse16(int v) {
    if (v < 0 || (v >> 12) > 15) return BR_RANGE;
    return (int)(int16_t)(bit4Reverse[v & 0xF] << 12 | bit4Reverse[(v >> 4) & 0xF] << 8 |
                          bit4Reverse[(v >> 8) & 0xF] << 4 | bit4Reverse[v >> 12]);
}
typedef struct {
    int16_t *tree;
    int treeSize;
} IHT;
static int g_szo_quirk_sets = 0; /* see iht_build */
int szo_quirk_sets_seen(int reset) { int v = g_szo_quirk_sets; if (reset) g_szo_quirk_sets = 0; return v; }

static int iht_build(IHT *t, const uint8_t *codeLengths, int n) { /* :87 */
    int blCount[16] = {0}, nextCode[16] = {0};
    for (int i = 0; i < n; i++) {
        int bits = codeLengths[i];
        if (bits > 0) blCount[bits]++;
    }
    int code = 0;
    int treeSize = 512;
    for (int bits = 1; bits <= 15; bits++) {
        nextCode[bits] = code;
        code += blCount[bits] << (16 - bits);
        if (bits >= 10) {
            int start = nextCode[bits] & 0x1ff80;
            int end = code & 0x1ff80;
            treeSize += (end - start) >> (16 - bits);
        }
    }
    if (treeSize < 512) treeSize = 512;
    /* Test aid: an INCOMPLETE set that holds codes of 10+ bits makes the reference's table differ from any canonical decoder —
     * unassigned second-level slots decode as "symbol 0, 0 bits" (:200-203) and codes in the last, partial 9-bit prefix are
     * written into the primary table instead (:153-163).  The differential tests record that such a set was seen. */
    if (code < 65536) { int longc = 0; for (int b = 10; b <= 15; b++) longc += blCount[b]; if (longc) g_szo_quirk_sets++; }
    /* The reference is lenient about over-subscribed sets (:116-121) and would index out of
     * range (exception) in the fill loops below; we allocate generously and bounds-check. */
    int cap = treeSize;
    t->tree = (int16_t *)calloc((size_t)cap, sizeof(int16_t));
    t->treeSize = cap;
    int treePtr = 512;
    for (int bits = 15; bits >= 10; bits--) {
        int end = code & 0x1ff80;
        code -= blCount[bits] << (16 - bits);
        int start = code & 0x1ff80;
        for (int i = start; i < end; i += 1 << 7) {
            int idx = bit_reverse16(i);
            if (idx == BR_RANGE) return SZO_ERR_INDEX_RANGE;
            if (idx < 0 || idx >= cap) return SZO_ERR_INDEX_RANGE; /* tree[] index outside new short[treeSize] */
            t->tree[idx] = (int16_t)(((-treePtr) * 16) | bits);
            treePtr += 1 << (bits - 9);
        }
    }
    for (int i = 0; i < n; i++) {
        int bits = codeLengths[i];
        if (bits == 0) continue;
        code = nextCode[bits];
        int revcode = bit_reverse16(code);
        if (revcode == BR_RANGE) return SZO_ERR_INDEX_RANGE;
        if (bits <= 9) {
            do {
                if (revcode < 0 || revcode >= cap) return SZO_ERR_INDEX_RANGE;
                t->tree[revcode] = (int16_t)((i << 4) | bits);
                revcode += 1 << bits;
            } while (revcode < 512);
        } else {
            int subTree = t->tree[revcode & 511];
            int treeLen = 1 << (subTree & 15);
            subTree = -(subTree >> 4);
            do {
                int idx = subTree | (revcode >> 9);
                if (idx < 0 || idx >= cap) return SZO_ERR_INDEX_RANGE;
                t->tree[idx] = (int16_t)((i << 4) | bits);
                revcode += 1 << bits;
            } while (revcode < treeLen);
        }
        nextCode[bits] = code + (1 << (16 - bits));
    }
    return 0;
}
static void iht_free(IHT *t) { free(t->tree); t->tree = NULL; }

/* returns symbol >=0, -1 = need input, <= -100 error */
static int iht_get_symbol(const IHT *t, SM *input) { /* :181 */
    int lookahead, symbol;
    if ((lookahead = sm_peek(input, 9)) >= 0) {
        symbol = t->tree[lookahead];
        int bitlen = symbol & 15;
        if (symbol >= 0) {
            if (bitlen == 0) return -100 + SZO_ERR_CODELEN_ZERO;
            sm_drop(input, bitlen);
            return symbol >> 4;
        }
        int subtree = -(symbol >> 4);
        if ((lookahead = sm_peek(input, bitlen)) >= 0) {
            symbol = t->tree[subtree | (lookahead >> 9)];
            sm_drop(input, symbol & 15);
            return symbol >> 4;
        } else {
            int bits = input->bitsInBuffer_;
            lookahead = sm_peek(input, bits);
            symbol = t->tree[subtree | (lookahead >> 9)];
            if ((symbol & 15) <= bits) {
                sm_drop(input, symbol & 15);
                return symbol >> 4;
            } else return -1;
        }
    } else {
        int bits = input->bitsInBuffer_;
        lookahead = sm_peek(input, bits);
        symbol = t->tree[lookahead];
        if (symbol >= 0 && (symbol & 15) <= bits) {
            sm_drop(input, symbol & 15);
            return symbol >> 4;
        } else return -1;
    }
}

/* Test hooks (tests/test_reftree.py): the lookup table BuildTree leaves behind and one GetSymbol on a bit pattern of which
 * `avail` bits exist — what the device's exact-table mode (csrc/szl_inflate_reftree.h) is checked against. */
int szo_iht_table(const uint8_t *codeLengths, int n, int16_t *out, int cap) {
    IHT t; t.tree = NULL; t.treeSize = 0;
    int rc = iht_build(&t, codeLengths, n);
    if (rc < 0) { free(t.tree); return rc; }
    int sz = t.treeSize;
    if (sz <= cap) memcpy(out, t.tree, (size_t)sz * sizeof(int16_t));
    iht_free(&t);
    return sz;
}
int szo_iht_symbol(const int16_t *tree, int treeSize, uint32_t bits, int avail, int *dropped) {
    uint8_t buf[4];
    for (int i = 0; i < 4; i++) buf[i] = (uint8_t)(bits >> (8 * i));
    IHT t; t.tree = (int16_t *)tree; t.treeSize = treeSize;
    SM in; sm_reset(&in);
    /* `avail` bits: whole bytes through the window, the rest pre-loaded the way an odd SetInput leaves them */
    int nbytes = avail >> 3, rest = avail & 7;
    (void)rest;
    in.window_ = buf; in.windowStart_ = 0; in.windowEnd_ = 0; in.buffer_ = avail >= 32 ? bits : (bits & ((1u << avail) - 1)); in.bitsInBuffer_ = avail;
    (void)nbytes;
    int before = in.bitsInBuffer_;
    int sym = iht_get_symbol(&t, &in);
    *dropped = before - in.bitsInBuffer_;
    return sym;
}

/* Test hook (tests/test_reftree.py): a script of StreamManipulator / GetSymbol operations on one SetInput(buf, 0, n) — the device's
 * emulation of the bit buffer (csrc/szl_inflate_reftree.h ExSM) runs the same script.  ops[2i] = 0 PeekBits(arg), 1 DropBits(arg),
 * 2 SkipToByteBoundary, 3 AvailableBits, 4 AvailableBytes, 5 GetSymbol(tree); results[i] = the value returned (0 for void). */
int szo_sm_script(const uint8_t *buf, int n, const int32_t *ops, int nops, const int16_t *tree, int treeSize, int32_t *results) {
    SM in; sm_reset(&in);
    if (sm_set_input(&in, buf, 0, n) < 0) return -1;
    IHT t; t.tree = (int16_t *)tree; t.treeSize = treeSize;
    for (int i = 0; i < nops; i++) {
        const int op = ops[2 * i], arg = ops[2 * i + 1];
        int r = 0;
        switch (op) {
        case 0: r = sm_peek(&in, arg); break;
        case 1: sm_drop(&in, arg); break;
        case 2: sm_skip_to_byte(&in); break;
        case 3: r = in.bitsInBuffer_; break;
        case 4: r = sm_available_bytes(&in); break;
        case 5: r = iht_get_symbol(&t, &in); break;
        default: return -2;
        }
        results[i] = r;
    }
    return 0;
}

/* ================================================================= InflaterDynHeader.cs
 * The C# iterator state machine (:42-120) restated as an explicit resumable state machine. */
static const int MetaCodeLengthIndex[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
typedef struct {
    int step; /* 0 litlen count,1 dist count,2 meta count,3 meta lens,4 symbols,5 repeat bits,6 done */
    int litLenCodeCount, distanceCodeCount, metaCodeCount, dataCodeCount;
    int i, index, symbol;
    uint8_t codeLengths[286 + 30];
    uint8_t codeLength;
    IHT meta;
    int have_meta;
    IHT litLen, dist;
} DynHeader;

static void dyn_init(DynHeader *h) { memset(h, 0, sizeof(*h)); }
static void dyn_free_meta(DynHeader *h) { if (h->have_meta) { iht_free(&h->meta); h->have_meta = 0; } }
/* returns 1 done, 0 need input, <0 error */
static int dyn_attempt_read(DynHeader *h, SM *input) {
    int bits;
    for (;;) {
        switch (h->step) {
        case 0:
            if ((bits = sm_peek(input, 5)) < 0) return 0;
            sm_drop(input, 5); h->litLenCodeCount = bits + 257; h->step = 1; break;
        case 1:
            if ((bits = sm_peek(input, 5)) < 0) return 0;
            sm_drop(input, 5); h->distanceCodeCount = bits + 1; h->step = 2; break;
        case 2:
            if ((bits = sm_peek(input, 4)) < 0) return 0;
            sm_drop(input, 4); h->metaCodeCount = bits + 4;
            h->dataCodeCount = h->litLenCodeCount + h->distanceCodeCount;
            if (h->litLenCodeCount > 286) return SZO_ERR_DYN_HEADER;
            if (h->distanceCodeCount > 30) return SZO_ERR_DYN_HEADER;
            if (h->metaCodeCount > 19) return SZO_ERR_DYN_HEADER;
            h->i = 0; h->step = 3; break;
        case 3:
            while (h->i < h->metaCodeCount) {
                if ((bits = sm_peek(input, 3)) < 0) return 0;
                sm_drop(input, 3);
                h->codeLengths[MetaCodeLengthIndex[h->i]] = (uint8_t)bits;
                h->i++;
            }
            { /* new InflaterHuffmanTree(codeLengths) — over the whole 316-entry array :64 */
                int rc = iht_build(&h->meta, h->codeLengths, 286 + 30);
                h->have_meta = 1;
                if (rc < 0) return rc;
            }
            h->index = 0; h->step = 4; break;
        case 4:
            if (h->index >= h->dataCodeCount) { h->step = 6; break; }
            {
                int symbol = iht_get_symbol(&h->meta, input);
                if (symbol == -1) return 0;
                if (symbol < -1) return symbol + 100;
                if (symbol < 16) { h->codeLengths[h->index++] = (uint8_t)symbol; break; }
                h->symbol = symbol;
                if (symbol == 16) {
                    if (h->index == 0) return SZO_ERR_DYN_HEADER;
                    h->codeLength = h->codeLengths[h->index - 1];
                } else h->codeLength = 0;
                h->step = 5;
            }
            break;
        case 5: {
            int nb = h->symbol == 16 ? 2 : (h->symbol == 17 ? 3 : 7);
            int base = h->symbol == 18 ? 11 : 3;
            if ((bits = sm_peek(input, nb)) < 0) return 0;
            sm_drop(input, nb);
            int repeatCount = bits + base;
            if (h->index + repeatCount > h->dataCodeCount) return SZO_ERR_DYN_HEADER;
            while (repeatCount-- > 0) h->codeLengths[h->index++] = h->codeLength;
            h->step = 4;
        } break;
        case 6:
            if (h->codeLengths[256] == 0) return SZO_ERR_DYN_HEADER;
            {
                int rc = iht_build(&h->litLen, h->codeLengths, h->litLenCodeCount);
                if (rc < 0) return rc;
                rc = iht_build(&h->dist, h->codeLengths + h->litLenCodeCount, h->distanceCodeCount);
                if (rc < 0) { iht_free(&h->litLen); return rc; }
            }
            dyn_free_meta(h);
            h->step = 7;
            return 1;
        default:
            return 1;
        }
    }
}

/* ================================================================= Inflater.cs */
static const int CPLENS[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const int CPLEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const int CPDIST[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const int CPDEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
enum { DECODE_HEADER = 0, DECODE_DICT, DECODE_BLOCKS, DECODE_STORED_LEN1, DECODE_STORED_LEN2, DECODE_STORED,
       DECODE_DYN_HEADER, DECODE_HUFFMAN, DECODE_HUFFMAN_LENBITS, DECODE_HUFFMAN_DIST, DECODE_HUFFMAN_DISTBITS,
       DECODE_CHKSUM, FINISHED };

static IHT defLitLen, defDist;
static int defs_ready = 0;
static void init_defs(void) { /* InflaterHuffmanTree.cs:34-70 */
    if (defs_ready) return;
    uint8_t cl[288];
    int i = 0;
    while (i < 144) cl[i++] = 8;
    while (i < 256) cl[i++] = 9;
    while (i < 280) cl[i++] = 7;
    while (i < 288) cl[i++] = 8;
    iht_build(&defLitLen, cl, 288);
    for (i = 0; i < 32; i++) cl[i] = 5;
    iht_build(&defDist, cl, 32);
    defs_ready = 1;
}

struct szo_inflater {
    int mode, readAdler, neededBits, repLength, repDist, uncomprLen, isLastBlock;
    int64_t totalOut, totalIn;
    int noHeader;
    SM input;
    OW outputWindow;
    DynHeader dyn;
    int dyn_live;        /* dyn.litLen/dist allocated */
    const IHT *litlenTree, *distTree;
    uint32_t adler;
    int err;             /* sticky error */
};

static void inf_drop_trees(szo_inflater *s) {
    if (s->dyn_live) {
        if (s->dyn.step == 7) { iht_free(&s->dyn.litLen); iht_free(&s->dyn.dist); }
        dyn_free_meta(&s->dyn);
        s->dyn_live = 0;
    }
    s->litlenTree = s->distTree = NULL;
}
void szo_inflater_reset(szo_inflater *s) { /* :188 */
    s->mode = s->noHeader ? DECODE_BLOCKS : DECODE_HEADER;
    s->totalIn = 0; s->totalOut = 0;
    sm_reset(&s->input);
    s->outputWindow.windowFilled = s->outputWindow.windowEnd = 0;
    inf_drop_trees(s);
    s->isLastBlock = 0;
    s->adler = 1;
    s->err = 0;
}
szo_inflater *szo_inflater_new(int noHeader) { /* :156 */
    init_defs();
    szo_inflater *s = (szo_inflater *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->noHeader = noHeader;
    s->adler = 1;
    s->mode = noHeader ? DECODE_BLOCKS : DECODE_HEADER;
    return s;
}
void szo_inflater_free(szo_inflater *s) { if (s) { inf_drop_trees(s); free(s); } }

/* Decode helpers return: 1 true, 0 false, <0 error */
static int inf_decode_header(szo_inflater *s) { /* :211 */
    int header = sm_peek(&s->input, 16);
    if (header < 0) return 0;
    sm_drop(&s->input, 16);
    header = ((header << 8) | (header >> 8)) & 0xffff;
    if (header % 31 != 0) return SZO_ERR_HEADER_CHECKSUM;
    if ((header & 0x0f00) != (8 << 8)) return SZO_ERR_METHOD_UNKNOWN;
    if ((header & 0x0020) == 0) s->mode = DECODE_BLOCKS;
    else { s->mode = DECODE_DICT; s->neededBits = 32; }
    return 1;
}
static int inf_decode_dict(szo_inflater *s) { /* :254 */
    while (s->neededBits > 0) {
        int b = sm_peek(&s->input, 8);
        if (b < 0) return 0;
        sm_drop(&s->input, 8);
        s->readAdler = (int)(((uint32_t)s->readAdler << 8) | (uint32_t)b);
        s->neededBits -= 8;
    }
    return 0;
}
static int inf_decode_huffman(szo_inflater *s) { /* :283 */
    int free_ = OW_SIZE - s->outputWindow.windowFilled;
    while (free_ >= 258) {
        int symbol;
        switch (s->mode) {
        case DECODE_HUFFMAN:
            while (((symbol = iht_get_symbol(s->litlenTree, &s->input)) & ~0xff) == 0) {
                int rc = ow_write(&s->outputWindow, symbol);
                if (rc < 0) return rc;
                if (--free_ < 258) return 1;
            }
            if (symbol < 257) {
                if (symbol < 0) {
                    if (symbol < -1) return symbol + 100;
                    return 0;
                } else {
                    inf_drop_trees(s);
                    s->mode = DECODE_BLOCKS;
                    return 1;
                }
            }
            if (symbol - 257 >= 29) return SZO_ERR_ILLEGAL_LEN_CODE;
            s->repLength = CPLENS[symbol - 257];
            s->neededBits = CPLEXT[symbol - 257];
            /* fall through */
        case DECODE_HUFFMAN_LENBITS:
            if (s->neededBits > 0) {
                s->mode = DECODE_HUFFMAN_LENBITS;
                int i = sm_peek(&s->input, s->neededBits);
                if (i < 0) return 0;
                sm_drop(&s->input, s->neededBits);
                s->repLength += i;
            }
            s->mode = DECODE_HUFFMAN_DIST;
            /* fall through */
        case DECODE_HUFFMAN_DIST:
            symbol = iht_get_symbol(s->distTree, &s->input);
            if (symbol < 0) {
                if (symbol < -1) return symbol + 100;
                return 0;
            }
            if (symbol >= 30) return SZO_ERR_ILLEGAL_DIST_CODE;
            s->repDist = CPDIST[symbol];
            s->neededBits = CPDEXT[symbol];
            /* fall through */
        case DECODE_HUFFMAN_DISTBITS:
            if (s->neededBits > 0) {
                s->mode = DECODE_HUFFMAN_DISTBITS;
                int i = sm_peek(&s->input, s->neededBits);
                if (i < 0) return 0;
                sm_drop(&s->input, s->neededBits);
                s->repDist += i;
            }
            {
                int rc = ow_repeat(&s->outputWindow, s->repLength, s->repDist);
                if (rc < 0) return rc;
            }
            free_ -= s->repLength;
            s->mode = DECODE_HUFFMAN;
            break;
        default:
            return SZO_ERR_STATE;
        }
    }
    return 1;
}
static int inf_decode_chksum(szo_inflater *s) { /* :397 */
    while (s->neededBits > 0) {
        int b = sm_peek(&s->input, 8);
        if (b < 0) return 0;
        sm_drop(&s->input, 8);
        s->readAdler = (int)(((uint32_t)s->readAdler << 8) | (uint32_t)b);
        s->neededBits -= 8;
    }
    if ((int)s->adler != s->readAdler) return SZO_ERR_ADLER_MISMATCH;
    s->mode = FINISHED;
    return 0;
}
static int inf_decode(szo_inflater *s) { /* :429 */
    switch (s->mode) {
    case DECODE_HEADER: return inf_decode_header(s);
    case DECODE_DICT: return inf_decode_dict(s);
    case DECODE_CHKSUM: return inf_decode_chksum(s);
    case DECODE_BLOCKS: {
        if (s->isLastBlock) {
            if (s->noHeader) { s->mode = FINISHED; return 0; }
            sm_skip_to_byte(&s->input);
            s->neededBits = 32;
            s->mode = DECODE_CHKSUM;
            return 1;
        }
        int type = sm_peek(&s->input, 3);
        if (type < 0) return 0;
        sm_drop(&s->input, 3);
        s->isLastBlock |= (type & 1) != 0;
        switch (type >> 1) {
        case 0: sm_skip_to_byte(&s->input); s->mode = DECODE_STORED_LEN1; break;
        case 1: s->litlenTree = &defLitLen; s->distTree = &defDist; s->mode = DECODE_HUFFMAN; break;
        case 2: inf_drop_trees(s); dyn_init(&s->dyn); s->dyn_live = 1; s->mode = DECODE_DYN_HEADER; break;
        default: return SZO_ERR_UNKNOWN_BLOCK;
        }
        return 1;
    }
    case DECODE_STORED_LEN1:
        if ((s->uncomprLen = sm_peek(&s->input, 16)) < 0) return 0;
        sm_drop(&s->input, 16);
        s->mode = DECODE_STORED_LEN2;
        /* fall through */
    case DECODE_STORED_LEN2: {
        int nlen = sm_peek(&s->input, 16);
        if (nlen < 0) return 0;
        sm_drop(&s->input, 16);
        if (nlen != (s->uncomprLen ^ 0xffff)) return SZO_ERR_BROKEN_STORED;
        s->mode = DECODE_STORED;
    }
        /* fall through */
    case DECODE_STORED: {
        int more = ow_copy_stored(&s->outputWindow, &s->input, s->uncomprLen);
        s->uncomprLen -= more;
        if (s->uncomprLen == 0) { s->mode = DECODE_BLOCKS; return 1; }
        return !sm_needs_input(&s->input);
    }
    case DECODE_DYN_HEADER: {
        int rc = dyn_attempt_read(&s->dyn, &s->input);
        if (rc <= 0) return rc;
        s->litlenTree = &s->dyn.litLen;
        s->distTree = &s->dyn.dist;
        s->mode = DECODE_HUFFMAN;
    }
        /* fall through */
    case DECODE_HUFFMAN:
    case DECODE_HUFFMAN_LENBITS:
    case DECODE_HUFFMAN_DIST:
    case DECODE_HUFFMAN_DISTBITS:
        return inf_decode_huffman(s);
    case FINISHED: return 0;
    default: return SZO_ERR_STATE;
    }
}

int szo_inflater_needs_dictionary(const szo_inflater *s) { return s->mode == DECODE_DICT && s->neededBits == 0; } /* :794 */
int szo_inflater_set_dictionary(szo_inflater *s, const uint8_t *p, int n) { /* :563 */
    if (n < 0) return SZO_ERR_ARG;
    if (!szo_inflater_needs_dictionary(s)) return SZO_ERR_STATE;
    if (!s->noHeader) {
        s->adler = szo_adler32(s->adler, p, (size_t)n);
        if ((int)s->adler != s->readAdler) return SZO_ERR_ADLER_MISMATCH;
        s->adler = 1;
    }
    int rc = ow_copy_dict(&s->outputWindow, p, 0, n);
    if (rc < 0) return rc;
    s->mode = DECODE_BLOCKS;
    return 0;
}
int szo_inflater_set_input(szo_inflater *s, const uint8_t *p, int n) { /* :629 */
    int rc = sm_set_input(&s->input, p, 0, n);
    if (rc < 0) return rc;
    s->totalIn += (int64_t)n;
    return 0;
}
int szo_inflater_inflate(szo_inflater *s, uint8_t *buffer, int count) { /* :715 */
    if (s->err) return s->err;
    if (count < 0) return SZO_ERR_ARG;
    int offset = 0;
    if (count == 0) {
        if (!szo_inflater_is_finished(s)) {
            int rc = inf_decode(s);
            if (rc < 0) { s->err = rc; return rc; }
        }
        return 0;
    }
    int bytesCopied = 0;
    int dec;
    do {
        if (s->mode != DECODE_CHKSUM) {
            int more = ow_copy_output(&s->outputWindow, buffer, offset, count);
            if (more > 0) {
                if (!s->noHeader) s->adler = szo_adler32(s->adler, buffer + offset, (size_t)more);
                offset += more;
                bytesCopied += more;
                s->totalOut += (int64_t)more;
                count -= more;
                if (count == 0) return bytesCopied;
            }
        }
        dec = inf_decode(s);
        if (dec < 0) { s->err = dec; return dec; }
    } while (dec || (s->outputWindow.windowFilled > 0 && s->mode != DECODE_CHKSUM));
    return bytesCopied;
}
int szo_inflater_needs_input(const szo_inflater *s) { return sm_needs_input(&s->input); }                  /* :783 */
int szo_inflater_is_finished(const szo_inflater *s) { return s->mode == FINISHED && s->outputWindow.windowFilled == 0; } /* :806 */
int szo_inflater_remaining_input(const szo_inflater *s) { return sm_available_bytes(&s->input); }         /* :878 */
int64_t szo_inflater_total_in(const szo_inflater *s) { return s->totalIn - (int64_t)sm_available_bytes(&s->input); } /* :862 */
int64_t szo_inflater_total_out(const szo_inflater *s) { return s->totalOut; }
uint32_t szo_inflater_adler(const szo_inflater *s) { /* :823 */
    if (szo_inflater_needs_dictionary(s)) return (uint32_t)s->readAdler;
    return s->noHeader ? 0 : s->adler;
}

/* Mirrors InflaterInputStream.Read's loop (CS/InflaterInputStream.cs:658-688) with the whole
 * compressed buffer handed over in <=1 GiB SetInput calls. */
int64_t szo_inflate_oneshot(const uint8_t *in, size_t n, int noHeader, uint8_t *out, size_t out_cap, size_t *consumed) {
    szo_inflater *s = szo_inflater_new(noHeader);
    if (!s) return SZO_ERR_ARG;
    size_t ip = 0, op = 0;
    int64_t rc = 0;
    uint8_t scratch[8];
    for (;;) {
        size_t room = out_cap - op;
        int want = room > ((size_t)1 << 30) ? (1 << 30) : (int)room;
        int k;
        if (want == 0) {
            k = szo_inflater_inflate(s, scratch, 1); /* any further output means overflow */
            if (k > 0) { rc = -100; break; }
        } else {
            k = szo_inflater_inflate(s, out + op, want);
        }
        if (k < 0) { rc = k; break; }
        op += (size_t)k;
        if (szo_inflater_is_finished(s)) { rc = (int64_t)op; break; }
        if (k == 0) {
            if (szo_inflater_needs_dictionary(s)) { rc = SZO_ERR_STATE; break; }
            if (szo_inflater_needs_input(s)) {
                if (ip >= n) { rc = -102; break; } /* "Unexpected EOF" CS/InflaterInputStream.cs:494 */
                size_t chunk = n - ip > ((size_t)1 << 30) ? ((size_t)1 << 30) : n - ip;
                szo_inflater_set_input(s, in + ip, (int)chunk);
                ip += chunk;
            } else if (want != 0) { rc = -103; break; } /* "Invalid input data" :683 */
        }
    }
    if (consumed) *consumed = (size_t)szo_inflater_total_in(s);
    szo_inflater_free(s);
    return rc;
}

/* Test helper: like szo_inflate_oneshot, but output is requested ONE byte per Inflate() call, which is the finest
 * grain at which the reference hands out bytes (C/Inflater.cs:749-774: copy from the window first, Decode() only when the
 * window is drained) — so *produced is the longest prefix a caller of the reference can have received before the
 * exception that ends the stream.  Returns total bytes (finished) or the negative error. */
int64_t szo_inflate_probe(const uint8_t *in, size_t n, int noHeader, uint8_t *out, size_t out_cap, size_t *consumed,
                          size_t *produced) {
    szo_inflater *s = szo_inflater_new(noHeader);
    if (!s) return SZO_ERR_ARG;
    size_t op = 0;
    int64_t rc = 0;
    int fed = 0;
    uint8_t scratch[2];
    for (;;) {
        int k = szo_inflater_inflate(s, op < out_cap ? out + op : scratch, 1);
        if (k < 0) { rc = k; break; }
        if (k > 0 && op >= out_cap) { rc = -100; break; }
        op += (size_t)k;
        if (szo_inflater_is_finished(s)) { rc = (int64_t)op; break; }
        if (k == 0) {
            if (szo_inflater_needs_dictionary(s)) { rc = SZO_ERR_STATE; break; }
            if (szo_inflater_needs_input(s)) {
                if (fed || n >= ((size_t)1 << 30)) { rc = -102; break; }
                szo_inflater_set_input(s, in, (int)n);
                fed = 1;
            } else { rc = -103; break; }
        }
    }
    if (consumed) *consumed = (size_t)szo_inflater_total_in(s);
    if (produced) *produced = op;
    szo_inflater_free(s);
    return rc;
}
