/*
 * szl_deflate_oracle.c — CPU restatement of SharpZipLib's Deflater (compress side).
 * TEST INFRASTRUCTURE ONLY (see szl_oracle.h).  Encoder parity: "parity unpinned" by the
 * reference's tests; pinned by line-by-line restatement + vectors (see header).
 *
 * Follows (paths under /root/reference/src/ICSharpCode.SharpZipLib/Zip/Compression/):
 *   DeflaterConstants.cs (all) · PendingBuffer.cs · DeflaterPending.cs · DeflaterHuffman.cs ·
 *   DeflaterEngine.cs · Deflater.cs
 */
#include "szl_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---- DeflaterConstants.cs:44-144 */
enum {
    STORED_BLOCK = 0, STATIC_TREES = 1, DYN_TREES = 2, PRESET_DICT = 0x20,
    MAX_MATCH = 258, MIN_MATCH = 3, MAX_WBITS = 15, WSIZE = 1 << 15, WMASK = WSIZE - 1,
    HASH_BITS = 15, HASH_SIZE = 1 << 15, HASH_MASK = HASH_SIZE - 1, HASH_SHIFT = 5,
    MIN_LOOKAHEAD = MAX_MATCH + MIN_MATCH + 1, MAX_DIST = WSIZE - MIN_LOOKAHEAD,
    PENDING_BUF_SIZE = 1 << 16, MAX_BLOCK_SIZE = 65531, /* Math.Min(65535, PENDING_BUF_SIZE-5) :104 */
    DEFLATE_STORED = 0, DEFLATE_FAST = 1, DEFLATE_SLOW = 2
};
static const int GOOD_LENGTH[10] = {0, 4, 4, 4, 4, 8, 8, 8, 32, 32};         /* :124 */
static const int MAX_LAZY[10] = {0, 4, 5, 6, 4, 16, 16, 32, 128, 258};       /* :129 */
static const int NICE_LENGTH[10] = {0, 8, 16, 32, 16, 32, 128, 128, 258, 258}; /* :134 */
static const int MAX_CHAIN[10] = {0, 4, 8, 32, 16, 32, 128, 256, 1024, 4096};  /* :139 */
static const int COMPR_FUNC[10] = {0, 1, 1, 1, 1, 2, 2, 2, 2, 2};              /* :144 */

/* ================================================================= PendingBuffer.cs */
/* The reference allocates exactly PENDING_BUF_SIZE and would throw IndexOutOfRange beyond it
 * (never guarded).  We keep 4x so a pathological block cannot corrupt memory; the
 * byte stream produced is the same whenever the reference would not have thrown. */
#define PEND_CAP (4 * PENDING_BUF_SIZE)
typedef struct {
    uint8_t buffer[PEND_CAP];
    int start, end;
    uint32_t bits;
    int bitCount;
    int64_t total_bits; /* trace only: bits ever written (for block bit_start) */
} Pending;

static void pend_reset(Pending *p) { p->start = p->end = p->bitCount = 0; p->total_bits = 0; } /* :43 ; NB bits not cleared */
static void pend_write_short(Pending *p, int v) { /* :81 */
    p->buffer[p->end++] = (uint8_t)v;
    p->buffer[p->end++] = (uint8_t)(v >> 8);
    p->total_bits += 16;
}
static void pend_write_block(Pending *p, const uint8_t *b, int off, int len) { /* :117 */
    memcpy(p->buffer + p->end, b + off, (size_t)len);
    p->end += len;
    p->total_bits += 8 * (int64_t)len;
}
static void pend_align(Pending *p) { /* :143 */
    if (p->bitCount > 0) {
        p->buffer[p->end++] = (uint8_t)p->bits;
        if (p->bitCount > 8) p->buffer[p->end++] = (uint8_t)(p->bits >> 8);
    }
    p->total_bits += (-p->bitCount) & 7;
    p->bits = 0;
    p->bitCount = 0;
}
static void pend_write_bits(Pending *p, int b, int count) { /* :168 */
    p->bits |= (uint32_t)(b << p->bitCount);
    p->bitCount += count;
    p->total_bits += count;
    if (p->bitCount >= 16) {
        p->buffer[p->end++] = (uint8_t)p->bits;
        p->buffer[p->end++] = (uint8_t)(p->bits >> 8);
        p->bits >>= 16;
        p->bitCount -= 16;
    }
}
static void pend_write_short_msb(Pending *p, int s) { /* :195 */
    p->buffer[p->end++] = (uint8_t)(s >> 8);
    p->buffer[p->end++] = (uint8_t)s;
    p->total_bits += 16;
}
static int pend_is_flushed(const Pending *p) { return p->end == 0; } /* :212 */
static int pend_flush(Pending *p, uint8_t *out, int offset, int length) { /* :226 */
    if (p->bitCount >= 8) {
        p->buffer[p->end++] = (uint8_t)p->bits;
        p->bits >>= 8;
        p->bitCount -= 8;
    }
    if (length > p->end - p->start) {
        length = p->end - p->start;
        memcpy(out + offset, p->buffer + p->start, (size_t)length);
        p->start = 0;
        p->end = 0;
    } else {
        memcpy(out + offset, p->buffer + p->start, (size_t)length);
        p->start += length;
    }
    return length;
}

/* ================================================================= DeflaterHuffman.cs */
enum { BUFSIZE = 1 << 14, LITERAL_NUM = 286, DIST_NUM = 30, BITLEN_NUM = 19,
       REP_3_6 = 16, REP_3_10 = 17, REP_11_138 = 18, EOF_SYMBOL = 256 };
static const int BL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15}; /* :37 */
static const uint8_t bit4Reverse[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};      /* :39 */

static int16_t bit_reverse(int v) { /* :924 */
    return (int16_t)(bit4Reverse[v & 0xF] << 12 | bit4Reverse[(v >> 4) & 0xF] << 8 |
                     bit4Reverse[(v >> 8) & 0xF] << 4 | bit4Reverse[v >> 12]);
}
static int Lcode(int length) { /* :932 */
    if (length == 255) return 285;
    int code = 257;
    while (length >= 8) { code += 4; length >>= 1; }
    return code + length;
}
static int Dcode(int distance) { /* :948 */
    int code = 0;
    while (distance >= 4) { code += 2; distance >>= 1; }
    return code + distance;
}

typedef struct {
    int nsyms;             /* freqs.Length */
    int16_t freqs[LITERAL_NUM];
    uint8_t length_own[LITERAL_NUM];
    int16_t codes_own[LITERAL_NUM];
    const uint8_t *length; /* may alias static tables (SetStaticCodes :134) */
    const int16_t *codes;
    int minNumCodes, numCodes, maxLength;
    int bl_counts[15];
} Tree;

static int16_t staticLCodes[LITERAL_NUM], staticDCodes[DIST_NUM];
static uint8_t staticLLength[LITERAL_NUM], staticDLength[DIST_NUM];
static int statics_ready = 0;
static void init_statics(void) { /* static ctor :596-643 */
    if (statics_ready) return;
    int i = 0;
    while (i < 144) { staticLCodes[i] = bit_reverse((0x030 + i) << 8); staticLLength[i++] = 8; }
    while (i < 256) { staticLCodes[i] = bit_reverse((0x190 - 144 + i) << 7); staticLLength[i++] = 9; }
    while (i < 280) { staticLCodes[i] = bit_reverse((0x000 - 256 + i) << 9); staticLLength[i++] = 7; }
    while (i < LITERAL_NUM) { staticLCodes[i] = bit_reverse((0x0c0 - 280 + i) << 8); staticLLength[i++] = 8; }
    for (i = 0; i < DIST_NUM; i++) { staticDCodes[i] = bit_reverse(i << 11); staticDLength[i] = 5; }
    statics_ready = 1;
}

static void tree_init(Tree *t, int elems, int minCodes, int maxLength) { /* :84 */
    memset(t, 0, sizeof(*t));
    t->nsyms = elems; t->minNumCodes = minCodes; t->maxLength = maxLength;
    t->length = t->length_own; t->codes = t->codes_own;
}
static void tree_reset(Tree *t) { /* :98 */
    for (int i = 0; i < t->nsyms; i++) t->freqs[i] = 0;
    t->codes = t->codes_own; t->length = t->length_own;
}
static void tree_write_symbol(const Tree *t, Pending *p, int code) { /* :108 */
    pend_write_bits(p, t->codes[code] & 0xffff, t->length[code]);
}
static void tree_build_codes(Tree *t) { /* :151 */
    int nextCode[15];
    int code = 0;
    memset(t->codes_own, 0, sizeof(t->codes_own));
    for (int bits = 0; bits < t->maxLength; bits++) {
        nextCode[bits] = code;
        code += t->bl_counts[bits] << (15 - bits);
    }
    for (int i = 0; i < t->numCodes; i++) {
        int bits = t->length[i];
        if (bits > 0) {
            t->codes_own[i] = bit_reverse(nextCode[bits - 1]);
            nextCode[bits - 1] += 1 << (16 - bits);
        }
    }
    t->codes = t->codes_own;
}

static void tree_build_length(Tree *t, const int *childs, int childsLen) { /* :475 */
    memset(t->length_own, 0, sizeof(t->length_own));
    t->length = t->length_own;
    int numNodes = childsLen / 2;
    int numLeafs = (numNodes + 1) / 2;
    int overflow = 0;
    int maxLength = t->maxLength;
    for (int i = 0; i < maxLength; i++) t->bl_counts[i] = 0;

    int lengths[2 * LITERAL_NUM];
    lengths[numNodes - 1] = 0;
    for (int i = numNodes - 1; i >= 0; i--) {
        if (childs[2 * i + 1] != -1) {
            int bitLength = lengths[i] + 1;
            if (bitLength > maxLength) { bitLength = maxLength; overflow++; }
            lengths[childs[2 * i]] = lengths[childs[2 * i + 1]] = bitLength;
        } else {
            int bitLength = lengths[i];
            t->bl_counts[bitLength - 1]++;
            t->length_own[childs[2 * i]] = (uint8_t)lengths[i];
        }
    }
    if (overflow == 0) return;

    int incrBitLen = maxLength - 1;
    do {
        while (t->bl_counts[--incrBitLen] == 0) { }
        do {
            t->bl_counts[incrBitLen]--;
            t->bl_counts[++incrBitLen]++;
            overflow -= 1 << (maxLength - 1 - incrBitLen);
        } while (overflow > 0 && incrBitLen < maxLength - 1);
    } while (overflow > 0);

    t->bl_counts[maxLength - 1] += overflow;
    t->bl_counts[maxLength - 2] -= overflow;

    int nodePtr = 2 * numLeafs;
    for (int bits = maxLength; bits != 0; bits--) {
        int n = t->bl_counts[bits - 1];
        while (n > 0) {
            int childPtr = 2 * childs[nodePtr++];
            if (childs[childPtr + 1] == -1) {
                t->length_own[childs[childPtr]] = (uint8_t)bits;
                n--;
            }
        }
    }
}

static void tree_build_tree(Tree *t) { /* :196 */
    int numSymbols = t->nsyms;
    int heap[LITERAL_NUM];
    int heapLen = 0, maxCode = 0;
    for (int n = 0; n < numSymbols; n++) {
        int freq = t->freqs[n];
        if (freq != 0) {
            int pos = heapLen++;
            int ppos;
            while (pos > 0 && t->freqs[heap[ppos = (pos - 1) / 2]] > freq) {
                heap[pos] = heap[ppos];
                pos = ppos;
            }
            heap[pos] = n;
            maxCode = n;
        }
    }
    while (heapLen < 2) {
        int node = maxCode < 2 ? ++maxCode : 0;
        heap[heapLen++] = node;
    }
    t->numCodes = (maxCode + 1 > t->minNumCodes) ? maxCode + 1 : t->minNumCodes;

    int numLeafs = heapLen;
    int childs[4 * LITERAL_NUM];
    int values[2 * LITERAL_NUM];
    int childsLen = 4 * heapLen - 2;
    int numNodes = numLeafs;
    for (int i = 0; i < heapLen; i++) {
        int node = heap[i];
        childs[2 * i] = node;
        childs[2 * i + 1] = -1;
        values[i] = t->freqs[node] << 8;
        heap[i] = i;
    }
    do {
        int first = heap[0];
        int last = heap[--heapLen];
        int ppos = 0;
        int path = 1;
        while (path < heapLen) {
            if (path + 1 < heapLen && values[heap[path]] > values[heap[path + 1]]) path++;
            heap[ppos] = heap[path];
            ppos = path;
            path = path * 2 + 1;
        }
        int lastVal = values[last];
        while ((path = ppos) > 0 && values[heap[ppos = (path - 1) / 2]] > lastVal) heap[path] = heap[ppos];
        heap[path] = last;

        int second = heap[0];
        last = numNodes++;
        childs[2 * last] = first;
        childs[2 * last + 1] = second;
        int d1 = values[first] & 0xff, d2 = values[second] & 0xff;
        int mindepth = d1 < d2 ? d1 : d2;
        values[last] = lastVal = values[first] + values[second] - mindepth + 1;

        ppos = 0;
        path = 1;
        while (path < heapLen) {
            if (path + 1 < heapLen && values[heap[path]] > values[heap[path + 1]]) path++;
            heap[ppos] = heap[path];
            ppos = path;
            path = ppos * 2 + 1;
        }
        while ((path = ppos) > 0 && values[heap[ppos = (path - 1) / 2]] > lastVal) heap[path] = heap[ppos];
        heap[path] = last;
    } while (heapLen > 1);
    /* heap[0] == childsLen/2 - 1 invariant (:323) */
    tree_build_length(t, childs, childsLen);
}

/* test tap: the code lengths DeflaterHuffman.Tree.BuildTree (:196-329) + BuildLength (:475-579) give a frequency vector (the device's
 * wave-parallel build is checked against this on random histograms, tests/test_gpu_tree_build.py) */
int szo_tree_lengths(const int16_t *freqs, int nsyms, int minCodes, int maxLen, uint8_t *len_out, int *numCodes_out) {
    static Tree t;
    if (nsyms < 1 || nsyms > LITERAL_NUM) return -1;
    tree_init(&t, nsyms, minCodes, maxLen);
    memcpy(t.freqs, freqs, sizeof(int16_t) * (size_t)nsyms);
    tree_build_tree(&t);
    memcpy(len_out, t.length_own, (size_t)nsyms);
    if (numCodes_out) *numCodes_out = t.numCodes;
    return 0;
}

static int tree_encoded_length(const Tree *t) { /* :331 */
    int len = 0;
    for (int i = 0; i < t->nsyms; i++) len += t->freqs[i] * t->length[i];
    return len;
}

static void tree_calc_bl_freq(const Tree *t, Tree *bl) { /* :349 */
    int max_count, min_count, count, curlen = -1;
    int i = 0;
    while (i < t->numCodes) {
        count = 1;
        int nextlen = t->length[i];
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else {
            max_count = 6; min_count = 3;
            if (curlen != nextlen) { bl->freqs[nextlen]++; count = 0; }
        }
        curlen = nextlen;
        i++;
        while (i < t->numCodes && curlen == t->length[i]) {
            i++;
            if (++count >= max_count) break;
        }
        if (count < min_count) bl->freqs[curlen] += (int16_t)count;
        else if (curlen != 0) bl->freqs[REP_3_6]++;
        else if (count <= 10) bl->freqs[REP_3_10]++;
        else bl->freqs[REP_11_138]++;
    }
}

static void tree_write_tree(const Tree *t, const Tree *bl, Pending *p) { /* :411 */
    int max_count, min_count, count, curlen = -1;
    int i = 0;
    while (i < t->numCodes) {
        count = 1;
        int nextlen = t->length[i];
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else {
            max_count = 6; min_count = 3;
            if (curlen != nextlen) { tree_write_symbol(bl, p, nextlen); count = 0; }
        }
        curlen = nextlen;
        i++;
        while (i < t->numCodes && curlen == t->length[i]) {
            i++;
            if (++count >= max_count) break;
        }
        if (count < min_count) { while (count-- > 0) tree_write_symbol(bl, p, curlen); }
        else if (curlen != 0) { tree_write_symbol(bl, p, REP_3_6); pend_write_bits(p, count - 3, 2); }
        else if (count <= 10) { tree_write_symbol(bl, p, REP_3_10); pend_write_bits(p, count - 3, 3); }
        else { tree_write_symbol(bl, p, REP_11_138); pend_write_bits(p, count - 11, 7); }
    }
}

typedef struct {
    Pending *pending;
    Tree literalTree, distTree, blTree;
    int16_t d_buf[BUFSIZE];
    uint8_t l_buf[BUFSIZE];
    int last_lit, extra_bits;
    szo_trace *trace;
    int64_t trace_tokens_done;
} Huffman;

static void huff_init(Huffman *h, Pending *p) { /* :645 */
    init_statics();
    h->pending = p;
    tree_init(&h->literalTree, LITERAL_NUM, 257, 15);
    tree_init(&h->distTree, DIST_NUM, 1, 15);
    tree_init(&h->blTree, BITLEN_NUM, 4, 7);
    h->last_lit = 0; h->extra_bits = 0; h->trace = NULL; h->trace_tokens_done = 0;
}
static void huff_reset(Huffman *h) { /* :660 */
    h->last_lit = 0; h->extra_bits = 0;
    tree_reset(&h->literalTree); tree_reset(&h->distTree); tree_reset(&h->blTree);
}
static void huff_send_all_trees(Huffman *h, int blTreeCodes) { /* :676 */
    tree_build_codes(&h->blTree);
    tree_build_codes(&h->literalTree);
    tree_build_codes(&h->distTree);
    pend_write_bits(h->pending, h->literalTree.numCodes - 257, 5);
    pend_write_bits(h->pending, h->distTree.numCodes - 1, 5);
    pend_write_bits(h->pending, blTreeCodes - 4, 4);
    for (int rank = 0; rank < blTreeCodes; rank++) pend_write_bits(h->pending, h->blTree.length[BL_ORDER[rank]], 3);
    tree_write_tree(&h->literalTree, &h->blTree, h->pending);
    tree_write_tree(&h->distTree, &h->blTree, h->pending);
}
static void huff_compress_block(Huffman *h) { /* :701 */
    for (int i = 0; i < h->last_lit; i++) {
        int litlen = h->l_buf[i] & 0xff;
        int dist = h->d_buf[i];
        if (dist-- != 0) {
            int lc = Lcode(litlen);
            tree_write_symbol(&h->literalTree, h->pending, lc);
            int bits = (lc - 261) / 4; /* C# and C99 both truncate toward zero */
            if (bits > 0 && bits <= 5) pend_write_bits(h->pending, litlen & ((1 << bits) - 1), bits);
            int dc = Dcode(dist);
            tree_write_symbol(&h->distTree, h->pending, dc);
            bits = dc / 2 - 1;
            if (bits > 0) pend_write_bits(h->pending, dist & ((1 << bits) - 1), bits);
        } else {
            tree_write_symbol(&h->literalTree, h->pending, litlen);
        }
    }
    tree_write_symbol(&h->literalTree, h->pending, EOF_SYMBOL);
}

static void huff_trace_block(Huffman *h, int type, int last, int soff, int slen, int opt, int stat, int64_t bit_start) {
    szo_trace *t = h->trace;
    if (!t) return;
    if (t->blk && t->blk_n < t->blk_cap) {
        szo_block_info *b = &t->blk[t->blk_n];
        b->first_token = h->trace_tokens_done; b->ntokens = h->last_lit; b->type = type; b->last = last;
        b->stored_offset = soff; b->stored_len = slen; b->opt_len = opt; b->static_len = stat; b->bit_start = bit_start;
    }
    t->blk_n++;
    h->trace_tokens_done += h->last_lit;
}

static void huff_flush_stored_block(Huffman *h, const uint8_t *stored, int storedOffset, int storedLength, int lastBlock) { /* :766 */
    pend_write_bits(h->pending, (STORED_BLOCK << 1) + (lastBlock ? 1 : 0), 3);
    pend_align(h->pending);
    pend_write_short(h->pending, storedLength);
    pend_write_short(h->pending, ~storedLength);
    pend_write_block(h->pending, stored, storedOffset, storedLength);
    huff_reset(h);
}

static void huff_flush_block(Huffman *h, const uint8_t *stored, int storedOffset, int storedLength, int lastBlock) { /* :788 */
    h->literalTree.freqs[EOF_SYMBOL]++;
    tree_build_tree(&h->literalTree);
    tree_build_tree(&h->distTree);
    tree_calc_bl_freq(&h->literalTree, &h->blTree);
    tree_calc_bl_freq(&h->distTree, &h->blTree);
    tree_build_tree(&h->blTree);

    int blTreeCodes = 4;
    for (int i = 18; i > blTreeCodes; i--)
        if (h->blTree.length[BL_ORDER[i]] > 0) blTreeCodes = i + 1;
    int opt_len = 14 + blTreeCodes * 3 + tree_encoded_length(&h->blTree) + tree_encoded_length(&h->literalTree) +
                  tree_encoded_length(&h->distTree) + h->extra_bits;
    int static_len = h->extra_bits;
    for (int i = 0; i < LITERAL_NUM; i++) static_len += h->literalTree.freqs[i] * staticLLength[i];
    for (int i = 0; i < DIST_NUM; i++) static_len += h->distTree.freqs[i] * staticDLength[i];
    if (opt_len >= static_len) opt_len = static_len;

    int64_t bit_start = h->pending->total_bits;
    if (storedOffset >= 0 && storedLength + 4 < opt_len >> 3) {
        huff_trace_block(h, 0, lastBlock, storedOffset, storedLength, opt_len, static_len, bit_start);
        huff_flush_stored_block(h, stored, storedOffset, storedLength, lastBlock);
    } else if (opt_len == static_len) {
        huff_trace_block(h, 1, lastBlock, storedOffset, storedLength, opt_len, static_len, bit_start);
        pend_write_bits(h->pending, (STATIC_TREES << 1) + (lastBlock ? 1 : 0), 3);
        h->literalTree.codes = staticLCodes; h->literalTree.length = staticLLength;
        h->distTree.codes = staticDCodes; h->distTree.length = staticDLength;
        huff_compress_block(h);
        huff_reset(h);
    } else {
        huff_trace_block(h, 2, lastBlock, storedOffset, storedLength, opt_len, static_len, bit_start);
        pend_write_bits(h->pending, (DYN_TREES << 1) + (lastBlock ? 1 : 0), 3);
        huff_send_all_trees(h, blTreeCodes);
        huff_compress_block(h);
        huff_reset(h);
    }
}
static int huff_is_full(const Huffman *h) { return h->last_lit >= BUFSIZE; } /* :863 */
static void huff_trace_tok(Huffman *h, uint32_t packed) {
    szo_trace *t = h->trace;
    if (!t) return;
    if (t->tok && t->tok_n < t->tok_cap) t->tok[t->tok_n] = packed;
    t->tok_n++;
}
static int huff_tally_lit(Huffman *h, int literal) { /* :873 */
    h->d_buf[h->last_lit] = 0;
    h->l_buf[h->last_lit++] = (uint8_t)literal;
    h->literalTree.freqs[literal]++;
    huff_trace_tok(h, (uint32_t)literal);
    return huff_is_full(h);
}
static int huff_tally_dist(Huffman *h, int distance, int length) { /* :894 */
    h->d_buf[h->last_lit] = (int16_t)distance;
    h->l_buf[h->last_lit++] = (uint8_t)(length - 3);
    int lc = Lcode(length - 3);
    h->literalTree.freqs[lc]++;
    if (lc >= 265 && lc < 285) h->extra_bits += (lc - 261) / 4;
    int dc = Dcode(distance - 1);
    h->distTree.freqs[dc]++;
    if (dc >= 4) h->extra_bits += dc / 2 - 1;
    huff_trace_tok(h, ((uint32_t)distance << 16) | (uint32_t)length);
    return huff_is_full(h);
}

/* ================================================================= DeflaterEngine.cs */
enum { TooFar = 4096 }; /* :51 */
typedef struct {
    int ins_h;
    uint16_t head[HASH_SIZE]; /* C# short[], always read back with & 0xffff */
    uint16_t prev[WSIZE];
    int matchStart, matchLen, prevAvailable, blockStart, strstart, lookahead;
    uint8_t window[2 * WSIZE + 8];
    int strategy;
    int max_chain, max_lazy, niceLength, goodLength, compressionFunction;
    const uint8_t *inputBuf;
    int64_t totalIn;
    int inputOff, inputEnd;
    Pending *pending;
    Huffman huffman;
    int has_adler;
    uint32_t adler;
} Engine;

static void eng_update_hash(Engine *e) { /* :402 */
    e->ins_h = (e->window[e->strstart] << HASH_SHIFT) ^ e->window[e->strstart + 1];
}
static int eng_insert_string(Engine *e) { /* :417 */
    uint16_t match;
    int hash = ((e->ins_h << HASH_SHIFT) ^ e->window[e->strstart + (MIN_MATCH - 1)]) & HASH_MASK;
    e->prev[e->strstart & WMASK] = match = e->head[hash];
    e->head[hash] = (uint16_t)e->strstart;
    e->ins_h = hash;
    return match & 0xffff;
}
static void eng_slide_window(Engine *e) { /* :441 */
    memcpy(e->window, e->window + WSIZE, WSIZE);
    e->matchStart -= WSIZE;
    e->strstart -= WSIZE;
    e->blockStart -= WSIZE;
    for (int i = 0; i < HASH_SIZE; ++i) {
        int m = e->head[i] & 0xffff;
        e->head[i] = (uint16_t)(m >= WSIZE ? (m - WSIZE) : 0);
    }
    for (int i = 0; i < WSIZE; i++) {
        int m = e->prev[i] & 0xffff;
        e->prev[i] = (uint16_t)(m >= WSIZE ? (m - WSIZE) : 0);
    }
}
static void eng_fill_window(Engine *e) { /* :366 */
    if (e->strstart >= WSIZE + MAX_DIST) eng_slide_window(e);
    if (e->lookahead < MIN_LOOKAHEAD && e->inputOff < e->inputEnd) {
        int more = 2 * WSIZE - e->lookahead - e->strstart;
        if (more > e->inputEnd - e->inputOff) more = e->inputEnd - e->inputOff;
        memcpy(e->window + e->strstart + e->lookahead, e->inputBuf + e->inputOff, (size_t)more);
        if (e->has_adler) e->adler = szo_adler32(e->adler, e->inputBuf + e->inputOff, (size_t)more);
        e->inputOff += more;
        e->totalIn += more;
        e->lookahead += more;
    }
    if (e->lookahead >= MIN_MATCH) eng_update_hash(e);
}

static int eng_find_longest_match(Engine *e, int curMatch) { /* :474 */
    int match;
    int scan = e->strstart;
    int scanMax = scan + (MAX_MATCH < e->lookahead ? MAX_MATCH : e->lookahead) - 1;
    int limit = scan - MAX_DIST > 0 ? scan - MAX_DIST : 0;
    const uint8_t *window = e->window;
    const uint16_t *prev = e->prev;
    int chainLength = e->max_chain;
    int niceLength = e->niceLength < e->lookahead ? e->niceLength : e->lookahead;

    e->matchLen = e->matchLen > MIN_MATCH - 1 ? e->matchLen : MIN_MATCH - 1;
    if (scan + e->matchLen > scanMax) return 0;

    uint8_t scan_end1 = window[scan + e->matchLen - 1];
    uint8_t scan_end = window[scan + e->matchLen];
    if (e->matchLen >= e->goodLength) chainLength >>= 2;

    do {
        match = curMatch;
        scan = e->strstart;
        if (window[match + e->matchLen] != scan_end || window[match + e->matchLen - 1] != scan_end1 ||
            window[match] != window[scan] || window[++match] != window[++scan]) {
            continue;
        }
        /* :518-591 — the unrolled switch + 8-way loop is "advance while equal, up to scanMax":
         * after it, scan-strstart = length of the common prefix capped at scanMax+1-strstart. */
        {
            int rem = (scanMax - scan) % 8;
            int ok = 1;
            for (int k = 0; k < rem; k++) {
                if (window[++scan] != window[++match]) { ok = 0; break; }
            }
            if (ok && window[scan] == window[match]) {
                for (;;) {
                    if (scan == scanMax) { ++scan; ++match; break; }
                    int k, brk = 0;
                    for (k = 0; k < 8; k++) {
                        if (window[++scan] != window[++match]) { brk = 1; break; }
                    }
                    if (brk) break;
                }
            }
        }
        if (scan - e->strstart > e->matchLen) {
            e->matchStart = curMatch;
            e->matchLen = scan - e->strstart;
            if (e->matchLen >= niceLength) break;
            scan_end1 = window[scan - 1];
            scan_end = window[scan];
        }
    } while ((curMatch = (prev[curMatch & WMASK] & 0xffff)) > limit && 0 != --chainLength);

    return e->matchLen >= MIN_MATCH;
}

static int eng_deflate_stored(Engine *e, int flush, int finish) { /* :614 */
    if (!flush && e->lookahead == 0) return 0;
    e->strstart += e->lookahead;
    e->lookahead = 0;
    int storedLength = e->strstart - e->blockStart;
    if (storedLength >= MAX_BLOCK_SIZE || (e->blockStart < WSIZE && storedLength >= MAX_DIST) || flush) {
        int lastBlock = finish;
        if (storedLength > MAX_BLOCK_SIZE) { storedLength = MAX_BLOCK_SIZE; lastBlock = 0; }
        int64_t bs = e->pending->total_bits;
        huff_trace_block(&e->huffman, 0, lastBlock, e->blockStart, storedLength, 0, 0, bs);
        huff_flush_stored_block(&e->huffman, e->window, e->blockStart, storedLength, lastBlock);
        e->blockStart += storedLength;
        return !(lastBlock || storedLength == 0);
    }
    return 1;
}

static int eng_deflate_fast(Engine *e, int flush, int finish) { /* :651 */
    if (e->lookahead < MIN_LOOKAHEAD && !flush) return 0;
    while (e->lookahead >= MIN_LOOKAHEAD || flush) {
        if (e->lookahead == 0) {
            huff_flush_block(&e->huffman, e->window, e->blockStart, e->strstart - e->blockStart, finish);
            e->blockStart = e->strstart;
            return 0;
        }
        if (e->strstart > 2 * WSIZE - MIN_LOOKAHEAD) eng_slide_window(e);
        int hashHead;
        if (e->lookahead >= MIN_MATCH && (hashHead = eng_insert_string(e)) != 0 && e->strategy != 2 &&
            e->strstart - hashHead <= MAX_DIST && eng_find_longest_match(e, hashHead)) {
            int full = huff_tally_dist(&e->huffman, e->strstart - e->matchStart, e->matchLen);
            e->lookahead -= e->matchLen;
            if (e->matchLen <= e->max_lazy && e->lookahead >= MIN_MATCH) {
                while (--e->matchLen > 0) { ++e->strstart; eng_insert_string(e); }
                ++e->strstart;
            } else {
                e->strstart += e->matchLen;
                if (e->lookahead >= MIN_MATCH - 1) eng_update_hash(e);
            }
            e->matchLen = MIN_MATCH - 1;
            if (!full) continue;
        } else {
            huff_tally_lit(&e->huffman, e->window[e->strstart] & 0xff);
            ++e->strstart;
            --e->lookahead;
        }
        if (huff_is_full(&e->huffman)) {
            int lastBlock = finish && (e->lookahead == 0);
            huff_flush_block(&e->huffman, e->window, e->blockStart, e->strstart - e->blockStart, lastBlock);
            e->blockStart = e->strstart;
            return !lastBlock;
        }
    }
    return 1;
}

static int eng_deflate_slow(Engine *e, int flush, int finish) { /* :741 */
    if (e->lookahead < MIN_LOOKAHEAD && !flush) return 0;
    while (e->lookahead >= MIN_LOOKAHEAD || flush) {
        if (e->lookahead == 0) {
            if (e->prevAvailable) huff_tally_lit(&e->huffman, e->window[e->strstart - 1] & 0xff);
            e->prevAvailable = 0;
            huff_flush_block(&e->huffman, e->window, e->blockStart, e->strstart - e->blockStart, finish);
            e->blockStart = e->strstart;
            return 0;
        }
        if (e->strstart >= 2 * WSIZE - MIN_LOOKAHEAD) eng_slide_window(e);

        int prevMatch = e->matchStart;
        int prevLen = e->matchLen;
        if (e->lookahead >= MIN_MATCH) {
            int hashHead = eng_insert_string(e);
            if (e->strategy != 2 && hashHead != 0 && e->strstart - hashHead <= MAX_DIST &&
                eng_find_longest_match(e, hashHead)) {
                if (e->matchLen <= 5 &&
                    (e->strategy == 1 || (e->matchLen == MIN_MATCH && e->strstart - e->matchStart > TooFar))) {
                    e->matchLen = MIN_MATCH - 1;
                }
            }
        }
        if (prevLen >= MIN_MATCH && e->matchLen <= prevLen) {
            huff_tally_dist(&e->huffman, e->strstart - 1 - prevMatch, prevLen);
            prevLen -= 2;
            do {
                e->strstart++;
                e->lookahead--;
                if (e->lookahead >= MIN_MATCH) eng_insert_string(e);
            } while (--prevLen > 0);
            e->strstart++;
            e->lookahead--;
            e->prevAvailable = 0;
            e->matchLen = MIN_MATCH - 1;
        } else {
            if (e->prevAvailable) huff_tally_lit(&e->huffman, e->window[e->strstart - 1] & 0xff);
            e->prevAvailable = 1;
            e->strstart++;
            e->lookahead--;
        }
        if (huff_is_full(&e->huffman)) {
            int len = e->strstart - e->blockStart;
            if (e->prevAvailable) len--;
            int lastBlock = (finish && (e->lookahead == 0) && !e->prevAvailable);
            huff_flush_block(&e->huffman, e->window, e->blockStart, len, lastBlock);
            e->blockStart += len;
            return !lastBlock;
        }
    }
    return 1;
}

static int eng_deflate(Engine *e, int flush, int finish) { /* :104 */
    int progress;
    do {
        eng_fill_window(e);
        int canFlush = flush && (e->inputOff == e->inputEnd);
        switch (e->compressionFunction) {
        case DEFLATE_STORED: progress = eng_deflate_stored(e, canFlush, finish); break;
        case DEFLATE_FAST: progress = eng_deflate_fast(e, canFlush, finish); break;
        default: progress = eng_deflate_slow(e, canFlush, finish); break;
        }
    } while (pend_is_flushed(e->pending) && progress);
    return progress;
}

static void eng_init(Engine *e, Pending *p, int noAdler) { /* ctor :80-94 */
    memset(e, 0, sizeof(*e));
    e->pending = p;
    huff_init(&e->huffman, p);
    e->has_adler = !noAdler;
    e->adler = 1;
    e->blockStart = e->strstart = 1;
}
static void eng_reset(Engine *e) { /* :234 */
    huff_reset(&e->huffman);
    e->adler = 1;
    e->blockStart = e->strstart = 1;
    e->lookahead = 0;
    e->totalIn = 0;
    e->prevAvailable = 0;
    e->matchLen = MIN_MATCH - 1;
    memset(e->head, 0, sizeof(e->head));
    memset(e->prev, 0, sizeof(e->prev));
}
static void eng_set_dictionary(Engine *e, const uint8_t *buffer, int offset, int length) { /* :198 */
    if (e->has_adler) e->adler = szo_adler32(e->adler, buffer + offset, (size_t)length);
    if (length < MIN_MATCH) return;
    if (length > MAX_DIST) { offset += length - MAX_DIST; length = MAX_DIST; }
    memcpy(e->window + e->strstart, buffer + offset, (size_t)length);
    eng_update_hash(e);
    --length;
    while (--length > 0) { eng_insert_string(e); e->strstart++; }
    e->strstart += 2;
    e->blockStart = e->strstart;
}
static void eng_set_level(Engine *e, int level) { /* :304 */
    e->goodLength = GOOD_LENGTH[level];
    e->max_lazy = MAX_LAZY[level];
    e->niceLength = NICE_LENGTH[level];
    e->max_chain = MAX_CHAIN[level];
    if (COMPR_FUNC[level] != e->compressionFunction) {
        switch (e->compressionFunction) {
        case DEFLATE_STORED:
            if (e->strstart > e->blockStart) {
                int64_t bs = e->pending->total_bits;
                huff_trace_block(&e->huffman, 0, 0, e->blockStart, e->strstart - e->blockStart, 0, 0, bs);
                huff_flush_stored_block(&e->huffman, e->window, e->blockStart, e->strstart - e->blockStart, 0);
                e->blockStart = e->strstart;
            }
            eng_update_hash(e);
            break;
        case DEFLATE_FAST:
            if (e->strstart > e->blockStart) {
                huff_flush_block(&e->huffman, e->window, e->blockStart, e->strstart - e->blockStart, 0);
                e->blockStart = e->strstart;
            }
            break;
        case DEFLATE_SLOW:
            if (e->prevAvailable) huff_tally_lit(&e->huffman, e->window[e->strstart - 1] & 0xff);
            if (e->strstart > e->blockStart) {
                huff_flush_block(&e->huffman, e->window, e->blockStart, e->strstart - e->blockStart, 0);
                e->blockStart = e->strstart;
            }
            e->prevAvailable = 0;
            e->matchLen = MIN_MATCH - 1;
            break;
        }
        e->compressionFunction = COMPR_FUNC[level];
    }
}

/* ================================================================= Deflater.cs */
enum { IS_SETDICT = 0x01, IS_FLUSHING = 0x04, IS_FINISHING = 0x08, INIT_STATE = 0x00, SETDICT_STATE = 0x01,
       BUSY_STATE = 0x10, FLUSHING_STATE = 0x14, FINISHING_STATE = 0x1c, FINISHED_STATE = 0x1e, CLOSED_STATE = 0x7f };

struct szo_deflater {
    int level, noZlibHeaderOrFooter, state;
    int64_t totalOut;
    Pending pending;
    Engine engine;
};

void szo_deflater_reset(szo_deflater *d) { /* :204 */
    d->state = d->noZlibHeaderOrFooter ? BUSY_STATE : INIT_STATE;
    d->totalOut = 0;
    pend_reset(&d->pending);
    eng_reset(&d->engine);
    d->engine.huffman.trace_tokens_done = 0;
}
int szo_deflater_set_level(szo_deflater *d, int level) { /* :349 */
    if (level == -1) level = 6;
    else if (level < 0 || level > 9) return SZO_ERR_ARG;
    if (d->level != level) { d->level = level; eng_set_level(&d->engine, level); }
    return 0;
}
szo_deflater *szo_deflater_new(int level, int nowrap) { /* :178 */
    if (level == -1) level = 6;
    else if (level < 0 || level > 9) return NULL;
    szo_deflater *d = (szo_deflater *)calloc(1, sizeof(*d));
    if (!d) return NULL;
    d->pending.bits = 0;
    eng_init(&d->engine, &d->pending, nowrap);
    d->noZlibHeaderOrFooter = nowrap;
    d->engine.strategy = 0;
    d->level = 0; /* C# field default */
    szo_deflater_set_level(d, level);
    szo_deflater_reset(d);
    return d;
}
void szo_deflater_free(szo_deflater *d) { free(d); }
void szo_deflater_set_strategy(szo_deflater *d, int s) { d->engine.strategy = s; } /* :385 */
void szo_deflater_set_trace(szo_deflater *d, szo_trace *t) { d->engine.huffman.trace = t; }
int szo_deflater_set_dictionary(szo_deflater *d, const uint8_t *p, int n) { /* :559 */
    if (d->state != INIT_STATE) return SZO_ERR_STATE;
    d->state = SETDICT_STATE;
    eng_set_dictionary(&d->engine, p, 0, n);
    return 0;
}
int szo_deflater_set_input(szo_deflater *d, const uint8_t *p, int n) { /* :331 + Engine.SetInput :146 */
    if ((d->state & IS_FINISHING) != 0) return SZO_ERR_STATE;
    if (n < 0) return SZO_ERR_ARG;
    if (d->engine.inputOff < d->engine.inputEnd) return SZO_ERR_STATE;
    d->engine.inputBuf = p; d->engine.inputOff = 0; d->engine.inputEnd = n;
    return 0;
}
void szo_deflater_flush(szo_deflater *d) { d->state |= IS_FLUSHING; }                    /* :252 */
void szo_deflater_finish(szo_deflater *d) { d->state |= (IS_FLUSHING | IS_FINISHING); }  /* :262 */
int szo_deflater_is_finished(const szo_deflater *d) { return d->state == FINISHED_STATE && pend_is_flushed(&d->pending); } /* :271 */
int szo_deflater_needs_input(const szo_deflater *d) { return d->engine.inputEnd == d->engine.inputOff; } /* :285 */
int64_t szo_deflater_total_in(const szo_deflater *d) { return d->engine.totalIn; }
int64_t szo_deflater_total_out(const szo_deflater *d) { return d->totalOut; }
uint32_t szo_deflater_adler(const szo_deflater *d) { return d->engine.has_adler ? d->engine.adler : 0; }

int szo_deflater_deflate(szo_deflater *d, uint8_t *output, int length) { /* :427 */
    int offset = 0;
    int origLength = length;
    if (d->state == CLOSED_STATE) return SZO_ERR_STATE;
    if (d->state < BUSY_STATE) {
        int header = (8 + ((MAX_WBITS - 8) << 4)) << 8;
        int level_flags = (d->level - 1) >> 1;
        if (level_flags < 0 || level_flags > 3) level_flags = 3;
        header |= level_flags << 6;
        if ((d->state & IS_SETDICT) != 0) header |= PRESET_DICT;
        header += 31 - (header % 31);
        pend_write_short_msb(&d->pending, header);
        if ((d->state & IS_SETDICT) != 0) {
            int chksum = (int)d->engine.adler;
            d->engine.adler = 1;
            pend_write_short_msb(&d->pending, chksum >> 16);
            pend_write_short_msb(&d->pending, chksum & 0xffff);
        }
        d->state = BUSY_STATE | (d->state & (IS_FLUSHING | IS_FINISHING));
    }
    for (;;) {
        int count = pend_flush(&d->pending, output, offset, length);
        offset += count;
        d->totalOut += count;
        length -= count;
        if (length == 0 || d->state == FINISHED_STATE) break;
        if (!eng_deflate(&d->engine, (d->state & IS_FLUSHING) != 0, (d->state & IS_FINISHING) != 0)) {
            switch (d->state) {
            case BUSY_STATE:
                return origLength - length;
            case FLUSHING_STATE:
                if (d->level != 0) {
                    int neededbits = 8 + ((-d->pending.bitCount) & 7);
                    while (neededbits > 0) {
                        pend_write_bits(&d->pending, 2, 10);
                        neededbits -= 10;
                    }
                }
                d->state = BUSY_STATE;
                break;
            case FINISHING_STATE:
                pend_align(&d->pending);
                if (!d->noZlibHeaderOrFooter) {
                    int adler = (int)d->engine.adler;
                    pend_write_short_msb(&d->pending, adler >> 16);
                    pend_write_short_msb(&d->pending, adler & 0xffff);
                }
                d->state = FINISHED_STATE;
                break;
            }
        }
    }
    return origLength - length;
}

/* ---- convenience driver mirroring DeflaterOutputStream.Write/Flush/Finish (CS/DeflaterOutputStream.cs:506,388,100) */
int64_t szo_deflate_oneshot(const uint8_t *in, size_t n, int level, int nowrap, int strategy,
                            int flush_before_finish, uint8_t *out, size_t out_cap, szo_trace *trace) {
    szo_deflater *d = szo_deflater_new(level, nowrap);
    if (!d) return SZO_ERR_ARG;
    szo_deflater_set_strategy(d, strategy);
    szo_deflater_set_trace(d, trace);
    size_t op = 0, ip = 0;
    int64_t rc = 0;
    uint8_t buf[4096];
    do { /* Write in <=1 GiB pieces (C# arrays < 2 GiB) */
        size_t chunk = n - ip > ((size_t)1 << 30) ? ((size_t)1 << 30) : n - ip;
        szo_deflater_set_input(d, in + ip, (int)chunk);
        ip += chunk;
        while (!szo_deflater_needs_input(d)) {
            int k = szo_deflater_deflate(d, buf, (int)sizeof buf);
            if (k <= 0) break;
            if (op + (size_t)k > out_cap) { rc = -100; goto done; }
            memcpy(out + op, buf, (size_t)k); op += (size_t)k;
        }
    } while (ip < n);
    if (flush_before_finish) {
        szo_deflater_flush(d);
        for (;;) {
            int k = szo_deflater_deflate(d, buf, (int)sizeof buf);
            if (k <= 0) break;
            if (op + (size_t)k > out_cap) { rc = -100; goto done; }
            memcpy(out + op, buf, (size_t)k); op += (size_t)k;
        }
    }
    szo_deflater_finish(d);
    while (!szo_deflater_is_finished(d)) {
        int k = szo_deflater_deflate(d, buf, (int)sizeof buf);
        if (k <= 0) break;
        if (op + (size_t)k > out_cap) { rc = -100; goto done; }
        memcpy(out + op, buf, (size_t)k); op += (size_t)k;
    }
    rc = szo_deflater_is_finished(d) ? (int64_t)op : -101;
done:
    szo_deflater_free(d);
    return rc;
}
