/* CPU model of the block finder of the chunk-parallel inflate (k_find_blocks): how often does a complete, consistent dynamic block
 * header parse at a bit position that is NOT a block start, and what does that do to members of ~1 MiB cut into 16 / 32 / 64 KiB chunks?
 *   gcc -O2 -o /tmp/finder_model tools/finder_model.c -Ioracle -Loracle -lszl_oracle -Wl,-rpath,$PWD/oracle && /tmp/finder_model [members] [KiB]
 * Input: the corpus generator's text is not linked here — the program compresses pseudo-text of its own (word soup with a Zipf-like
 * vocabulary) with the oracle at level 6 and uses the oracle's trace for the true block starts. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "szl_oracle.h"

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 32); }

static void make_text(uint8_t *d, size_t n) {
    static char vocab[4096][12]; static int init = 0;
    if (!init) { init = 1; for (int i = 0; i < 4096; i++) { int l = 2 + rnd() % 9; for (int k = 0; k < l; k++) vocab[i][k] = "etaoinshrdlucmfwypvbgkqjxz"[(rnd() % 26) * (rnd() % 26) / 26]; vocab[i][l] = 0; } }
    size_t p = 0;
    while (p < n) {
        uint32_t r = rnd(); int w = (int)((uint64_t)(r % 4096) * (rnd() % 4096) / 4096);   /* skewed */
        const char *s = vocab[w];
        while (*s && p < n) d[p++] = (uint8_t)*s++;
        if (p < n) d[p++] = (rnd() % 17 == 0) ? '\n' : ' ';
    }
}

static uint64_t bits_at(const uint8_t *in, uint64_t len, uint64_t bp) {
    uint64_t v = 0; uint64_t b = bp >> 3;
    for (int k = 0; k < 9; k++) { uint64_t by = b + k < len ? in[b + k] : 0; int pos = 8 * k - (int)(bp & 7); if (pos >= 0 && pos < 64) v |= by << pos; else if (pos < 0) v |= by >> (-pos); }
    return v;
}
static const int ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
static uint32_t bitrev(uint32_t v, int l) { uint32_t r = 0; for (int i = 0; i < l; i++) if (v >> i & 1) r |= 1u << (l - 1 - i); return r; }

/* the device's header_ok: 0 = no, 1 = complete and consistent; *syms_out = code-length symbols decoded before the verdict */
static int header_ok(const uint8_t *in, uint64_t len, uint64_t p, int *syms_out) {
    uint64_t bp = p, w = bits_at(in, len, bp);
    *syms_out = 0;
    if ((w & 7) != 4) return 0;
    uint32_t nl = ((w >> 3) & 31) + 257, nd = ((w >> 8) & 31) + 1, nm = ((w >> 13) & 15) + 4;
    if (nl > 286 || nd > 30) return 0;
    bp += 17; w = bits_at(in, len, bp);
    int ml[19] = {0}, kraft = 0;
    for (uint32_t i = 0; i < nm; i++) { int l = (int)(w >> (3 * i)) & 7; ml[ORDER[i]] = l; if (l) kraft += 128 >> l; }
    if (kraft != 128) return 0;
    bp += 3 * nm;
    if (bp + 64 > len * 8) return 0;
    uint16_t mlut[128] = {0}; int code = 0;
    for (int l = 1; l < 8; l++) { for (int i = 0; i < 19; i++) { if (ml[i] != l) continue; uint32_t rev = bitrev((uint32_t)code++, l); for (uint32_t j = rev; j < 128; j += 1u << l) mlut[j] = (uint16_t)((i << 4) | l); } code <<= 1; }
    uint8_t lens[320]; uint32_t idx = 0, total = nl + nd; int kl = 0, kd = 0, ndist = 0;
    while (idx < total) {
        if (bp + 16 > len * 8) return 0;
        w = bits_at(in, len, bp);
        uint32_t e = mlut[w & 127]; if (!e) return 0;
        uint32_t sl = e & 15, sym = e >> 4; bp += sl; w >>= sl; (*syms_out)++;
        uint32_t rep = 1, val = sym;
        if (sym == 16) { if (!idx) return 0; val = lens[idx - 1]; rep = 3 + (w & 3); bp += 2; }
        else if (sym == 17) { val = 0; rep = 3 + (w & 7); bp += 3; }
        else if (sym == 18) { val = 0; rep = 11 + (w & 127); bp += 7; }
        if (idx + rep > total) return 0;
        for (uint32_t r = 0; r < rep; r++, idx++) { lens[idx] = (uint8_t)val; if (val) { if (idx < nl) kl += 32768 >> val; else { kd += 32768 >> val; ndist++; } } }
        if (kl > 32768 || kd > 32768) return 0;
    }
    if (!lens[256] || kl != 32768 || !(kd == 32768 || ndist <= 1)) return 0;
    return 1;
}

int main(int argc, char **argv) {
    int members = argc > 1 ? atoi(argv[1]) : 64; size_t msz = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 10;
    uint8_t *d = malloc(msz), *z = malloc(msz + 65536);
    szo_block_info *blk = malloc(sizeof(szo_block_info) * 4096);
    uint64_t cand_total = 0, attempts_syms = 0, bits_total = 0, false_valid = 0, true_found = 0, true_dyn = 0;
    int fail16 = 0, fail32 = 0, fail64 = 0, nostart[3] = {0, 0, 0}, nchunks[3] = {0, 0, 0};
    for (int m = 0; m < members; m++) {
        make_text(d, msz);
        szo_trace tr; memset(&tr, 0, sizeof tr); tr.blk = blk; tr.blk_cap = 4096;
        int64_t zl = szo_deflate_oneshot(d, msz, 6, 1, 0, 0, z, msz + 65536, &tr);
        if (zl < 0) return 1;
        uint64_t nbits = (uint64_t)zl * 8; bits_total += nbits;
        /* every bit position: the cheap tests, then the full parse */
        uint8_t *valid = calloc((size_t)zl * 8 + 8, 1);
        for (uint64_t p = 0; p + 128 < nbits; p++) {
            uint64_t w = bits_at(z, (uint64_t)zl, p);
            if ((w & 7) != 4 || ((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;
            uint32_t nm = (uint32_t)((w >> 13) & 15) + 4; uint64_t mw = bits_at(z, (uint64_t)zl, p + 17); int kraft = 0;
            for (uint32_t i = 0; i < nm; i++) { int l = (int)(mw >> (3 * i)) & 7; if (l) kraft += 128 >> l; }
            if (kraft != 128) continue;
            cand_total++;
            int syms; if (header_ok(z, (uint64_t)zl, p, &syms)) valid[p] = 1;
            attempts_syms += (uint64_t)syms;
        }
        uint8_t *truth = calloc((size_t)zl * 8 + 8, 1);
        for (size_t b = 0; b < tr.blk_n; b++) { truth[blk[b].bit_start] = 1; if (blk[b].type == 2 && !blk[b].last) true_dyn++; }
        for (uint64_t p = 0; p < nbits; p++) if (valid[p]) { if (truth[p]) true_found++; else false_valid++; }
        /* chunking: first valid header at or behind every chunk start; the member fails the single pass if any of those is not a block start */
        const uint64_t cs[3] = {16384, 32768, 65536};
        for (int c = 0; c < 3; c++) {
            int bad = 0;
            for (uint64_t s0 = cs[c] * 8; s0 < nbits; s0 += cs[c] * 8) {
                uint64_t p = s0, e = s0 + cs[c] * 8 < nbits ? s0 + cs[c] * 8 : nbits;
                while (p < e && !valid[p]) p++;
                nchunks[c]++;
                if (p >= e) { nostart[c]++; continue; }
                if (!truth[p]) bad = 1;
            }
            if (c == 0) fail16 += bad; else if (c == 1) fail32 += bad; else fail64 += bad;
        }
        free(valid); free(truth);
    }
    printf("%d members of %zu KiB: %.1f MiB compressed, %llu dynamic non-final blocks (%.1f KB each)\n", members, msz >> 10, bits_total / 8.0 / 1048576,
           (unsigned long long)true_dyn, bits_total / 8.0 / (double)(true_dyn ? true_dyn : 1) / 1000);
    printf("cheap tests pass at 1 of %.0f bit positions; a full parse decodes %.1f code-length symbols on average before its verdict\n",
           (double)bits_total / (double)cand_total, (double)attempts_syms / (double)cand_total);
    printf("complete + consistent headers: %llu at true block starts, %llu elsewhere (1 per %.1f MiB of compressed data)\n", (unsigned long long)true_found,
           (unsigned long long)false_valid, false_valid ? bits_total / 8.0 / 1048576 / (double)false_valid : 0.0);
    printf("members whose chunk starts include a false one:  16 KiB chunks %d, 32 KiB %d, 64 KiB %d of %d\n", fail16, fail32, fail64, members);
    printf("chunks without any start: 16 KiB %d of %d, 32 KiB %d of %d, 64 KiB %d of %d\n", nostart[0], nchunks[0], nostart[1], nchunks[1], nostart[2], nchunks[2]);
    return 0;
}
