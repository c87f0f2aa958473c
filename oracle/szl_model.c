/*
 * szl_model.c — CPU model of the parallel decomposition of DeflateSlow (see szl_model.h).
 * TEST INFRASTRUCTURE ONLY.  Every rule cites the reference line it restates
 * (C/ = /root/reference/src/ICSharpCode.SharpZipLib/Zip/Compression/).
 */
#include "szl_model.h"
#include <stdlib.h>
#include <string.h>

enum { WSIZE = 32768, MAX_DIST = 32506, MAX_MATCH = 258, MIN_MATCH = 3, TOO_FAR = 4096, BLOCK_TOKENS = 16384 };

int szm_level_params(int level, szm_params *out) { /* C/DeflaterConstants.cs:124-144 */
    static const int GOOD[10] = {0, 4, 4, 4, 4, 8, 8, 8, 32, 32};
    static const int NICE[10] = {0, 8, 16, 32, 16, 32, 128, 128, 258, 258};
    static const int CHAIN[10] = {0, 4, 8, 32, 16, 32, 128, 256, 1024, 4096};
    if (level == -1) level = 6;
    if (level < 5 || level > 9) return -1;
    out->good = GOOD[level]; out->nice = NICE[level]; out->max_chain = CHAIN[level]; out->strategy = 0;
    return 0;
}

/* Window base for an iteration that starts at absolute position s: the engine slides (base += 32768)
 * whenever an iteration starts at window index >= 65274 (C/DeflaterEngine.cs:371, :771-778; the
 * window index of absolute position a is a + 1 - base because strstart starts at 1, :93). */
int64_t szm_base_of(int64_t s) {
    int64_t idx = s + 1;
    if (idx <= 65273) return 0;
    return ((idx - 65273 + 32767) / 32768) * 32768;
}

static inline uint32_t hash3(const uint8_t *d, size_t q) { /* UpdateHash+InsertString :402-419 */
    return (((uint32_t)d[q] << 10) ^ ((uint32_t)d[q + 1] << 5) ^ d[q + 2]) & 0x7FFF;
}

void szm_links(const uint8_t *d, size_t n, const size_t *seg_ends, size_t nseg, uint16_t *link) {
    int64_t *head = (int64_t *)malloc(sizeof(int64_t) * 32768);
    for (int i = 0; i < 32768; i++) head[i] = -1;
    size_t seg = 0;
    for (size_t q = 0; q < n; q++) {
        while (seg < nseg && q >= seg_ends[seg]) seg++;
        size_t e = seg < nseg ? seg_ends[seg] : n;
        link[q] = 0;
        if (e - q < 3) continue; /* InsertString only runs while lookahead >= MIN_MATCH (:780, :817) */
        uint32_t h = hash3(d, q);
        int64_t r = head[h];
        if (r >= 0 && (int64_t)q - r <= 32767) link[q] = (uint16_t)((int64_t)q - r);
        head[h] = (int64_t)q;
    }
    free(head);
}

static inline int lcp_cap(const uint8_t *d, size_t c, size_t p, int cap) {
    int l = 0;
    while (l < cap && d[c + l] == d[p + l]) l++;
    return l;
}

/* One FindLongestMatch walk (:474-612) at position p, inside the segment ending at seg_end.
 *   best0      = initial matchLen (already max(prevLen,2))
 *   budget     = chainLength after the goodLength quartering
 *   snap_at    = if >0, *snap receives the state after that many candidates (quarter-budget result)
 * Returns final (len | dist<<16) if improved beyond best0, else 0. */
static uint32_t flm_walk(const uint8_t *d, size_t p, size_t seg_end, const uint16_t *link, const szm_params *P,
                         int best0, int budget, int snap_at, uint32_t *snap) {
    if (snap) *snap = 0;
    size_t rem = seg_end - p;
    if (rem < MIN_MATCH) return 0; /* :780 */
    if (P->strategy == 2) return 0; /* HuffmanOnly :786 */
    uint32_t l0 = link[p];
    if (l0 == 0) return 0; /* hashHead == 0 */
    int64_t base = szm_base_of((int64_t)p);
    int64_t idx_p = (int64_t)p + 1 - base;
    int64_t c = (int64_t)p - l0;
    if ((int64_t)p - c > MAX_DIST) return 0;            /* strstart - hashHead <= MAX_DIST :788 */
    if (c + 1 - base < 1) return 0;                      /* hashHead != 0 (entry clamped by a slide :450-461) */
    int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;    /* scanMax :479 */
    int nice = rem < (size_t)P->nice ? (int)rem : P->nice;  /* :485 */
    int best = best0;
    if (best >= cap) return 0;                           /* scan + matchLen > scanMax :489 */
    int64_t limit_idx = idx_p - MAX_DIST > 0 ? idx_p - MAX_DIST : 0; /* :480 */
    uint32_t res = 0;
    int count = 0;
    for (;;) {
        int L = lcp_cap(d, (size_t)c, p, cap);
        if (L > best) {
            best = L;
            res = (uint32_t)L | ((uint32_t)((int64_t)p - c) << 16);
            if (best >= nice) { /* :604 */
                if (snap && snap_at > 0 && count < snap_at) *snap = res;
                return res;
            }
        }
        count++;
        if (snap && count == snap_at) *snap = res;
        uint32_t l = link[c];
        if (l == 0) break;
        int64_t c2 = c - l;
        if (c2 + 1 - base <= limit_idx) break;           /* (curMatch = prev[..]) > limit :609 */
        if (--budget == 0) break;                        /* 0 != --chainLength :609 */
        c = c2;
    }
    if (snap && snap_at > 0 && count < snap_at) *snap = res;
    return res;
}

void szm_match_tables(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link,
                      const szm_params *P, uint32_t *m2, uint32_t *mq) {
    for (size_t p = seg_start; p < seg_end; p++) {
        uint32_t snap = 0;
        m2[p] = flm_walk(d, p, seg_end, link, P, 2, P->max_chain, P->max_chain >> 2, &snap);
        mq[p] = snap;
    }
}

/* --- chain compression (not in the product yet) -------------------------------------------------------------------
 * Once best_len >= 3 only a candidate that shares the position's first FOUR bytes can be strictly longer, and those candidates
 * form a sub-chain of the 3-byte-hash chain.  link4[q] = distance to the previous inserted position with q's four bytes (0 = none
 * within the window), skip4[q] = how many elements of the hash chain that hop passes; a walk that jumps along link4 and charges
 * skip4 against the chain budget examines exactly the candidates of the full walk that could matter, at the same chain indices —
 * so the limit test, the budget (max_chain, :609) and the quarter-budget snapshot (:495) stay exact.  While best_len is still 2
 * (the first candidates were hash collisions) the full chain is walked.  szm_match_tables_c4 must equal szm_match_tables. */
void szm_links4(const uint8_t *d, size_t n, const uint16_t *link, uint16_t *link4, uint16_t *skip4) {
    for (size_t q = 0; q < n; q++) {
        link4[q] = 0; skip4[q] = 0;
        if (q + 4 > n || link[q] == 0) continue;
        size_t c = q; uint32_t hops = 0;
        for (;;) {
            uint32_t l = link[c];
            if (l == 0) break;
            c -= l; hops++;
            if (q - c > 32767 || hops > 65535) break;
            if (memcmp(d + c, d + q, 4) == 0) { link4[q] = (uint16_t)(q - c); skip4[q] = (uint16_t)hops; break; }
        }
    }
}

static uint32_t flm_walk_c4(const uint8_t *d, size_t n, size_t p, size_t seg_end, const uint16_t *link, const uint16_t *link4,
                            const uint16_t *skip4, const szm_params *P, int budget, int snap_at, uint32_t *snap, uint64_t *steps) {
    if (snap) *snap = 0;
    size_t rem = seg_end - p;
    if (rem < MIN_MATCH) return 0;
    if (P->strategy == 2) return 0;
    uint32_t l0 = link[p];
    if (l0 == 0) return 0;
    int64_t base = szm_base_of((int64_t)p);
    int64_t idx_p = (int64_t)p + 1 - base;
    int64_t c = (int64_t)p - l0;
    if ((int64_t)p - c > MAX_DIST) return 0;
    if (c + 1 - base < 1) return 0;
    int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
    int nice = rem < (size_t)P->nice ? (int)rem : P->nice;
    int best = 2;
    if (best >= cap) return 0;
    int64_t limit_idx = idx_p - MAX_DIST > 0 ? idx_p - MAX_DIST : 0;
    const int has4 = p + 4 <= n;                          /* the position has four bytes to share */
    uint32_t res = 0;
    int64_t k = 1;                                        /* index of candidate c in the position's hash chain */
    for (;;) {
        if (steps) (*steps)++;
        int L = lcp_cap(d, (size_t)c, p, cap);
        if (L > best) {
            best = L;
            res = (uint32_t)L | ((uint32_t)((int64_t)p - c) << 16);
            if (best >= nice) { if (snap && snap_at > 0 && k <= snap_at) *snap = res; return res; }
        }
        if (snap && k == snap_at) *snap = res;            /* the quarter-budget walk ends with this candidate */
        /* next candidate that can matter, and its chain index */
        int64_t c2, k2;
        if (best == 2 || !has4) {
            uint32_t l = link[c];
            if (l == 0) break;
            c2 = c - l; k2 = k + 1;
        } else if ((size_t)c + 4 <= n && memcmp(d + c, d + p, 4) == 0) {
            if (link4[c] == 0) break;
            c2 = c - link4[c]; k2 = k + skip4[c];
        } else {                                          /* c is not on the sub-chain: its first element below c, counted from p */
            int64_t y = (int64_t)p, acc = 0;
            int ended = 0;
            do {
                if (link4[y] == 0) { ended = 1; break; }
                acc += skip4[y]; y -= link4[y];
            } while (y >= c);
            if (ended) break;
            c2 = y; k2 = acc;
        }
        if (snap && k < snap_at && k2 > snap_at) *snap = res;    /* the quarter-budget walk ends between the two */
        if (c2 + 1 - base <= limit_idx) break;
        if (k2 > budget) break;
        c = c2; k = k2;
    }
    if (snap && snap_at > 0 && k < snap_at) *snap = res;
    return res;
}

void szm_match_tables_c4(const uint8_t *d, size_t n, size_t seg_start, size_t seg_end, const uint16_t *link, const uint16_t *link4,
                         const uint16_t *skip4, const szm_params *P, uint32_t *m2, uint32_t *mq, uint64_t *steps) {
    for (size_t p = seg_start; p < seg_end; p++) {
        uint32_t snap = 0;
        m2[p] = flm_walk_c4(d, n, p, seg_end, link, link4, skip4, P, P->max_chain, P->max_chain >> 2, &snap, steps);
        mq[p] = snap;
    }
}

/* The same walk in the shape the device kernel runs it (csrc/szl_kernels_match3.hip, k_match6): only link4 / skip4 (saturating at
 * 255: the form is used for max_chain <= 128) are at hand for the chain walk; the 3-byte link is read once, for the first
 * candidate; while best_len is still 2 after it, a slow routine walks the 3-byte chain (global memory on the device) until
 * best_len >= 3 and then re-enters the four-byte sub-chain at its first element below the candidate it stands on. */
static uint32_t flm_walk_k6(const uint8_t *d, size_t n, size_t p, size_t seg_end, const uint16_t *link, const uint16_t *link4,
                            const uint8_t *skip8, const szm_params *P, uint32_t *snap, uint64_t *steps) {
    *snap = 0;
    const int max_chain = P->max_chain, snap_at = P->max_chain >> 2;
    size_t rem = seg_end - p;
    if (rem < MIN_MATCH || P->strategy == 2) return 0;
    uint32_t l3 = link[p];
    if (l3 == 0) return 0;
    int64_t base = szm_base_of((int64_t)p);
    int64_t idx_p = (int64_t)p + 1 - base;
    int64_t c1 = (int64_t)p - l3;
    if ((int64_t)p - c1 > MAX_DIST) return 0;
    if (c1 + 1 - base < 1) return 0;
    int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
    int nice = rem < (size_t)P->nice ? (int)rem : P->nice;
    int best = 2;
    if (best >= cap) return 0;
    int64_t limit_idx = idx_p - MAX_DIST > 0 ? idx_p - MAX_DIST : 0;
    const int has4 = p + 4 <= n;
    uint32_t res = 0, resq = 0;
    /* FETCH: the first candidate goes to VERIFY; the chain position is the candidate itself if it is on the four-byte sub-chain,
     * else the position p (virtual: nothing of the budget is spent yet, the hop from p accounts for c1 as well) */
    int64_t vcl = c1, cl;
    int left;                                            /* max_chain - (chain index of cl) */
    if (has4 && link4[p] != 0 && link4[p] == l3) { cl = c1; left = max_chain - 1; }
    else { cl = (int64_t)p; left = max_chain; }
    int first = 1;
    for (;;) {
        /* VERIFY + COMPLETE of vcl */
        if (steps) (*steps)++;
        int L = lcp_cap(d, (size_t)vcl, p, cap);
        if (L > best) {
            best = L;
            res = (uint32_t)L | ((uint32_t)((int64_t)p - vcl) << 16);
            if ((first ? max_chain - 1 : left) >= max_chain - snap_at) resq = res;    /* chain index <= snap_at */
            if (best >= nice) { *snap = resq; return res; }
        }
        if (best == 2 || !has4) {
            /* SLOW: the 3-byte chain from the candidate just examined */
            int64_t cur3 = vcl;
            int k3 = first ? 1 : max_chain - left;
            int done = 0;
            for (;;) {
                uint32_t l = link[cur3];
                if (l == 0) { done = 1; break; }
                int64_t nx = cur3 - l;
                if (nx + 1 - base <= limit_idx) { done = 1; break; }
                if (k3 + 1 > max_chain) { done = 1; break; }
                cur3 = nx; k3++;
                if (steps) (*steps)++;
                int L2 = lcp_cap(d, (size_t)cur3, p, cap);
                if (L2 > best) {
                    best = L2;
                    res = (uint32_t)L2 | ((uint32_t)((int64_t)p - cur3) << 16);
                    if (k3 <= snap_at) resq = res;
                    if (best >= nice) { *snap = resq; return res; }
                }
                if (best >= 3 && has4) break;
            }
            if (done) break;
            /* re-enter the sub-chain: its first element below cur3, with its chain index */
            int64_t y = (int64_t)p; int acc = 0, ended = 0;
            do {
                if (link4[y] == 0) { ended = 1; break; }
                acc += skip8[y]; y -= link4[y];
                if (acc > max_chain) { ended = 1; break; }       /* (saturated hop counts only ever end the walk) */
            } while (y >= cur3);
            if (ended) break;
            if (y + 1 - base <= limit_idx) break;
            cl = y; left = max_chain - acc;
            first = 0;
        } else {
            first = 0;
            /* advance along the sub-chain from cl */
            if (link4[cl] == 0) break;
            int64_t nx = cl - link4[cl];
            left -= skip8[cl];
            if (left < 0) break;
            if (nx + 1 - base <= limit_idx) break;
            cl = nx;
        }
        /* QUICK: test candidates until one passes the scan_end / scan_end1 filter */
        for (;;) {
            if (d[cl + best] == d[p + best] && d[cl + best - 1] == d[p + best - 1]) break;
            if (steps) (*steps)++;
            if (link4[cl] == 0) { cl = -1; break; }
            int64_t nx = cl - link4[cl];
            left -= skip8[cl];
            if (left < 0) { cl = -1; break; }
            if (nx + 1 - base <= limit_idx) { cl = -1; break; }
            cl = nx;
        }
        if (cl < 0) break;
        vcl = cl;
    }
    *snap = resq;
    return res;
}

void szm_match_tables_k6(const uint8_t *d, size_t n, size_t seg_start, size_t seg_end, const uint16_t *link, const uint16_t *link4,
                         const uint16_t *skip4, const szm_params *P, uint32_t *m2, uint32_t *mq, uint64_t *steps) {
    uint8_t *sk = (uint8_t *)malloc(n + 8);
    for (size_t i = 0; i < n; i++) sk[i] = skip4[i] > 255 ? 255 : (uint8_t)skip4[i];
    for (size_t p = seg_start; p < seg_end; p++) {
        uint32_t snap = 0;
        m2[p] = flm_walk_k6(d, n, p, seg_end, link, link4, sk, P, &snap, steps);
        mq[p] = snap;
    }
    free(sk);
}

/* The same without a slow routine (device form k_match6).  While best_len is 2 a candidate changes the walk's state only if its
 * first three bytes equal the position's (a hash collision just uses up budget), so the first candidate that matters is the first
 * chain element with the same three bytes: e3 (distance e3d, chain index e3h), found by the pass that builds link4.  The walk
 * compares e3 and then follows the four-byte sub-chain: from e3 itself if it is its first element, else from p. */
void szm_links4e(const uint8_t *d, size_t n, const uint16_t *link, uint16_t *link4, uint8_t *skip8, uint16_t *e3d, uint8_t *e3h, int dist_cap) {
    for (size_t q = 0; q < n; q++) {
        link4[q] = 0; skip8[q] = 0; e3d[q] = 0; e3h[q] = 0;
        if (q + 3 > n || link[q] == 0) continue;
        const int has4 = q + 4 <= n;
        size_t c = q;
        for (uint32_t hops = 1; hops <= 255; hops++) {
            uint32_t l = link[c];
            if (l == 0) break;
            c -= l;
            if (q - c > (size_t)dist_cap) break;
            if (memcmp(d + c, d + q, 3) != 0) continue;
            if (e3d[q] == 0) { e3d[q] = (uint16_t)(q - c); e3h[q] = (uint8_t)hops; }
            if (!has4) break;
            if (d[c + 3] == d[q + 3]) { link4[q] = (uint16_t)(q - c); skip8[q] = (uint8_t)hops; break; }
        }
    }
}

static uint32_t flm_walk_k7(const uint8_t *d, size_t p, size_t seg_end, const uint16_t *link, const uint16_t *link4, const uint8_t *skip8,
                            const uint16_t *e3d, const uint8_t *e3h, const szm_params *P, uint32_t *snap, uint64_t *steps) {
    *snap = 0;
    const int max_chain = P->max_chain, snap_at = P->max_chain >> 2;
    size_t rem = seg_end - p;
    if (rem < MIN_MATCH || P->strategy == 2) return 0;
    uint32_t l3 = link[p];
    if (l3 == 0) return 0;
    int64_t base = szm_base_of((int64_t)p);
    int64_t idx_p = (int64_t)p + 1 - base;
    int64_t c1 = (int64_t)p - l3;
    if ((int64_t)p - c1 > MAX_DIST) return 0;                 /* :788 */
    if (c1 + 1 - base < 1) return 0;
    int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
    int nice = rem < (size_t)P->nice ? (int)rem : P->nice;
    int64_t limit_idx = idx_p - MAX_DIST > 0 ? idx_p - MAX_DIST : 0;
    /* FETCH */
    if (e3d[p] == 0 || e3h[p] > max_chain) return 0;
    int64_t vcl = (int64_t)p - e3d[p];
    if (e3d[p] != l3 && vcl + 1 - base <= limit_idx) return 0;  /* :609 for every candidate but the first */
    int best = 2;
    int64_t cl; int left = max_chain - e3h[p], kadj;
    if (link4[p] == e3d[p]) { cl = vcl; kadj = 0; } else { cl = (int64_t)p; kadj = e3h[p]; }
    uint32_t res = 0, resq = 0;
    for (;;) {
        if (steps) (*steps)++;
        int L = lcp_cap(d, (size_t)vcl, p, cap);
        if (L > best) {
            best = L;
            res = (uint32_t)L | ((uint32_t)((int64_t)p - vcl) << 16);
            if (left >= max_chain - snap_at) resq = res;
            if (best >= nice) break;
        }
        /* advance along the sub-chain from cl */
        if (link4[cl] == 0) break;
        int64_t nx = cl - link4[cl];
        if (nx + 1 - base <= limit_idx) break;
        left += kadj; kadj = 0;
        left -= skip8[cl];
        if (left < 0) break;
        cl = nx;
        for (;;) { /* QUICK */
            if (d[cl + best] == d[p + best] && d[cl + best - 1] == d[p + best - 1]) break;
            if (steps) (*steps)++;
            if (link4[cl] == 0) { cl = -1; break; }
            nx = cl - link4[cl];
            if (nx + 1 - base <= limit_idx) { cl = -1; break; }
            left -= skip8[cl];
            if (left < 0) { cl = -1; break; }
            cl = nx;
        }
        if (cl < 0) break;
        vcl = cl;
    }
    *snap = resq;
    return res;
}

void szm_match_tables_k7(const uint8_t *d, size_t n, size_t seg_start, size_t seg_end, const uint16_t *link, int dist_cap,
                         const szm_params *P, uint32_t *m2, uint32_t *mq, uint64_t *steps) {
    uint16_t *l4 = (uint16_t *)malloc((n + 8) * 2), *e3d = (uint16_t *)malloc((n + 8) * 2);
    uint8_t *s8 = (uint8_t *)malloc(n + 8), *e3h = (uint8_t *)malloc(n + 8);
    szm_links4e(d, n, link, l4, s8, e3d, e3h, dist_cap);
    for (size_t p = seg_start; p < seg_end; p++) {
        uint32_t snap = 0;
        m2[p] = flm_walk_k7(d, p, seg_end, link, l4, s8, e3d, e3h, P, &snap, steps);
        mq[p] = snap;
    }
    free(l4); free(e3d); free(s8); free(e3h);
}

/* --- the parse as a functional graph ------------------------------------------------------
 * A "clean" iteration is one entered with matchLen == 2 (after a match was emitted, :826-827, or
 * after a literal step with no match pending).  From a clean iteration at p everything up to the
 * next clean iteration is a pure function of p: the node owns its tokens and J(p). */
typedef struct { uint32_t tok[260]; int ntok; size_t next; } node_t;

static int research(const uint8_t *d, size_t x, size_t seg_end, int L, const uint16_t *link, const uint32_t *m2,
                    const uint32_t *mq, const szm_params *P, uint32_t *out, uint64_t *stats) {
    size_t rem = seg_end - x;
    if (rem < MIN_MATCH) return 0;
    int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
    if (L >= cap) return 0;
    int nice = rem < (size_t)P->nice ? (int)rem : P->nice;
    uint32_t e;
    if (L < P->good) e = m2[x];          /* same budget, same stop: records > L are the same (App. A.3) */
    else if (L < nice) e = mq[x];        /* chainLength >>= 2 when matchLen >= goodLength :495 */
    else {                               /* prevLen >= niceLength': first strictly longer candidate wins */
        if (stats) stats[0]++;
        e = flm_walk(d, x, seg_end, link, P, L, P->max_chain >> 2, 0, NULL);
    }
    if ((int)(e & 0xFFFF) > L) {
        if (P->strategy == 1 && (e & 0xFFFF) <= 5) return 0; /* Filtered :794-797 => matchLen=2 <= prevLen */
        *out = e;
        return 1;
    }
    return 0;
}

static void node_eval(const uint8_t *d, size_t p, size_t seg_end, const uint16_t *link, const uint32_t *m2,
                      const uint32_t *mq, const szm_params *P, node_t *nd, uint64_t *stats) {
    nd->ntok = 0;
    uint32_t m = m2[p];
    int len = (int)(m & 0xFFFF), dist = (int)(m >> 16);
    if (len && len <= 5 && (P->strategy == 1 || (len == MIN_MATCH && dist > TOO_FAR))) len = 0; /* :794-797 */
    if (!len) { /* literal step :830-839 ; the byte is tallied by the NEXT iteration (or the final flush :752) */
        nd->tok[nd->ntok++] = d[p];
        nd->next = p + 1;
        return;
    }
    size_t x = p + 1;
    uint32_t cur = (uint32_t)len | ((uint32_t)dist << 16);
    for (;;) { /* lazy evaluation :802 "previous match was better" */
        uint32_t better;
        if (!research(d, x, seg_end, (int)(cur & 0xFFFF), link, m2, mq, P, &better, stats)) break;
        nd->tok[nd->ntok++] = d[x - 1];
        cur = better;
        x++;
    }
    nd->tok[nd->ntok++] = cur; /* TallyDist(strstart-1-prevMatch, prevLen) :815 */
    nd->next = x - 1 + (cur & 0xFFFF);
}

size_t szm_parse(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link, const uint32_t *m2,
                 const uint32_t *mq, const szm_params *P, uint32_t *tok, uint64_t *stats) {
    size_t p = seg_start, nt = 0;
    node_t nd;
    while (p < seg_end) {
        node_eval(d, p, seg_end, link, m2, mq, P, &nd, stats);
        memcpy(tok + nt, nd.tok, sizeof(uint32_t) * (size_t)nd.ntok);
        nt += (size_t)nd.ntok;
        p = nd.next;
        if (stats) stats[1]++;
    }
    return nt;
}

/* First clean iteration at or after `at_least` of the parse that starts, clean, at `from` (hand-over between the parts of one
 * stream on several engines, and between the windows of a long stream: DESIGN §3, §6). */
size_t szm_first_node(const uint8_t *d, size_t seg_end, const uint16_t *link, const uint32_t *m2, const uint32_t *mq,
                      const szm_params *P, size_t from, size_t at_least) {
    size_t p = from;
    node_t nd;
    while (p < seg_end && p < at_least) {
        node_eval(d, p, seg_end, link, m2, mq, P, &nd, NULL);
        p = nd.next;
    }
    return p;
}

size_t szm_parse_ranges(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link, const uint32_t *m2,
                        const uint32_t *mq, const szm_params *P, size_t R, uint32_t *tok, uint64_t *stats) {
    size_t n = seg_end - seg_start;
    if (n == 0) return 0;
    size_t nr = (n + R - 1) / R;
    uint8_t *visited = (uint8_t *)calloc(n + 1, 1);
    size_t *exitSpec = (size_t *)malloc(sizeof(size_t) * nr);
    size_t *entry = (size_t *)malloc(sizeof(size_t) * nr);
    size_t *mergeAt = (size_t *)malloc(sizeof(size_t) * nr); /* (size_t)-1 = went through */
    size_t *exitThrough = (size_t *)malloc(sizeof(size_t) * nr);
    node_t nd;
    /* C1: speculative walk of every range from a fresh (clean) state at its first position */
    for (size_t r = 0; r < nr; r++) {
        size_t s = seg_start + r * R, e = s + R < seg_end ? s + R : seg_end;
        size_t p = s;
        while (p < e) {
            visited[p - seg_start] = 1;
            node_eval(d, p, seg_end, link, m2, mq, P, &nd, NULL);
            p = nd.next;
        }
        exitSpec[r] = p;
    }
    /* C2: fix-up walk from the predecessor's speculative exit until it lands on this range's spec path */
    entry[0] = seg_start; mergeAt[0] = seg_start; exitThrough[0] = 0;
    for (size_t r = 1; r < nr; r++) {
        size_t s = seg_start + r * R, e = s + R < seg_end ? s + R : seg_end;
        size_t p = exitSpec[r - 1];
        entry[r] = p;
        mergeAt[r] = (size_t)-1;
        while (p < e) {
            if (visited[p - seg_start]) { mergeAt[r] = p; break; }
            node_eval(d, p, seg_end, link, m2, mq, P, &nd, NULL);
            p = nd.next;
        }
        exitThrough[r] = p;
        (void)s;
    }
    /* C3: sequential validation; redo ranges whose assumed entry was wrong */
    size_t trueExitPrev = (mergeAt[0] != (size_t)-1) ? exitSpec[0] : exitThrough[0];
    for (size_t r = 1; r < nr; r++) {
        size_t s = seg_start + r * R, e = s + R < seg_end ? s + R : seg_end;
        if (entry[r] != trueExitPrev) {
            size_t p = trueExitPrev;
            entry[r] = p;
            mergeAt[r] = (size_t)-1;
            while (p < e) {
                if (p >= s && visited[p - seg_start]) { mergeAt[r] = p; break; }
                node_eval(d, p, seg_end, link, m2, mq, P, &nd, NULL);
                p = nd.next;
            }
            exitThrough[r] = p;
        }
        if (mergeAt[r] == (size_t)-1 && stats) stats[0]++;
        trueExitPrev = (mergeAt[r] != (size_t)-1) ? exitSpec[r] : exitThrough[r];
    }
    /* emission: fix-up prefix, then the speculative suffix from the merge point */
    size_t nt = 0;
    for (size_t r = 0; r < nr; r++) {
        size_t s = seg_start + r * R, e = s + R < seg_end ? s + R : seg_end;
        size_t p = entry[r];
        size_t stop = mergeAt[r] != (size_t)-1 ? mergeAt[r] : e;
        while (p < stop) {
            node_eval(d, p, seg_end, link, m2, mq, P, &nd, NULL);
            memcpy(tok + nt, nd.tok, sizeof(uint32_t) * (size_t)nd.ntok);
            nt += (size_t)nd.ntok;
            p = nd.next;
        }
        if (mergeAt[r] != (size_t)-1) {
            p = mergeAt[r];
            while (p < e) {
                node_eval(d, p, seg_end, link, m2, mq, P, &nd, NULL);
                memcpy(tok + nt, nd.tok, sizeof(uint32_t) * (size_t)nd.ntok);
                nt += (size_t)nd.ntok;
                p = nd.next;
            }
        }
        (void)s;
    }
    free(visited); free(exitSpec); free(entry); free(mergeAt); free(exitThrough);
    return nt;
}

size_t szm_block_table(const uint32_t *tok, size_t ntok, int finish, int64_t *first, int32_t *count, int32_t *last) {
    /* A block is cut after every 16384th token emitted by a real iteration (IsFull, :841-852); the
     * final FlushBlock (:750-768) closes whatever remains — even nothing — unless the 16384th token was
     * a match ending exactly at the end of input during Finish (lastBlock computed true at :847). */
    size_t full = ntok / BLOCK_TOKENS, rem = ntok % BLOCK_TOKENS, nb = 0;
    for (size_t b = 0; b < full; b++) { first[nb] = (int64_t)(b * BLOCK_TOKENS); count[nb] = BLOCK_TOKENS; last[nb] = 0; nb++; }
    if (rem > 0 || ntok == 0) {
        first[nb] = (int64_t)(full * BLOCK_TOKENS); count[nb] = (int32_t)rem; last[nb] = 0; nb++;
    } else if ((tok[ntok - 1] >> 16) != 0 && !finish) { /* last token a match, sync flush: extra empty block */
        first[nb] = (int64_t)ntok; count[nb] = 0; last[nb] = 0; nb++;
    }
    if (finish) last[nb - 1] = 1;
    return nb;
}

/* =====================================================================================================
 * DeflateFast (levels 1-4, C/DeflaterEngine.cs:651-739) — model of the device formulation.
 *
 * The greedy parse inserts a position into the hash chains only if the parse visits it or it lies inside a
 * match of length <= max_lazy (:697-708); positions inside longer matches are skipped.  The reference's
 * head/prev chain of a position is therefore the ALL-positions chain of stage A (szm_links) filtered by an
 * "inserted" flag: a non-inserted hop costs no chain budget.  The set of flags depends on the parse, so the
 * device runs a fixpoint iteration over ranges (szm_fast_parse_fixpoint): every range is parsed from the
 * exit of its predecessor in the previous iteration, reading flags below its entry from the previous
 * iteration.  When an iteration reproduces (flags, exits) exactly, it IS the sequential parse (induction over
 * positions); range k is exact after iteration k+1 at the latest, in practice after a handful.
 * ===================================================================================================== */
int szm_fast_level_params(int level, szm_fast_params *out) { /* C/DeflaterConstants.cs:124-144 */
    static const int LAZY[5] = {0, 4, 5, 6, 4}, NICE[5] = {0, 8, 16, 32, 16}, CHAIN[5] = {0, 4, 8, 32, 16};
    if (level < 1 || level > 4) return -1;
    out->nice = NICE[level]; out->max_chain = CHAIN[level]; out->max_lazy = LAZY[level]; out->strategy = 0;
    return 0;
}

/* DeflateFast slides when an iteration starts at window index > 65274 (:680, strict), not >= as DeflateSlow. */
int64_t szm_base_of_fast(int64_t s) {
    int64_t idx = s + 1;
    if (idx <= 65274) return 0;
    return ((idx - 65274 + 32767) / 32768) * 32768;
}

/* FindLongestMatch at p over the filtered chain.  ins(q) = is q inserted.  Returns len | dist<<16, 0 = none. */
typedef struct { const uint8_t *fa, *fb; size_t split; } flag_view; /* q >= split ? fb[q] : fa[q] */
static inline int fv_get(const flag_view *v, size_t q) { return q >= v->split ? v->fb[q] : v->fa[q]; }

static uint32_t flm_fast(const uint8_t *d, size_t p, size_t seg_end, const uint16_t *link, const flag_view *fv,
                         const szm_fast_params *P) {
    size_t rem = seg_end - p;
    if (P->strategy == 2) return 0; /* HuffmanOnly :686 */
    int64_t base = szm_base_of_fast((int64_t)p);
    int64_t idx_p = (int64_t)p + 1 - base;
    int64_t limit_idx = idx_p - MAX_DIST > 0 ? idx_p - MAX_DIST : 0; /* :480 */
    /* hashHead = newest inserted position with this hash (:686): skip hops over positions never inserted */
    int64_t c = (int64_t)p;
    for (;;) {
        uint32_t l = link[c];
        if (l == 0) return 0;
        c -= l;
        if ((int64_t)p - c > MAX_DIST) return 0;   /* strstart - hashHead <= MAX_DIST :687 (older ones are farther still) */
        if (c + 1 - base < 1) return 0;            /* entry clamped to 0 by a slide :450-461 */
        if (fv_get(fv, (size_t)c)) break;
    }
    int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
    int nice = rem < (size_t)P->nice ? (int)rem : P->nice;
    int best = 2, budget = P->max_chain; /* matchLen == 2 < goodLength: full budget */
    if (best >= cap) return 0;
    uint32_t res = 0;
    for (;;) {
        int L = lcp_cap(d, (size_t)c, p, cap);
        if (L > best) {
            best = L;
            res = (uint32_t)L | ((uint32_t)((int64_t)p - c) << 16);
            if (best >= nice) return res;
        }
        /* curMatch = prev[curMatch] : next INSERTED position down the chain */
        int64_t c2 = c;
        for (;;) {
            uint32_t l = link[c2];
            if (l == 0) return res;
            c2 -= l;
            if (c2 + 1 - base <= limit_idx) return res; /* > limit :609 (monotone: older hops fail too) */
            if (fv_get(fv, (size_t)c2)) break;
        }
        if (--budget == 0) return res;
        c = c2;
    }
}

/* One greedy step at x.  Writes the token, updates flags in fnew[x ..), returns the next position. */
static size_t fast_step(const uint8_t *d, size_t x, size_t seg_end, const uint16_t *link, const flag_view *fv,
                        uint8_t *fnew, const szm_fast_params *P, uint32_t *tok) {
    size_t rem = seg_end - x;
    uint32_t m = 0;
    fnew[x] = 0;
    if (rem >= MIN_MATCH) { /* InsertString :686 */
        m = flm_fast(d, x, seg_end, link, fv, P);
        fnew[x] = 1;
    }
    if (!m) { *tok = d[x]; return x + 1; }
    size_t len = m & 0xFFFF;
    *tok = m;
    int ins = (int)len <= P->max_lazy && rem - len >= MIN_MATCH; /* :697 */
    for (size_t k = 1; k < len; k++) fnew[x + k] = (uint8_t)ins;
    return x + len;
}

size_t szm_fast_parse(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link,
                      const szm_fast_params *P, uint8_t *flags, uint32_t *tok) {
    flag_view fv = {flags, flags, 0};
    size_t x = seg_start, nt = 0;
    while (x < seg_end) x = fast_step(d, x, seg_end, link, &fv, flags, P, &tok[nt++]);
    return nt;
}

size_t szm_fast_parse_fixpoint(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link,
                               const szm_fast_params *P, size_t R, uint8_t *flags /* in: history flags below seg_start; out: all */,
                               uint32_t *tok, uint64_t *iters) {
    size_t n = seg_end - seg_start, nr = (n + R - 1) / R;
    if (nr == 0) return 0;
    uint8_t *fold = (uint8_t *)malloc(seg_end + 1), *fnew = (uint8_t *)malloc(seg_end + 1);
    size_t *entry = (size_t *)malloc(sizeof(size_t) * nr), *exitp = (size_t *)malloc(sizeof(size_t) * nr);
    uint32_t *stage = (uint32_t *)malloc(sizeof(uint32_t) * (nr * R + 1));
    uint32_t *cnt = (uint32_t *)malloc(sizeof(uint32_t) * nr);
    memcpy(fold, flags, seg_start);
    memset(fold + seg_start, 1, n); /* first guess: everything inserted */
    for (size_t r = 0; r < nr; r++) exitp[r] = seg_start + (r + 1) * R < seg_end ? seg_start + (r + 1) * R : seg_end;
    uint64_t it = 0;
    for (;;) {
        it++;
        memcpy(fnew, flags, seg_start);
        memset(fnew + seg_start, 0, n);
        int changed = 0;
        for (size_t r = 0; r < nr; r++) entry[r] = r == 0 ? seg_start : exitp[r - 1]; /* exits of the previous iteration */
        for (size_t r = 0; r < nr; r++) { /* independent of each other: this is the parallel loop */
            size_t re = seg_start + (r + 1) * R < seg_end ? seg_start + (r + 1) * R : seg_end;
            size_t x = entry[r], k = 0;
            flag_view fv = {fold, fnew, entry[r]};
            while (x < re) x = fast_step(d, x, seg_end, link, &fv, fnew, P, &stage[r * R + k++]);
            cnt[r] = (uint32_t)k;
            size_t ex = x > re ? x : re; /* a range whose entry is already past its end is skipped */
            if (entry[r] >= re) ex = entry[r];
            if (ex != exitp[r]) changed = 1;
            exitp[r] = ex;
        }
        if (!changed && memcmp(fold + seg_start, fnew + seg_start, n) == 0) break;
        uint8_t *t = fold; fold = fnew; fnew = t;
    }
    size_t nt = 0;
    for (size_t r = 0; r < nr; r++) { memcpy(tok + nt, stage + r * R, sizeof(uint32_t) * cnt[r]); nt += cnt[r]; }
    memcpy(flags + seg_start, fnew + seg_start, n);
    if (iters) *iters = it;
    free(fold); free(fnew); free(entry); free(exitp); free(stage); free(cnt);
    return nt;
}

/* ---- analysis helper (round-2 planning, not used by any kernel): which positions does the parse read? -----------------
 * needed[p] = 1 if DeflateSlow's parse reads FindLongestMatch's result at p (clean iteration or lazy re-search). */
size_t szm_parse_needed(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link, const uint32_t *m2,
                        const uint32_t *mq, const szm_params *P, uint8_t *needed) {
    size_t p = seg_start, count = 0;
    while (p < seg_end) {
        if (!needed[p]) { needed[p] = 1; count++; }
        uint32_t m = m2[p];
        int len = (int)(m & 0xFFFF), dist = (int)(m >> 16);
        if (len && len <= 5 && (P->strategy == 1 || (len == MIN_MATCH && dist > TOO_FAR))) len = 0;
        if (!len) { p++; continue; }
        size_t x = p + 1;
        uint32_t cur = (uint32_t)len | ((uint32_t)dist << 16);
        for (;;) {
            uint32_t better;
            if (x < seg_end && !needed[x]) { needed[x] = 1; count++; }
            if (!research(d, x, seg_end, (int)(cur & 0xFFFF), link, m2, mq, P, &better, NULL)) break;
            cur = better;
            x++;
        }
        p = x - 1 + (cur & 0xFFFF);
    }
    return count;
}

/* ---- model of stage B's on-demand form (k_match_lazy): which positions do the tile walkers evaluate? ------------------
 * Inside every tile of `tile` positions a walker starts at each multiple of `stride`, follows the DeflateSlow step from
 * position to position and stops on a clean position another walker has passed or when it leaves the tile.  The union
 * does not depend on the order in which walkers run (each path is cut only where another one continues it). */
size_t szm_lazy_eval_set(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link, const uint32_t *m2,
                         const uint32_t *mq, const szm_params *P, size_t tile, size_t stride, uint8_t *evaluated) {
    size_t count = 0;
    uint8_t *clean = (uint8_t *)calloc(seg_end + 1, 1);
    for (size_t t0 = seg_start; t0 < seg_end; t0 += tile) {
        size_t t1 = t0 + tile < seg_end ? t0 + tile : seg_end;
        for (size_t st = t0; st < t1; st += stride) {
            size_t p = st;
            while (p < t1) {
                if (clean[p]) break;                 /* merge */
                clean[p] = 1;
                if (!evaluated[p]) { evaluated[p] = 1; count++; }
                uint32_t m = m2[p];
                int len = (int)(m & 0xFFFF), dist = (int)(m >> 16);
                if (len && len <= 5 && (P->strategy == 1 || (len == MIN_MATCH && dist > TOO_FAR))) len = 0;
                if (!len) { p++; continue; }
                size_t x = p + 1;
                uint32_t cur = (uint32_t)len | ((uint32_t)dist << 16);
                int left_tile = 0;
                for (;;) {
                    if (x >= t1) { left_tile = 1; break; }   /* the lazy look belongs to the next tile: not evaluated here */
                    if (!evaluated[x]) { evaluated[x] = 1; count++; }
                    uint32_t better;
                    if (!research(d, x, seg_end, (int)(cur & 0xFFFF), link, m2, mq, P, &better, NULL)) break;
                    cur = better;
                    x++;
                }
                if (left_tile) break;
                p = x - 1 + (cur & 0xFFFF);
            }
        }
    }
    free(clean);
    return count;
}
