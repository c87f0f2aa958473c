/* tools/ondemand_model.c — CPU count for stage B's on-demand form (round 3 planning; analysis only, not product code).
 *
 * Question: how many hash-chain steps does a tile of walkers need if every walker evaluates FindLongestMatch only where its
 * parse stands AND only with the budget its state asks for (full max_chain at a clean iteration or a lazy look after a
 * match shorter than goodLength; max_chain>>2 after a longer one, C/DeflaterEngine.cs:495), against the 100 %-of-positions /
 * full-budget search of k_match4 — and how long is the slowest dependent chain of a tile (its tail)?
 *
 *   gcc -O2 -o /tmp/lab/ondemand_model tools/ondemand_model.c && /tmp/lab/ondemand_model file level tile stride [stride...]
 */
#include "../oracle/szl_model.c"
#include <stdio.h>

static int steps_of(const uint8_t *d, size_t p, size_t seg_end, const uint16_t *link, const szm_params *P, int budget) {
    /* candidates a walk at p examines (the walk of flm_walk, counting) */
    size_t rem = seg_end - p;
    if (rem < MIN_MATCH || P->strategy == 2) return 0;
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
    int64_t limit_idx = idx_p - MAX_DIST > 0 ? idx_p - MAX_DIST : 0;
    int count = 0;
    for (;;) {
        int L = lcp_cap(d, (size_t)c, p, cap);
        count++;
        if (L > best) { best = L; if (best >= nice) return count; }
        uint32_t l = link[c];
        if (l == 0) break;
        int64_t c2 = c - l;
        if (c2 + 1 - base <= limit_idx) break;
        if (--budget == 0) break;
        c = c2;
    }
    return count;
}

static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s file level tile stride...\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    fseek(f, 0, SEEK_END); size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *d = (uint8_t *)calloc(n + 600, 1);
    if (fread(d, 1, n, f) != n) return 1;
    fclose(f);
    szm_params P;
    if (szm_level_params(atoi(argv[2]), &P)) return 2;
    size_t tile = (size_t)atol(argv[3]);
    uint16_t *link = (uint16_t *)calloc(n + 8, 2);
    uint32_t *m2 = (uint32_t *)calloc(n + 8, 4), *mq = (uint32_t *)calloc(n + 8, 4);
    size_t ends[1] = {n};
    szm_links(d, n, ends, 1, link);
    szm_match_tables(d, 0, n, link, &P, m2, mq);
    uint8_t *s2 = (uint8_t *)calloc(n + 8, 1), *sq = (uint8_t *)calloc(n + 8, 1);
    uint64_t tot2 = 0, totq = 0;
    for (size_t p = 0; p < n; p++) { s2[p] = (uint8_t)steps_of(d, p, n, link, &P, P.max_chain); sq[p] = (uint8_t)steps_of(d, p, n, link, &P, P.max_chain >> 2); tot2 += s2[p]; totq += sq[p]; }
    printf("n %zu level %s good %d nice %d chain %d: full search %.2f steps/byte (quarter budget everywhere: %.2f)\n", n, argv[2], P.good, P.nice, P.max_chain, (double)tot2 / n, (double)totq / n);
    {   /* histogram of walk lengths (full) */
        uint64_t h[9] = {0}, hs[9] = {0};
        for (size_t p = 0; p < n; p++) { int b = s2[p] == 0 ? 0 : s2[p] <= 2 ? 1 : s2[p] <= 4 ? 2 : s2[p] <= 8 ? 3 : s2[p] <= 16 ? 4 : s2[p] <= 32 ? 5 : s2[p] <= 64 ? 6 : s2[p] < P.max_chain ? 7 : 8; h[b]++; hs[b] += s2[p]; }
        const char *nm[9] = {"0", "1-2", "3-4", "5-8", "9-16", "17-32", "33-64", "65-<max", "max"};
        for (int b = 0; b < 9; b++) printf("  walk %-8s %5.1f %% of positions, %5.1f %% of steps\n", nm[b], 100.0 * h[b] / n, 100.0 * hs[b] / tot2);
    }
    /* the true parse: positions read, and steps with state-dependent budgets */
    {
        size_t p = 0; uint64_t ev = 0, st = 0, evq = 0, nclean = 0;
        while (p < n) {
            ev++; nclean++; st += s2[p];
            uint32_t m = m2[p]; int len = (int)(m & 0xFFFF), dist = (int)(m >> 16);
            if (len && len <= 5 && (P.strategy == 1 || (len == MIN_MATCH && dist > TOO_FAR))) len = 0;
            if (!len) { p++; continue; }
            size_t x = p + 1; uint32_t cur = (uint32_t)len | ((uint32_t)dist << 16);
            for (;;) {
                if (x < n) { ev++; if ((int)(cur & 0xFFFF) < P.good) st += s2[x]; else { st += sq[x]; evq++; } }
                uint32_t better;
                if (!research(d, x, n, (int)(cur & 0xFFFF), link, m2, mq, &P, &better, NULL)) break;
                cur = better; x++;
            }
            p = x - 1 + (cur & 0xFFFF);
        }
        printf("true parse: reads %.1f %% of positions (%.1f %% clean, %.1f %% of reads quarter-budget): %.2f steps/byte\n", 100.0 * ev / n, 100.0 * nclean / n, 100.0 * evq / ev, (double)st / n);
    }
    for (int a = 4; a < argc; a++) {
        size_t stride = (size_t)atol(argv[a]);
        uint8_t *clean = (uint8_t *)calloc(n + 1, 1), *done = (uint8_t *)calloc(n + 1, 1); /* done: bit0 full evaluated, bit1 quarter evaluated */
        uint64_t evals = 0, evals_dup = 0, steps = 0, steps_nodup = 0, uniq = 0;
        uint64_t sum_ideal = 0, sum_sched = 0, sum_maxw = 0, ntiles = 0;
        const int NCTX = 2048, EV_COST = 6; /* contexts per tile; fixed cost of an evaluation in step units (fetch, first compare, store, consume) */
        uint32_t *wcost = (uint32_t *)malloc(sizeof(uint32_t) * (tile / stride + 2));
        for (size_t t0 = 0; t0 < n; t0 += tile) {
            size_t t1 = t0 + tile < n ? t0 + tile : n;
            size_t nw = 0;
            /* walkers in reverse start order: on the device they run concurrently, each one behind the next, so a walker
             * ends where its parse first stands on a clean position of a path ahead of it */
            size_t nst = (t1 - t0 + stride - 1) / stride;
            for (size_t si = nst; si-- > 0;) {
                size_t st = t0 + si * stride;
                size_t p = st; uint64_t wc = 0;
                #define EVAL(x, quarter) do { int s_ = (quarter) ? sq[x] : s2[x]; evals++; steps += s_; wc += s_ + EV_COST; \
                    uint8_t bit_ = (quarter) ? 2 : 1; if (!done[x]) uniq++; if (done[x] & (bit_ | 1)) evals_dup++; else steps_nodup += s_; done[x] |= bit_; } while (0)
                while (p < t1) {
                    if (clean[p]) break;
                    clean[p] = 1;
                    EVAL(p, 0);
                    uint32_t m = m2[p]; int len = (int)(m & 0xFFFF), dist = (int)(m >> 16);
                    if (len && len <= 5 && (P.strategy == 1 || (len == MIN_MATCH && dist > TOO_FAR))) len = 0;
                    if (!len) { p++; continue; }
                    size_t x = p + 1; uint32_t cur = (uint32_t)len | ((uint32_t)dist << 16); int left_tile = 0;
                    for (;;) {
                        if (x >= t1) { left_tile = 1; break; }
                        EVAL(x, (int)(cur & 0xFFFF) >= P.good);
                        uint32_t better;
                        if (!research(d, x, n, (int)(cur & 0xFFFF), link, m2, mq, &P, &better, NULL)) break;
                        cur = better; x++;
                    }
                    if (left_tile) break;
                    p = x - 1 + (cur & 0xFFFF);
                }
                wcost[nw++] = (uint32_t)wc;
            }
            /* schedule the walkers on NCTX contexts in start order (a context takes the next start when its walker ends) */
            uint64_t tot = 0; uint32_t mx = 0;
            for (size_t i = 0; i < nw; i++) { tot += wcost[i]; if (wcost[i] > mx) mx = wcost[i]; }
            uint64_t finish = 0;
            if (nw <= (size_t)NCTX) finish = mx;
            else { /* greedy: min-heap would be exact; with nw < 8*NCTX a simple array is fine */
                static uint64_t ctx[2048];
                for (int i = 0; i < NCTX; i++) ctx[i] = 0;
                for (size_t i = 0; i < nw; i++) { int b = 0; for (int k = 1; k < NCTX; k++) if (ctx[k] < ctx[b]) b = k; ctx[b] += wcost[i]; }
                for (int i = 0; i < NCTX; i++) if (ctx[i] > finish) finish = ctx[i];
            }
            sum_ideal += (tot + NCTX - 1) / NCTX; sum_sched += finish; sum_maxw += mx; ntiles++;
        }
        printf("tile %zu stride %3zu: evaluates %.1f %% of positions (%.1f %% evaluations incl. repeats), %.2f steps/byte (%.2f without repeats); per tile: ideal %.0f, scheduled %.0f, longest walker %.0f step-units (x%.2f)\n",
               tile, stride, 100.0 * uniq / n, 100.0 * evals / n, (double)steps / n, (double)steps_nodup / n, (double)sum_ideal / ntiles, (double)sum_sched / ntiles, (double)sum_maxw / ntiles, (double)sum_sched / sum_ideal);
        free(clean); free(done); free(wcost);
    }
    return 0;
}
