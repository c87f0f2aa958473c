/* tools/bucket_model.c — CPU model of stage B in BUCKET ORDER (round 3; analysis + exactness check, not product code).
 *
 * The positions of a window (history + tile) are sorted by (3-byte hash, position): the hash chain of a position is then the
 * run of entries in front of it — no prev[] hops.  A wavefront takes 64 consecutive tile entries (mostly one bucket, so their
 * walks are almost equally long) and every lane walks backwards through the sorted array.  FindLongestMatch's result is the
 * maximum of an order-free key over the candidates (first candidate with length >= nice wins; otherwise the longest, the
 * nearest among equals), so a lane may look at its candidates in any grouping as long as the key is right.
 *
 * Checked here: the tables equal szm_match_tables (the restated reference walk).  Counted: candidates per position, the lane
 * efficiency of 64-entry wavefronts, how many candidates share >= 7 bytes with the position (they need the window's bytes; the
 * rest is decided from the 7 content bytes the sorted entry carries).
 *
 *   gcc -O2 -o /tmp/lab/bucket_model tools/bucket_model.c && /tmp/lab/bucket_model file level window_KiB
 */
#include "../oracle/szl_model.c"
#include <stdio.h>

typedef struct { uint32_t h, pos; } ent_t;
static int cmp_ent(const void *a, const void *b) {
    const ent_t *x = (const ent_t *)a, *y = (const ent_t *)b;
    if (x->h != y->h) return x->h < y->h ? -1 : 1;
    return x->pos < y->pos ? -1 : x->pos > y->pos;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s file level window_KiB\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    fseek(f, 0, SEEK_END); size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *d = (uint8_t *)calloc(n + 600, 1);
    if (fread(d, 1, n, f) != n) return 1;
    fclose(f);
    szm_params P;
    if (szm_level_params(atoi(argv[2]), &P)) return 2;
    const size_t W = (size_t)atol(argv[3]) * 1024, H = 32768, T = W - H;
    uint16_t *link = (uint16_t *)calloc(n + 8, 2);
    uint32_t *m2 = (uint32_t *)calloc(n + 8, 4), *mq = (uint32_t *)calloc(n + 8, 4);
    size_t ends[1] = {n};
    szm_links(d, n, ends, 1, link);
    szm_match_tables(d, 0, n, link, &P, m2, mq);
    ent_t *S = (ent_t *)malloc(sizeof(ent_t) * W);
    uint64_t n_pos = 0, n_steps = 0, n_wave_slots = 0, n_ge7 = 0, n_ge7_filtered = 0, n_ext_bytes8 = 0, bad = 0, n_waves = 0, n_mq_differs = 0, n_lane_idle_hist = 0;
    uint64_t n_upd = 0, slots_b[4] = {0, 0, 0, 0};
    static int bsteps[4][1024], bnav[4][1024]; int bcnt[4] = {0, 0, 0, 0}; const size_t NB[4] = {64, 128, 256, 512};
    const int SNAP = P.max_chain >> 2;
    for (size_t t0 = 0; t0 < n; t0 += T) {
        const size_t t1 = t0 + T < n ? t0 + T : n;
        const size_t w0 = t0 >= H ? t0 - H : 0;
        size_t ns = 0;
        for (size_t q = w0; q < t1; q++) if (n - q >= 3) { S[ns].h = hash3(d, q); S[ns].pos = (uint32_t)q; ns++; }   /* inserted positions only */
        qsort(S, ns, sizeof(ent_t), cmp_ent);
        /* wavefronts of 64 consecutive entries; lanes whose entry is a history position idle */
        for (size_t i0 = 0; i0 < ns; i0 += 64) {
            int wave_max = 0, any_tile = 0;
            for (size_t i = i0; i < i0 + 64 && i < ns; i++) {
                const size_t p = S[i].pos;
                if (p < t0) { continue; }
                any_tile = 1;
                n_pos++;
                /* the walk of lane (i - i0): candidates S[i-1], S[i-2], ... of the same bucket */
                const size_t rem = n - p;
                uint32_t r2 = 0, rq = 0;
                int steps = 0;
                if (rem >= MIN_MATCH && P.strategy != 2) {
                    const int64_t base = szm_base_of((int64_t)p);
                    const int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
                    const int nice = rem < (size_t)P.nice ? (int)rem : P.nice;
                    int best = 2, budget = P.max_chain;
                    for (size_t k = 1; k <= i; k++) {
                        const ent_t *c = &S[i - k];
                        if (c->h != S[i].h) break;
                        const int64_t dist = (int64_t)p - c->pos;
                        if (k == 1 ? dist > MAX_DIST : dist > MAX_DIST - 1) break;      /* :788 / :609 */
                        if ((int64_t)c->pos + 1 - base < 1) break;                      /* below window index 1: clamped by a slide */
                        if (budget-- == 0) break;
                        steps++;
                        const int L = lcp_cap(d, c->pos, p, cap);
                        if (L >= 7) { n_ge7++; if (best < 7 || (d[c->pos + best] == d[p + best] && d[c->pos + best - 1] == d[p + best - 1])) { n_ge7_filtered++; n_ext_bytes8 += (uint64_t)(L - 7) / 8 + 1; } }
                        if (L > best) {
                            best = L; n_upd++;
                            r2 = (uint32_t)L | ((uint32_t)dist << 16);
                            if (steps <= SNAP) rq = r2;
                            if (L >= nice) break;
                        }
                    }
                }
                if (r2 != m2[p] || rq != mq[p]) { if (bad < 5) fprintf(stderr, "MISMATCH at %zu: got %08x/%08x want %08x/%08x\n", p, r2, rq, m2[p], mq[p]); bad++; }
                if (r2 != rq) n_mq_differs++;
                n_steps += (uint64_t)steps;
                {   /* batch models */
                    size_t bs = i; while (bs > 0 && S[bs - 1].h == S[i].h && i - bs < 128) bs--;
                    for (int b = 0; b < 4; b++) { bsteps[b][bcnt[b]] = steps; bnav[b][bcnt[b]] = (int)(i - bs); bcnt[b]++; }
                }
                if (steps > wave_max) wave_max = steps;
            }
            if (any_tile) { n_waves++; n_wave_slots += 64ull * (uint64_t)wave_max; }
            for (int b = 0; b < 4; b++) if (((i0 + 64) % NB[b]) == 0 || i0 + 64 >= ns) {
                /* order the batch's tile entries by navail (descending), cut into waves of 64 */
                int cnt = bcnt[b];
                for (int x = 1; x < cnt; x++) { int s_ = bsteps[b][x], v_ = bnav[b][x], y = x - 1; while (y >= 0 && bnav[b][y] < v_) { bnav[b][y + 1] = bnav[b][y]; bsteps[b][y + 1] = bsteps[b][y]; y--; } bnav[b][y + 1] = v_; bsteps[b][y + 1] = s_; }
                for (int x = 0; x < cnt; x += 64) { int mx = 0; for (int y = x; y < x + 64 && y < cnt; y++) if (bsteps[b][y] > mx) mx = bsteps[b][y]; slots_b[b] += 64ull * (uint64_t)mx; }
                bcnt[b] = 0;
            }
        }
    }
    printf("window %zu KiB (tile %zu): %llu positions, %llu mismatches vs szm_match_tables\n", W >> 10, T, (unsigned long long)n_pos, (unsigned long long)bad);
    printf("  %.2f candidates per position; wavefronts of 64 sorted entries: %.1f %% of the lane-steps do work (%.2f lane-steps per position)\n",
           (double)n_steps / n_pos, 100.0 * n_steps / n_wave_slots, (double)n_wave_slots / n_pos);
    for (int b = 0; b < 4; b++) printf("  batches of %zu entries, tile entries ordered by bucket rank, waves of 64: %.1f %% of the lane-steps do work\n", NB[b], 100.0 * n_steps / slots_b[b]);
    printf("  candidates with lcp >= 7: %.3f per position (%.2f %% of candidates); after the scan_end filter at the running best: %.3f per position, %.2f eight-byte steps each\n",
           (double)n_ge7 / n_pos, 100.0 * n_ge7 / n_steps, (double)n_ge7_filtered / n_pos, (double)n_ext_bytes8 / (n_ge7_filtered ? n_ge7_filtered : 1));
    printf("  best_len updates: %.2f per position; Mq differs from M2 at %.2f %% of the positions\n", (double)n_upd / n_pos, 100.0 * n_mq_differs / n_pos);
    return bad != 0;
}
