/* tile_time_sim.c — research tool (not product, not oracle): a timing model of one stage-B tile on one CU, to see what the end of
 * a tile costs and what a different hand-out order of its positions would change (DESIGN §4.2 "what a tile costs", §8 1(d)).
 *
 * The exact FindLongestMatch walk (C/DeflaterEngine.cs:474-612, with the two-byte filter of k_match4) runs for every position of
 * a sample under k_match4's scheduling: 16 waves, two contexts per lane, census -> QUICK (three chain steps per iteration while
 * >= QKEEP contexts walk) / VERIFY (8 bytes per step while >= VKEEP compare) / exit to FETCH when >= FTH contexts are free,
 * positions handed out in slices from one counter.  Time: a SIMD (4 waves) issues one wave's step at a time; a step occupies the
 * issue port for its instruction cost and the wave for max(cost, latency) — so four busy waves are issue-bound and a lone wave
 * at the end of a tile is latency-bound.  Costs are read off the ISA (cycles at 4 per VALU instruction) and one scale factor is
 * left to calibration against the measured tile times (profiles/r02/lab_s46_tile_length.log: 99 / 136 / 206 us for 4 / 8 / 16 Ki).
 *
 *   gcc -O2 -o /tmp/tile_sim tools/tile_time_sim.c && /tmp/tile_sim sample.bin [TILE=16384] [ORDER=0|1] [TH=4096] [key=value ...]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { MAX_DIST = 32506, MAX_MATCH = 258, MAXWAVES = 16, LANES = 64, NCTX = 128 };
static int WAVES = 16;
enum { IDLE = 0, QUICK = 1, VERIFY = 2 };
typedef struct { int mode, p, cl, best, left, off; } Ctx;

static uint8_t *d; static uint16_t *lk; static size_t n;
static int max_chain = 128, nice = 128;
static int QSTEPS = 3, RING = 0, SPAN = 18400, TILE = 16384, ORDER = 0, TH = 4096, FTH = 32, VTH = 2, QKEEP = 64, VKEEP = 2, SLICE = 128, QKEEP_TAIL = -1;
/* issue cost / latency of the engine's steps, in cycles */
static double C_CENSUS = 40, C_QITER = 190, L_QITER = 360, C_CLASSIFY = 50, C_VITER = 150, L_VITER = 140, C_COMPLETE = 210, L_COMPLETE = 240,
              C_FETCH = 330, L_FETCH = 420, C_FETCH_ORD = 130, C_STAGE_US = 21, SCALE = 1.0, GHZ = 2.4;

static void links_build(void) {
    int32_t *head = malloc(sizeof(int32_t) * 32768);
    for (int i = 0; i < 32768; i++) head[i] = -1;
    for (size_t q = 0; q + 3 <= n; q++) {
        uint32_t h = (((uint32_t)d[q] << 10) ^ ((uint32_t)d[q + 1] << 5) ^ d[q + 2]) & 0x7FFF;
        int32_t prev = head[h];
        uint32_t dist = prev < 0 ? 0 : (uint32_t)(q - (size_t)prev);
        lk[q] = dist > 32767 ? 0 : (uint16_t)dist;
        head[h] = (int32_t)q;
    }
    free(head);
}
static int walk_start(Ctx *c, int64_t p) {
    c->p = (int)p;
    if ((int64_t)n - p < 3 || lk[p] == 0 || lk[p] > MAX_DIST) return 0;
    c->cl = (int)(p - lk[p]); c->best = 2; c->left = max_chain - 1; c->off = 0;
    return 1;
}
static int walk_next(Ctx *c) { /* 1: there is a next candidate */
    uint32_t l = lk[c->cl];
    if (!l || c->p - ((int64_t)c->cl - l) >= MAX_DIST || c->left == 0) return 0;
    c->left--; c->cl -= (int)l;
    return 1;
}
static void quick_step(Ctx *c) {
    if (d[c->cl + c->best] == d[c->p + c->best] && d[c->cl + c->best - 1] == d[c->p + c->best - 1]) { c->off = 0; c->mode = VERIFY; return; }
    if (!walk_next(c)) c->mode = IDLE;
}
static int verify_step(Ctx *c) { /* 1: comparison complete */
    int64_t rem = (int64_t)n - c->p;
    int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH, k = 0, l = c->off;
    while (k < 8 && l < cap && d[c->cl + l] == d[c->p + l]) { l++; k++; }
    c->off = l;
    return !(k == 8 && l < cap);
}
static void complete(Ctx *c) {
    int64_t rem = (int64_t)n - c->p;
    int nc = rem < nice ? (int)rem : nice;
    if (c->off > c->best) { c->best = c->off; if (c->best >= nc) { c->mode = IDLE; return; } }
    c->off = 0;
    c->mode = walk_next(c) ? QUICK : IDLE;
}

typedef struct { Ctx c[NCTX]; int wnext, wend, pass, exhausted, done, starve_exit; double ready; } Wave;

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: tile_time_sim file [key=value...]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    fseek(f, 0, SEEK_END); n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    d = malloc(n + 512); memset(d, 0, n + 512);
    if (fread(d, 1, n, f) != n) return 2;
    fclose(f);
    lk = calloc(n + 8, 2);
    links_build();
    for (int i = 2; i < argc; i++) {
        char *eq = strchr(argv[i], '='); if (!eq) continue; *eq = 0;
        double v = atof(eq + 1); const char *k = argv[i];
#define P(name, var) if (!strcmp(k, name)) { var = v; continue; }
        P("QSTEPS", QSTEPS) P("WAVES", WAVES) P("RING", RING) P("SPAN", SPAN) P("TILE", TILE) P("ORDER", ORDER) P("TH", TH) P("FTH", FTH) P("VTH", VTH) P("QKEEP", QKEEP) P("VKEEP", VKEEP) P("SLICE", SLICE) P("QKEEP_TAIL", QKEEP_TAIL)
        P("CHAIN", max_chain) P("NICE", nice) P("SCALE", SCALE) P("C_QITER", C_QITER) P("L_QITER", L_QITER) P("C_VITER", C_VITER) P("L_VITER", L_VITER)
        P("C_COMPLETE", C_COMPLETE) P("L_COMPLETE", L_COMPLETE) P("C_FETCH", C_FETCH) P("L_FETCH", L_FETCH) P("C_CENSUS", C_CENSUS) P("C_FETCH_ORD", C_FETCH_ORD)
        fprintf(stderr, "unknown key %s\n", k); return 2;
    }
    if (RING) TILE = 48 * 16384;                               /* ring (k_match8): one stripe, positions handed out as the ring allows */
    size_t first = RING ? 1 : 4 * 16384 / (size_t)TILE + 1, ntiles = n / (size_t)TILE;
    size_t want = RING ? 1 : (size_t)(48 * 16384 / TILE);      /* the same 768 Ki positions whatever the tile length */
    if (ntiles > first + want) ntiles = first + want;
    if (RING && n < 2 * (size_t)TILE) { fprintf(stderr, "sample too short for RING\n"); return 2; }
    uint64_t n_starved = 0;
    double total = 0, busy_ctx_time = 0, t_drain = 0, t_first_exit = 0; uint64_t positions = 0, steps_q = 0;
    static Wave W[MAXWAVES];
    for (size_t t = first; t < ntiles; t++) {
        int64_t t0 = (int64_t)t * TILE;
        int counter[2] = {0, 0};
        memset(W, 0, sizeof W);
        for (int v = 0; v < WAVES; v++) W[v].starve_exit = -1;
        double simd_free[4] = {0, 0, 0, 0}, tend = 0, t_last_grab = 0, t_first_done = -1;
        int alive = WAVES;
        while (alive) {
            /* the wave that can go next: earliest max(ready, its SIMD free) */
            int w = -1; double best = 1e30;
            for (int i = 0; i < WAVES; i++) if (!W[i].done) { double s = W[i].ready > simd_free[i & 3] ? W[i].ready : simd_free[i & 3]; if (s < best) { best = s; w = i; } }
            Wave *X = &W[w];
            double start = best, cost = 0, lat = 0;
            int nq = 0, nv = 0;
            for (int i = 0; i < NCTX; i++) { nq += X->c[i].mode == QUICK; nv += X->c[i].mode == VERIFY; }
            int busy = nq + nv;
            int bexit = X->exhausted ? 0 : (X->starve_exit >= 0 ? X->starve_exit : NCTX - FTH);
            cost = C_CENSUS;
            if (busy <= bexit) {
                /* FETCH: both contexts' free lanes get positions */
                if (X->exhausted) { X->done = 1; alive--; if (start > tend) tend = start; if (t_first_done < 0) t_first_done = start; continue; }
                int got_any = 0, iters = 0, starved = 0;
                int limit = TILE;
                if (RING) {    /* lowest position in flight over all waves (reservations included) */
                    int mn = counter[0];
                    for (int v = 0; v < WAVES; v++) { if (W[v].wnext < W[v].wend && W[v].wnext < mn) mn = W[v].wnext; for (int i = 0; i < NCTX; i++) if (W[v].c[i].mode != IDLE && (int)(W[v].c[i].p - t0) < mn) mn = (int)(W[v].c[i].p - t0); }
                    limit = mn + SPAN < TILE ? mn + SPAN : TILE;
                }
                for (int i = 0; i < NCTX && !X->exhausted && !starved; i++) {
                    if (X->c[i].mode != IDLE) continue;
                    for (;;) {
                        if (X->wnext >= X->wend) {
                            if (RING && counter[0] + SLICE > limit && counter[0] < TILE) { starved = 1; break; }
                            int base = counter[X->pass]; counter[X->pass] += SLICE;
                            if (base < TILE) t_last_grab = start;
                            X->wnext = base < TILE ? base : TILE; X->wend = base + SLICE < TILE ? base + SLICE : TILE;
                            if (X->wnext >= X->wend) { if (ORDER && X->pass == 0) { X->pass = 1; X->wnext = X->wend = 0; continue; } X->exhausted = 1; break; }
                        }
                        int p = X->wnext++;
                        if (ORDER) { int a = lk[t0 + p] && lk[t0 + p] < TH; iters++; if (a != (X->pass == 0)) continue; }
                        positions++; got_any = 1;
                        if (walk_start(&X->c[i], t0 + p)) X->c[i].mode = VERIFY;
                        break;
                    }
                }
                (void)got_any;
                X->starve_exit = -1;
                if (starved) { /* the ring is not ahead: run until 16 more walks are over (k_match8), or sleep if there is nothing to run */
                    int b2 = 0;
                    for (int i = 0; i < NCTX; i++) b2 += X->c[i].mode != IDLE;
                    n_starved++;
                    if (b2 == 0) { X->ready = start + 600 * SCALE; simd_free[w & 3] = start + 40 * SCALE; continue; }
                    X->starve_exit = b2 > 16 ? b2 - 16 : 0;
                }
                cost += C_FETCH + (ORDER ? C_FETCH_ORD * (1 + iters / 64) : 0); lat = L_FETCH;
            } else if (nv >= VTH || nq == 0) {
                /* VERIFY phase: 8-byte steps while >= VKEEP compare, then COMPLETE */
                for (;;) {
                    int still = 0;
                    for (int i = 0; i < NCTX; i++) if (X->c[i].mode == VERIFY && X->c[i].off >= 0) { if (verify_step(&X->c[i])) X->c[i].off = -1 - X->c[i].off; else still++; }
                    cost += C_VITER; lat += L_VITER;
                    if (still < VKEEP) break;
                }
                for (int i = 0; i < NCTX; i++) if (X->c[i].mode == VERIFY && X->c[i].off < 0) { X->c[i].off = -1 - X->c[i].off; complete(&X->c[i]); }
                cost += C_COMPLETE; lat += L_COMPLETE;
            } else {
                int qk = X->exhausted && QKEEP_TAIL >= 0 ? QKEEP_TAIL : QKEEP;
                for (;;) {
                    for (int s = 0; s < QSTEPS; s++) for (int i = 0; i < NCTX; i++) if (X->c[i].mode == QUICK) { quick_step(&X->c[i]); steps_q++; }
                    cost += 16 + (C_QITER - 16) * QSTEPS / 3.0; lat += L_QITER * QSTEPS / 3.0;   /* (a fixed loop overhead + per-step instructions) */
                    int still = 0;
                    for (int i = 0; i < NCTX; i++) still += X->c[i].mode == QUICK;
                    if (still < qk || still == 0) break;
                }
                cost += C_CLASSIFY;
            }
            cost *= SCALE; lat *= SCALE;
            simd_free[w & 3] = start + cost;
            X->ready = start + (cost > lat ? cost : lat);
            busy_ctx_time += busy * (cost > lat ? cost : lat);
            if (X->ready > tend) tend = X->ready;
        }
        total += tend; t_drain += tend - t_last_grab; t_first_exit += tend - t_first_done;
    }
    double tiles = (double)(ntiles - first), us = total / tiles / (GHZ * 1e3) + C_STAGE_US;
    printf("TILE %d ORDER %d TH %d SLICE %d: %.1f us per tile (%.1f search + %.0f staging) = %.2f ns per position -> %.1f ms per GiB on 256 CUs; "
           "last slice taken %.1f us before the end, first wave done %.1f us before the end; contexts busy %.0f%%; chain steps per position %.1f; starved fetches %llu\n",
           TILE, ORDER, TH, SLICE, us, us - C_STAGE_US, C_STAGE_US, us * 1e3 / TILE, us * 1e-3 * (1073741824.0 / TILE) / 256,
           t_drain / tiles / (GHZ * 1e3), t_first_exit / tiles / (GHZ * 1e3), 100.0 * busy_ctx_time / (total * WAVES * NCTX), (double)steps_q / positions, (unsigned long long)n_starved);
    return 0;
}
