/* sched_sim.c — offline emulator of the stage-B wavefront scheduler (research tool; not product, not oracle).
 *
 * k_match is bound by VALU issue (profiles/r02: SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles, ~100 % of the SIMD's
 * issue slots at ≈44 wave-instructions per position), so what decides its speed is how many wave-instructions a policy
 * executes per position — a quantity that can be counted exactly on the CPU by replaying the policy on real data with
 * the per-phase instruction costs read off the ISA.  This emulator runs the exact FindLongestMatch walk
 * (C/DeflaterEngine.cs:474-612, as restated in szl_kernels_match*.hip) for every position of a sample under a
 * parametrised policy: K positions in flight per lane, phase thresholds, steps per visit, filter and compare widths.
 *
 *   gcc -O2 -o /tmp/sched_sim tools/sched_sim.c && /tmp/sched_sim sample.bin [key=value ...]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { WSIZE = 32768, MAX_DIST = 32506, MAX_MATCH = 258, TILE = 16384, WAVES = 16, LANES = 64, MAXK = 8 };
enum { NEED = 0, DONE = 1, QUICK = 2, VERIFY = 3 };

typedef struct {
    int mode, p, cl, best, left, off;
    uint32_t pb;
} Ctx;

static uint8_t *d;
static uint16_t *lk;
static size_t n;

/* policy / cost parameters */
static int K = 1, FTH = 16, VTH = 20, QSTEPS = 4, VSTEPS = 2, QKEEP = 0, VKEEP = 0, BATCH = 256, VBYTES = 4, FILT2 = 0, FIRSTV = 0;
static int max_chain = 128, nice = 128, POLICY = 0;
static double WV = 1.0, WF = 1.0, QFRAC = 0, VFRAC = 0;
static double C_CENSUS = 4, C_SWAP = 20, C_Q = 13, C_V = 22, C_VDONE = 15, C_F = 70, C_VISIT = 8;

typedef struct { double census, swap, quick, verify, fetch, visit; uint64_t qsteps, qlanes, vsteps, vlanes, fvis, flanes, visits; } Acc;

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

static int64_t base_of(int64_t s) { int64_t idx = s + 1; if (idx <= 65273) return 0; return ((idx - 65273 + 32767) >> 15) << 15; }

/* start a walk at absolute position p; returns 0 if there is nothing to search */
static int walk_start(Ctx *c, int64_t p) {
    int64_t rem = (int64_t)n - p;
    c->p = (int)p;
    if (rem < 3) return 0;
    uint32_t l0 = lk[p];
    if (l0 == 0) return 0;
    int64_t basem = base_of(p); /* abs0 == 0: window index 1 is position base */
    int64_t firstmin = p - MAX_DIST > basem ? p - MAX_DIST : basem;
    int64_t cand = p - l0;
    if (cand < firstmin) return 0;
    c->cl = (int)cand; c->best = 2; c->left = max_chain; c->off = 0; c->pb = d[p + 2];
    return 1;
}
static int64_t mincl_of(int64_t p) { int64_t basem = base_of(p); return p - (MAX_DIST - 1) > basem ? p - (MAX_DIST - 1) : basem; }

/* advance to the next candidate after the current one was rejected / compared; returns new mode */
static int walk_next(Ctx *c, int nicehit) {
    if (nicehit) return DONE;
    uint32_t l = lk[c->cl];
    int64_t c2 = l ? (int64_t)c->cl - l : -1000000;
    int left1 = c->left - 1;
    if (c2 < mincl_of(c->p) || left1 == 0) return DONE;
    c->left = left1; c->cl = (int)c2;
    return QUICK;
}

static int quick_step(Ctx *c) { /* returns new mode */
    int pass = d[c->cl + c->best] == c->pb;
    if (pass && FILT2 && c->best >= 3) pass = d[c->cl + c->best - 1] == d[c->p + c->best - 1];
    if (pass) { c->off = 0; return VERIFY; }
    return walk_next(c, 0);
}
static int verify_step(Ctx *c, int *completed) { /* compares VBYTES bytes */
    int64_t rem = (int64_t)n - c->p;
    int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
    int nc = rem < nice ? (int)rem : nice;
    int l = c->off, k = 0;
    while (k < VBYTES && l < cap && d[c->cl + l] == d[c->p + l]) { l++; k++; }
    int more = (k == VBYTES) && l < cap;
    c->off = l;
    *completed = !more;
    if (more) return VERIFY;
    int nicehit = 0;
    if (l > c->best) { c->best = l; nicehit = l >= nc; if (!nicehit) c->pb = d[c->p + l]; }
    c->off = 0;
    return walk_next(c, nicehit);
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: sched_sim file [key=value...]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    fseek(f, 0, SEEK_END); n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    d = malloc(n + 512); memset(d, 0, n + 512);
    if (fread(d, 1, n, f) != n) return 2;
    fclose(f);
    lk = calloc(n + 8, 2);
    links_build();
    for (int i = 2; i < argc; i++) {
        char *eq = strchr(argv[i], '=');
        if (!eq) continue;
        *eq = 0;
        double v = atof(eq + 1);
        const char *k = argv[i];
#define P(name, var) if (!strcmp(k, name)) { var = v; continue; }
        P("K", K) P("FTH", FTH) P("VTH", VTH) P("QSTEPS", QSTEPS) P("VSTEPS", VSTEPS) P("QKEEP", QKEEP) P("VKEEP", VKEEP)
        P("BATCH", BATCH) P("VBYTES", VBYTES) P("FILT2", FILT2) P("FIRSTV", FIRSTV) P("CHAIN", max_chain) P("NICE", nice)
        P("POLICY", POLICY) P("WV", WV) P("WF", WF) P("QFRAC", QFRAC) P("VFRAC", VFRAC) P("C_CENSUS", C_CENSUS) P("C_SWAP", C_SWAP) P("C_Q", C_Q) P("C_V", C_V) P("C_VDONE", C_VDONE) P("C_F", C_F) P("C_VISIT", C_VISIT)
        fprintf(stderr, "unknown key %s\n", k); return 2;
    }
    Acc A; memset(&A, 0, sizeof A);
    uint64_t positions = 0, lane_q = 0, lane_v = 0;
    size_t t_first = 4, ntile = n / TILE; /* skip the first tiles (short history) */
    if (ntile > t_first + 48) ntile = t_first + 48;
    static Ctx ctx[WAVES][LANES][MAXK];
    for (size_t t = t_first; t < ntile; t++) {
        int64_t t0 = (int64_t)t * TILE;
        int counter = 0;
        int wnext[WAVES], wend[WAVES], exhausted[WAVES], finished[WAVES];
        memset(ctx, 0, sizeof ctx);
        for (int w = 0; w < WAVES; w++) { wnext[w] = wend[w] = 0; exhausted[w] = finished[w] = 0; }
        int nfin = 0;
        while (nfin < WAVES) {
            for (int w = 0; w < WAVES; w++) {
                if (finished[w]) continue;
                Ctx (*C)[MAXK] = ctx[w];
                /* census: lanes that have some context in each state */
                int ni = 0, nv = 0, nq = 0;
                for (int l = 0; l < LANES; l++) {
                    int hi = 0, hv = 0, hq = 0;
                    for (int k = 0; k < K; k++) { int m = C[l][k].mode; if (m < QUICK) hi = 1; else if (m == VERIFY) hv = 1; else hq = 1; }
                    ni += hi; nv += hv; nq += hq;
                }
                A.census += C_CENSUS * (K > 1 ? 1.5 : 1.0); A.visit += C_VISIT; A.visits++;
                int phase;
                if (POLICY == 1) { /* the phase with the most lanes available (weighted); thresholds only as minimums */
                    double sq = nq, sv = nv * WV, sf = exhausted[w] ? -1 : ni * WF;
                    if (nq == 0 && nv == 0) phase = NEED;
                    else if (sf >= sq && sf >= sv && ni >= FTH) phase = NEED;
                    else if (sv >= sq && nv >= VTH) phase = VERIFY;
                    else if (nq > 0) phase = QUICK;
                    else phase = VERIFY;
                } else
                if ((ni >= FTH && !exhausted[w]) || (nq == 0 && nv == 0)) phase = NEED;
                else if (nv >= VTH || nq == 0) phase = VERIFY;
                else phase = QUICK;
                /* bring a context of the phase's state to slot 0 */
                int swapped = 0, v0 = 0;
                for (int l = 0; l < LANES; l++) {
                    int want0 = phase == NEED ? C[l][0].mode < QUICK : C[l][0].mode == phase;
                    if (want0) continue;
                    for (int k = 1; k < K; k++) {
                        int ok = phase == NEED ? C[l][k].mode < QUICK : C[l][k].mode == phase;
                        if (ok) { Ctx tmp = C[l][0]; C[l][0] = C[l][k]; C[l][k] = tmp; swapped = 1; break; }
                    }
                }
                if (K > 1 && (swapped || 1)) A.swap += C_SWAP * (K - 1);
                if (phase == NEED) {
                    /* retire all DONE contexts; hand out positions to slot 0 of idle lanes */
                    for (int l = 0; l < LANES; l++) for (int k = 0; k < K; k++) if (C[l][k].mode == DONE) C[l][k].mode = NEED;
                    A.fetch += C_F; A.fvis++;
                    if (!exhausted[w]) {
                        if (wnext[w] >= wend[w]) {
                            int base = counter; counter += BATCH;
                            wnext[w] = base < TILE ? base : TILE; wend[w] = base + BATCH < TILE ? base + BATCH : TILE;
                            if (wnext[w] >= wend[w]) exhausted[w] = 1;
                        }
                        if (!exhausted[w]) {
                            for (int l = 0; l < LANES && wnext[w] < wend[w]; l++) {
                                if (C[l][0].mode != NEED) continue;
                                int64_t p = t0 + wnext[w]++;
                                positions++; A.flanes++;
                                if (walk_start(&C[l][0], p)) C[l][0].mode = FIRSTV ? VERIFY : QUICK;
                            }
                        }
                    }
                    if (exhausted[w]) {
                        int busy = 0;
                        for (int l = 0; l < LANES; l++) for (int k = 0; k < K; k++) if (C[l][k].mode != NEED) busy = 1;
                        if (!busy) { finished[w] = 1; nfin++; }
                    }
                } else if (phase == VERIFY) {
                    for (int s = 0;; s++) {
                        int act = 0, done_any = 0;
                        for (int l = 0; l < LANES; l++) if (C[l][0].mode == VERIFY) {
                            int comp; act++; C[l][0].mode = verify_step(&C[l][0], &comp); done_any |= comp;
                        }
                        if (!act) break;
                        A.verify += C_V * (VBYTES == 8 ? 1.35 : 1.0) + (done_any ? C_VDONE : 0); A.vsteps++; A.vlanes += act; lane_v += act;
                        int still = 0;
                        for (int l = 0; l < LANES; l++) still += C[l][0].mode == VERIFY;
                        if (s == 0) v0 = act;
                        if (VFRAC > 0 ? (still < VFRAC * v0 || s + 1 >= VSTEPS) : (VKEEP > 0 ? (still < VKEEP) : (s + 1 >= VSTEPS))) break;
                    }
                } else {
                    for (int s = 0;; s++) {
                        int act = 0;
                        for (int l = 0; l < LANES; l++) if (C[l][0].mode == QUICK) { act++; C[l][0].mode = quick_step(&C[l][0]); }
                        if (!act) break;
                        A.quick += C_Q + (FILT2 ? 4 : 0); A.qsteps++; A.qlanes += act; lane_q += act;
                        int still = 0;
                        for (int l = 0; l < LANES; l++) still += C[l][0].mode == QUICK;
                        if (s == 0) v0 = act;
                        if (QFRAC > 0 ? (still < QFRAC * v0 || s + 1 >= QSTEPS) : (QKEEP > 0 ? (still < QKEEP) : (s + 1 >= QSTEPS))) break;
                    }
                }
            }
        }
    }
    double P = (double)positions;
    double tot = A.census + A.swap + A.quick + A.verify + A.fetch + A.visit;
    printf("K=%d FTH=%d VTH=%d Q=%d/%d V=%d/%d VB=%d F2=%d FV=%d | VALU/pos %.1f = quick %.1f verify %.1f fetch %.1f census %.1f swap %.1f visit %.1f | q lanes/step %.1f (%.2f steps/pos, %.1f lane-steps) v lanes/step %.1f (%.2f, %.1f) fetch lanes/visit %.1f visits/pos %.2f\n",
           K, FTH, VTH, QSTEPS, QKEEP, VSTEPS, VKEEP, VBYTES, FILT2, FIRSTV, tot / P, A.quick / P, A.verify / P, A.fetch / P, A.census / P, A.swap / P, A.visit / P,
           (double)A.qlanes / A.qsteps, A.qsteps / P, lane_q / P, (double)A.vlanes / A.vsteps, A.vsteps / P, lane_v / P,
           (double)A.flanes / A.fvis, A.visits / P);
    return 0;
}
