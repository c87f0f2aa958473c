/* CPU model for a faster exact Huffman tree build on the device (next step for k_block_build: one lane runs the reference's binary heap
 * with an LDS round trip per sift level, 0.37 ms per block, 2 ms of the 1 GiB pass and 35 % of a 64 KiB call).
 *
 * The reference (C/DeflaterHuffman.cs Tree.BuildTree :234-330) removes the two smallest nodes and inserts their parent, 2 sifts per merge:
 *   hole at the root -> move the smaller child up, level by level, to a leaf (ties: the LEFT child, `>` is strict); then sift the inserted
 *   value up from that leaf while the parent is strictly greater.
 * Tie-breaking is by heap position, so the tree — and with equal frequencies the code LENGTHS per symbol — can only be reproduced by
 * reproducing the heap array after every operation.  This model checks a level-parallel evaluation of the same operation:
 *   1. the root-to-leaf path of smaller children is read off "winner" bits (bit n = right child of node n strictly smaller) kept per
 *      internal node — no value is read to find the path;
 *   2. the path's values v_1 <= ... <= v_m are read at once (one lane per level);
 *   3. the inserted value lands at level j = #(v_k <= inserted); v_1..v_j move up one level (v_{j+1}.. never moved in effect);
 *   4. the winner bits of the path nodes at levels 0..j-1 (and of the parent of a removed last slot) are recomputed from their children.
 * Three LDS round trips per sift instead of two per level.  The program runs both forms on random frequency vectors (heavy ties included)
 * and compares heap[] / values after EVERY sift, then the resulting code lengths.
 *   gcc -O2 -o /tmp/heap_model tools/heap_model.c && /tmp/heap_model [trials] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define N 286
static uint64_t rng = 88172645463325252ull;
static uint32_t rnd(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 32); }

typedef struct { int heap[N]; int hval[N]; int len; } Heap;

/* ---- reference form: insert (last, lastVal) into the heap whose root is a hole */
static void sift_ref(Heap *h, int last, int lastVal) {
    int ppos = 0, path = 1;
    while (path < h->len) {
        if (path + 1 < h->len && h->hval[path] > h->hval[path + 1]) path++;
        h->heap[ppos] = h->heap[path]; h->hval[ppos] = h->hval[path];
        ppos = path; path = path * 2 + 1;
    }
    while ((path = ppos) > 0) {
        ppos = (path - 1) / 2;
        if (!(h->hval[ppos] > lastVal)) break;
        h->heap[path] = h->heap[ppos]; h->hval[path] = h->hval[ppos];
    }
    h->heap[path] = last; h->hval[path] = lastVal;
}

/* ---- level-parallel form */
typedef struct { Heap h; uint8_t win[N]; } PHeap;      /* win[n] = 1: the right child of n exists and is strictly smaller than the left */
static void win_recompute(PHeap *p, int n) {
    const int l = 2 * n + 1, r = l + 1;
    p->win[n] = (r < p->h.len && p->h.hval[l] > p->h.hval[r]) ? 1 : 0;
}
static void win_init(PHeap *p) { for (int n = 0; n < N; n++) p->win[n] = 0; for (int n = 0; 2 * n + 1 < p->h.len; n++) win_recompute(p, n); }
static void sift_par(PHeap *p, int last, int lastVal) {
    int pos[16], m = 0;                   /* pos[k] = heap position of the path node at level k (pos[0] = 0 = the hole) */
    pos[0] = 0;
    for (int n = 0; 2 * n + 1 < p->h.len; ) { n = 2 * n + 1 + p->win[n]; pos[++m] = n; }      /* step 1: bits only */
    int v[16], id[16];
    for (int k = 1; k <= m; k++) { v[k] = p->h.hval[pos[k]]; id[k] = p->h.heap[pos[k]]; }    /* step 2: one read per level, together */
    int j = 0;
    for (int k = 1; k <= m; k++) j += (v[k] <= lastVal);                                        /* step 3: a ballot + popcount */
    for (int k = 1; k <= j; k++) { p->h.heap[pos[k - 1]] = id[k]; p->h.hval[pos[k - 1]] = v[k]; }
    p->h.heap[pos[j]] = last; p->h.hval[pos[j]] = lastVal;
    for (int k = 0; k < j; k++) win_recompute(p, pos[k]);                                       /* step 4 (the node at level j keeps its children) */
    if (j > 0 || m > 0) { /* the node that received the inserted value: its own children did not change, but ITS value did: its parent was recomputed above when j > 0 */ }
}

/* the whole merge loop of BuildTree on a prepared heap; returns the number of nodes; childs as in the reference */
static int build(int nleaf, const int *leafval, int parallel, short *childs, int *check_fail) {
    static PHeap P; static Heap R;
    Heap *h = parallel ? &P.h : &R;
    h->len = nleaf;
    for (int i = 0; i < nleaf; i++) { h->heap[i] = i; h->hval[i] = leafval[i]; }
    static int values[2 * N];
    for (int i = 0; i < nleaf; i++) { values[i] = leafval[i]; childs[2 * i] = (short)i; childs[2 * i + 1] = -1; }
    if (parallel) win_init(&P);
    int numNodes = nleaf;
    (void)check_fail;
    do {
        int first = h->heap[0], firstVal = h->hval[0];
        --h->len;
        int last = h->heap[h->len], lastVal = h->hval[h->len];
        if (parallel) { if (h->len > 0) win_recompute(&P, (h->len - 1) / 2); sift_par(&P, last, lastVal); } else sift_ref(h, last, lastVal);
        int second = h->heap[0], secondVal = h->hval[0];
        last = numNodes++;
        childs[2 * last] = (short)first; childs[2 * last + 1] = (short)second;
        int d1 = firstVal & 0xff, d2 = secondVal & 0xff, mind = d1 < d2 ? d1 : d2;
        lastVal = firstVal + secondVal - mind + 1;
        values[last] = lastVal;
        if (parallel) sift_par(&P, last, lastVal); else sift_ref(h, last, lastVal);
    } while (h->len > 1);
    return numNodes;
}

int main(int argc, char **argv) {
    long trials = argc > 1 ? atol(argv[1]) : 200000;
    long bad = 0, sifts = 0, levels_ref = 0;
    for (long t = 0; t < trials; t++) {
        int nleaf = 2 + rnd() % (N - 1);
        int leafval[N];
        const int mode = rnd() % 4;       /* 0: wide range, 1: tiny range (heavy ties), 2: all equal, 3: geometric */
        /* the leaves must come as a valid heap, as BuildTree's insertion loop leaves them (:246-262): build it the same way */
        int freqs[N], heap[N], hl = 0;
        for (int i = 0; i < nleaf; i++) freqs[i] = mode == 0 ? 1 + rnd() % 16384 : mode == 1 ? 1 + rnd() % 3 : mode == 2 ? 7 : 1 + (int)(16384.0 / (1 + rnd() % 4096));
        for (int n = 0; n < nleaf; n++) {
            int pos = hl++, ppos;
            while (pos > 0 && freqs[heap[ppos = (pos - 1) / 2]] > freqs[n]) { heap[pos] = heap[ppos]; pos = ppos; }
            heap[pos] = n;
        }
        for (int i = 0; i < nleaf; i++) leafval[i] = freqs[heap[i]] << 8;
        short ca[4 * N], cb[4 * N];
        int na = build(nleaf, leafval, 0, ca, NULL), nb = build(nleaf, leafval, 1, cb, NULL);
        sifts += 2L * (nleaf - 1);
        for (int l = nleaf; l > 1; l >>= 1) levels_ref++;
        if (na != nb || memcmp(ca, cb, sizeof(short) * 2 * (size_t)na)) { bad++; if (bad < 5) printf("MISMATCH at trial %ld (nleaf %d, mode %d)\n", t, nleaf, mode); }
    }
    printf("%ld trees (2..%d leaves; wide, tiny-range, all-equal and geometric frequencies), %ld sifts: %ld trees differ between the reference's heap and the level-parallel form\n",
           trials, N, sifts, bad);
    return bad != 0;
}
