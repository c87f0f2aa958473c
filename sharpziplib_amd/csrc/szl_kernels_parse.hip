// szl_kernels_parse.hip — stage C: the lazy-match parse of DeflateSlow (C/DeflaterEngine.cs:741-855)
// resolved as a path in a functional graph.
//
// State of the reference loop at an iteration start = (strstart, prevAvailable, matchLen, matchStart).
// With the match tables of stage B the loop body is a table lookup; an iteration entered with
// matchLen == 2 ("clean") has a future that does not depend on how it was reached, so two parses
// that are both clean at the same position are identical from there on.  Ranges of C_RANGE positions
// are therefore parsed speculatively from a clean state (k_spec), the predecessor's exit is walked
// into each range until it lands on that speculative path (k_fix), the rare ranges where that does
// not happen are chained sequentially (k_resolve), and the true path is replayed once to emit tokens.
#include <hip/hip_runtime.h>
#include "szl_internal.h"

namespace szl {

__device__ __forceinline__ int64_t base_of_c(int64_t s_abs) {
    int64_t idx = s_abs + 1;
    if (idx <= 65273) return 0;
    return ((idx - 65273 + 32767) >> 15) << 15;
}

struct ParseCtx {
    const uint8_t *d;      // stream buffer
    const uint16_t *lk;    // links
    const uint32_t *m2;    // M2 (full-budget search from matchLen 2)
    const uint32_t *mq;    // Mq (state after max_chain>>2 candidates); read only when L >= good
    int64_t seg_end;       // end of the parse ranges
    int64_t look_end;      // end of the input the engine has seen (lookahead); == seg_end except in the windows of a long stream
    int64_t tab_end;       // match-table entries exist for positions < tab_end
    int64_t abs0;
    LevelParams P;
    const SegDev *sg;      // the segment in device memory (parameter switches, read only when nsw != 0)
    uint32_t nsw;
    // parameters in force for the DeflateSlow iteration that starts at buffer position x
    __device__ __forceinline__ LevelParams par(int64_t x) const {
        if (nsw == 0) return P;
        LevelParams r = P;
        const int64_t *sp = sg->sw_pos; const LevelParams *sP = sg->sw_P;
        for (uint32_t k = 0; k < nsw; k++) if (x >= sp[k]) r = sP[k];
        return r;
    }
};

// FindLongestMatch entered with matchLen = L >= niceLength' (C/DeflaterEngine.cs:474-612): the first
// candidate among the first max_chain>>2 that is strictly longer wins (it is >= niceLength, :604).
// Rare (measured per call in szl_timing.fallback_walks) — walks global memory.
__device__ uint32_t slow_walk(const ParseCtx &c, int64_t p, int L, unsigned long long *fallbacks) {
    if (fallbacks) atomicAdd(fallbacks, 1ull);
    const int64_t rem = c.look_end - p;
    uint32_t l0 = c.lk[p];
    if (l0 == 0) return 0;
    const int64_t basem = base_of_c(c.abs0 + p) - c.abs0;
    int64_t cand = p - l0;
    int64_t firstmin = p - MAX_DIST > basem ? p - MAX_DIST : basem;
    if (cand < firstmin) return 0;
    const int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
    if (L >= cap) return 0;
    const int64_t minc = p - (MAX_DIST - 1) > basem ? p - (MAX_DIST - 1) : basem;
    int budget = c.par(p).max_chain >> 2;
    for (;;) {
        int l = 0;
        if (c.d[cand + L] == c.d[p + L]) { // quick reject :505
            while (l < cap && c.d[cand + l] == c.d[p + l]) l++;
        }
        if (l > L) return (uint32_t)l | ((uint32_t)(p - cand) << 16);
        uint32_t lnk = c.lk[cand];
        if (lnk == 0) return 0;
        int64_t c2 = cand - lnk;
        if (c2 < minc) return 0;
        if (--budget == 0) return 0;
        cand = c2;
    }
}

// Stage B can evaluate positions on demand (k_match_lazy): an entry still holding M_UNSET was never reached by any of
// its tile's walkers — e.g. the few positions where the true path enters a tile from its predecessor.  The parse then
// runs the same FindLongestMatch walk here, out of global memory (same rules as k_match; returns M2 and Mq).
__device__ void eval_global(const ParseCtx &c, int64_t p, uint32_t &m2, uint32_t &mq, unsigned long long *count) {
    if (count) atomicAdd(count, 1ull);
    m2 = 0; mq = 0;
    const LevelParams Px = c.par(p);
    const int64_t rem = c.look_end - p;
    if (rem < MIN_MATCH || Px.strategy == 2) return;       // :780, HuffmanOnly :786
    const uint32_t l0 = c.lk[p];
    if (l0 == 0) return;
    const int64_t basem = base_of_c(c.abs0 + p) - c.abs0;
    int64_t cand = p - l0;
    const int64_t firstmin = p - MAX_DIST > basem ? p - MAX_DIST : basem; // :788, :450-461
    if (cand < firstmin) return;
    const int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
    const int nice = rem < (int64_t)Px.nice ? (int)rem : Px.nice;
    const int64_t minc = p - (MAX_DIST - 1) > basem ? p - (MAX_DIST - 1) : basem; // :609
    const int snapleft = Px.max_chain - (Px.max_chain >> 2);
    int best = 2, left = Px.max_chain;
    for (;;) {
        int l = 0;
        if (c.d[cand + best] == c.d[p + best]) { // quick reject :505
            while (l < cap && c.d[cand + l] == c.d[p + l]) l++;
        }
        if (l > best) {
            best = l;
            m2 = (uint32_t)l | ((uint32_t)(p - cand) << 16);
            if (left > snapleft) mq = m2;
            if (l >= nice) return;
        }
        const uint32_t lnk = c.lk[cand];
        if (lnk == 0) return;
        const int64_t c2 = cand - lnk;
        if (c2 < minc) return;
        if (--left == 0) return;
        cand = c2;
    }
}

// One iteration of the DeflateSlow loop body at position x with pending match (L,D) (L==0: none).
// Returns the token emitted by this iteration (0xFFFFFFFF = none) and its input start via *tpos.
// `A` supplies M2 entries and literal bytes: straight from global memory (GlobalAcc) or from a window a wave
// staged in LDS (WinAcc, k_spec_win / k_emit_win).
struct GlobalAcc {
    const uint32_t *m2p, *mqp; const uint8_t *dp;
    // (entries are packed, szl_internal.h: M2 in the word itself, Mq by its code — the same, empty, or in the mq array)
    __device__ __forceinline__ uint32_t m2(int64_t x) const { const uint32_t e = m2p[x]; return e == M_UNSET ? e : mt_m2(e); }
    __device__ __forceinline__ uint32_t mq(int64_t x) const { const uint32_t e = m2p[x]; const uint32_t k = mt_code(e); return k == 0u ? mt_m2(e) : (k == 1u ? 0u : mqp[x]); }
    __device__ __forceinline__ uint32_t lit(int64_t x) const { return dp[x]; }
};

template <typename A>
__device__ __forceinline__ uint32_t parse_step(const ParseCtx &c, const A &acc, int64_t &x, int &L, int &D, int64_t *tpos,
                                               bool want_lit, unsigned long long *fallbacks) {
    const LevelParams Px = c.par(x);   // the parameters this iteration runs with (SetLevel / SetStrategy in mid-segment)
    uint32_t e2 = acc.m2(x), eq = 0;
    const bool unset = e2 == M_UNSET;
    if (unset) eval_global(c, x, e2, eq, fallbacks ? fallbacks + 6 : nullptr);
    if (L == 0) {
        int len = (int)(e2 & 0xFFFF), dist = (int)(e2 >> 16);
        if (len != 0 && len <= 5 && (Px.strategy == 1 || (len == MIN_MATCH && dist > TOO_FAR))) len = 0; // :794-797
        if (len == 0) { // literal step :830-839 (tallied by the next iteration / final flush :752)
            uint32_t t = want_lit ? acc.lit(x) : 0u;
            *tpos = x;
            x += 1;
            return t;
        }
        L = len; D = dist;
        x += 1;
        return 0xFFFFFFFFu;
    }
    // lazy evaluation at x: is there a strictly longer match than the one found at x-1 ?
    const int64_t rem = c.look_end - x;
    uint32_t better = 0;
    if (rem >= MIN_MATCH) {
        const int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
        if (L < cap) {
            const int nice = rem < (int64_t)Px.nice ? (int)rem : Px.nice;
            uint32_t cand;
            if (L < Px.good) cand = e2;
            else if (L < nice) cand = unset ? eq : acc.mq(x);              // chainLength >>= 2 (:495)
            else cand = slow_walk(c, x, L, fallbacks);
            if ((int)(cand & 0xFFFF) > L && !(Px.strategy == 1 && (cand & 0xFFFF) <= 5)) better = cand;
        }
    }
    if (better) { // previous position becomes a literal (:830-835)
        uint32_t t = want_lit ? acc.lit(x - 1) : 0u;
        *tpos = x - 1;
        L = (int)(better & 0xFFFF); D = (int)(better >> 16);
        x += 1;
        return t;
    }
    // "previous match was better" :802-828
    uint32_t t = ((uint32_t)D << 16) | (uint32_t)L;
    *tpos = x - 1;
    x = x - 1 + L;
    L = 0;
    return t;
}

__device__ __forceinline__ uint32_t parse_step(const ParseCtx &c, int64_t &x, int &L, int &D, int64_t *tpos,
                                               bool want_lit, unsigned long long *fallbacks) {
    const GlobalAcc acc{c.m2, c.mq, c.d};
    return parse_step(c, acc, x, L, D, tpos, want_lit, fallbacks);
}

__device__ __forceinline__ uint32_t find_seg(const SegDev *segs, uint32_t nseg, uint64_t r) {
    // largest s with segs[s].range_off <= r  (segments with zero ranges share an offset with their successor)
    uint32_t lo = 0, hi = nseg - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) >> 1;
        if (segs[mid].range_off <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ ParseCtx make_ctx(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev &s,
                                             LevelParams P, const SegDev *sg) {
    ParseCtx c;
    c.d = in + s.buf_off; c.lk = link + s.buf_off; c.m2 = mtab.m2 + s.buf_off; c.mq = mtab.mq + s.buf_off;
    c.seg_end = s.seg_end; c.look_end = s.look_end; c.abs0 = (int64_t)s.abs0; c.P = P; c.sg = sg; c.nsw = s.sw_cnt;
    // windows of a long stream keep a short tail of (unset) table entries past their parse end for the walk that crosses it
    c.tab_end = s.look_end > s.seg_end ? (s.seg_end + (int64_t)C_WIN_HALO < s.look_end ? s.seg_end + (int64_t)C_WIN_HALO : s.look_end) : s.seg_end;
    return c;
}

// C1: speculative walk of each range from a clean state at its first position.
#if SZL_LAB   // (laboratory library only)
__global__ __launch_bounds__(256) void k_spec(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs,
                                              uint32_t nseg, uint64_t nranges, LevelParams P, RangeDev *ranges,
                                              uint32_t *visited, unsigned long long *counters) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nranges) return;
    uint32_t si = find_seg(segs, nseg, r);
    const SegDev s = segs[si];
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs + si);
    uint64_t lr = r - s.range_off;
    int64_t rs = s.seg_start + (int64_t)lr * (int64_t)s.range_len;
    int64_t re = rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end;
    // each range owns whole words of its segment's bitmap (C_RANGE is a multiple of 32): plain stores, no atomics
    uint32_t *vis = visited + s.vis_word_off;
    int64_t x = rs;
    int L = 0, D = 0;
    uint32_t count = 0;
    int64_t tp;
    uint64_t cw = (uint64_t)(rs - s.seg_start) >> 5;
    uint32_t acc = 0;
    for (;;) {
        if (L == 0) {
            if (x >= re) break;
            uint64_t bit = (uint64_t)(x - s.seg_start);
            if ((bit >> 5) != cw) { if (acc) vis[cw] = acc; cw = bit >> 5; acc = 0; }
            acc |= 1u << (bit & 31);
        }
        uint32_t t = parse_step(c, x, L, D, &tp, false, counters + 1);
        count += (t != 0xFFFFFFFFu);
    }
    if (acc) vis[cw] = acc;
    RangeDev rd;
    rd.exit_spec = x; rd.exit_true = x; rd.entry = rs; rd.spec_count = count; rd.true_count = count; rd.merged = 1; rd.pad = 0;
    ranges[r] = rd;
}
#endif   // SZL_LAB

__device__ __forceinline__ bool is_visited(const uint32_t *vis /* segment bitmap */, uint64_t i) { return (vis[i >> 5] >> (i & 31)) & 1u; }

// Walk from clean position `entry` into range [rs,re): returns through `merged`, `exit`, `count` where count =
// tokens owned by the true nodes in [entry, re).  If the walk lands on the range's speculative path at y the
// remainder equals the speculative parse: count = fix + spec_count - (tokens of spec nodes in [rs,y)).
__device__ void fixup_range(const ParseCtx &c, const uint32_t *vis, int64_t seg_start, int64_t rs, int64_t re,
                            int64_t entry, uint32_t spec_count, int64_t exit_spec, uint32_t *out_merged,
                            int64_t *out_exit, uint32_t *out_count, unsigned long long *fallbacks) {
    int64_t x = entry, tp;
    int L = 0, D = 0;
    uint32_t fix = 0;
    bool merged = false;
    for (;;) {
        if (L == 0) {
            if (x >= re) break;
            if (x >= rs && is_visited(vis, (uint64_t)(x - seg_start))) { merged = true; break; }
        }
        uint32_t t = parse_step(c, x, L, D, &tp, false, fallbacks);
        fix += (t != 0xFFFFFFFFu);
    }
    if (!merged) { *out_merged = 0; *out_exit = x; *out_count = fix; return; }
    const int64_t y = x;
    uint32_t skip = 0;
    x = rs; L = 0; D = 0;
    for (;;) {
        if (L == 0 && x >= y) break;
        uint32_t t = parse_step(c, x, L, D, &tp, false, fallbacks);
        skip += (t != 0xFFFFFFFFu);
    }
    *out_merged = 1; *out_exit = exit_spec; *out_count = fix + spec_count - skip;
}

// C2: assume the predecessor's true exit is its speculative exit.
__global__ __launch_bounds__(256) void k_fix(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs,
                                             uint32_t nseg, uint64_t nranges, LevelParams P, RangeDev *ranges,
                                             const uint32_t *visited, unsigned long long *counters, uint32_t *bad_slot,
                                             uint64_t *bad_range) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nranges) return;
    uint32_t si = find_seg(segs, nseg, r);
    const SegDev s = segs[si];
    uint64_t lr = r - s.range_off;
    bad_slot[r] = 0xFFFFFFFFu;
    if (lr == 0) return; // first range of a segment: entry = range start, speculative parse is the true one
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs + si);
    int64_t rs = s.seg_start + (int64_t)lr * (int64_t)s.range_len;
    int64_t re = rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end;
    const int64_t entry = ranges[r - 1].exit_spec;
    uint32_t merged, count;
    int64_t ex;
    fixup_range(c, visited + s.vis_word_off, s.seg_start, rs, re, entry, ranges[r].spec_count, ranges[r].exit_spec, &merged, &ex, &count,
                counters + 1);
    ranges[r].entry = entry;
    ranges[r].merged = merged;
    ranges[r].exit_true = ex;
    ranges[r].true_count = count;
    if (!merged) { // remember it: if there are many, their entry->exit maps are built instead of chaining walks
        unsigned long long slot = atomicAdd(counters + 0, 1ull);
        bad_slot[r] = (uint32_t)slot;
        bad_range[slot] = r;
    }
}

// A range that merged under k_fix's assumption but FOLLOWS a never-merging one will be entered somewhere else (the true exit of
// its predecessor is only known in the chain), and without a map the chain walks it with one lane, ~100 us a time, one after the
// other — mixed data (stretches of zeros or periodic bytes between ordinary ones) has hundreds of them.  They get a map too.
__global__ __launch_bounds__(256) void k_flag_followers(const SegDev *segs, uint32_t nseg, uint64_t nranges, const RangeDev *ranges,
                                                        unsigned long long *counters, uint32_t *bad_slot, uint64_t *bad_range) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nranges || r == 0) return;
    const uint32_t si = find_seg(segs, nseg, r);
    if (r - segs[si].range_off < 2) return;                  // (range 0 of a segment is never flagged by k_fix)
    if (ranges[r].merged != 0 && ranges[r - 1].merged == 0 && bad_slot[r - 1] != 0xFFFFFFFFu) {
        unsigned long long slot = atomicAdd(counters + 0, 1ull);
        bad_slot[r] = (uint32_t)slot;
        bad_range[slot] = r;
    }
}

// C3: one wavefront per segment chains the ranges whose assumed entry was wrong (sequential; rare).
__global__ __launch_bounds__(64) void k_resolve(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs,
                                                uint32_t nseg, LevelParams P, RangeDev *ranges, const uint32_t *visited,
                                                unsigned long long *counters) {
    if (counters[0] == 0) return; // every range merged: nothing to do
    uint32_t si = blockIdx.x;
    if (si >= nseg) return;
    const SegDev s = segs[si];
    if (s.range_cnt < 2) return;
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs + si);
    const int lane = threadIdx.x;
    RangeDev *R = ranges + s.range_off;
    // Invariant: ranges [0, k) are final.  Range k is final iff its assumed entry == exit_true of k-1.
    for (uint32_t k0 = 1; k0 < s.range_cnt; k0 += 64) {
        uint32_t k = k0 + lane;
        for (;;) {
            bool bad = false;
            if (k < s.range_cnt) bad = R[k].entry != R[k - 1].exit_true;
            uint64_t m = __ballot(bad);
            if (m == 0) break;
            int l = __builtin_ctzll(m);
            if (lane == l) { // redo this range from the true exit of its predecessor
                int64_t rs = s.seg_start + (int64_t)k * (int64_t)s.range_len;
                int64_t re = rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end;
                const int64_t entry = R[k - 1].exit_true;
                uint32_t merged, count;
                int64_t ex;
                fixup_range(c, visited + s.vis_word_off, s.seg_start, rs, re, entry, R[k].spec_count, R[k].exit_spec, &merged, &ex, &count,
                            counters + 1);
                R[k].entry = entry; R[k].merged = merged; R[k].exit_true = ex; R[k].true_count = count;
                __threadfence();
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- Ranges that never merge (all-zero / period-258 data lock the two parses into different phases).
// For each such range the map "entry position -> (exit position, tokens)" is computed for every possible entry in
// parallel (a true entry lies < 513 positions past the range start: a node is at most 255 lazy literals + a 258 match),
// then the ranges are chained by table lookups instead of by re-walking them one after the other.
enum : int { X_W = 576 }; // entries per map (multiple of 64, > 513)

__global__ __launch_bounds__(64) void k_exitmap(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs,
                                                uint32_t nseg, LevelParams P, const uint64_t *bad_range, uint64_t nbad,
                                                uint16_t *exmap, uint16_t *cnmap, unsigned long long *counters,
                                                const RangeDev *ranges, const uint32_t *visited) {
    const uint64_t slot = blockIdx.x / (X_W / 64);
    const int j = (int)(blockIdx.x % (X_W / 64)) * 64 + (int)threadIdx.x;
    if (slot >= nbad) return;
    const uint64_t r = bad_range[slot];
    const uint32_t si = find_seg(segs, nseg, r);
    const SegDev s = segs[si];
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs + si);
    const uint64_t lr = r - s.range_off;
    const int64_t rs = s.seg_start + (int64_t)lr * (int64_t)s.range_len;
    const int64_t re = rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end;
    int64_t x = rs + j, tp;
    uint32_t count = 0;
    if (ranges[r].merged != 0) {      // a follower (k_flag_followers): walk until the range's speculative path is met, as k_fix does
        uint32_t merged = 0; int64_t ex = x;
        if (x < s.seg_end) fixup_range(c, visited + s.vis_word_off, s.seg_start, rs, re, x, ranges[r].spec_count, ranges[r].exit_spec, &merged, &ex, &count, counters + 1);
        x = ex;
    } else
    if (x < s.seg_end) {
        int L = 0, D = 0;
        for (;;) {
            if (L == 0 && x >= re) break;
            uint32_t t = parse_step(c, x, L, D, &tp, false, counters + 1);
            count += (t != 0xFFFFFFFFu);
        }
    }
    int64_t off = x - re;
    if (off < 0) off = 0;            // entry beyond the segment end: nothing to do
    exmap[slot * X_W + j] = (uint16_t)(off > 65535 ? 65535 : off);
    cnmap[slot * X_W + j] = (uint16_t)(count > 65535 ? 65535 : count);
}

enum : int { CH_RANGES = 32 };
// The chain over the ranges of a segment: range k is entered where range k-1 was left.  Where the entry is the one k_fix assumed
// the range's exit stands; where it is not, the exit map of the range (k_exitmap: exit as a function of the entry offset) gives it,
// and a range without a map is walked (rare).  With all-zero or periodic input EVERY range has a map and the chain is as long as the
// stream: 64 Ki ranges for 64 MiB, one global round trip each — 67 ms of a 70 ms call.  Hence two levels for a long segment:
//   k_chain_maps   every chunk of CH_RANGES ranges: exit of the chunk as a function of the entry offset into it (all X_W of them,
//                  through the chunk's maps in LDS; an entry that would need a walk is marked);
//   k_chain_top    one wavefront: chunk to chunk through those chunk maps (a marked entry runs the chunk's ranges one by one);
//   k_chain_apply  every chunk again, from its true entry: the ranges' records.
struct ChainLds {
    uint16_t ex[CH_RANGES][X_W];
    int64_t entry[CH_RANGES], exit[CH_RANGES];
    uint32_t slot[CH_RANGES], chg[CH_RANGES], cnt[CH_RANGES], mrg[CH_RANGES];
};
// stage the chunk's range records and exit maps (whole wavefront)
__device__ __forceinline__ void chain_stage(ChainLds &S, const RangeDev *R, const uint32_t *slots, const uint16_t *exmap, uint32_t k0, uint32_t nk, int lane) {
    if ((uint32_t)lane < nk) {
        S.entry[lane] = R[k0 + lane].entry; S.exit[lane] = R[k0 + lane].exit_true;
        S.slot[lane] = slots[k0 + lane]; S.chg[lane] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    {   // all loads of the chunk are issued before the first store (map by map, every map was a global round trip of its own)
        constexpr int PER = X_W * 2 / 16, TOT = CH_RANGES * PER, NL = (TOT + 63) / 64;
        uint4 v[NL];
#pragma unroll
        for (int t = 0; t < NL; t++) {
            const int q = lane + 64 * t, i = q / PER;
            v[t] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (q < TOT && (uint32_t)i < nk && S.slot[i] != 0xFFFFFFFFu) v[t] = ((const uint4 *)(exmap + (uint64_t)S.slot[i] * X_W))[q - i * PER];
        }
#pragma unroll
        for (int t = 0; t < NL; t++) { const int q = lane + 64 * t; if (q < TOT) ((uint4 *)&S.ex[0][0])[q] = v[t]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}
// the ranges [k0, k0 + nk) of segment s from the entry prev_exit on: updates their records, returns the exit of the last one
__device__ int64_t chain_chunk(ChainLds &S, ParseCtx &c, const SegDev &s, RangeDev *R, const uint32_t *slots, const uint32_t *visited,
                               const uint16_t *exmap, const uint16_t *cnmap, unsigned long long *counters, uint32_t k0, uint32_t nk,
                               int64_t prev_exit, int lane) {
    chain_stage(S, R, slots, exmap, k0, nk, lane);
        if (lane == 0) {
            for (uint32_t i = 0; i < nk; i++) {
                const uint32_t k = k0 + i;
                const int64_t rs = s.seg_start + (int64_t)k * (int64_t)s.range_len;
                const int64_t re = rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end;
                const int64_t e = prev_exit;
                if (e == S.entry[i]) { prev_exit = S.exit[i]; continue; }           // the assumption of k_fix holds
                S.entry[i] = e;
                const int64_t j = e - rs;
                if (S.slot[i] != 0xFFFFFFFFu && j >= 0 && j < X_W && S.ex[i][j] != 65535) {
                    prev_exit = (e >= re ? e : re + S.ex[i][j]);
                    S.exit[i] = prev_exit; S.chg[i] = 1;
                } else { // unexpected entry into a range without a map: walk it (rare)
                    uint32_t merged, count; int64_t ex;
                    fixup_range(c, visited + s.vis_word_off, s.seg_start, rs, re, e, R[k].spec_count, R[k].exit_spec, &merged, &ex, &count, counters + 1);
                    prev_exit = ex; S.exit[i] = ex; S.cnt[i] = count; S.mrg[i] = merged; S.chg[i] = 2;
                }
            }
        }
        prev_exit = ((int64_t)__builtin_amdgcn_readfirstlane((int)(prev_exit >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)prev_exit);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if ((uint32_t)lane < nk && S.chg[lane]) {
            const uint32_t k = k0 + lane;
            const int64_t rs = s.seg_start + (int64_t)k * (int64_t)s.range_len;
            uint32_t cnt = S.cnt[lane], mrg = S.mrg[lane];
            if (S.chg[lane] == 1) {
                const int64_t j = S.entry[lane] - rs;
                cnt = S.entry[lane] >= (rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end) ? 0u : cnmap[(uint64_t)S.slot[lane] * X_W + j];
                mrg = 0;
            }
            R[k].entry = S.entry[lane]; R[k].exit_true = S.exit[lane]; R[k].true_count = cnt; R[k].merged = mrg;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    return prev_exit;
}

__global__ __launch_bounds__(64) void k_chain(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs,
                                              uint32_t nseg, LevelParams P, RangeDev *ranges, const uint32_t *visited,
                                              const uint32_t *bad_slot, const uint16_t *exmap, const uint16_t *cnmap,
                                              unsigned long long *counters) {
    __shared__ ChainLds S;
    const uint32_t si = blockIdx.x;
    if (si >= nseg) return;
    const SegDev s = segs[si];
    if (s.range_cnt < 2) return;
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs + si);
    const int lane = threadIdx.x;
    RangeDev *R = ranges + s.range_off;
    const uint32_t *slots = bad_slot + s.range_off;
    int64_t prev_exit = R[0].exit_true;
    for (uint32_t k0 = 1; k0 < s.range_cnt; k0 += CH_RANGES) {
        const uint32_t nk = s.range_cnt - k0 < CH_RANGES ? s.range_cnt - k0 : CH_RANGES;
        prev_exit = chain_chunk(S, c, s, R, slots, visited, exmap, cnmap, counters, k0, nk, prev_exit, lane);
    }
}

// (single long segment) chunk map: cm[chunk * X_W + j] = exit of the chunk relative to the start of the range behind it, for the
// entry offset j into the chunk's first range; 0xFFFFFFFF = some range of the chunk needs a walk for that entry
__global__ __launch_bounds__(64) void k_chain_maps(const SegDev *segs, const RangeDev *ranges, const uint32_t *bad_slot, const uint16_t *exmap,
                                                   uint32_t *cm) {
    __shared__ ChainLds S;
    const SegDev s = segs[0];
    const uint32_t k0 = 1 + blockIdx.x * CH_RANGES;
    if (k0 >= s.range_cnt) return;
    const uint32_t nk = s.range_cnt - k0 < CH_RANGES ? s.range_cnt - k0 : CH_RANGES;
    const int lane = threadIdx.x;
    chain_stage(S, ranges + s.range_off, bad_slot + s.range_off, exmap, k0, nk, lane);
    const int64_t rs0 = s.seg_start + (int64_t)k0 * (int64_t)s.range_len;
    const int64_t rs_next = s.seg_start + (int64_t)(k0 + nk) * (int64_t)s.range_len;
    for (int j0 = lane; j0 < X_W; j0 += 64) {
        int64_t e = rs0 + j0;
        bool ok = true;
        for (uint32_t i = 0; i < nk && ok; i++) {
            const int64_t rs = rs0 + (int64_t)i * (int64_t)s.range_len;
            const int64_t re = rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end;
            if (e == S.entry[i]) { e = S.exit[i]; continue; }
            const int64_t j = e - rs;
            if (S.slot[i] != 0xFFFFFFFFu && j >= 0 && j < X_W && S.ex[i][j] != 65535) e = e >= re ? e : re + S.ex[i][j];
            else ok = false;
        }
        const int64_t off = e - rs_next;
        cm[(uint64_t)blockIdx.x * X_W + j0] = (ok && off >= -(int64_t)0x40000000 && off < (int64_t)0x40000000) ? (uint32_t)(int32_t)off : 0xFFFFFFFFu;
    }
}
// one wavefront: the entry of every chunk (ein[chunk]); a chunk whose map does not cover the entry is run range by range
__global__ __launch_bounds__(64) void k_chain_top(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, LevelParams P,
                                                  RangeDev *ranges, const uint32_t *visited, const uint32_t *bad_slot, const uint16_t *exmap,
                                                  const uint16_t *cnmap, unsigned long long *counters, const uint32_t *cm, int64_t *ein) {
    __shared__ ChainLds S;
    const SegDev s = segs[0];
    if (s.range_cnt < 2) return;
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs);
    const int lane = threadIdx.x;
    RangeDev *R = ranges + s.range_off;
    const uint32_t *slots = bad_slot + s.range_off;
    int64_t prev_exit = R[0].exit_true;
    uint32_t ch = 0;
    for (uint32_t k0 = 1; k0 < s.range_cnt; k0 += CH_RANGES, ch++) {
        const uint32_t nk = s.range_cnt - k0 < CH_RANGES ? s.range_cnt - k0 : CH_RANGES;
        if (lane == 0) ein[ch] = prev_exit;
        const int64_t rs0 = s.seg_start + (int64_t)k0 * (int64_t)s.range_len;
        const int64_t rs_next = s.seg_start + (int64_t)(k0 + nk) * (int64_t)s.range_len;
        const int64_t j = prev_exit - rs0;
        uint32_t v = 0xFFFFFFFFu;
        if (j >= 0 && j < X_W) v = __builtin_nontemporal_load(cm + (uint64_t)ch * X_W + j);
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        if (v != 0xFFFFFFFFu) prev_exit = rs_next + (int64_t)(int32_t)v;
        else prev_exit = chain_chunk(S, c, s, R, slots, visited, exmap, cnmap, counters, k0, nk, prev_exit, lane);
    }
}
__global__ __launch_bounds__(64) void k_chain_apply(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, LevelParams P,
                                                    RangeDev *ranges, const uint32_t *visited, const uint32_t *bad_slot, const uint16_t *exmap,
                                                    const uint16_t *cnmap, unsigned long long *counters, const int64_t *ein) {
    __shared__ ChainLds S;
    const SegDev s = segs[0];
    const uint32_t k0 = 1 + blockIdx.x * CH_RANGES;
    if (k0 >= s.range_cnt) return;
    const uint32_t nk = s.range_cnt - k0 < CH_RANGES ? s.range_cnt - k0 : CH_RANGES;
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs);
    (void)chain_chunk(S, c, s, ranges + s.range_off, bad_slot + s.range_off, visited, exmap, cnmap, counters, k0, nk, ein[blockIdx.x], threadIdx.x);
}

// Token counts per range -> scan input
__global__ void k_range_counts(const RangeDev *ranges, uint64_t nranges, uint32_t *counts) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nranges) counts[r] = ranges[r].true_count;
}

// Per-segment token totals and block counts (C/DeflaterEngine.cs:841-852, :750-768)
__global__ void k_seg_tokens(const SegDev *segs, uint32_t nseg, const uint64_t *range_tok /* exclusive scan, nranges+1 */,
                             SegOut *so, uint32_t *blk_counts) {
    uint32_t si = blockIdx.x * blockDim.x + threadIdx.x;
    if (si >= nseg) return;
    const SegDev s = segs[si];
    uint64_t first = range_tok[s.range_off];
    uint64_t cnt = range_tok[s.range_off + s.range_cnt] - first;
    so[si].tok_first = first;
    so[si].tok_count = cnt;
    // The number of blocks also depends on whether the very last token is a match; that is only known after
    // emission, so reserve the maximum here (one more than ceil) and let k_block_table fix blk_count.
    blk_counts[si] = (uint32_t)(cnt / BLOCK_TOKENS) + 1;
}

// C5: replay the true path of each range and write its tokens; record block edges.
#if SZL_LAB   // (laboratory library only)
__global__ __launch_bounds__(256) void k_emit(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs,
                                              uint32_t nseg, uint64_t nranges, LevelParams P, const RangeDev *ranges,
                                              const uint64_t *range_tok, const SegOut *so, uint32_t *tokens,
                                              const uint64_t *blk_off, int64_t *blk_start_pos, int64_t *blk_lasttok_pos,
                                              unsigned long long *counters) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nranges) return;
    uint32_t si = find_seg(segs, nseg, r);
    const SegDev s = segs[si];
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs + si);
    uint64_t lr = r - s.range_off;
    int64_t rs = s.seg_start + (int64_t)lr * (int64_t)s.range_len;
    int64_t re = rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end;
    int64_t x = lr == 0 ? rs : ranges[r].entry;
    int L = 0, D = 0;
    uint64_t ti = range_tok[r];
    const uint64_t seg_tok0 = so[si].tok_first, seg_ntok = so[si].tok_count;
    const uint64_t b0 = blk_off[si];
    int64_t tp;
    for (;;) {
        if (L == 0 && x >= re) break;
        uint32_t t = parse_step(c, x, L, D, &tp, true, nullptr);
        if (t != 0xFFFFFFFFu) {
            tokens[ti] = t;
            uint64_t li = ti - seg_tok0; // token index inside the segment
            if ((li & (BLOCK_TOKENS - 1)) == 0) blk_start_pos[b0 + li / BLOCK_TOKENS] = tp;
            if ((li & (BLOCK_TOKENS - 1)) == BLOCK_TOKENS - 1 || li == seg_ntok - 1) blk_lasttok_pos[b0 + li / BLOCK_TOKENS] = tp;
            ti++;
        }
    }
    (void)counters;
}
#endif   // SZL_LAB

// ============================================================================================
// Windowed variants of C1 / C5.  The per-lane walks above touch 4 bytes of a cache line per step and
// come back to the same line dozens of times, by which time 64 lanes x 16 waves have evicted it from the
// CU's 32 KiB vector cache (measured: k_emit fetched 6x its algorithmic bytes).  Here a wave works in
// rounds: it stages, for each of its 64 ranges, the next W entries of M2 and Mq (and the bytes) with
// coalesced loads into LDS, every lane then walks its own window out of LDS, and the tokens produced
// are written back range by range, again coalesced.  Tokens are staged in the M2 window itself: the
// k-th token of a round is produced by a step that has already consumed M2 entries 0..k.
// ============================================================================================
template <int W> struct WinCfg { enum : int { STRIDE = W + 1, BSTRIDE = W + 8 }; };

struct WinAcc {
    const uint32_t *m2p, *mqg; const uint8_t *bp; int64_t w0; // m2p: the staged window of packed entries; mqg: the GLOBAL mq array (read only where an entry says so); bp[0] is the byte at w0-1
    __device__ __forceinline__ uint32_t m2(int64_t x) const { const uint32_t e = m2p[(int)(x - w0)]; return e == M_UNSET ? e : mt_m2(e); }
    __device__ __forceinline__ uint32_t mq(int64_t x) const { const uint32_t e = m2p[(int)(x - w0)]; const uint32_t k = mt_code(e); return k == 0u ? mt_m2(e) : (k == 1u ? 0u : mqg[x]); }
    __device__ __forceinline__ uint32_t lit(int64_t x) const { return bp[(int)(x - w0) + 1]; }
};

__device__ __forceinline__ int64_t shfl64(int64_t v, int l) {
    uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, l), hi = (uint32_t)__shfl((int)(uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int64_t readlane64(int64_t v, int l) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Stage the window [x, x+W) of every active lane: one load instruction covers 64/W windows.
template <int W, bool WITH_BYTES>
__device__ __forceinline__ void win_refill(const ParseCtx &c, int64_t x, bool active, int lane, uint32_t *sm2, uint8_t *sb) {
    constexpr int PER = 64 / W;          // windows per load instruction
    constexpr int STRIDE = WinCfg<W>::STRIDE, BSTRIDE = WinCfg<W>::BSTRIDE;
    const int64_t navail = active ? c.tab_end - x : 0; // entries valid from x on
    const int64_t pm_l = (int64_t)(c.m2 + x), pd_l = (int64_t)(c.d + x) - 1;
    const int sub = lane / W, i = lane % W;
    constexpr int BATCH = 8;
    for (int kb = 0; kb < W; kb += BATCH) { // W instructions per array, BATCH of them in flight
        uint32_t v[BATCH], bv[BATCH], be[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            const int j = (kb + u) * PER + sub;
            const int64_t nv = shfl64(navail, j);
            const uint32_t *pm = (const uint32_t *)shfl64(pm_l, j);
            const bool ok = (int64_t)i < nv;
            v[u] = ok ? pm[i] : 0u;
            if (WITH_BYTES) {
                const uint8_t *pd = (const uint8_t *)shfl64(pd_l, j);
                const int64_t xj = shfl64(x, j);
                bv[u] = ((i > 0 || xj > 0) && (int64_t)i - 1 < nv) ? (uint32_t)pd[i] : 0u; // byte x-1+i
                be[u] = (i == 0 && (int64_t)W - 1 < nv) ? (uint32_t)pd[W] : 0u;            // byte x+W-1
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            const int j = (kb + u) * PER + sub;
            sm2[j * STRIDE + i] = v[u];
            if (WITH_BYTES) {
                sb[j * BSTRIDE + i] = (uint8_t)bv[u];
                if (i == 0) sb[j * BSTRIDE + W] = (uint8_t)be[u];
            }
        }
    }
}

// WB4 (the product's form since round 6; SZL_SPEC_WB=0 in the laboratory library is the other): the write-back of a round's tokens takes four ranges per store instruction
// (16 lanes each) out of a compacted list instead of one range per iteration of a loop over up to 64 of them — on the interpreter the loop
// is 17-20 % of this kernel's instructions (profiles/r04/gfxsim_srcprof_k_spec_win.log).
template <int W, bool WB4 = false>
__global__ __launch_bounds__(64) void k_spec_win(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs,
                                                 uint32_t nseg, uint64_t nranges, LevelParams P, RangeDev *ranges,
                                                 uint32_t *visited, unsigned long long *counters, uint32_t *spec_tok) {
    constexpr int STRIDE = WinCfg<W>::STRIDE;
    __shared__ uint32_t sm2[64 * STRIDE];
    const int lane = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * 64 + lane;
    const bool mine = r0 < nranges;
    bool active = mine;
    const uint64_t r = mine ? r0 : nranges - 1;
    const uint32_t si = find_seg(segs, nseg, r);
    const SegDev s = segs[si];
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs + si);
    const uint64_t lr = r - s.range_off;
    const int64_t rs = s.seg_start + (int64_t)lr * (int64_t)s.range_len;
    const int64_t re = rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end;
    uint32_t *vis = visited + s.vis_word_off;
    int64_t x = rs;
    int L = 0, D = 0;
    uint32_t count = 0;
    int64_t tp;
    uint64_t cw = (uint64_t)(rs - s.seg_start) >> 5;
    uint32_t acc = 0;
    if (x >= re) active = false;
    while (__any(active)) {
        win_refill<W, false>(c, x, active, lane, sm2, nullptr);
        __syncthreads();
        uint32_t *win = sm2 + lane * STRIDE;
        const WinAcc wa{win, c.mq, nullptr, x};
        const int64_t wend = x + W;
        int nt = 0;
        while (active && x < wend) {
            if (L == 0) {
                if (x >= re) { active = false; break; }
                uint64_t bit = (uint64_t)(x - s.seg_start);
                if ((bit >> 5) != cw) { if (acc) vis[cw] = acc; cw = bit >> 5; acc = 0; }
                acc |= 1u << (bit & 31);
            }
            uint32_t t = parse_step(c, wa, x, L, D, &tp, false, counters + 1);
            if (t != 0xFFFFFFFFu) { if (spec_tok) win[nt] = t; nt++; } // literal = 0, match = dist<<16|len; entry nt of the window is dead by now
        }
        if (active && L == 0 && x >= re) active = false;
        __syncthreads();
        if (spec_tok) { // the k-th token of this range's speculative path goes to spec_tok[range start + k] (k_emit_copy reads them back)
            // A range has range_len slots and its path can hold MORE tokens than that: literals at (nearly) every position and then
            // a run of lazy literals past the range's end (a node that starts in the range belongs to it to its last token).  Such a
            // range's surplus is not stored — it would land in the next range's slots, and whichever lane stored last would win — and
            // k_emit_copy walks a range whose spec_count exceeds range_len instead of copying it (found by tools/lab/small_call_soak.py:
            // level 7, Filtered, ranges of 64, 66 tokens in one of them; ranges of 4096 need 4095 literals and the run at the end).
            const int nt_all = nt;
            nt = (int)min((uint32_t)nt, s.range_len - min(s.range_len, count));
            if constexpr (WB4) {
                __shared__ uint32_t wb_j[64], wb_n[64];
                __shared__ uint64_t wb_dst[64];
                const uint64_t m = __ballot(nt > 0);
                const int nact = __builtin_popcountll(m);
                if (nt > 0) {
                    const int rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
                    wb_j[rank] = (uint32_t)lane; wb_n[rank] = (uint32_t)nt;
                    wb_dst[rank] = (uint64_t)(uintptr_t)(spec_tok + (s.buf_off + (uint64_t)rs + count));
                }
                __syncthreads();
                const int g = lane >> 4, k0 = lane & 15;
                for (int i = 0; i < nact; i += 4) {
                    const int r = i + g;
                    if (r < nact) {
                        const uint32_t j = wb_j[r], n = wb_n[r];
                        uint32_t *dst = (uint32_t *)(uintptr_t)wb_dst[r];
                        for (uint32_t k = (uint32_t)k0; k < n; k += 16) dst[k] = sm2[j * STRIDE + k];
                    }
                }
            } else {
                for (uint64_t m = __ballot(nt > 0); m; m &= m - 1) {
                    const int j = __builtin_ctzll(m);
                    const int ntj = __builtin_amdgcn_readlane(nt, j);
                    uint32_t *dst = spec_tok + readlane64((int64_t)(s.buf_off + (uint64_t)rs + count), j);
                    if (lane < ntj) dst[lane] = sm2[j * STRIDE + lane];
                }
            }
            __syncthreads();
            nt = nt_all;
        }
        count += (uint32_t)nt;
    }
    if (!mine) return;
    if (acc) vis[cw] = acc;
    RangeDev rd;
    rd.exit_spec = x; rd.exit_true = x; rd.entry = rs; rd.spec_count = count; rd.true_count = count; rd.merged = 1; rd.pad = 0;
    ranges[r] = rd;
}

template <int W>
__global__ __launch_bounds__(64) void k_emit_win(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs,
                                                 uint32_t nseg, uint64_t nranges, LevelParams P, const RangeDev *ranges,
                                                 const uint64_t *range_tok, const SegOut *so, uint32_t *tokens,
                                                 const uint64_t *blk_off, int64_t *blk_start_pos, int64_t *blk_lasttok_pos) {
    constexpr int STRIDE = WinCfg<W>::STRIDE, BSTRIDE = WinCfg<W>::BSTRIDE;
    __shared__ uint32_t sm2[64 * STRIDE];
    __shared__ uint8_t sb[64 * BSTRIDE];
    const int lane = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * 64 + lane;
    bool active = r0 < nranges;
    const uint64_t r = active ? r0 : nranges - 1;
    const uint32_t si = find_seg(segs, nseg, r);
    const SegDev s = segs[si];
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs + si);
    const uint64_t lr = r - s.range_off;
    const int64_t rs = s.seg_start + (int64_t)lr * (int64_t)s.range_len;
    const int64_t re = rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end;
    int64_t x = lr == 0 ? rs : ranges[r].entry;
    int L = 0, D = 0;
    uint64_t ti = range_tok[r];
    const uint64_t seg_tok0 = so[si].tok_first, seg_ntok = so[si].tok_count;
    const uint64_t b0 = blk_off[si];
    int64_t tp;
    if (x >= re) active = false;
    while (__any(active)) {
        win_refill<W, true>(c, x, active, lane, sm2, sb);
        __syncthreads();
        uint32_t *win = sm2 + lane * STRIDE;
        const WinAcc wa{win, c.mq, sb + lane * BSTRIDE, x};
        const int64_t wend = x + W;
        int nt = 0;
        while (active && x < wend) {
            if (L == 0 && x >= re) { active = false; break; }
            uint32_t t = parse_step(c, wa, x, L, D, &tp, true, nullptr);
            if (t != 0xFFFFFFFFu) {
                win[nt] = t; // entry nt of the M2 window is dead by now (see the header comment)
                const uint64_t li = ti + nt - seg_tok0; // token index inside the segment
                if ((li & (BLOCK_TOKENS - 1)) == 0) blk_start_pos[b0 + li / BLOCK_TOKENS] = tp;
                if ((li & (BLOCK_TOKENS - 1)) == BLOCK_TOKENS - 1 || li == seg_ntok - 1) blk_lasttok_pos[b0 + li / BLOCK_TOKENS] = tp;
                nt++;
            }
        }
        if (active && L == 0 && x >= re) active = false;
        __syncthreads();
        for (uint64_t m = __ballot(nt > 0); m; m &= m - 1) {
            const int j = __builtin_ctzll(m);
            const int ntj = __builtin_amdgcn_readlane(nt, j);
            uint32_t *dst = tokens + readlane64((int64_t)ti, j);
            if (lane < ntj) dst[lane] = sm2[j * STRIDE + lane];
        }
        ti += nt;
        __syncthreads();
    }
}

// C5, copy form (default).  The speculative walk already produced every range's tokens (k_spec_win -> spec_tok); the true
// path of a merged range is a short fix-up prefix (entry -> merge point y, walked here like k_fix does) followed by the
// speculative tokens from y on, so emission is a coalesced copy instead of a second walk: each lane does the prefix of its
// range, then the wave copies range by range, turning literal placeholders into bytes (their positions are a running sum of
// the token lengths) and recording block edges.  A range the true path never merges with is walked in full by its lane.
__global__ __launch_bounds__(64) void k_emit_copy(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs,
                                                  uint32_t nseg, uint64_t nranges, LevelParams P, const RangeDev *ranges,
                                                  const uint32_t *visited, const uint32_t *spec_tok,
                                                  const uint64_t *range_tok, const SegOut *so, uint32_t *tokens,
                                                  const uint64_t *blk_off, int64_t *blk_start_pos, int64_t *blk_lasttok_pos) {
    const int lane = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * 64 + lane;
    const bool mine = r0 < nranges;
    const uint64_t r = mine ? r0 : nranges - 1;
    const uint32_t si = find_seg(segs, nseg, r);
    const SegDev s = segs[si];
    ParseCtx c = make_ctx(in, link, mtab, s, P, segs + si);
    const uint64_t lr = r - s.range_off;
    const int64_t rs = s.seg_start + (int64_t)lr * (int64_t)s.range_len;
    const int64_t re = rs + (int64_t)s.range_len < s.seg_end ? rs + (int64_t)s.range_len : s.seg_end;
    const uint32_t *vis = visited + s.vis_word_off;
    const RangeDev rd = ranges[r];
    uint64_t ti = range_tok[r];
    const uint64_t seg_tok0 = so[si].tok_first, seg_ntok = so[si].tok_count;
    const uint64_t b0 = blk_off[si];
    auto edges = [&](uint64_t tindex, int64_t tpos) { // block starts / last-token positions (C/DeflaterEngine.cs:841-852)
        const uint64_t li = tindex - seg_tok0;
        if ((li & (BLOCK_TOKENS - 1)) == 0) blk_start_pos[b0 + li / BLOCK_TOKENS] = tpos;
        if ((li & (BLOCK_TOKENS - 1)) == BLOCK_TOKENS - 1 || li == seg_ntok - 1) blk_lasttok_pos[b0 + li / BLOCK_TOKENS] = tpos;
    };
    // ---- per lane: fix-up prefix from the true entry to the merge point y
    int64_t y = -1;
    uint32_t ncopy = 0, skip = 0;
    if (mine && rs < re) {
        int64_t x = lr == 0 ? rs : rd.entry, tp;
        int L = 0, D = 0;
        const bool fits = rd.spec_count <= s.range_len;   // (k_spec_win stored no more than range_len tokens: the rest of such a path is walked here)
        for (;;) {
            if (L == 0) {
                if (x >= re) break;
                if (x >= rs && fits && is_visited(vis, (uint64_t)(x - s.seg_start))) { y = x; break; }
            }
            const uint32_t t = parse_step(c, x, L, D, &tp, true, nullptr);
            if (t != 0xFFFFFFFFu) { tokens[ti] = t; edges(ti, tp); ti++; }
        }
        if (y >= 0) { // tokens of speculative nodes before y are not on the true path
            x = rs; L = 0; D = 0;
            for (;;) {
                if (L == 0 && x >= y) break;
                const uint32_t t = parse_step(c, x, L, D, &tp, false, nullptr);
                skip += (t != 0xFFFFFFFFu);
            }
            ncopy = rd.spec_count - skip;
        }
    }
    // ---- the wave: copy each range's speculative tokens from y on, 64 tokens a step.  A step's two trips to memory — its tokens, then
    // the bytes of its literals at positions that follow from the tokens — used to be waited for one after the other, 64 ranges of ~20
    // tokens one after the other when the call is ONE 64 KiB entry (77 us, round 6): now the tokens of the step after it are on their way
    // before a step begins, and a step's tokens are stored at the end of the step after it, when its bytes have arrived.
    uint64_t m = __ballot(ncopy > 0);
    int j = 0; uint32_t n = 0, k0 = 0; const uint32_t *src = nullptr; bool more = false;   // the step the iterator stands on (all uniform)
    auto next = [&]() {
        if (more && k0 + 64 < n) { k0 += 64; return; }
        if (!m) { more = false; return; }
        j = __builtin_ctzll(m); m &= m - 1;
        n = (uint32_t)__builtin_amdgcn_readlane((int)ncopy, j);
        src = spec_tok + readlane64((int64_t)(s.buf_off + (uint64_t)rs + skip), j);
        k0 = 0; more = true;
    };
    next();
    uint32_t t_pre = (more && k0 + lane < n) ? src[k0 + lane] : 0u;
    bool p_on = false, p_lit = false; uint32_t p_t = 0, p_byte = 0; uint64_t p_tindex = 0;      // the step whose store is pending
    uint64_t dst0 = 0, j_tok0 = 0, j_ntok = 0, j_b0 = 0; int64_t pos = 0; const uint8_t *dj = nullptr;   // the current range (uniform)
    while (more) {
        const int cj = j; const uint32_t cn = n, ck0 = k0;
        uint32_t t = t_pre;
        next();
        if (more) t_pre = (k0 + lane < n) ? src[k0 + lane] : 0u;
        if (ck0 == 0) {
            dst0 = (uint64_t)readlane64((int64_t)ti, cj);
            pos = readlane64(y, cj);                                        // input position of the next token
            dj = (const uint8_t *)readlane64((int64_t)c.d, cj);
            j_tok0 = (uint64_t)readlane64((int64_t)seg_tok0, cj); j_ntok = (uint64_t)readlane64((int64_t)seg_ntok, cj);
            j_b0 = (uint64_t)readlane64((int64_t)b0, cj);
        }
        const bool on = ck0 + lane < cn;
        if (!on) t = 0u;
        const uint32_t len = on ? ((t >> 16) ? (t & 0xFFFF) : 1u) : 0u;
        uint32_t incl = len;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        const int64_t tpos = pos + (incl - len);
        const bool lit = on && (t >> 16) == 0;
        const uint32_t byte = lit ? (uint32_t)dj[tpos] : 0u;               // literal placeholder -> the byte
        if (p_on) tokens[p_tindex] = p_lit ? p_byte : p_t;
        const uint64_t tindex = dst0 + ck0 + lane;
        if (on) {
            const uint64_t li = tindex - j_tok0;
            if ((li & (BLOCK_TOKENS - 1)) == 0) blk_start_pos[j_b0 + li / BLOCK_TOKENS] = tpos;
            if ((li & (BLOCK_TOKENS - 1)) == BLOCK_TOKENS - 1 || li == j_ntok - 1) blk_lasttok_pos[j_b0 + li / BLOCK_TOKENS] = tpos;
        }
        p_on = on; p_lit = lit; p_t = t; p_byte = byte; p_tindex = tindex;
        pos += (int64_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (p_on) tokens[p_tindex] = p_lit ? p_byte : p_t;
}

int knob(const char *name, int dflt);
static int cwin_mode() {
    // (measured again in round 3, after the match tables were packed: 32 positions of window per lane — 1 GiB: parse 6.7 -> 5.8 ms, a
    // 64 KiB call 0.38 -> 0.34 ms; 64 is better still for small calls and worse for large ones)
#if SZL_LAB
    static const int m = getenv("SZL_CWIN") ? atoi(getenv("SZL_CWIN")) : 32;
    return m;
#else
    return 32;    // (round 5: the other window lengths, the per-lane global walks k_spec / k_emit and the second-walk emit live in the laboratory library)
#endif
}

void launch_spec(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg,
                 uint64_t nranges, LevelParams P, RangeDev *ranges, uint32_t *visited, unsigned long long *counters,
                 uint32_t *spec_tok, hipStream_t st) {
    if (nranges == 0) return;
    const int cw = cwin_mode();
    const dim3 wg((unsigned)((nranges + 63) / 64));
#if SZL_LAB
    if (cw == 64) { hipLaunchKernelGGL(k_spec_win<64>, wg, dim3(64), 0, st, in, link, mtab, segs, nseg, nranges, P, ranges, visited, counters, spec_tok); return; }
#endif
    // (round 6, measured at last — alternating calls in one process, tools/lab/knob_ab.py: the write-back that takes four ranges per store
    // instruction, stage C of 1 GiB of text 5.46 -> 5.07 ms, of logs 2.16 -> 1.97; the product holds that form only, the laboratory
    // library both: SZL_SPEC_WB=0 is the loop over the ranges)
#if SZL_LAB
    if (cw == 32 && SZL_LABKNOB("SZL_SPEC_WB", 1) == 0) { hipLaunchKernelGGL(k_spec_win<32>, wg, dim3(64), 0, st, in, link, mtab, segs, nseg, nranges, P, ranges, visited, counters, spec_tok); return; }
#endif
    if (cw == 32) { hipLaunchKernelGGL((k_spec_win<32, true>), wg, dim3(64), 0, st, in, link, mtab, segs, nseg, nranges, P, ranges, visited, counters, spec_tok); return; }
#if SZL_LAB
    if (cw == 16) { hipLaunchKernelGGL(k_spec_win<16>, wg, dim3(64), 0, st, in, link, mtab, segs, nseg, nranges, P, ranges, visited, counters, spec_tok); return; }
    if (cw == 8) { hipLaunchKernelGGL(k_spec_win<8>, wg, dim3(64), 0, st, in, link, mtab, segs, nseg, nranges, P, ranges, visited, counters, spec_tok); return; }
    hipLaunchKernelGGL(k_spec, dim3((unsigned)((nranges + 255) / 256)), dim3(256), 0, st, in, link, mtab, segs, nseg, nranges, P,
                       ranges, visited, counters);
#endif
}
void launch_fix(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, uint64_t nranges,
                LevelParams P, RangeDev *ranges, const uint32_t *visited, unsigned long long *counters, uint32_t *bad_slot,
                uint64_t *bad_range, hipStream_t st) {
    if (nranges == 0) return;
    hipLaunchKernelGGL(k_fix, dim3((unsigned)((nranges + 255) / 256)), dim3(256), 0, st, in, link, mtab, segs, nseg, nranges, P,
                       ranges, visited, counters, bad_slot, bad_range);
    hipLaunchKernelGGL(k_flag_followers, dim3((unsigned)((nranges + 255) / 256)), dim3(256), 0, st, segs, nseg, nranges, (const RangeDev *)ranges, counters,
                       bad_slot, bad_range);
}
void launch_resolve(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, LevelParams P,
                    RangeDev *ranges, const uint32_t *visited, unsigned long long *counters, hipStream_t st) {
    hipLaunchKernelGGL(k_resolve, dim3(nseg), dim3(64), 0, st, in, link, mtab, segs, nseg, P, ranges, visited, counters);
}
int exitmap_width() { return X_W; }
void launch_exitmaps(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, LevelParams P,
                     RangeDev *ranges, const uint32_t *visited, const uint32_t *bad_slot, const uint64_t *bad_range, uint64_t nbad,
                     uint16_t *exmap, uint16_t *cnmap, unsigned long long *counters, hipStream_t st, void *chain_scratch, uint32_t range_cnt0) {
    if (nbad == 0) return;
    hipLaunchKernelGGL(k_exitmap, dim3((unsigned)(nbad * (X_W / 64))), dim3(64), 0, st, in, link, mtab, segs, nseg, P, bad_range, nbad,
                       exmap, cnmap, counters, (const RangeDev *)ranges, visited);
    if (nseg == 1 && chain_scratch && range_cnt0 > 8u * CH_RANGES) {   // one long segment: the chain in two levels
        const uint32_t nch = (range_cnt0 - 1 + CH_RANGES - 1) / CH_RANGES;
        uint32_t *cm = (uint32_t *)chain_scratch;
        int64_t *ein = (int64_t *)((uint8_t *)chain_scratch + (((size_t)nch * X_W * 4 + 63) & ~(size_t)63));
        hipLaunchKernelGGL(k_chain_maps, dim3(nch), dim3(64), 0, st, segs, (const RangeDev *)ranges, bad_slot, (const uint16_t *)exmap, cm);
        hipLaunchKernelGGL(k_chain_top, dim3(1), dim3(64), 0, st, in, link, mtab, segs, P, ranges, visited, bad_slot, (const uint16_t *)exmap, (const uint16_t *)cnmap, counters, (const uint32_t *)cm, ein);
        hipLaunchKernelGGL(k_chain_apply, dim3(nch), dim3(64), 0, st, in, link, mtab, segs, P, ranges, visited, bad_slot, (const uint16_t *)exmap, (const uint16_t *)cnmap, counters, (const int64_t *)ein);
        return;
    }
    hipLaunchKernelGGL(k_chain, dim3(nseg), dim3(64), 0, st, in, link, mtab, segs, nseg, P, ranges, visited, bad_slot, exmap, cnmap,
                       counters);
}
size_t exitchain_scratch_bytes(uint64_t range_cnt) {
    const uint64_t nch = (range_cnt + CH_RANGES - 1) / CH_RANGES + 1;
    return (size_t)(((nch * X_W * 4 + 63) & ~63ull) + nch * 8 + 64);
}
// ---- exclusive prefix sum of 32-bit counts into 64-bit offsets (tokens in front of every range; block slots in front of every segment).
// Through round 5 this was hipcub::DeviceScan::ExclusiveSum — the one library kernel on the path.  A block scans tiles of 8192 counts
// (1024 threads x 8: serial in the thread, __shfl_up inside the wavefront, the sixteen wavefront totals through LDS); up to eight tiles
// one block does alone with a running carry (a small call: one launch); more go tile sums -> their scan (one block) -> tiles again.
enum : int { SCAN_T = 1024, SCAN_I = 8, SCAN_TILE = SCAN_T * SCAN_I };
__device__ __forceinline__ uint64_t scan_tile(const uint32_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t base, uint64_t n, uint64_t carry, uint64_t *wsum) {
    // exclusive scan of in[base .. base + SCAN_TILE) (clipped at n) + carry into out (if out); returns the tile's total
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    uint32_t v[SCAN_I];
    uint64_t mine = 0;
    const uint64_t i0 = base + (uint64_t)t * SCAN_I;
#pragma unroll
    for (int k = 0; k < SCAN_I; k++) { v[k] = i0 + k < n ? in[i0 + k] : 0u; mine += v[k]; }
    uint64_t inc = mine;                                       // inclusive over the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint64_t before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < SCAN_T / 64; k++) { const uint64_t x = wsum[k]; if (k < w) before += x; total += x; }
    __syncthreads();
    if (out) {
        uint64_t run = carry + before + inc - mine;
#pragma unroll
        for (int k = 0; k < SCAN_I; k++) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
    }
    return total;
}
__global__ __launch_bounds__(SCAN_T) void k_exscan_small(const uint32_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t n) {
    __shared__ uint64_t wsum[SCAN_T / 64];
    uint64_t carry = 0;
    for (uint64_t base = 0; base < n; base += SCAN_TILE) carry += scan_tile(in, out, base, n, carry, wsum);
}
__global__ __launch_bounds__(SCAN_T) void k_exscan_sums(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ sums) {
    __shared__ uint64_t wsum[SCAN_T / 64];
    const uint64_t total = scan_tile(in, nullptr, (uint64_t)blockIdx.x * SCAN_TILE, n, 0, wsum);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(SCAN_T) void k_exscan_top(uint64_t *__restrict__ sums, uint32_t nb) {     // exclusive scan of the tile sums, in place (nb <= 65536)
    __shared__ uint64_t wsum[SCAN_T / 64];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += SCAN_T) {
        const uint32_t i = base + threadIdx.x;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const uint64_t x = i < nb ? sums[i] : 0;
        uint64_t inc = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint64_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < SCAN_T / 64; k++) { const uint64_t y = wsum[k]; if (k < w) before += y; total += y; }
        const uint64_t carry = s_carry;
        if (i < nb) sums[i] = carry + before + inc - x;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + total;
        __syncthreads();
    }
}
__global__ __launch_bounds__(SCAN_T) void k_exscan_apply(const uint32_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t n, const uint64_t *__restrict__ sums) {
    __shared__ uint64_t wsum[SCAN_T / 64];
    (void)scan_tile(in, out, (uint64_t)blockIdx.x * SCAN_TILE, n, sums[blockIdx.x], wsum);
}
size_t exscan_tmp_bytes(uint64_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE + 1) * 8; }
// out[i] = in[0] + ... + in[i - 1] for i in [0, n); tmp: exscan_tmp_bytes(n)
hipError_t launch_exscan(const uint32_t *in, uint64_t *out, uint64_t n, void *tmp, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb <= 8) hipLaunchKernelGGL(k_exscan_small, dim3(1), dim3(SCAN_T), 0, st, in, out, n);
    else {
        if (nb > 65536ull * 1024ull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_exscan_sums, dim3((unsigned)nb), dim3(SCAN_T), 0, st, in, n, (uint64_t *)tmp);
        hipLaunchKernelGGL(k_exscan_top, dim3(1), dim3(SCAN_T), 0, st, (uint64_t *)tmp, (uint32_t)nb);
        hipLaunchKernelGGL(k_exscan_apply, dim3((unsigned)nb), dim3(SCAN_T), 0, st, in, out, n, (const uint64_t *)tmp);
    }
    return hipGetLastError();
}

// A small call's four steps from the ranges' token counts to the block slots — k_range_counts, the scan, k_seg_tokens, the scan — in one
// workgroup (round 6: each of them was 2-3 us of work and 5-8 us of launch for ONE 64 KiB entry).  counts[nranges] and blk_counts[nseg]
// are zero (the call's memsets), as the scans of the separate launches expect.
__global__ __launch_bounds__(SCAN_T) void k_counts_to_blocks(const RangeDev *__restrict__ ranges, uint64_t nranges, const SegDev *__restrict__ segs, uint32_t nseg,
                                                             uint32_t *counts, uint64_t *range_tok, SegOut *so, uint32_t *blk_counts, uint64_t *blk_off) {
    __shared__ uint64_t wsum[SCAN_T / 64];
    for (uint64_t r = threadIdx.x; r < nranges; r += SCAN_T) counts[r] = ranges[r].true_count;
    __threadfence_block();
    __syncthreads();
    uint64_t carry = 0;
    for (uint64_t base = 0; base < nranges + 1; base += SCAN_TILE) carry += scan_tile(counts, range_tok, base, nranges + 1, carry, wsum);
    __threadfence_block();
    __syncthreads();
    for (uint32_t si = threadIdx.x; si < nseg; si += SCAN_T) {     // (k_seg_tokens)
        const uint64_t off = segs[si].range_off;
        const uint64_t first = range_tok[off], cnt = range_tok[off + segs[si].range_cnt] - first;
        so[si].tok_first = first;
        so[si].tok_count = cnt;
        blk_counts[si] = (uint32_t)(cnt / BLOCK_TOKENS) + 1;
    }
    __threadfence_block();
    __syncthreads();
    carry = 0;
    for (uint64_t base = 0; base < (uint64_t)nseg + 1; base += SCAN_TILE) carry += scan_tile(blk_counts, blk_off, base, (uint64_t)nseg + 1, carry, wsum);
}
bool counts_to_blocks_fits(uint64_t nranges, uint32_t nseg) { return nranges + 1 <= 4ull * SCAN_TILE && (uint64_t)nseg + 1 <= 4ull * SCAN_TILE; }
void launch_counts_to_blocks(const RangeDev *ranges, uint64_t nranges, const SegDev *segs, uint32_t nseg, uint32_t *counts, uint64_t *range_tok, SegOut *so,
                             uint32_t *blk_counts, uint64_t *blk_off, hipStream_t st) {
    hipLaunchKernelGGL(k_counts_to_blocks, dim3(1), dim3(SCAN_T), 0, st, ranges, nranges, segs, nseg, counts, range_tok, so, blk_counts, blk_off);
}
void launch_range_counts(const RangeDev *ranges, uint64_t nranges, uint32_t *counts, hipStream_t st) {
    if (nranges == 0) return;
    hipLaunchKernelGGL(k_range_counts, dim3((unsigned)((nranges + 255) / 256)), dim3(256), 0, st, ranges, nranges, counts);
}
void launch_seg_tokens(const SegDev *segs, uint32_t nseg, const uint64_t *range_tok, SegOut *so, uint32_t *blk_counts,
                       hipStream_t st) {
    hipLaunchKernelGGL(k_seg_tokens, dim3((nseg + 255) / 256), dim3(256), 0, st, segs, nseg, range_tok, so, blk_counts);
}
bool emit_copy_enabled() {
#if SZL_LAB
    static const bool on = cwin_mode() != 0 && !(getenv("SZL_EMIT_COPY") && atoi(getenv("SZL_EMIT_COPY")) == 0);
    return on;
#else
    return true;
#endif
}
void launch_emit_copy(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, uint64_t nranges,
                      LevelParams P, const RangeDev *ranges, const uint32_t *visited, const uint32_t *spec_tok,
                      const uint64_t *range_tok, const SegOut *so, uint32_t *tokens, const uint64_t *blk_off, int64_t *blk_start_pos,
                      int64_t *blk_lasttok_pos, hipStream_t st) {
    if (nranges == 0) return;
    hipLaunchKernelGGL(k_emit_copy, dim3((unsigned)((nranges + 63) / 64)), dim3(64), 0, st, in, link, mtab, segs, nseg, nranges, P, ranges,
                       visited, spec_tok, range_tok, so, tokens, blk_off, blk_start_pos, blk_lasttok_pos);
}
void launch_emit(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, uint64_t nranges,
                 LevelParams P, const RangeDev *ranges, const uint64_t *range_tok, const SegOut *so, uint32_t *tokens,
                 const uint64_t *blk_off, int64_t *blk_start_pos, int64_t *blk_lasttok_pos, unsigned long long *counters,
                 hipStream_t st) {
#if SZL_LAB
    if (nranges == 0) return;
    const int cw = cwin_mode();
    const dim3 wg((unsigned)((nranges + 63) / 64));
    if (cw == 64) { hipLaunchKernelGGL(k_emit_win<64>, wg, dim3(64), 0, st, in, link, mtab, segs, nseg, nranges, P, ranges, range_tok, so, tokens, blk_off, blk_start_pos, blk_lasttok_pos); return; }
    if (cw == 32) { hipLaunchKernelGGL(k_emit_win<32>, wg, dim3(64), 0, st, in, link, mtab, segs, nseg, nranges, P, ranges, range_tok, so, tokens, blk_off, blk_start_pos, blk_lasttok_pos); return; }
    if (cw == 16) { hipLaunchKernelGGL(k_emit_win<16>, wg, dim3(64), 0, st, in, link, mtab, segs, nseg, nranges, P, ranges, range_tok, so, tokens, blk_off, blk_start_pos, blk_lasttok_pos); return; }
    if (cw == 8) { hipLaunchKernelGGL(k_emit_win<8>, wg, dim3(64), 0, st, in, link, mtab, segs, nseg, nranges, P, ranges, range_tok, so, tokens, blk_off, blk_start_pos, blk_lasttok_pos); return; }
    hipLaunchKernelGGL(k_emit, dim3((unsigned)((nranges + 255) / 256)), dim3(256), 0, st, in, link, mtab, segs, nseg, nranges, P,
                       ranges, range_tok, so, tokens, blk_off, blk_start_pos, blk_lasttok_pos, counters);
#else
    (void)in; (void)link; (void)mtab; (void)segs; (void)nseg; (void)nranges; (void)P; (void)ranges; (void)range_tok; (void)so; (void)tokens; (void)blk_off;
    (void)blk_start_pos; (void)blk_lasttok_pos; (void)counters; (void)st;   // (never reached: emit_copy_enabled() is true in the product library)
#endif
}


// ------------------------------------------------------------------------------------------------
// SetLevel to another compression function while DeflateSlow has bytes pending (C/DeflaterEngine.cs:338-351).  The segment was
// parsed as if nothing happened; the reference's engine, however, STANDS at the first iteration start at or behind cut_pos (its loop
// stopped for want of lookahead) when the call arrives.  In terms of the token stream: iteration starts are the token starts plus,
// for every match token, the position after its first byte (the lazy look).  So the cut lands on the first token start >= cut_pos —
// unless the token in front of it is a match that starts at cut_pos - 1: then the engine stands on that match's lazy look with the
// match pending, SetLevel tallies its first byte as a literal and drops the match (`if (prevAvailable) TallyLit`, :340), and the
// next function starts one byte behind the match start.  (Iterations in front of the cut had MIN_LOOKAHEAD bytes to look at: what
// follows the cut never influenced them.)  One wavefront, segment 0.
__global__ __launch_bounds__(64) void k_switch_cut(const uint8_t *in, const SegDev *segs, uint32_t *tokens, SegOut *so, const uint64_t *blk_off,
                                                   const int64_t *blk_start_pos, int64_t *blk_lasttok_pos) {
    const int lane = threadIdx.x;
    const SegDev s = segs[0];
    const uint8_t *d = in + s.buf_off;
    const int64_t T = s.cut_pos;
    const uint64_t ntok = so[0].tok_count, tf = so[0].tok_first, b0 = blk_off[0];
    uint64_t keep = 0;           // tokens that stay
    int64_t X = s.seg_start;     // where the next function starts
    int64_t last_pos = -1;       // start of the last token that stays (for the block's stored-offset rule)
    bool patch = false;
    if (ntok > 0 && blk_start_pos[b0] < T) {
        const uint64_t nblk = (ntok + BLOCK_TOKENS - 1) / BLOCK_TOKENS;
        uint64_t lo = 0, hi = nblk - 1;                          // last block whose first token starts below T
        while (lo < hi) { const uint64_t mid = (lo + hi + 1) >> 1; if (blk_start_pos[b0 + mid] < T) lo = mid; else hi = mid - 1; }
        const uint64_t kb = lo;
        const uint64_t base_i = kb * BLOCK_TOKENS, cnt = ntok - base_i < (uint64_t)BLOCK_TOKENS ? ntok - base_i : (uint64_t)BLOCK_TOKENS;
        int64_t pos = blk_start_pos[b0 + kb];                    // start of token base_i
        uint64_t ci = base_i + cnt;                              // first token of the block that starts at or behind T (none: the next block's first)
        int64_t cstart = 0, pstart = -1; uint32_t ptok = 0;      // its start; start and value of the token in front of it
        bool found = false;
        for (uint64_t i0 = 0; i0 < cnt && !found; i0 += 64) {
            const uint64_t i = i0 + lane;
            const uint32_t t = i < cnt ? tokens[tf + base_i + i] : 0u;
            const int len = i < cnt ? ((t >> 16) ? (int)(t & 0xFFFF) : 1) : 0;
            int incl = len;
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            const int64_t start = pos + (int64_t)(incl - len);
            const uint64_t m = __ballot(i < cnt && start >= T);
            if (m) {
                const int l = __builtin_ctzll(m);
                found = true;
                ci = base_i + i0 + (uint64_t)l;
                cstart = shfl64(start, l);
                if (l > 0) { pstart = shfl64(start, l - 1); ptok = (uint32_t)__shfl((int)t, l - 1); }
                // (l == 0: the token in front is the previous round's last one — kept in pstart / ptok below)
            } else {
                pstart = shfl64(start, 63 < (int)(cnt - i0) - 1 ? 63 : (int)(cnt - i0) - 1);
                ptok = (uint32_t)__shfl((int)t, 63 < (int)(cnt - i0) - 1 ? 63 : (int)(cnt - i0) - 1);
                pos += (int64_t)__shfl(incl, 63);
            }
        }
        if (!found) cstart = pos;                                // behind the block's last token: the next block's first token (or the segment's end)
        keep = ci; X = cstart; last_pos = pstart;
        if (ci > 0 && (ptok >> 16) != 0 && pstart + 1 >= T) {    // the engine stands on the lazy look of that match
            patch = true;
            X = pstart + 1;
        }
        if (lane == 0) {
            if (patch) tokens[tf + ci - 1] = (uint32_t)d[pstart];
            if (keep > 0) blk_lasttok_pos[b0 + (keep - 1) / BLOCK_TOKENS] = last_pos;
        }
    }   // (else: the first iteration start is already at or behind cut_pos — nothing was parsed, X = seg_start)
    if (lane == 0) { so[0].tok_count = keep; so[0].cut_x = X; }
}

void launch_switch_cut(const uint8_t *in, const SegDev *segs, uint32_t *tokens, SegOut *so, const uint64_t *blk_off, const int64_t *bsp, int64_t *blp, hipStream_t st) {
    hipLaunchKernelGGL(k_switch_cut, dim3(1), dim3(64), 0, st, in, segs, tokens, so, blk_off, bsp, blp);
}

} // namespace szl
