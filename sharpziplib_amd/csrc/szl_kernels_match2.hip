// szl_kernels_match2.hip — stage B (FindLongestMatch for every position), second form of the full search.
//
// Reference being restated: FindLongestMatch, C/DeflaterEngine.cs:474-612 (same results as k_match in
// szl_kernels_match.hip: M2 = walk entered with matchLen 2 and the full max_chain budget, Mq = that walk's state after
// max_chain>>2 candidates).
//
// Why a second form.  k_match is bound by VALU issue at ≈46 wave-instructions per position, of which the chain steps and
// compares themselves need ≈12: the rest is lane under-use.  A wavefront runs three kinds of work — FETCH (start a position),
// QUICK (one chain step), VERIFY (4-byte compare step) — and a lane can only take part in the kind its one position is in,
// so each kind runs with about half of the lanes (measured: profiles/r02).  Here every lane holds TWO positions in flight
// (contexts A and B, all in registers: LDS is full with the window).  Before a phase runs, lanes whose A context is not in
// that phase's state but whose B context is exchange the two (one v_swap per field under the lanes' exec mask), so the phase
// sees a lane as busy if EITHER of its positions can use it.  The phases themselves are loops that keep going while enough
// lanes remain in them, instead of a fixed number of predicated steps per scheduler visit.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <cstdint>
#include "szl_internal.h"

namespace szl {
int knob(const char *name, int dflt);

__device__ __forceinline__ int64_t base_of2(int64_t s_abs) { // window base of an iteration starting at s (App. A.2; C/DeflaterEngine.cs:371,:771,:93)
    int64_t idx = s_abs + 1;
    if (idx <= 65273) return 0;
    return ((idx - 65273 + 32767) >> 15) << 15;
}
__device__ __forceinline__ uint32_t load_u32_unaligned2(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

enum : int { B2_THREADS = 1024 };
enum : int { B2_DATA_BYTES = B_HIST + B_TILE + B_TAIL + 8, B2_LINKS = B_HIST + B_TILE };
enum : int { B2_LDS_BYTES = B2_DATA_BYTES + B2_LINKS * 2 + 16 };

// One FindLongestMatch walk in flight (a lane holds NCTX of them).  Kept small: every field is one v_swap_b32 per exchange.
struct WalkCtx {
    int p;          // tile position being searched
    int cl;         // LDS data index of the current candidate (curMatch)
    int best;       // best_len (:483)
    int left;       // chainLength still available (:477)
    int off;        // VERIFY: bytes of the candidate already known to match
    int mincl;      // candidates below this LDS index end the walk (limit / window index 0, :609)
    int mode;
    uint32_t pb;    // byte of the position at offset best (scan_end, :505)
    uint32_t res2, resq;
};

__device__ __forceinline__ void vswap(int &a, int &b) { asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void vswap(uint32_t &a, uint32_t &b) { asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

// EDGE: the tile reaches within 258 bytes of the end of its segment, so min(258, lookahead) / min(niceLength, lookahead)
// (:479,:485) depend on the position; everywhere else they are the constants 258 and niceLength.
template <int NCTX, bool EDGE, bool DBG>
__device__ __forceinline__ void match2_body(uint8_t *smem, const TileDev &tile, const SegDev &seg, const uint8_t *__restrict__ in,
                                            const uint16_t *__restrict__ link, MTab mtab, LevelParams P, unsigned long long *dbg,
                                            int fth, int vth, int qkeep, int vkeep) {
    uint32_t *sdata32 = (uint32_t *)smem;                          // B2_DATA_BYTES
    uint16_t *slink = (uint16_t *)(smem + B2_DATA_BYTES);          // B2_LINKS entries
    int *s_counter = (int *)(smem + B2_DATA_BYTES + B2_LINKS * 2);
    const uint8_t *d = in + seg.buf_off;
    const uint16_t *lk = link + seg.buf_off;
    uint32_t *__restrict__ mt2 = mtab.m2 + seg.buf_off;
    uint32_t *__restrict__ mtq = mtab.mq + seg.buf_off;
    const int64_t t0 = tile.start;
    const int tlen = tile.len;
    const int64_t dlo = t0 - B_HIST; // buffer position of LDS data byte 0 (may be negative)
    const int64_t seg_end = seg.seg_end;

    // ---- stage the window into LDS (bytes and links of history + tile)
    for (int i = threadIdx.x; i < B2_DATA_BYTES / 4; i += B2_THREADS) {
        int64_t pos = dlo + 4 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos + 4 <= seg_end) w = load_u32_unaligned2(d + pos);
        else {
            for (int k = 0; k < 4; k++) {
                int64_t pk = pos + k;
                if (pk >= 0 && pk < seg_end) w |= (uint32_t)d[pk] << (8 * k);
            }
        }
        sdata32[i] = w;
    }
    for (int i = threadIdx.x; i < B2_LINKS / 2; i += B2_THREADS) {
        int64_t pos = dlo + 2 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos + 2 <= t0 + tlen) {
            uint16_t a = lk[pos], c = lk[pos + 1];
            w = (uint32_t)a | ((uint32_t)c << 16);
        } else {
            if (pos >= 0 && pos < t0 + tlen) w |= lk[pos];
            if (pos + 1 >= 0 && pos + 1 < t0 + tlen) w |= (uint32_t)lk[pos + 1] << 16;
        }
        // "no previous position" (0) is staged as 0xFFFF: cl - 0xFFFF is below every limit (real links are <= 32767)
        if ((w & 0xFFFFu) == 0) w |= 0xFFFFu;
        if ((w >> 16) == 0) w |= 0xFFFF0000u;
        ((uint32_t *)slink)[i] = w;
    }
    if (threadIdx.x == 0) *s_counter = 0;
    __syncthreads();

    auto ldsdw = [&](int i) -> uint32_t { // unaligned 32-bit read of LDS data byte i from two aligned dwords
        uint32_t w0 = sdata32[i >> 2], w1 = sdata32[(i >> 2) + 1];
        return __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)(i & 3));
    };
    const uint8_t *sdata8 = smem;

    const int64_t base_lo = base_of2((int64_t)seg.abs0 + t0), base_hi = base_of2((int64_t)seg.abs0 + t0 + tlen - 1);
    const int64_t sw64 = base_lo == base_hi ? (int64_t)1 << 30 : (base_lo + 65273) - (int64_t)seg.abs0 - t0; // first tile position on base_hi
    const int sw = sw64 > (int64_t)B_TILE ? B_TILE : (int)sw64;
    const int basem_lo = (int)(base_lo - (int64_t)seg.abs0 - dlo), basem_hi = (int)(base_hi - (int64_t)seg.abs0 - dlo);
    // lookahead at tile position 0, clamped (only its value below 258 matters): rem(p) = rem0 - p
    const int64_t rem0_64 = seg_end - t0;
    const int rem0 = rem0_64 > (int64_t)(1 << 24) ? (1 << 24) : (int)rem0_64;
    const int lane = threadIdx.x & 63;
    const uint64_t lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int SNAPLEFT = P.max_chain - (P.max_chain >> 2); // `left` value at which the quarter-budget walk would stop
    enum { NEED = 0, DONE = 1, QUICK = 2, VERIFY = 3 };       // idle contexts are those with mode < QUICK
    int wnext = 0, wend = 0;       // wave-uniform slice of tile positions being handed out
    bool exhausted = false;

    WalkCtx A, B;
    A.p = 0; A.cl = B_HIST; A.best = 2; A.left = 1; A.off = 0; A.mincl = 0; A.mode = NEED;
    A.pb = 0; A.res2 = 0; A.resq = 0;
    B = A;
    auto cap_of = [&](int p) -> int { if (!EDGE) return MAX_MATCH; const int rem = rem0 - p; return rem < MAX_MATCH ? rem : MAX_MATCH; };   // scanMax :479
    auto nice_of = [&](int p) -> int { if (!EDGE) return P.nice; const int rem = rem0 - p; return rem < P.nice ? rem : P.nice; };          // :485
    unsigned long long c_qvis = 0, c_qsteps = 0, c_qlanes = 0, c_vvis = 0, c_vsteps = 0, c_vlanes = 0, c_fvis = 0, c_flanes = 0,
                       c_swaps = 0, c_swaplanes = 0;

    auto swap_ab = [&](bool need) { // branch-free exchange (two selects per field)
        if (NCTX == 2) {
#define SZL_SW(f) { const auto ta = A.f, tb = B.f; A.f = need ? tb : ta; B.f = need ? ta : tb; }
            SZL_SW(p) SZL_SW(cl) SZL_SW(best) SZL_SW(left) SZL_SW(off) SZL_SW(mincl) SZL_SW(mode) SZL_SW(pb) SZL_SW(res2) SZL_SW(resq)
#undef SZL_SW
        }
    };
    auto retire = [&](WalkCtx &C) { // results of a finished walk
        if (C.mode == DONE) { mt2[t0 + C.p] = C.res2; mtq[t0 + C.p] = C.resq; C.mode = NEED; }
    };

    for (;;) {
        uint64_t ia = __ballot(A.mode < QUICK), va = __ballot(A.mode == VERIFY);
        uint64_t ib = 0, vb = 0;
        if (NCTX == 2) { ib = __ballot(B.mode < QUICK); vb = __ballot(B.mode == VERIFY); }
        const uint64_t qa = ~(ia | va), qb = NCTX == 2 ? ~(ib | vb) : 0ull;
        const int ni = __builtin_popcountll(ia | ib), nv = __builtin_popcountll(va | vb), nq = __builtin_popcountll(qa | qb);
        if ((ni >= fth && !exhausted) || (nq == 0 && nv == 0)) {
            // ---------------- FETCH: retire finished walks, hand out new positions (lanes whose A is busy but B is idle bring B forward)
            retire(A);
            if (NCTX == 2) retire(B);
            if (!exhausted) {
                if (NCTX == 2) {
                    const bool sw_need = A.mode >= QUICK && B.mode < QUICK;
                    if (DBG) { const uint64_t m = __ballot(sw_need); if (m) { c_swaps++; c_swaplanes += __builtin_popcountll(m); } }
                    swap_ab(sw_need);
                }
                const uint64_t idle = __ballot(A.mode == NEED);
                const int nidle = __builtin_popcountll(idle);
                if (wnext >= wend) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(s_counter, 512);
                    base = __builtin_amdgcn_readfirstlane(base);
                    wnext = base < tlen ? base : tlen;
                    wend = base + 512 < tlen ? base + 512 : tlen;
                    if (wnext >= wend) exhausted = true;
                }
                if (!exhausted) {
                    const int rank = __builtin_popcountll(idle & lanemask_lt);
                    if (DBG) { c_fvis++; c_flanes += (wend - wnext) < nidle ? (wend - wnext) : nidle; }
                    if (A.mode == NEED && wnext + rank < wend) {
                        const int p = wnext + rank;
                        A.p = p;
                        const int rem = rem0 - p;                                   // lookahead (clamped high)
                        A.res2 = 0; A.resq = 0;
                        bool ok = rem >= MIN_MATCH && P.strategy != 2;              // :780, HuffmanOnly :786
                        if (ok) {
                            const int pl = p + B_HIST;
                            const int l0 = (int)slink[pl];                           // hashHead (:782)
                            const int basem = p >= sw ? basem_hi : basem_lo;        // LDS index of window index 1 (entries below were clamped by a slide, :450-461)
                            const int firstmin = pl - MAX_DIST > basem ? pl - MAX_DIST : basem; // first candidate: strstart - hashHead <= MAX_DIST (:788)
                            const int c = pl - l0;
                            ok = c >= firstmin;                                      // l0 == 0xFFFF (none) fails this too
                            if (ok) {
                                A.cl = c;
                                A.mincl = pl - (MAX_DIST - 1) > basem ? pl - (MAX_DIST - 1) : basem; // chain: curMatch > limit (:609)
                                A.best = 2; A.left = P.max_chain;
                                A.pb = sdata8[pl + 2];
                                // the first candidate shares the position's 3-byte hash, so the scan_end test at offset 2 (:505) all but
                                // always passes: compare it right away instead of spending a chain step on that test
                                A.off = 0;
                                A.mode = VERIFY;
                            }
                        }
                        if (!ok) { mt2[t0 + p] = 0u; mtq[t0 + p] = 0u; }
                    }
                    wnext = wnext + nidle < wend ? wnext + nidle : wend;
                }
            }
            if (exhausted) {
                const bool busy = A.mode != NEED || (NCTX == 2 && B.mode != NEED);
                if (!__any(busy)) break;
            }
            continue;
        }
        if (nv >= vth || nq == 0) {
            // ---------------- VERIFY: dword-by-dword comparison of the candidates that passed the scan_end test
            if (NCTX == 2) {
                const bool sw_need = A.mode != VERIFY && B.mode == VERIFY;
                if (DBG) { const uint64_t m = __ballot(sw_need); if (m) { c_swaps++; c_swaplanes += __builtin_popcountll(m); } }
                swap_ab(sw_need);
            }
            if (DBG) c_vvis++;
            uint64_t vm = __ballot(A.mode == VERIFY);
            do {
                if (DBG) { c_vsteps++; c_vlanes += __builtin_popcountll(vm); }
                if (__builtin_amdgcn_inverse_ballot_w64(vm)) {
                    const int pl = A.p + B_HIST;
                    const int cap = cap_of(A.p);
                    const uint32_t x = ldsdw(A.cl + A.off) ^ ldsdw(pl + A.off);
                    const bool eq = x == 0;
                    const int l = A.off + (eq ? 4 : (__builtin_ctz(x) >> 3));
                    const bool more = eq & (l < cap);
                    A.off = l;
                    if (!more) {
                        const int L = l < cap ? l : cap;
                        const int lnk = (int)slink[A.cl];          // prev[] hop of this candidate
                        bool nicehit = false;
                        if (L > A.best) { // :593-607
                            A.best = L;
                            A.res2 = (uint32_t)L | ((uint32_t)(pl - A.cl) << 16);
                            if (A.left > SNAPLEFT) A.resq = A.res2; // among the first max_chain>>2 candidates: the quarter walk sees it too
                            nicehit = L >= nice_of(A.p);
                            if (!nicehit) A.pb = sdata8[pl + L];
                        }
                        const int left1 = A.left - 1;
                        const int c2 = A.cl - lnk;
                        const bool end = (c2 < A.mincl) | (left1 == 0);
                        A.left = nicehit ? A.left : left1;
                        A.cl = (nicehit | end) ? A.cl : c2;
                        A.off = 0;
                        A.mode = (nicehit | end) ? DONE : QUICK;
                    }
                }
                vm = __ballot(A.mode == VERIFY);
            } while (__builtin_popcountll(vm) >= vkeep);
            continue;
        }
        // ---------------- QUICK: chain steps — two LDS reads (scan_end byte of the candidate, its prev[] hop) and a dozen VALU
        if (NCTX == 2) {
            const bool sw_need = A.mode != QUICK && B.mode == QUICK;
            if (DBG) { const uint64_t m = __ballot(sw_need); if (m) { c_swaps++; c_swaplanes += __builtin_popcountll(m); } }
            swap_ab(sw_need);
        }
        if (DBG) c_qvis++;
        {
            uint64_t qm = __ballot(A.mode == QUICK);
            do {
                if (DBG) { c_qsteps++; c_qlanes += __builtin_popcountll(qm); }
                if (__builtin_amdgcn_inverse_ballot_w64(qm)) {
                    const uint32_t qbyte = sdata8[A.cl + A.best];  // a longer match must agree at offset `best` (scan_end, :505)
                    const int lnk = (int)slink[A.cl];
                    const bool pass = qbyte == A.pb;
                    const int left1 = A.left - 1;
                    const int c2 = A.cl - lnk;                    // next candidate of the chain, or the end of this position (:609)
                    const bool end = (c2 < A.mincl) | (left1 == 0);
                    A.left = pass ? A.left : left1;
                    A.cl = (pass | end) ? A.cl : c2;
                    A.mode = pass ? VERIFY : (end ? DONE : QUICK);
                }
                qm = __ballot(A.mode == QUICK);
            } while (__builtin_popcountll(qm) >= qkeep);
        }
    }
    if (DBG && dbg) {
        if (lane == 0) {
            atomicAdd(dbg + 8, c_qvis); atomicAdd(dbg + 9, c_qsteps); atomicAdd(dbg + 10, c_qlanes);
            atomicAdd(dbg + 11, c_vvis); atomicAdd(dbg + 12, c_vsteps); atomicAdd(dbg + 13, c_vlanes);
            atomicAdd(dbg + 14, c_fvis); atomicAdd(dbg + 15, c_flanes); atomicAdd(dbg + 16, c_swaps); atomicAdd(dbg + 17, c_swaplanes);
        }
    }
}

template <int NCTX, bool DBG>
__global__ __launch_bounds__(B2_THREADS) void k_match2(const uint8_t *__restrict__ in, const SegDev *__restrict__ segs,
                                                       const TileDev *__restrict__ tiles, const uint16_t *__restrict__ link,
                                                       MTab mtab, LevelParams P, unsigned long long *dbg, int fth, int vth, int qkeep, int vkeep) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_k2[];
    const TileDev tile = tiles[blockIdx.x];
    const SegDev seg = segs[tile.seg];
    // lookahead of the tile's last position >= 258 (wave-uniform: one of the two bodies runs)
    if (seg.seg_end - (tile.start + tile.len - 1) >= (int64_t)MAX_MATCH)
        match2_body<NCTX, false, DBG>(smem_k2, tile, seg, in, link, mtab, P, dbg, fth, vth, qkeep, vkeep);
    else
        match2_body<NCTX, true, DBG>(smem_k2, tile, seg, in, link, mtab, P, dbg, fth, vth, qkeep, vkeep);
}

static bool lds_attr_needed2(std::atomic<uint64_t> &mask, uint64_t &bit) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bit = 1ull << (dev & 63);
    return (mask.load(std::memory_order_acquire) & bit) == 0;
}

hipError_t launch_match2(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int ntiles, const uint16_t *link,
                         MTab mtab, LevelParams P, unsigned long long *dbg, hipStream_t st) {
    static std::atomic<uint64_t> attr_mask{0};
    uint64_t attr_bit = 0;
    const bool want_dbg = knob("SZL_DEBUG", 0) != 0;
    const int nctx = knob("SZL_NCTX", 2);
    const int fth = knob("SZL_FTH2", 16), vth = knob("SZL_VTH2", 24), qkeep = knob("SZL_QKEEP", 40), vkeep = knob("SZL_VKEEP", 16);
    // the kernel body exists in two forms: tiles that touch the end of their segment (EDGE) and all the others
    auto attr = [&](const void *f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, B2_LDS_BYTES); };
    if (lds_attr_needed2(attr_mask, attr_bit)) {
        hipError_t e = attr((const void *)k_match2<1, false>);
        if (e == hipSuccess) e = attr((const void *)k_match2<2, false>);
        if (e == hipSuccess) e = attr((const void *)k_match2<1, true>);
        if (e == hipSuccess) e = attr((const void *)k_match2<2, true>);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(attr_bit, std::memory_order_release);
    }
    const int qk = qkeep < 1 ? 1 : qkeep, vk = vkeep < 1 ? 1 : vkeep;
    if (ntiles > 0) {
        const dim3 g(ntiles), b(B2_THREADS);
        if (nctx == 2) {
            if (want_dbg) hipLaunchKernelGGL((k_match2<2, true>), g, b, B2_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, qk, vk);
            else hipLaunchKernelGGL((k_match2<2, false>), g, b, B2_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, qk, vk);
        } else {
            if (want_dbg) hipLaunchKernelGGL((k_match2<1, true>), g, b, B2_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, qk, vk);
            else hipLaunchKernelGGL((k_match2<1, false>), g, b, B2_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, qk, vk);
        }
    }
    return hipGetLastError();
}

} // namespace szl
