// szl_kernels_fast.hip — levels 1-4: DeflateFast (C/DeflaterEngine.cs:651-739) on the device.
//
// DeflateFast is greedy AND its dictionary depends on its own output: a position enters the hash chains only if
// the parse visits it or it lies inside a match of length <= max_lazy (:697-708); everything inside a longer match
// is skipped.  That makes the parse a true sequential recurrence per stream (a range-parallel fixpoint over the
// "inserted" flags converges only one or two ranges per sweep — measured in oracle/szl_model.c before this was
// written), so the unit of parallelism here is the stream: ONE WAVEFRONT PER SEGMENT, many segments per launch.
//
// What the wavefront does with its 64 lanes:
//   * the reference's head/prev chain of a position is the all-positions chain of stage A (k_links, reused as is)
//     filtered by an "inserted" bit — a hop over a never-inserted position costs no chain budget;
//   * the last 32 Ki links (64 KiB), a 64 KiB ring of input bytes and the 32 Ki inserted bits live in LDS, so a
//     chain hop is an LDS read, not a trip to HBM;
//   * the byte comparison of a candidate (up to 258 bytes, :518-591) is one step: lane i compares bytes 4i..4i+3
//     and a ballot finds the first mismatch.
// Tokens are written at tokens[tok_base + k] (tok_base = bytes of the preceding segments), block starts / the window
// base at each FlushBlock go to the stage-D tables; stage D then runs unchanged apart from the two DeflateFast
// rules flagged by LevelParams.fast (k_seg_blocks, k_block_build).
#include <hip/hip_runtime.h>
#include <atomic>
#include "szl_internal.h"

namespace szl {

// hipFuncSetAttribute is per device: remember which devices have the large-LDS attribute for a kernel group
static bool lds_attr_needed(std::atomic<uint64_t> &mask, uint64_t &bit) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bit = 1ull << (dev & 63);
    return (mask.load(std::memory_order_acquire) & bit) == 0;
}

enum : int { F_DATA = 65536, F_LINKS = 32768, F_FLAGW = 1024 };

struct FastLds {
    uint32_t data[F_DATA / 4];   // byte q at data[q & 65535]
    uint16_t link[F_LINKS];      // link of position q at [q & 32767]
    uint32_t flag[F_FLAGW];      // inserted bit of position q: word (q>>5)&1023, bit q&31
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// window base after FillWindow ran at absolute position s (>= rule, :371)
__device__ __forceinline__ int64_t base_ge(int64_t s_abs) {
    int64_t idx = s_abs + 1;
    if (idx <= 65273) return 0;
    return ((idx - 65273 + 32767) >> 15) << 15;
}

__global__ __launch_bounds__(64) void k_fast(const uint8_t *__restrict__ in, const uint16_t *__restrict__ link,
                                             const SegDev *__restrict__ segs, uint32_t nseg, LevelParams P,
                                             uint32_t *fbits, SegOut *so, uint32_t *tokens, const uint64_t *blk_off,
                                             int64_t *blk_start_pos, int64_t *blk_base) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    FastLds &S = *(FastLds *)smem;
    const uint32_t si = blockIdx.x;
    if (si >= nseg) return;
    const SegDev s = segs[si];
    const int lane = threadIdx.x;
    const uint8_t *d = in + s.buf_off;
    const uint16_t *lk = link + s.buf_off;
    uint32_t *fb = fbits + s.vis_word_off;            // bit q = buffer position q of this segment's stream window
    const int64_t seg_start = s.seg_start, seg_end = s.seg_end;
    const uint64_t tok_base = s.range_off;            // fast mode: token base of this segment
    const uint64_t b0 = blk_off[si];
    uint8_t *sdata8 = (uint8_t *)S.data;

    // ---- preload history: bytes, links, inserted bits of [h0, seg_start)
    const int64_t h0 = seg_start > WSIZE ? seg_start - WSIZE : 0;
    int64_t dend = h0;   // bytes [dend-65536, dend) resident
    int64_t lend = h0;   // links [lend-32768, lend) resident
    auto load_data = [&](int64_t upto) { // make bytes < upto resident (whole dwords)
        while (dend < upto) {
            // 64 lanes x 16 bytes
            const int64_t q = dend + 16 * (int64_t)lane;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int64_t qq = q + 4 * k;
                uint32_t w = 0;
                if (qq + 4 <= seg_end) {
                    w = (uint32_t)d[qq] | ((uint32_t)d[qq + 1] << 8) | ((uint32_t)d[qq + 2] << 16) | ((uint32_t)d[qq + 3] << 24);
                } else {
                    for (int b = 0; b < 4; b++) if (qq + b < seg_end) w |= (uint32_t)d[qq + b] << (8 * b);
                }
                S.data[((uint32_t)qq & (F_DATA - 1)) >> 2] = w;
            }
            dend += 1024;
        }
    };
    // dend must stay dword aligned relative to the ring: start it at a multiple of 4
    dend = h0 & ~(int64_t)3;
    auto load_links = [&](int64_t upto) { // make links < upto resident
        while (lend < upto) {
            const int64_t q = lend + lane;
            S.link[(uint32_t)q & (F_LINKS - 1)] = q < seg_end ? lk[q] : (uint16_t)0;
            lend += 64;
        }
    };
    for (int i = lane; i < F_FLAGW; i += 64) S.flag[i] = 0;
    __syncthreads();
    if (seg_start > h0) { // history flags
        const int64_t w0 = h0 >> 5, w1 = (seg_start + 31) >> 5;
        for (int64_t w = w0 + lane; w < w1; w += 64) S.flag[(uint32_t)w & (F_FLAGW - 1)] = fb[w];
    }
    load_links(seg_start);
    __syncthreads();

    // set bits [q0, q1) to v (q1 - q0 <= 258): lane k owns word (q0>>5)+k
    auto flag_set = [&](int64_t q0, int64_t q1, int v) {
        const int64_t w = (q0 >> 5) + lane;
        const int64_t lo = w << 5, hi = lo + 32;
        const int64_t a = q0 > lo ? q0 : lo, b = q1 < hi ? q1 : hi;
        if (a < b) {
            const int nb = (int)(b - a);
            const uint32_t mask = (nb == 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u)) << (int)(a - lo);
            uint32_t &word = S.flag[(uint32_t)w & (F_FLAGW - 1)];
            word = v ? (word | mask) : (word & ~mask);
        }
    };
    auto ldsdw = [&](uint32_t i) -> uint32_t { // unaligned dword at ring byte index i
        const uint32_t w0 = S.data[(i & (F_DATA - 1)) >> 2], w1 = S.data[((i + 4) & (F_DATA - 1)) >> 2];
        return __builtin_amdgcn_alignbyte(w1, w0, i & 3);
    };

    int64_t x = seg_start;
    int64_t base = base_ge((int64_t)s.abs0 + x); // FillWindow runs when input arrives (:371)
    uint64_t ntok = 0;
    bool refilled = true; // FillWindow ran just before this iteration (segment start, or right after a block flush)
    LevelParams Pc = P;                   // parameters of the running iteration (SetLevel / SetStrategy inside the segment: SegDev.sw_*)
    bool search = Pc.strategy != 2;       // HuffmanOnly :686
    uint32_t swk = 0;

    const bool cut = (s.flags & SEG_SWITCH_CUT) != 0;   // SetLevel to another compression function: the engine stands at the first iteration start >= cut_pos (:681)
    while (x < seg_end) {
        if (cut && x >= s.cut_pos) break;
        // entries of sw_pos are ascending: a cursor.  An entry whose `fast` has bit 1 set marks a SetInput boundary: the engine had run
        // out of lookahead there (it stops at the first iteration start within MIN_LOOKAHEAD - 1 of the input it has, :681) and the
        // next Deflate() call began with FillWindow() — whose slide test is `>=` where this loop's own is `>` (:371 vs :680)
        while (swk < s.sw_cnt && x >= s.sw_pos[swk]) {
            Pc = s.sw_P[swk++];
            if (Pc.fast & 2) refilled = true;
            Pc.fast &= 1;
            search = Pc.strategy != 2;
        }
        // ---- window slide (:680 strict; FillWindow :371 non-strict)
        {
            const int64_t idx = (int64_t)s.abs0 + x + 1 - base;
            if (idx > 65274 || (refilled && idx >= 65274)) base += WSIZE;
            refilled = false;
        }
        if ((ntok & (BLOCK_TOKENS - 1)) == 0 && lane == 0) blk_start_pos[b0 + ntok / BLOCK_TOKENS] = x;
        const int64_t rem = seg_end - x;
        // ---- residency
        if (dend < x + 272 && dend < ((seg_end + 3) & ~(int64_t)3)) { load_data(x + 272 + 1024 < seg_end + 4 ? x + 272 + 1024 : seg_end + 4); __syncthreads(); }
        if (lend <= x) { load_links(x + 1); __syncthreads(); }

        uint32_t best = 2, bdist = 0;
        if (rem >= MIN_MATCH && search) {
            const int64_t idx_p = (int64_t)s.abs0 + x + 1 - base;
            const int64_t limit_idx = idx_p - MAX_DIST > 0 ? idx_p - MAX_DIST : 0; // :480
            const int64_t basem = base - (int64_t)s.abs0;                            // buffer position of window index 0
            const int cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
            const int nice = rem < (int64_t)Pc.nice ? (int)rem : Pc.nice;
            // One LDS round trip per chain hop: the hop's link, its inserted bit and its bytes (lane i: bytes 4i..4i+3,
            // compared against the position's own bytes fetched once) are requested together.
            const uint32_t off = 4u * (uint32_t)lane;
            const uint32_t xdw = ldsdw((uint32_t)x + off);
            int lnk = rfl((int)S.link[(uint32_t)x & (F_LINKS - 1)]);
            int64_t c = x;
            bool first = true;   // still looking for hashHead, the newest INSERTED position of this hash (:686)
            int budget = Pc.max_chain;
            // Path compression: the inserted bits of positions behind x are final, so a run of never-inserted positions between two
            // chain elements is walked once — the element in front of it (`from`) then links straight to the next inserted one in
            // this wavefront's LDS copy of the links (distances < 65536: both lie inside the 32 Ki links resident here).
            int64_t from = x;
            bool skipped = false;
            for (;;) {
                if (lnk == 0) { if (skipped && lane == 0) S.link[(uint32_t)from & (F_LINKS - 1)] = 0; break; }
                c -= lnk;
                if (first) {
                    if (x - c > MAX_DIST) break;              // strstart - hashHead <= MAX_DIST :687 (older hops are farther still)
                    if (c + 1 - basem < 1) break;             // head entry clamped to 0 by a slide :450-461
                } else if (c + 1 - basem <= limit_idx) break; // (curMatch = prev[..]) > limit :609
                const uint32_t l2 = S.link[(uint32_t)c & (F_LINKS - 1)];
                const uint32_t fw = S.flag[((uint32_t)c >> 5) & (F_FLAGW - 1)];
                const uint32_t cdw = ldsdw((uint32_t)c + off);
                lnk = rfl((int)l2);
                if (((rfl((int)fw) >> ((uint32_t)c & 31)) & 1) == 0) { skipped = true; continue; } // never inserted: not part of the reference's chain
                if (skipped && lane == 0) S.link[(uint32_t)from & (F_LINKS - 1)] = (uint16_t)(from - c);
                from = c; skipped = false;
                first = false;
                // length of the common prefix of c and x, capped (:505-591)
                const uint32_t xr = cdw ^ xdw;
                const uint64_t ne = __ballot(xr != 0);
                int L;
                if (ne) {
                    const int f = __builtin_ctzll(ne);
                    const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)xr, f);
                    L = 4 * f + (__builtin_ctz(xf) >> 3);
                } else {
                    L = 256;
                    if (cap > 256) { // bytes 256, 257
                        const uint32_t xt = rfl((int)(ldsdw((uint32_t)c + 256u) ^ ldsdw((uint32_t)x + 256u)));
                        L += (xt & 0xFFu) ? 0 : ((xt & 0xFF00u) ? 1 : 2);
                    }
                }
                if (L > cap) L = cap;
                if ((uint32_t)L > best) {
                    best = (uint32_t)L; bdist = (uint32_t)(x - c);
                    if (L >= nice) break; // :604
                }
                if (--budget == 0) break; // 0 != --chainLength :609
            }
        }
        // ---- token, inserted bits, advance (:689-725)
        uint32_t tok;
        int64_t nx;
        if (best >= MIN_MATCH) {
            tok = (bdist << 16) | best;
            const int ins = ((int)best <= Pc.max_lazy && rem - (int64_t)best >= MIN_MATCH) ? 1 : 0; // :697
            if (ins) flag_set(x, x + best, 1);
            else { // (two calls: the bits of x and of the interior can share a word, and flag_set gives each lane one word)
                flag_set(x, x + 1, 1);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                flag_set(x + 1, x + best, 0);
            }
            nx = x + best;
        } else {
            tok = (uint32_t)sdata8[(uint32_t)x & (F_DATA - 1)];
            flag_set(x, x + 1, rem >= MIN_MATCH ? 1 : 0);
            nx = x + 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (lane == 0) tokens[tok_base + ntok] = tok;
        ntok++;
        x = nx;
        if ((ntok & (BLOCK_TOKENS - 1)) == 0) { // IsFull -> FlushBlock (:727-736): storedOffset is judged against the current base
            if (lane == 0) blk_base[b0 + ntok / BLOCK_TOKENS - 1] = base;
            refilled = true; // Deflate() returns to drain pending; the next call starts with FillWindow
        }
    }
    // final (possibly empty) block :664-668
    if (lane == 0) {
        if ((ntok & (BLOCK_TOKENS - 1)) != 0 || ntok == 0) blk_base[b0 + ntok / BLOCK_TOKENS] = base;
        else if (!s.finish) blk_base[b0 + ntok / BLOCK_TOKENS] = base;
        so[si].tok_first = tok_base;
        so[si].tok_count = ntok;
        so[si].cut_x = x;                 // (SEG_SWITCH_CUT: where the next function starts)
    }
    // inserted bits of the last 32 Ki positions -> global (history of the next segment of this stream)
    __syncthreads();
    {
        const int64_t t0 = seg_end > WSIZE ? seg_end - WSIZE : 0;
        const int64_t w0 = t0 >> 5, w1 = (seg_end + 31) >> 5;
        for (int64_t w = w0 + lane; w < w1; w += 64) fb[w] = S.flag[(uint32_t)w & (F_FLAGW - 1)];
    }
}

int fast_lds_bytes() { return (int)sizeof(FastLds); }

hipError_t launch_fast(const uint8_t *in, const uint16_t *link, const SegDev *segs, uint32_t nseg, LevelParams P, uint32_t *fbits,
                       SegOut *so, uint32_t *tokens, const uint64_t *blk_off, int64_t *bsp, int64_t *blp, hipStream_t st) {
    static std::atomic<uint64_t> attr_mask{0};
    uint64_t attr_bit = 0;
    if (lds_attr_needed(attr_mask, attr_bit)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(FastLds));
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(attr_bit, std::memory_order_release);
    }
    if (nseg) hipLaunchKernelGGL(k_fast, dim3(nseg), dim3(64), sizeof(FastLds), st, in, link, segs, nseg, P, fbits, so, tokens, blk_off, bsp, blp);
    return hipSuccess;
}

} // namespace szl
