// szl_kernels_match5.hip — stage B (FindLongestMatch for every position) in BUCKET ORDER: k_match5.
//
// Reference being restated: FindLongestMatch, C/DeflaterEngine.cs:474-612, on the hash chains InsertString builds
// (:402-424).  Same tables as k_match4: M2 = the walk entered with matchLen 2 and the full max_chain budget, Mq = that walk's
// state after max_chain>>2 candidates (:495).
//
// Why another form (round 3; profiles/r02/pmc_sq_final_1gib.json, profiles/r03/lab_r3a_*.log).  k_match4 walks prev[] links:
// every chain step is a dependent LDS round trip at a random address (three LDS reads per step, 41 % of them bank conflicts),
// the walks of a wavefront's lanes have unrelated lengths (27 % lane efficiency), and history + tile + links at 3 bytes per
// position limit a tile to 21504 positions (a third of the kernel's time is tile overhead).  At levels 5-9 every position
// is inserted, so the chain of p is simply "the earlier positions with p's hash, nearest first".  Here each tile SORTS its
// window (32512 positions of history + up to 65536 of tile) by (hash bucket, position) first — a counting sort in LDS,
// written to a per-workgroup scratch array S in global memory — and the search then runs in S order:
//   * the chain of the entry at S[i] is S[i-1], S[i-2], ... down to the start of its bucket: consecutive addresses, known
//     in advance, read coalesced from S and staged per wavefront in LDS (conflict-free ds_read_b64 per step, no hops);
//   * the 64 lanes of a wavefront hold 64 consecutive entries — mostly ONE bucket — so their walks have almost the same
//     length and run in lockstep (CPU model tools/bucket_model.c: 59-66 % of the lane-steps do work with no refill logic at
//     all, and the tables equal the restated reference walk on every position);
//   * an entry carries 47 content bits besides its position: the 9 bits of bytes 0-2 the hash does not determine, bytes
//     3-6 and six bits of byte 7.  The first differing bit gives the exact common prefix up to 7 bytes from the entries
//     alone; only candidates that agree on all 47 bits (about a fifth on text) touch the window's bytes in LDS;
//   * LDS holds the window's BYTES only (1 byte per position instead of 3): a tile is 65536 positions.
// FindLongestMatch's result is order-free once written as a maximum: the first candidate (in chain order) that reaches
// niceLength wins, otherwise the longest, the nearest among equals; the budget, the `limit` test, the window-base rule
// (entries clamped by a slide are null, App. A.2) and the quarter-budget snapshot are position tests on the sorted run.
//
// The sort: (1) histogram of the window's inserted positions by hash (LDS atomics, order-free); (2) exclusive scan of the
// 32768 counters — thread t owns buckets t, t+1024, ...: the ORDER OF THE BUCKETS in S is irrelevant, only each bucket's
// entries must be contiguous and ascending — plus a bitmap of bucket starts; (3) ordered scatter: the window is cut into
// 64-position slices, four slices form a group, an LDS ticket passes from group to group (as in k_links3), the holder adds
// each bucket's count to its running offset (ds_add_rtn) and passes the ticket on.  No assumption about the order in which
// the LDS unit serves the lanes of one instruction: the lanes of a slice that share a bucket are found by a ballot
// multisplit and only the lowest of them issues the add.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <climits>
#include "szl_internal.h"

namespace szl {
int knob(const char *name, int dflt);

enum : int { B5_THREADS = 1024, B5_WAVES = 16, B5_TMAX = 65536, B5_WMAX = B_HIST + B5_TMAX };
// LDS.  Sort phase: the 32768 bucket counters, then (above everything the search phase touches) the bucket-start bitmap and the control
// words.  Search phase: the window's bytes, then one private area per wavefront — 192 staged entries of S, the queue of candidates
// that need the window's bytes (candidate position << 13 | lane << 7 | step), 64 + 64 result slots and the positions' own data.
enum : int { B5_CNT_BYTES = 32768 * 4 };
enum : int { B5_DATA_BYTES = (B5_WMAX + B_TAIL + 8 + 15) & ~15, B5_STG_ENTRIES = 192, B5_QCAP = 192 };
enum : int { B5_WV_STG = 0, B5_WV_QUEUE = B5_STG_ENTRIES * 8, B5_WV_SLOT2 = B5_WV_QUEUE + B5_QCAP * 4, B5_WV_SLOTQ = B5_WV_SLOT2 + 256, B5_WV_PINFO = B5_WV_SLOTQ + 256, B5_WV_BYTES = B5_WV_PINFO + 256 };
enum : int { B5_WV_OFF = B5_DATA_BYTES, B5_BITMAP_WORDS = (B5_WMAX + 63) / 64 + 4,
              B5_BITMAP_OFF = (B5_WV_OFF + B5_WAVES * B5_WV_BYTES > B5_CNT_BYTES ? B5_WV_OFF + B5_WAVES * B5_WV_BYTES : B5_CNT_BYTES),
              B5_CTL_OFF = B5_BITMAP_OFF + B5_BITMAP_WORDS * 8, B5_LDS_BYTES = B5_CTL_OFF + 64 };
static_assert(B5_WV_BYTES % 16 == 0 && B5_BITMAP_OFF % 8 == 0, "alignment of the per-wave areas");
static_assert(B5_LDS_BYTES <= 160 * 1024 && B5_WMAX < (1 << 17), "window: 17-bit positions, one CU's LDS");
enum : int { B5_CTL_TICKET = 0, B5_CTL_NS = 1, B5_CTL_SLICE = 2, B5_CTL_TILE = 3 };
// per-workgroup scratch: S (sorted entries), RES (the results, in S order) and RANK (S index of every window position)
enum : size_t { B5_S_BYTES = ((size_t)B5_WMAX * 8 + 4095) & ~(size_t)4095, B5_RANK_BYTES = ((size_t)B5_WMAX * 4 + 4095) & ~(size_t)4095,
                B5_SLOT_BYTES = 2 * B5_S_BYTES + B5_RANK_BYTES };
enum : int { B5_U = 4 };   // 64-position slices per ticket

typedef __attribute__((address_space(3))) uint8_t b5_lds_u8;

__device__ __forceinline__ int64_t base_of5(int64_t s_abs) { // window base of an iteration starting at s (App. A.2; C/DeflaterEngine.cs:371,:771,:93)
    int64_t idx = s_abs + 1;
    if (idx <= 65273) return 0;
    return ((idx - 65273 + 32767) >> 15) << 15;
}
// maximum over the wavefront without the LDS (DPP row shifts and broadcasts); the result is wave-uniform
__device__ __forceinline__ int wave_max_i32(int x) {
    int t;
    t = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); x = t > x ? t : x;   // row_shr:1
    t = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); x = t > x ? t : x;   // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); x = t > x ? t : x;   // row_shr:4
    t = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); x = t > x ? t : x;   // row_shr:8  (lane 15 of every row: the row's maximum)
    t = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); x = t > x ? t : x;   // row_bcast:15 into rows 1 and 3
    t = __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); x = t > x ? t : x;   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ uint32_t ffbl_m1(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }   // 0xFFFFFFFF for 0

// "is position q inserted into the hash chains": InsertString only runs while lookahead >= MIN_MATCH (:780,:817), so the
// last two positions in front of every segment end are not.  Same rule, same inputs as k_links3 (szl_kernels_match.hip).
struct InsRule5 {
    const uint64_t *b; int nb; int bi; int64_t bcur;
    __device__ __forceinline__ void init(const uint64_t *bb, int n) { b = bb; nb = n; bi = 0; bcur = nb > 0 ? (int64_t)b[0] : INT64_MAX; }
    __device__ __forceinline__ bool ins(int64_t q0, int64_t q, int64_t qend) {   // q0: slice start (wave-uniform, ascending between calls)
        while (bcur <= q0) { bi++; bcur = bi < nb ? (int64_t)b[bi] : INT64_MAX; }
        bool r = q < qend;
        if (bi >= nb) r = false;
        else if (bcur < q0 + 64 + 2) {
            int j = bi;
            while (j < nb && (int64_t)b[j] <= q) j++;
            r = r && j < nb && (int64_t)b[j] - q >= 3;
        }
        return r;
    }
};

// eight bytes at buffer position q (bytes past the arena's end read as 0)
__device__ __forceinline__ uint64_t load8(const uint8_t *d, int64_t q, uint64_t avail) {
    uint64_t v = 0;
    if ((uint64_t)q + 8 <= avail) __builtin_memcpy(&v, d + q, 8);
    else for (int k = 0; k < 8; k++) if ((uint64_t)(q + k) < avail) v |= (uint64_t)d[q + k] << (8 * k);
    return v;
}
__device__ __forceinline__ uint32_t hash_of(uint64_t w) { return ((((uint32_t)w & 0xFF) << 10) ^ ((((uint32_t)w >> 8) & 0xFF) << 5) ^ (((uint32_t)w >> 16) & 0xFF)) & 0x7FFF; }
// the sorted entry of window position `rel` whose eight bytes are w:
//   lo  [8:0] the bits of bytes 0-2 the hash does not determine, [16:9] byte 3, [24:17] byte 4, [31:25] byte 5 bits 6..0
//   hi  [0] byte 5 bit 7, [8:1] byte 6, [13:9] byte 7 bits 4..0, [14] zero, [31:15] window position
__device__ __forceinline__ uint2 entry_of(uint64_t w, uint32_t rel) {
    const uint32_t b0 = (uint32_t)w & 0xFF, b1 = ((uint32_t)w >> 8) & 0xFF, b2 = ((uint32_t)w >> 16) & 0xFF, b3 = (uint32_t)w >> 24;
    const uint32_t h = (uint32_t)(w >> 32), b4 = h & 0xFF, b5 = (h >> 8) & 0xFF, b6 = (h >> 16) & 0xFF, b7 = h >> 24;
    uint2 e;
    e.x = (b0 >> 5) | ((b1 >> 5) << 3) | ((b2 >> 5) << 6) | (b3 << 9) | (b4 << 17) | ((b5 & 0x7F) << 25);
    e.y = (b5 >> 7) | (b6 << 1) | ((b7 & 0x1F) << 9) | (rel << 15);
    return e;
}

// The byte pass of k_match5: up to 64 queued candidates (window position of the candidate << 13 | lane of the position << 7 |
// step), oldest first.  All 46 content bits of such a candidate agree with the position's, so their common prefix is 7 or more:
// test the two bytes at the position's running best (scan_end / scan_end1, :505-506), compare the window's bytes from offset 7 on,
// and raise the position's slot with an LDS atomic max — key = length (capped at niceLength) << 22 | (127 - step) << 15 | distance:
// the first longest candidate wins whatever the order inside a batch, and the length a slot holds when a candidate is tested comes
// from earlier batches only, i.e. from lower steps (a later equal candidate never hides an earlier one).
// The LDS pipe is what this pass costs: one read for the pair, two for the position's data and slot, the two scan_end bytes of each
// side only when the running best is 8 or more, and the bytes themselves only for the candidates that pass.
__device__ __noinline__ void b5_byte_pass(const uint8_t *smem, const uint32_t *queue, uint32_t *slot2, uint32_t *slotq, const uint32_t *pinfo,
                                          int nb, int pnice, int snapk, unsigned long long *cnt) {
    const int lane = threadIdx.x & 63;
    const uint8_t *sdata8 = smem;
    const uint32_t *sdata32 = (const uint32_t *)smem;
    const bool v = lane < nb;
    const uint32_t pr = v ? queue[lane] : 0u;
    const int pl = (int)((pr >> 7) & 63u), kk = (int)(pr & 127u), crel = (int)(pr >> 13);
    const uint32_t info = pinfo[pl], key0 = slot2[pl];
    const int pp = (int)(info & 0x1FFFFu), pc = (int)(info >> 17);
    const int pn = pc < pnice ? pc : pnice;
    const int b2 = (int)(key0 >> 22);
    bool go = v && b2 < pn;                                  // (a position that has reached niceLength takes nothing more, :603)
    if (go && b2 >= 8) {
        const uint32_t fc = ((uint32_t)sdata8[crel + b2] << 8) | sdata8[crel + b2 - 1], fp = ((uint32_t)sdata8[pp + b2] << 8) | sdata8[pp + b2 - 1];
        go = fc == fp;
    }
    if (go) {
        const int ca = crel + 7, pa = pp + 7;
        const uint32_t *cw = sdata32 + (ca >> 2), *pw = sdata32 + (pa >> 2);
        const uint32_t c0 = cw[0], c1 = cw[1], c2 = cw[2], c3 = cw[3], c4 = cw[4];
        const uint32_t p0 = pw[0], p1 = pw[1], p2 = pw[2], p3 = pw[3], p4 = pw[4];
        const uint32_t cs = (uint32_t)ca & 3u, ps = (uint32_t)pa & 3u;
        const uint32_t x0 = __builtin_amdgcn_alignbyte(c1, c0, cs) ^ __builtin_amdgcn_alignbyte(p1, p0, ps);
        const uint32_t x1 = __builtin_amdgcn_alignbyte(c2, c1, cs) ^ __builtin_amdgcn_alignbyte(p2, p1, ps);
        const uint32_t x2 = __builtin_amdgcn_alignbyte(c3, c2, cs) ^ __builtin_amdgcn_alignbyte(p3, p2, ps);
        const uint32_t x3 = __builtin_amdgcn_alignbyte(c4, c3, cs) ^ __builtin_amdgcn_alignbyte(p4, p3, ps);
        int L;
        if (x0) L = 7 + (int)(__builtin_ctz(x0) >> 3);
        else if (x1) L = 11 + (int)(__builtin_ctz(x1) >> 3);
        else if (x2) L = 15 + (int)(__builtin_ctz(x2) >> 3);
        else if (x3) L = 19 + (int)(__builtin_ctz(x3) >> 3);
        else {                                               // longer than 16 bytes past offset 7: the rest in a loop (rare on text)
            L = 23;
            while (L < pc) {
                const int i0 = crel + L, i1 = pp + L;
                const uint32_t a0 = sdata32[i0 >> 2], a1 = sdata32[(i0 >> 2) + 1], q0 = sdata32[i1 >> 2], q1 = sdata32[(i1 >> 2) + 1];
                const uint32_t x = __builtin_amdgcn_alignbyte(a1, a0, (uint32_t)(i0 & 3)) ^ __builtin_amdgcn_alignbyte(q1, q0, (uint32_t)(i1 & 3));
                if (x) { L += (int)(__builtin_ctz(x) >> 3); break; }
                L += 4;
            }
        }
        if (L > pc) L = pc;
        if (L > b2) {
            const uint32_t key = ((uint32_t)(L < pn ? L : pn) << 22) | ((uint32_t)(127 - kk) << 15) | (uint32_t)(pp - crel);
            atomicMax(&slot2[pl], key);
            if (kk < snapk) atomicMax(&slotq[pl], key);
            if (cnt) cnt[1]++;
        }
        if (cnt) cnt[0]++;
    }
}

template <bool DBG>
__global__ __launch_bounds__(B5_THREADS) void k_match5(const uint8_t *__restrict__ in, uint64_t in_total, const SegDev *__restrict__ segs,
                                                       const uint64_t *__restrict__ bnds, const TileDev *__restrict__ tiles, int ntiles,
                                                       MTab mtab, LevelParams P, uint8_t *__restrict__ scratch, unsigned int *__restrict__ tile_counter,
                                                       unsigned long long *dbg, int lab) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *s_cnt = (uint32_t *)smem;
    unsigned long long *s_bitmap = (unsigned long long *)(smem + B5_BITMAP_OFF);
    volatile int *s_ctl = (volatile int *)(smem + B5_CTL_OFF);
    int *s_ctl_a = (int *)(smem + B5_CTL_OFF);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint2 *S = (uint2 *)(scratch + (size_t)blockIdx.x * B5_SLOT_BYTES);
    uint2 *RES = (uint2 *)(scratch + (size_t)blockIdx.x * B5_SLOT_BYTES + B5_S_BYTES);
    uint32_t *RANK = (uint32_t *)(scratch + (size_t)blockIdx.x * B5_SLOT_BYTES + 2 * B5_S_BYTES);
    const uint32_t cnt_a = (uint32_t)(uintptr_t)(b5_lds_u8 *)smem;
    const uint32_t turn_a = cnt_a + (uint32_t)B5_CTL_OFF + 4u * B5_CTL_TICKET;
    unsigned long long c_steps = 0, c_slots = 0, c_ext = 0, c_upd = 0, bp_cnt[2] = {0, 0};
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, tk0 = 0, wk[6] = {0, 0, 0, 0, 0, 0}, wk0 = 0;   // (lab) clock ticks per phase, thread 0 of each workgroup
#define B5_WTICK(n) do { if (DBG) { const unsigned long long now_ = wall_clock64(); wk[n] += now_ - wk0; wk0 = now_; } } while (0)
#define B5_TICK(n) do { if (DBG && threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); tk[n] += now_ - tk0; tk0 = now_; } } while (0)

    for (;;) {
        __syncthreads();   // (the previous tile's search is over: the control words and the LDS may be reused)
        if (threadIdx.x == 0) s_ctl[B5_CTL_TILE] = (int)atomicAdd(tile_counter, 1u);
        __syncthreads();
        const int ti = s_ctl[B5_CTL_TILE];
        if (ti >= ntiles) break;
        const TileDev tile = tiles[ti];
        const SegDev seg = segs[tile.seg];
        const uint8_t *d = in + seg.buf_off;
        const uint64_t avail = in_total - seg.buf_off;
        const int64_t t0 = tile.start, t1 = tile.start + tile.len;
        const int64_t w0 = t0 - B_HIST > 0 ? t0 - B_HIST : 0;
        const int wlen = (int)(t1 - w0);                  // window positions
        const int hoff = (int)(t0 - w0);                  // window position of the tile's first position
        const int nslices = (wlen + 63) >> 6;
        const uint64_t *b = bnds + seg.bnd_off;
        const int nb = (int)seg.bnd_cnt;

        if (DBG && threadIdx.x == 0) tk0 = wall_clock64();
        // ---- S1: clear the counters, the bitmap and the ticket
        {
            uint4 *c4 = (uint4 *)smem;
            for (int i = threadIdx.x; i < B5_CNT_BYTES / 16; i += B5_THREADS) c4[i] = make_uint4(0, 0, 0, 0);
            for (int i = threadIdx.x; i < B5_BITMAP_WORDS; i += B5_THREADS) s_bitmap[i] = 0ull;
            if (threadIdx.x == 0) { s_ctl[B5_CTL_TICKET] = 0; s_ctl[B5_CTL_SLICE] = 0; }
        }
        __syncthreads();
        // ---- S2: histogram of the inserted positions by hash
        {
            InsRule5 R; R.init(b, nb);
            for (int s = wave; s < nslices; s += B5_WAVES) {
                const int64_t q0 = w0 + 64 * (int64_t)s, q = q0 + lane;
                if (R.ins(q0, q, t1)) {
                    uint32_t w;
                    if ((uint64_t)q + 4 <= avail) __builtin_memcpy(&w, d + q, 4);
                    else w = (uint32_t)d[q] | ((uint32_t)d[q + 1] << 8) | ((uint32_t)d[q + 2] << 16);
                    atomicAdd(&s_cnt[hash_of(w)], 1u);
                }
            }
        }
        __syncthreads();
        B5_TICK(0);
        // ---- S3: exclusive scan (thread t owns buckets t + 1024 j), bucket-start bitmap
        {
            uint32_t c[32], tot = 0;
#pragma unroll
            for (int j = 0; j < 32; j++) { c[j] = s_cnt[threadIdx.x + 1024 * j]; tot += c[j]; }
            uint32_t inc = tot;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += v; }
            __syncthreads();                                    // every counter is in registers now: the first 16 slots carry the wave totals
            uint32_t *s_row = (uint32_t *)smem;                 // (all counters are rewritten from registers below)
            if (lane == 63) s_row[wave] = inc;
            __syncthreads();
            uint32_t wbase = 0;
            for (int w = 0; w < wave; w++) wbase += s_row[w];
            if (threadIdx.x == B5_THREADS - 1) s_ctl[B5_CTL_NS] = (int)(wbase + inc);
            __syncthreads();                                    // (the row has been read: the counters come back)
            uint32_t off = wbase + inc - tot;
#pragma unroll
            for (int j = 0; j < 32; j++) {
                s_cnt[threadIdx.x + 1024 * j] = off;
                if (c[j]) atomicOr(&s_bitmap[off >> 6], 1ull << (off & 63));
                off += c[j];
            }
        }
        __syncthreads();
        const int ns = s_ctl[B5_CTL_NS];
        B5_TICK(1);
        // ---- S4: ordered scatter.  Group t (four slices) belongs to wave t mod 16; the ticket passes from group to group.
        {
            InsRule5 R; R.init(b, nb);
            const int ngr = (nslices + B5_U - 1) / B5_U;
            for (int t = wave; t < ngr; t += B5_WAVES) {
                uint32_t haddr[B5_U], addv[B5_U], old[B5_U], rank[B5_U], lead[B5_U];
                uint64_t m_lead[B5_U], m_ins[B5_U];
                uint2 ent[B5_U];
#pragma unroll
                for (int u = 0; u < B5_U; u++) {
                    const int s = B5_U * t + u;
                    const int64_t q0 = w0 + 64 * (int64_t)s, q = q0 + lane;
                    const bool ins = s < nslices && R.ins(q0, q, t1);
                    uint64_t w = 0;
                    if (ins) w = load8(d, q, avail);
                    const uint32_t h = hash_of(w);
                    ent[u] = entry_of(w, (uint32_t)(q - w0));
                    // the lanes of this slice that share my bucket (ballot multisplit over the 15 hash bits)
                    uint64_t peers = __ballot(ins);
                    m_ins[u] = peers;
#pragma unroll
                    for (int bit = 0; bit < 15; bit++) {
                        const bool mine = (h >> bit) & 1u;
                        const uint64_t bm = __ballot(mine);
                        peers &= mine ? bm : ~bm;
                    }
                    rank[u] = (uint32_t)__builtin_popcountll(peers & lanemask_lt);
                    addv[u] = (uint32_t)__builtin_popcountll(peers);
                    lead[u] = (uint32_t)__builtin_ctzll(peers | (1ull << 63));   // lowest lane of my group
                    m_lead[u] = __ballot(ins && rank[u] == 0);
                    haddr[u] = cnt_a + 4u * h;
                    old[u] = 0;
                }
                {
                    static_assert(B5_U == 4, "the ticket section is written out for four slices");
                    uint32_t c, sc;
                    uint64_t sv;
                    const uint32_t tt = (uint32_t)__builtin_amdgcn_readfirstlane(t), tn = tt + 1u;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n"
                        "1:\n\t"
                        "ds_read_b32 %[c], %[ta]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_readfirstlane_b32 %[sc], %[c]\n\t"
                        "s_cmp_eq_u32 %[sc], %[tt]\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "s_sub_u32 %[sc], %[tt], %[sc]\n\t"
                        "s_cmp_lt_u32 %[sc], 3\n\t"
                        "s_cbranch_scc1 1b\n\t"
                        "s_sleep 3\n\t"
                        "s_branch 1b\n"
                        "2:\n\t"
                        "s_mov_b64 exec, %[ml0]\n\t" "ds_add_rtn_u32 %[e0], %[a0], %[v0]\n\t"
                        "s_mov_b64 exec, %[ml1]\n\t" "ds_add_rtn_u32 %[e1], %[a1], %[v1]\n\t"
                        "s_mov_b64 exec, %[ml2]\n\t" "ds_add_rtn_u32 %[e2], %[a2], %[v2]\n\t"
                        "s_mov_b64 exec, %[ml3]\n\t" "ds_add_rtn_u32 %[e3], %[a3], %[v3]\n\t"
                        "s_mov_b64 exec, %[sv]\n\t"
                        "ds_write_b32 %[ta], %[tn]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        : [e0] "+&v"(old[0]), [e1] "+&v"(old[1]), [e2] "+&v"(old[2]), [e3] "+&v"(old[3]),
                          [c] "=&v"(c), [sc] "=&s"(sc), [sv] "=&s"(sv)
                        : [ta] "v"(turn_a), [tt] "s"(tt), [tn] "v"(tn),
                          [a0] "v"(haddr[0]), [a1] "v"(haddr[1]), [a2] "v"(haddr[2]), [a3] "v"(haddr[3]),
                          [v0] "v"(addv[0]), [v1] "v"(addv[1]), [v2] "v"(addv[2]), [v3] "v"(addv[3]),
                          [ml0] "s"(m_lead[0]), [ml1] "s"(m_lead[1]), [ml2] "s"(m_lead[2]), [ml3] "s"(m_lead[3])
                        : "scc", "memory");
                }
#pragma unroll
                for (int u = 0; u < B5_U; u++) {
                    const uint32_t basev = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lead[u] << 2), (int)old[u]);
                    const bool insd = (m_ins[u] >> lane) & 1;
                    if (insd) S[basev + rank[u]] = ent[u];
                    const int rel = 64 * (B5_U * t + u) + lane;
                    if (rel < wlen) RANK[rel] = insd ? basev + rank[u] : 0xFFFFFFFFu;   // (a position that is not inserted has no entry: empty tables, :780)
                }
            }
        }
        // S is written and read by this workgroup only (one CU, one vector L1): workgroup scope.  (Agent scope writes back and invalidates
        // the XCD's L2 — for all the workgroups that share it — once per tile: measured, it made every S read a miss.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        B5_TICK(2);
        // ---- B1: the window's bytes (+ the lookahead tail) into LDS
        {
            const int64_t look_end = seg.look_end;
            uint32_t *sdata32 = (uint32_t *)smem;
            const int ndw = (wlen + B_TAIL + 8 + 3) >> 2;
            for (int i = threadIdx.x; i < ndw; i += B5_THREADS) {
                const int64_t pos = w0 + 4 * (int64_t)i;
                uint32_t w = 0;
                if (pos + 4 <= look_end && (uint64_t)pos + 4 <= avail) __builtin_memcpy(&w, d + pos, 4);
                else for (int k = 0; k < 4; k++) { const int64_t pk = pos + k; if (pk < look_end && (uint64_t)pk < avail) w |= (uint32_t)d[pk] << (8 * k); }
                sdata32[i] = w;
            }
        }
        __syncthreads();

        B5_TICK(3);
        // ---- B2: the search, in S order
        {
            const uint8_t *sdata8 = smem;
            const uint32_t *sdata32 = (const uint32_t *)smem;
            uint8_t *wv = smem + B5_WV_OFF + wave * B5_WV_BYTES;
            uint2 *stg = (uint2 *)(wv + B5_WV_STG);
            uint32_t *slot2 = (uint32_t *)(wv + B5_WV_SLOT2), *slotq = (uint32_t *)(wv + B5_WV_SLOTQ);
            const int64_t A = (int64_t)seg.abs0 + w0;          // absolute stream position of window position 0
            const int64_t rem0 = seg.look_end - w0;             // lookahead at window position 0
            const int SNAP = P.max_chain >> 2;
            const int kmax = P.max_chain;
            auto ldsdw = [&](int i) -> uint32_t {               // four bytes at byte index i from two aligned dwords
                const uint32_t a0 = sdata32[i >> 2], a1 = sdata32[(i >> 2) + 1];
                return __builtin_amdgcn_alignbyte(a1, a0, (uint32_t)(i & 3));
            };
            auto wave_sync = [&]() {                            // LDS written by one lane of the wave, read by another
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            };
            // candidates in front of entry (128 + lane) of a 192-entry range whose first entry is S[lo_idx] (a multiple of 64, may be
            // negative): distance to the highest bucket start at or below it; INT_MAX if there is none in the range
            auto avail_in = [&](int lo_idx) -> int {
                const int wi = lo_idx >> 6;
                const unsigned long long w0b = wi >= 0 ? s_bitmap[wi] : 0ull, w1b = wi + 1 >= 0 ? s_bitmap[wi + 1] : 0ull, w2b = s_bitmap[wi + 2];
                const unsigned long long m2b = w2b & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
                int j;
                if (m2b) j = 128 + 63 - __builtin_clzll(m2b);
                else if (w1b) j = 64 + 63 - __builtin_clzll(w1b);
                else if (w0b) j = 63 - __builtin_clzll(w0b);
                else return INT_MAX;
                return (128 + lane) - j;
            };
            // the entries of a slice (own + the 128 in front) are loaded one slice ahead: a global load is ~2 us away
            auto grab = [&]() -> int {
                int sl = 0;
                if (lane == 0) sl = atomicAdd(&s_ctl_a[B5_CTL_SLICE], 64);
                return __builtin_amdgcn_readfirstlane(sl);
            };
            auto ldS = [&](int j) -> uint2 { return (j >= 0 && j < ns) ? S[j] : make_uint2(0, 0); };
            int nslice = grab();
            uint2 nOwn = ldS(nslice + lane), nP0 = ldS(nslice - 128 + lane), nP1 = ldS(nslice - 64 + lane);
            for (;;) {
                if (DBG) wk0 = wall_clock64();
                const int slice = nslice;
                if (slice >= ns) break;
                const int i = slice + lane;
                const uint2 own = nOwn, pre0 = nP0, pre1 = nP1;
                nslice = grab();
                nOwn = ldS(nslice + lane); nP0 = ldS(nslice - 128 + lane); nP1 = ldS(nslice - 64 + lane);
                const int prel = (int)(own.y >> 15);
                const bool mine = i < ns && prel >= hoff;        // this lane's entry is a position of the tile
                if (!__any(mine) || (lab & 16)) continue;
                stg[lane] = pre0; stg[64 + lane] = pre1; stg[128 + lane] = own;   // stage S[slice-128 .. slice+63]
                wave_sync();
                int nav = avail_in(slice - 128);                 // candidates of my bucket in front of me (INT_MAX: 128 or more)
                const int rem = (int)(rem0 - prel > (int64_t)(1 << 24) ? (int64_t)(1 << 24) : rem0 - prel);
                const int cap = rem < MAX_MATCH ? rem : MAX_MATCH;                              // :479
                const int nice = cap < P.nice ? cap : P.nice;                                   // :485 (min(niceLength, lookahead))
                const int64_t basem64 = base_of5(A + prel) - A;
                const int basem = basem64 < 0 ? 0 : (int)basem64;
                const int firstmin = prel - MAX_DIST > basem ? prel - MAX_DIST : basem;        // strstart - hashHead <= MAX_DIST (:788)
                const int mincl = prel - (MAX_DIST - 1) > basem ? prel - (MAX_DIST - 1) : basem; // curMatch > limit (:609)
                const uint32_t mincs = (uint32_t)mincl << 15;
                const uint32_t mylo = own.x, myhi = (own.y & 0x3FFFu) | 0x4000u;   // (bit 14 of an entry is 0: the xor below is never 0)
                bool act = mine && rem >= MIN_MATCH && nav > 0;
                const bool slow = act && cap < 7;                // the last positions of a stream: the lookahead caps every length (generic walk below)
                if (slow) act = false;
                int bestL = 2;                                   // running best_len over the chunks walked so far
                uint32_t res2 = 0, resq = 0;
                bool q_open = true;                              // the quarter-budget snapshot (:495) is still to be taken
                B5_WTICK(0);

                for (int chunk = 0;; chunk++) {
                    // ---- how many candidates does this chunk hold for me: bucket start, chain budget, distance
                    const int kbase = 128 * chunk;
                    int nact = 0;
                    if (act) {
                        int lim = nav < kmax ? nav : kmax;       // (nav == INT_MAX: no bucket start seen yet)
                        lim -= kbase;
                        nact = lim > 128 ? 128 : lim;
                    }
                    int count = 0;
                    {   // the staged positions descend with kk: how many of my nact candidates are at or above mincl?  Three rounds of
                        // independent probes (16-, 4-, 1-spaced) instead of a binary search: an LDS round trip is ~400 cycles under load
                        const uint2 *top = stg + lane + 127;
                        int c1 = 0;
#pragma unroll
                        for (int j = 0; j < 8; j++) { const int kk = 16 * j + 15; c1 += (kk < nact && top[-kk].y >= mincs) ? 1 : 0; }
                        const int b1 = 16 * c1;                  // every candidate below b1 is in range; b1 + 15 is not (or does not exist)
                        int c2 = 0;
#pragma unroll
                        for (int j = 0; j < 3; j++) { const int kk = b1 + 4 * j + 3; c2 += (kk < nact && top[-(kk < 127 ? kk : 127)].y >= mincs) ? 1 : 0; }
                        const int b2s = b1 + 4 * c2;
                        int c3 = 0;
#pragma unroll
                        for (int j = 0; j < 3; j++) { const int kk = b2s + j; c3 += (kk < nact && top[-(kk < 127 ? kk : 127)].y >= mincs) ? 1 : 0; }
                        count = b2s + c3;
                        if (count > nact) count = nact;
                        if (lab & 8) count = nact;
                        if (chunk == 0 && count == 0 && nact > 0 && (int)(top[0].y >> 15) >= firstmin) count = 1;   // a first candidate at exactly MAX_DIST (:788 vs :609)
                    }
                    const bool ends_here = count < 128 || kbase + 128 >= kmax;   // my walk ends inside this chunk
                    int kw = wave_max_i32(count);
                    if (lab & 2) kw = kw < 2 ? kw : 2;          // (lab: timing without the candidate loop; results are wrong)
                    if (kw == 0) break;
                    if (DBG) { c_slots += (unsigned long long)kw; c_steps += (unsigned long long)count; }
                    const int snapk = SNAP - kbase;              // the quarter-budget snapshot falls in front of this chunk's candidate `snapk`
                    // result slots of this chunk: key = length (capped at nice) << 22 | (127 - kk) << 15 | distance; a slot starts at the
                    // running best_len, so that "longer than everything before" is one comparison
                    slot2[lane] = (uint32_t)bestL << 22;
                    slotq[lane] = (uint32_t)bestL << 22;
                    B5_WTICK(1);
                    // ---- pass 1, lockstep: the first differing content bit t of every candidate; running maximum of (length class, nearest
                    // first); a candidate that agrees on all 46 content bits (length >= 7) is queued for the byte pass by its lane, in step
                    // order; 64 queued candidates are one byte pass
                    uint32_t best = 0, bestq = 0;
                    int qlen = 0;
                    uint32_t *queue = (uint32_t *)(wv + B5_WV_QUEUE), *pinfo = (uint32_t *)(wv + B5_WV_PINFO);
                    pinfo[lane] = (uint32_t)prel | ((uint32_t)cap << 17);
                    const uint2 *top = stg + lane + 127;
                    uint2 cur[8], nxt[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) cur[j] = top[-j];
                    for (int k0 = 0; k0 < kw; k0 += 8) {
                        {   // the next eight entries are in flight while these eight are looked at
                            const int kn = k0 + 8 < 120 ? k0 + 8 : 120;
#pragma unroll
                            for (int j = 0; j < 8; j++) nxt[j] = top[-(kn + j)];
                        }
#pragma unroll
                        for (int h = 0; h < 4; h++) {
                            uint32_t tt[2];
#pragma unroll
                            for (int j = 0; j < 2; j++) {
                                const int kk = k0 + 2 * h + j;
                                if (kk == snapk) bestq = best;
                                const uint32_t xl = cur[2 * h + j].x ^ mylo, xh = cur[2 * h + j].y ^ myhi;
                                const uint32_t tl = ffbl_m1(xl), th = ffbl_m1(xh) + 32u;
                                uint32_t t = tl < th ? tl : th;
                                t = kk < count ? t : 0u;
                                tt[j] = t;
                                const uint32_t K = (((t + 7u) & ~7u) << 4) | (uint32_t)(127 - kk);
                                best = K > best ? K : best;
                            }
#pragma unroll
                            for (int j = 0; j < 2; j++) {
                                const bool f = tt[j] == 46u;
                                const uint64_t bal = __ballot(f);
                                if (bal) {
                                    if (f) queue[qlen + __builtin_popcountll(bal & lanemask_lt)] = ((cur[2 * h + j].y >> 15) << 13) | ((uint32_t)lane << 7) | (uint32_t)(k0 + 2 * h + j);
                                    qlen += __builtin_popcountll(bal);
                                }
                            }
                            while (qlen >= 64) {
                                wave_sync();
                                if (!(lab & 1)) b5_byte_pass(smem, queue, slot2, slotq, pinfo, 64, P.nice, snapk, DBG ? bp_cnt : nullptr);
                                const int r = qlen - 64;                              // (< 128: two steps were appended)
                                const uint32_t mv0 = queue[64 + lane], mv1 = queue[128 + lane < B5_QCAP ? 128 + lane : 0];
                                wave_sync();
                                if (lane < r) queue[lane] = mv0;
                                if (64 + lane < r) queue[64 + lane] = mv1;
                                qlen = r;
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 8; j++) cur[j] = nxt[j];
                    }
                    if (snapk >= kw) bestq = best;               // (the snapshot lies behind this chunk's last candidate: all of them count)
                    B5_WTICK(2);
                    wave_sync();
                    if (qlen > 0 && !(lab & 1)) b5_byte_pass(smem, queue, slot2, slotq, pinfo, qlen, P.nice, snapk, DBG ? bp_cnt : nullptr);
                    wave_sync();
                    B5_WTICK(3);
                    // ---- this chunk's winner: the longer of the two passes' bests, the nearer one among equals
                    auto settle = [&](uint32_t bK, uint32_t key, int &Lout, uint32_t &rout) {
                        int L1 = (int)(bK >> 7) + 1, k1 = 127 - (int)(bK & 127u);
                        if (L1 < 3) L1 = 0;
                        int Lb = (int)(key >> 22), kb = 127 - (int)((key >> 15) & 127u), db = (int)(key & 0x7FFFu);
                        if ((key & 0x3FFFFFu) == 0u) Lb = 0;     // (the slot still holds its starting value: nothing longer was found)
                        if (Lb >= nice && Lb < cap) {            // the slot's length is capped at niceLength: compare on from there
                            const int crel = prel - db;
                            int L = Lb;
                            while (L < cap) {
                                const uint32_t x = ldsdw(crel + L) ^ ldsdw(prel + L);
                                if (x) { L += (int)(__builtin_ctz(x) >> 3); break; }
                                L += 4;
                            }
                            Lb = L > cap ? cap : L;
                        }
                        int Lw = L1, dw = 0;
                        const bool second = Lb > L1 || (Lb == L1 && Lb > 0 && kb < k1);
                        if (second) { Lw = Lb; dw = db; }
                        else if (L1 > 0) dw = prel - (int)(stg[lane + 127 - k1].y >> 15);
                        if (Lw > Lout) { Lout = Lw; rout = (uint32_t)Lw | ((uint32_t)dw << 16); }
                    };
                    if (act) {
                        const uint32_t key2 = slot2[lane], keyq = slotq[lane];
                        if (q_open && snapk > 0) {               // the quarter-budget walk still sees candidates of this chunk
                            int Lq = bestL; uint32_t rq = res2;  // (its state in front of this chunk is the full walk's)
                            settle(bestq, keyq, Lq, rq);
                            resq = rq;
                            if (snapk <= 128) q_open = false;    // it ends inside (or with) this chunk
                        } else if (q_open) { resq = res2; q_open = false; }
                        settle(best, key2, bestL, res2);
                        if (bestL >= nice || ends_here) act = false;                 // :603 / end of the chain
                    }
                    if (!__any(act)) break;
                    // ---- the next 128 older entries: keep S[base .. base+63] as the top third, load two new thirds below
                    const uint2 keep = stg[lane];
                    const int nb0 = slice - 128 * (chunk + 2);
                    const int j0 = nb0 + lane, j1 = nb0 + 64 + lane;
                    const uint2 n0 = j0 >= 0 ? S[j0] : make_uint2(0, 0), n1 = j1 >= 0 ? S[j1] : make_uint2(0, 0);
                    wave_sync();
                    stg[128 + lane] = keep; stg[lane] = n0; stg[64 + lane] = n1;
                    wave_sync();
                    if (nav == INT_MAX) {                        // no bucket start seen yet: look at the bits of the new range
                        const int a2 = avail_in(nb0);
                        if (a2 != INT_MAX) nav = a2 + 128 * (chunk + 1);
                    }
                }
                if (q_open) resq = res2;
                if (SNAP == 0) resq = 0;                         // (flm: no snapshot is taken with a budget below 4)
                if (__any(slow)) {
                    // ---- generic walk for the few positions whose lookahead is shorter than the content bytes of an entry (S read directly)
                    if (slow) {
                        int best = 2, k = 0;
                        res2 = 0; resq = 0;
                        bool qo = true;
                        for (int idx = i - 1; idx >= 0; idx--, k++) {
                            if ((s_bitmap[(idx + 1) >> 6] >> ((idx + 1) & 63)) & 1ull) break;          // S[idx + 1] starts a bucket: S[idx] is not in mine
                            const int crel = (int)(S[idx].y >> 15);
                            if (k >= kmax || crel < (k == 0 ? firstmin : mincl)) break;
                            int L = 0;
                            while (L < cap && sdata8[crel + L] == sdata8[prel + L]) L++;
                            if (L > best) {
                                best = L; res2 = (uint32_t)L | ((uint32_t)(prel - crel) << 16);
                                if (L >= nice) { if (k < SNAP) resq = res2; qo = false; break; }
                            }
                            if (k + 1 == SNAP && qo) { resq = res2; qo = false; }
                        }
                        if (qo) resq = res2;
                        if (SNAP == 0) resq = 0;
                    }
                }
                if (mine && !(lab & 4)) RES[i] = make_uint2(res2, resq);     // coalesced; the tile's tables are written in position order below
                B5_WTICK(4);
            }
        }
        __syncthreads();
        B5_TICK(4);
        // ---- B3: the tile's tables in position order.  In S order a wavefront's 64 results belong to 64 positions all over the tile:
        // written straight to the tables they are 4-byte stores scattered over 512 KiB per workgroup — more open cache lines than the
        // L2 can merge, 25 ms per GiB (profiles/r03/lab_m5_knockout.log).  RES is written coalesced instead and gathered here through
        // RANK (a workgroup's 0.8 MiB of RES is read back while it is still in the L2).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (!(lab & 4)) {
            uint32_t *m2o = mtab.m2 + seg.buf_off + w0, *mqo = mtab.mq + seg.buf_off + w0;
            for (int rel = hoff + (int)threadIdx.x; rel < wlen; rel += B5_THREADS) {
                const uint32_t r = RANK[rel];
                uint2 v = make_uint2(0, 0);
                if (r != 0xFFFFFFFFu) v = RES[r];
                const uint32_t e = mt_pack(v.x, v.y);
                m2o[rel] = e;
                if (e >> 25) mqo[rel] = v.y;
            }
        }
    }
    if (DBG && dbg && threadIdx.x == 0) for (int i = 0; i < 5; i++) atomicAdd(dbg + 9 + i, tk[i]);
    if (DBG && dbg && lane == 0) for (int i = 0; i < 5; i++) atomicAdd(dbg + 17 + i, wk[i]);
    if (DBG) { c_ext += bp_cnt[0]; c_upd += bp_cnt[1]; }
    if (DBG && dbg) { for (int o = 32; o > 0; o >>= 1) { c_steps += __shfl_xor(c_steps, o); c_ext += __shfl_xor(c_ext, o); c_upd += __shfl_xor(c_upd, o); } }
    if (DBG && dbg && lane == 0) { atomicAdd(dbg + 30, c_steps); atomicAdd(dbg + 31, c_slots); atomicAdd(dbg + 29, c_ext); atomicAdd(dbg + 28, c_upd); }
}

size_t match5_scratch_bytes(int nslots) { return (size_t)nslots * B5_SLOT_BYTES + 256; }
int match5_tile() { return B5_TMAX; }

hipError_t launch_match5(const uint8_t *in, uint64_t in_total, const SegDev *segs, const uint64_t *bnds, const TileDev *tiles, int ntiles,
                         MTab mtab, LevelParams P, uint8_t *scratch, int nslots, unsigned long long *dbg, hipStream_t st) {
    static std::atomic<uint64_t> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if ((attr_mask.load(std::memory_order_acquire) & bit) == 0) {
        hipError_t e = hipFuncSetAttribute((const void *)k_match5<false>, hipFuncAttributeMaxDynamicSharedMemorySize, B5_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_match5<true>, hipFuncAttributeMaxDynamicSharedMemorySize, B5_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(bit, std::memory_order_release);
    }
    if (ntiles <= 0) return hipSuccess;
    unsigned int *counter = (unsigned int *)(scratch + (size_t)nslots * B5_SLOT_BYTES);   // (the last 256 bytes of the scratch area)
    hipError_t e = hipMemsetAsync(counter, 0, 4, st);
    if (e != hipSuccess) return e;
    const int grid = ntiles < nslots ? ntiles : nslots;
    if (knob("SZL_DEBUG", 0)) hipLaunchKernelGGL((k_match5<true>), dim3(grid), dim3(B5_THREADS), B5_LDS_BYTES, st, in, in_total, segs, bnds, tiles, ntiles, mtab, P, scratch, counter, dbg, SZL_LABKNOB("SZL_B5_LAB", 0));
    else hipLaunchKernelGGL((k_match5<false>), dim3(grid), dim3(B5_THREADS), B5_LDS_BYTES, st, in, in_total, segs, bnds, tiles, ntiles, mtab, P, scratch, counter, dbg, SZL_LABKNOB("SZL_B5_LAB", 0));
    return hipGetLastError();
}

} // namespace szl
