// szl_kernels_match.hip — stage A (hash links) and stage B (match tables) for gfx950.
//
// Reference being restated (C/ = /root/reference/src/ICSharpCode.SharpZipLib/Zip/Compression/):
//   stage A  InsertString / UpdateHash / SlideWindow      C/DeflaterEngine.cs:402-462
//   stage B  FindLongestMatch                              C/DeflaterEngine.cs:474-612
// Both are *parse-independent* at levels 5-9 (DeflateSlow inserts every position),
// so they are computed for every position in parallel; the lazy parse (stage C) only looks results up.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <cstdint>
#include "szl_internal.h"

namespace szl {
int knob(const char *name, int dflt);

// hipFuncSetAttribute is per device: remember which devices have the large-LDS attribute for a kernel group
static bool lds_attr_needed(std::atomic<uint64_t> &mask, uint64_t &bit) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bit = 1ull << (dev & 63);
    return (mask.load(std::memory_order_acquire) & bit) == 0;
}

__device__ __forceinline__ int64_t base_of(int64_t s_abs) {
    // Window base in effect for an iteration starting at absolute position s: the engine slides by 32768
    // whenever an iteration starts at window index >= 65274 (C/DeflaterEngine.cs:371,:771); index = s+1-base (:93).
    int64_t idx = s_abs + 1;
    if (idx <= 65273) return 0;
    return ((idx - 65273 + 32767) >> 15) << 15;
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// ============================================================================================
// Stage A: k_links.  One workgroup (16 wavefronts) walks a span of a stream front to back.
// The 32768-entry head table (C/DeflaterEngine.cs:87 `head`) lives in LDS as u16, partitioned by
// hash between the 16 wavefronts: wavefront w owns the 2048 buckets whose mixed hash nibble == w and
// handles ONLY positions that hash into them, so table updates need no inter-wave ordering; the span is
// hashed once per 1024-position chunk (each wavefront hashes one 64-position slice into LDS, one barrier).  Inside a
// 64-position batch, "previous lane with the same bucket" is found with a ballot loop over the
// distinct buckets present (≈4 per batch per wave).  Stale entries are aged out every 16384
// positions exactly like SlideWindow's clamp (:450-461), which keeps 16-bit entries unambiguous.
// ============================================================================================
enum : int { A_THREADS = 1024, A_WAVES = 16 };

#if SZL_LAB   // the first form (lab library only)
__global__ __launch_bounds__(A_THREADS) void k_links(const uint8_t *__restrict__ in, uint64_t in_total,
                                                     const SegDev *__restrict__ segs, const uint64_t *__restrict__ bnds,
                                                     const SpanDev *__restrict__ spans, uint16_t *__restrict__ link) {
    __shared__ uint16_t head[32768];
    __shared__ uint16_t hidx[2][A_THREADS]; // bucket index of each position of the current 1024-position chunk (0xFFFF: not inserted)
    const SpanDev span = spans[blockIdx.x];
    const SegDev seg = segs[span.seg];
    const uint8_t *d = in + seg.buf_off;
    uint16_t *lk = link + seg.buf_off;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t warm0 = span.start - WSIZE > 0 ? span.start - WSIZE : 0;
    const uint64_t avail = in_total - seg.buf_off; // bytes readable from d

    uint16_t *myhead = head + wave * 2048;
    for (int i = lane; i < 2048; i += 64) myhead[i] = (uint16_t)((warm0 - 40000) & 0xFFFF);

    const uint64_t *b = bnds + seg.bnd_off;
    const int nb = (int)seg.bnd_cnt;
    int bi = 0;
    const uint64_t lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    int buf = 0;
    for (int64_t c0 = warm0; c0 < span.end; c0 += A_THREADS, buf ^= 1) {
        if (c0 != warm0 && ((c0 - warm0) & 16383) == 0) {
            for (int i = lane; i < 2048; i += 64) {
                uint32_t dist = (uint32_t)(c0 - myhead[i]) & 0xFFFF;
                if (dist >= 32768u || dist == 0u) myhead[i] = (uint16_t)((c0 - 40000) & 0xFFFF);
            }
        }
        // ---- phase 1: every wavefront hashes ITS 64-position slice of the chunk once (instead of all 16 hashing everything)
        {
            const int64_t q = c0 + threadIdx.x;
            while (bi < nb && (int64_t)b[bi] <= c0) bi++; // uniform
            bool ins = q < span.end;
            if (bi >= nb) ins = false;
            else if ((int64_t)b[bi] < c0 + A_THREADS + 2) { // a segment end is near: InsertString needs lookahead >= 3 (:780,:817)
                int j = bi;
                while (j < nb && (int64_t)b[j] <= q) j++;
                ins = ins && j < nb && (int64_t)b[j] - q >= 3;
            }
            uint32_t idx = 0xFFFF;
            if (ins) {
                uint32_t w;
                if ((uint64_t)q + 4 <= avail) w = load_u32_unaligned(d + q);
                else w = (uint32_t)d[q] | ((uint32_t)d[q + 1] << 8) | ((uint32_t)d[q + 2] << 16);
                const uint32_t b0 = w & 0xFF, b1 = (w >> 8) & 0xFF, b2 = (w >> 16) & 0xFF;
                const uint32_t h = ((b0 << 10) ^ (b1 << 5) ^ b2) & 0x7FFF; // :404,:420
                const uint32_t o = (h ^ (h >> 5) ^ (h >> 10)) & 15;         // owner wavefront (bijective with h>>4)
                idx = (o << 11) | (h >> 4);
            }
            hidx[buf][threadIdx.x] = (uint16_t)idx;
        }
        __syncthreads(); // one barrier per chunk; double buffering makes the second one unnecessary
        // ---- phase 2: each wavefront walks the 16 slices in order and handles the positions whose bucket it owns
#pragma unroll 4
        for (int k = 0; k < A_WAVES; k++) {
            const int64_t q0 = c0 + 64 * k;
            if (q0 >= span.end) break;
            const int64_t q = q0 + lane;
            const uint32_t idx = hidx[buf][64 * k + lane];
            const bool owned = (idx >> 11) == (uint32_t)wave; // 0xFFFF>>11 == 31: never a wave id
            // Fast path: most batches hold no two owned positions with the same bucket.  All owned lanes store their
            // position into the bucket and read it back; a lane that reads something else lost to a same-bucket lane.
            // Only then is the exact "previous lane with my bucket" computed, with a ballot loop over the buckets in conflict.
            const uint32_t myq16 = (uint32_t)q & 0xFFFF;
            uint32_t e_old = 0, rb = myq16;
            if (owned) { // volatile: the read-back must really hit LDS (another lane of this wave may have overwritten it)
                volatile uint16_t *vh = head;
                e_old = vh[idx];
                vh[idx] = (uint16_t)myq16;
                rb = vh[idx];
            }
            int predlane = -1;
            bool islast = true;
            uint64_t mm = __ballot(owned && rb != myq16);
            while (mm) {
                int l = __builtin_ctzll(mm);
                uint32_t kk = __builtin_amdgcn_readlane(idx, l);
                bool mine = owned && idx == kk;
                uint64_t same = __ballot(mine);
                if (mine) {
                    uint64_t below = same & lanemask_lt;
                    predlane = below ? 63 - __builtin_clzll(below) : -1;
                    islast = ((same >> lane) >> 1) == 0;
                }
                mm &= ~same;
            }
            if (owned) {
                uint32_t dist = predlane >= 0 ? (uint32_t)(lane - predlane) : ((uint32_t)(q - e_old) & 0xFFFF);
                if (dist > 32767u) dist = 0; // candidates farther than the window are never followed (:609)
                if (q >= span.start) lk[q] = (uint16_t)dist;
                if (!islast || predlane >= 0 || rb != myq16) { if (islast) head[idx] = (uint16_t)myq16; }
            } else if (idx == 0xFFFF && wave == 0 && q >= span.start && q < span.end) {
                lk[q] = 0; // position never inserted (tail of a segment)
            }
        }
    }
}

#endif

// ============================================================================================
// Stage A, compacted form (default).  Same partition of the head table, but instead of every wavefront scanning
// all 1024 positions of a chunk for the ~64 it owns (6 % lane use), the chunk is first bucketed by owner:
//   phase 1  wave w hashes its 64 positions; 16 ballots give, per owner, this wave's count and every lane's rank;
//   (barrier) every wave scans the 16x16 count table privately (owner-major) -> list offsets; lanes scatter
//            (position, bucket) into one 1024-entry list, grouped by owner, in position order;
//   (barrier) phase 2  owner wave o walks ITS group (≈64 entries, dense lanes) with the same store/read-back
//            protocol as above.
// Lists and counts are double-buffered, so a chunk costs two barriers and ≈3.5x fewer wave instructions.
// ============================================================================================
__global__ __launch_bounds__(A_THREADS) void k_links2(const uint8_t *__restrict__ in, uint64_t in_total,
                                                      const SegDev *__restrict__ segs, const uint64_t *__restrict__ bnds,
                                                      const SpanDev *__restrict__ spans, uint16_t *__restrict__ link,
                                                      const uint32_t *__restrict__ hflags) {
    __shared__ uint16_t head[32768];
    __shared__ uint32_t cnt[2][A_WAVES * 16];  // [owner*16 + wave] = positions of that wave's slice owned by `owner`
    __shared__ uint32_t list[2][A_THREADS];    // (position in chunk) << 16 | bucket (11 bits), grouped by owner
    const SpanDev span = spans[blockIdx.x];
    const SegDev seg = segs[span.seg];
    const uint8_t *d = in + seg.buf_off;
    uint16_t *lk = link + seg.buf_off;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t warm0 = span.start - WSIZE > 0 ? span.start - WSIZE : 0;
    const uint64_t avail = in_total - seg.buf_off;

    uint16_t *myhead = head + wave * 2048;
    for (int i = lane; i < 2048; i += 64) myhead[i] = (uint16_t)((warm0 - 40000) & 0xFFFF);

    const uint64_t *b = bnds + seg.bnd_off;
    const int nb = (int)seg.bnd_cnt;
    int bi = 0;
    const uint64_t lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // The chunk's bytes are fetched one chunk ahead (a span is walked strictly in order by one workgroup, so the
    // HBM latency of a just-in-time load would be paid 2048 times per 2 MiB span), and the next segment boundary is
    // kept in a register.
    auto fetch = [&](int64_t qn) -> uint32_t { return (qn < span.end && (uint64_t)qn + 4 <= avail) ? load_u32_unaligned(d + qn) : 0u; };
    uint32_t wnext = fetch(warm0 + threadIdx.x);
    int64_t bcur = bi < nb ? (int64_t)b[bi] : INT64_MAX;
    int buf = 0;
    for (int64_t c0 = warm0; c0 < span.end; c0 += A_THREADS, buf ^= 1) {
        const uint32_t wcur = wnext;
        wnext = fetch(c0 + A_THREADS + threadIdx.x);
        if (c0 != warm0 && ((c0 - warm0) & 16383) == 0) {
            for (int i = lane; i < 2048; i += 64) {
                uint32_t dist = (uint32_t)(c0 - myhead[i]) & 0xFFFF;
                if (dist >= 32768u || dist == 0u) myhead[i] = (uint16_t)((c0 - 40000) & 0xFFFF);
            }
        }
        // ---- phase 1: hash my position, rank it among this wave's positions of the same owner
        const int64_t q = c0 + threadIdx.x;
        while (bcur <= c0) { bi++; bcur = bi < nb ? (int64_t)b[bi] : INT64_MAX; } // uniform
        bool ins = q < span.end;
        if (bi >= nb) ins = false;
        else if (bcur < c0 + A_THREADS + 2) { // a segment end is near: InsertString needs lookahead >= 3 (:780,:817)
            int j = bi;
            while (j < nb && (int64_t)b[j] <= q) j++;
            ins = ins && j < nb && (int64_t)b[j] - q >= 3;
        }
        if (hflags && ins && q < seg.seg_start) ins = (hflags[q >> 5] >> (q & 31)) & 1u; // history a DeflateFast level left: only inserted positions
        uint32_t owner = 16, bucket = 0;
        if (ins) {
            uint32_t w = wcur;
            if ((uint64_t)q + 4 > avail) w = (uint32_t)d[q] | ((uint32_t)d[q + 1] << 8) | ((uint32_t)d[q + 2] << 16);
            const uint32_t b0 = w & 0xFF, b1 = (w >> 8) & 0xFF, b2 = (w >> 16) & 0xFF;
            const uint32_t h = ((b0 << 10) ^ (b1 << 5) ^ b2) & 0x7FFF; // :404,:420
            owner = (h ^ (h >> 5) ^ (h >> 10)) & 15;                   // owner wavefront (bijective with h>>4)
            bucket = h >> 4;
        } else if (q >= span.start && q < span.end) {
            lk[q] = 0; // position never inserted (tail of a segment)
        }
        uint32_t rank = 0, mycnt = 0;
#pragma unroll
        for (int o = 0; o < 16; o++) {
            const uint64_t m = __ballot(owner == (uint32_t)o);
            if (owner == (uint32_t)o) rank = (uint32_t)__builtin_popcountll(m & lanemask_lt);
            if (lane == o) mycnt = (uint32_t)__builtin_popcountll(m);
        }
        if (lane < 16) cnt[buf][lane * 16 + wave] = mycnt;
        __syncthreads();
        // ---- private exclusive scan of the 256 counts in owner-major order: lane l holds elements 4l..4l+3 (all of owner l>>2)
        uint32_t c4[4], e4[4];
        {
            const uint4 v = *(const uint4 *)&cnt[buf][4 * lane];
            c4[0] = v.x; c4[1] = v.y; c4[2] = v.z; c4[3] = v.w;
            uint32_t sum = c4[0] + c4[1] + c4[2] + c4[3], incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            uint32_t ex = incl - sum;
            e4[0] = ex; e4[1] = ex + c4[0]; e4[2] = e4[1] + c4[1]; e4[3] = e4[2] + c4[2];
        }
        { // my slot: element owner*16 + wave, held by lane (owner*4 + wave/4), component wave&3 (shuffles stay convergent)
            const int src = owner < 16 ? (int)owner * 4 + (wave >> 2) : 0;
            const uint32_t a0 = __shfl(e4[0], src), a1 = __shfl(e4[1], src), a2 = __shfl(e4[2], src), a3 = __shfl(e4[3], src);
            const int comp = wave & 3;
            const uint32_t base = comp == 0 ? a0 : (comp == 1 ? a1 : (comp == 2 ? a2 : a3));
            if (owner < 16) list[buf][base + rank] = ((uint32_t)threadIdx.x << 16) | bucket;
        }
        // my group as owner wave: elements [wave*16, wave*16+16) = lanes 4*wave .. 4*wave+3
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)e4[0], 4 * wave);
        const uint32_t g1 = wave == 15 ? (uint32_t)(__builtin_amdgcn_readlane((int)e4[3], 63) + __builtin_amdgcn_readlane((int)c4[3], 63))
                                       : (uint32_t)__builtin_amdgcn_readlane((int)e4[0], 4 * (wave + 1) & 63);
        __syncthreads();
        // ---- phase 2: the positions whose bucket this wavefront owns, in position order, 64 at a time
        for (uint32_t k0 = g0; k0 < g1; k0 += 64) {
            const bool owned = k0 + lane < g1;
            const uint32_t e = owned ? list[buf][k0 + lane] : 0u;
            const int64_t qq = c0 + (int64_t)(e >> 16);
            const uint32_t idx = ((uint32_t)wave << 11) | (e & 0x7FF);
            const uint32_t myq16 = (uint32_t)qq & 0xFFFF;
            uint32_t e_old = 0, rb = myq16;
            if (owned) { // volatile: the read-back must really hit LDS (another lane of this wave may have overwritten it)
                volatile uint16_t *vh = head;
                e_old = vh[idx];
                vh[idx] = (uint16_t)myq16;
                rb = vh[idx];
            }
            int predlane = -1;
            bool islast = true;
            uint64_t mm = __ballot(owned && rb != myq16);
            while (mm) {
                int l = __builtin_ctzll(mm);
                uint32_t kk = __builtin_amdgcn_readlane(idx, l);
                bool mine = owned && idx == kk;
                uint64_t same = __ballot(mine);
                if (mine) {
                    uint64_t below = same & lanemask_lt;
                    predlane = below ? 63 - __builtin_clzll(below) : -1;
                    islast = ((same >> lane) >> 1) == 0;
                }
                mm &= ~same;
            }
            if (owned) {
                uint32_t dist;
                if (predlane >= 0) { // previous owned lane with my bucket: its position comes from its list entry
                    const uint32_t pe = list[buf][k0 + predlane];
                    dist = (uint32_t)((e >> 16) - (pe >> 16));
                } else dist = (uint32_t)(qq - e_old) & 0xFFFF;
                if (dist > 32767u) dist = 0; // candidates farther than the window are never followed (:609)
                if (qq >= span.start) lk[qq] = (uint16_t)dist;
                if (!islast || predlane >= 0 || rb != myq16) { if (islast) head[idx] = (uint16_t)myq16; }
            }
        }
    }
}

// ============================================================================================
// Stage A, pipelined form (default).  k_links2 spends ~12 wave instructions per position on bucketing a chunk by owner
// wavefront (16 ballots per slice, a 256-entry scan, a scatter, two workgroup barriers per 1024 positions) so that table
// updates need no ordering between wavefronts.  Here the order is kept instead: the head table is ONE 32768-entry array
// of 32-bit positions in LDS (no ageing: a span is far shorter than 2^32), the span is cut into 64-position slices, four
// slices form a group, group t belongs to wavefront t mod 16, and an LDS ticket passes from group to group.  Loading and
// hashing the slices and computing / storing the links happen outside the ticket; the holder issues one
// ds_wrxchg_rtn_b32 per slice — "old = head[h]; head[h] = position" for 64 positions at once — and passes the ticket on
// (LDS executes a wavefront's instructions in order, so the next holder sees the updates).
// Positions of one slice that share a bucket need no special care IF the LDS unit serves the lanes of one exchange
// instruction in ascending lane order: each lane then receives the position of the previous lane with its bucket, and the
// highest lane's position stays in the table — InsertString (C/DeflaterEngine.cs:404-424) executed 64 times in order.
// That order is a property of the hardware, not of the ISA manual, so k_probe_xchg_order checks it once per device
// (all-equal, paired, random and strided bucket patterns); launch_links falls back to k_links2 if the probe ever fails.
// Same results as k_links/k_links2: link = distance to the previous position with the same hash, 0 when there is none
// within 32767 or the position is not inserted (:780,:817).
// ============================================================================================
enum : int { A3_HEAD_BYTES = 32768 * 4, A3_LDS_BYTES = A3_HEAD_BYTES + 16 };
enum : int { A3_U = 4 };   // 64-position slices per ticket: a ticket covers 256 positions

typedef __attribute__((address_space(3))) uint8_t a3_lds_u8;

// `rounds` patterns: lane l exchanges (l + 1) into slot[pat(l)]; ok unless some lane got something else than "the lane below with my
// slot" or a slot does not end up holding its highest lane.  The probe runs the way k_links3 does: SIXTEEN wavefronts of one
// workgroup exchange into the same LDS at the same time (each into its own 64 slots, so that every wave can check what it received) —
// an idle LDS unit serving one wave in lane order proves nothing about a busy one.
__global__ __launch_bounds__(A_THREADS) void k_probe_xchg_order(int rounds, int *ok_out) {
    __shared__ uint32_t slots[A_WAVES * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *slot = slots + wave * 64;
    const uint32_t base = (uint32_t)(uintptr_t)(a3_lds_u8 *)slot;
    int ok = 1;
    uint32_t rng = 0x9E3779B9u * (uint32_t)(threadIdx.x + 1);
    for (int r = 0; r < rounds; r++) {
        slot[lane] = 0;
        __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        rng = rng * 1664525u + 1013904223u;
        uint32_t h;
        switch ((r + wave) & 7) {                                  // (the waves run different patterns at the same time)
            case 0: h = 0; break;                                  // every lane the same slot
            case 1: h = (uint32_t)lane >> 1; break;                // neighbours in pairs
            case 2: h = (uint32_t)lane & 1; break;                 // two interleaved chains
            case 3: h = (uint32_t)lane & 7; break;
            case 4: h = (uint32_t)(63 - lane) >> 2; break;
            case 5: h = (rng >> 16) & 3; break;
            case 6: h = (rng >> 16) & 15; break;
            default: h = (rng >> 16) & 63; break;
        }
        uint32_t got;
        const uint32_t addr = base + 4u * h, mine = (uint32_t)lane + 1u;
        asm volatile("ds_wrxchg_rtn_b32 %[g], %[a], %[m]\n\ts_waitcnt lgkmcnt(0)\n\t" : [g] "=&v"(got) : [a] "v"(addr), [m] "v"(mine) : "memory");
        __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        uint32_t want = 0, last = 0;
        for (int l = 0; l < 64; l++) {
            const uint32_t hl = (uint32_t)__builtin_amdgcn_readlane((int)h, l);
            if (hl == h) { if (l < lane) want = (uint32_t)l + 1u; last = (uint32_t)l + 1u; }
        }
        if (got != want || slot[h] != last) ok = 0;
        __builtin_amdgcn_wave_barrier();
    }
    const int all = __syncthreads_and(ok != 0);
    if (threadIdx.x == 0) *ok_out = all;
}

// The guard of every call: the probe above is evidence, not a guarantee, and the product's tolerance is 0.  One wavefront per
// SAMPLED position recomputes that position's link from first principles — scan backwards for the nearest inserted position with
// the same hash, 64 candidates per step — and raises *flag if k_links3 stored something else.  The engine reads the flag with the
// counters it reads anyway, distrusts the ticket form from then on (k_links2 needs no ordering assumption) and runs the call again.
__global__ __launch_bounds__(64) void k_links_guard(const uint8_t *__restrict__ in, uint64_t in_total, const SegDev *__restrict__ segs,
                                                    const uint64_t *__restrict__ bnds, const SpanDev *__restrict__ spans, int nspans,
                                                    const uint16_t *__restrict__ link, const uint32_t *__restrict__ hflags,
                                                    unsigned long long *flag) {
    const int lane = threadIdx.x;
    const uint32_t sidx = blockIdx.x;
    const SpanDev span = spans[(uint32_t)(((uint64_t)sidx * 2654435761u) >> 7) % (uint32_t)nspans];
    const SegDev seg = segs[span.seg];
    const uint8_t *d = in + seg.buf_off;
    const uint64_t avail = in_total - seg.buf_off;
    const uint64_t *b = bnds + seg.bnd_off;
    const int nb = (int)seg.bnd_cnt;
    if (nb > 64 || span.end <= span.start) return;                // (a stream flushed hundreds of times inside one window: not sampled)
    const int64_t q = span.start + (int64_t)((((uint64_t)sidx + 1) * 0x9E3779B97F4A7C15ull) >> 20) % (span.end - span.start);
    auto inserted = [&](int64_t x) -> bool {                      // InsertString ran at x: three bytes of lookahead in front of the next boundary
        if (x < 0) return false;
        int j = 0;
        while (j < nb && (int64_t)b[j] <= x) j++;
        if (j >= nb || (int64_t)b[j] - x < 3) return false;
        if (hflags && x < seg.seg_start) return (hflags[x >> 5] >> (x & 31)) & 1u;
        return true;
    };
    auto hash_at = [&](int64_t x) -> uint32_t {
        if ((uint64_t)x + 3 > avail) return 0xFFFFFFFFu;
        return (((uint32_t)d[x] << 10) ^ ((uint32_t)d[x + 1] << 5) ^ d[x + 2]) & 0x7FFFu;
    };
    uint32_t want = 0;
    if (inserted(q)) {
        const uint32_t hq = hash_at(q);
        for (int64_t base = 1; base <= 32767; base += 64) {
            const int64_t dist = base + lane, x = q - dist;
            const bool hit = dist <= 32767 && x >= 0 && inserted(x) && hash_at(x) == hq;
            const uint64_t m = __ballot(hit);
            if (m) { want = (uint32_t)(base + __builtin_ctzll(m)); break; }
        }
    }
    if (lane == 0 && (uint32_t)link[seg.buf_off + q] != want) atomicAdd(flag, 1ull);
}

__global__ __launch_bounds__(A_THREADS) void k_links3(const uint8_t *__restrict__ in, uint64_t in_total,
                                                      const SegDev *__restrict__ segs, const uint64_t *__restrict__ bnds,
                                                      const SpanDev *__restrict__ spans, uint16_t *__restrict__ link,
                                                      const uint32_t *__restrict__ hflags, unsigned long long *order_flag) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem3[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // LDS byte addresses for the hand-written part (the compiler turns volatile accesses through generic pointers into
    // FLAT instructions followed by s_waitcnt vmcnt(0), which is exactly what the ticket chain cannot afford)
    const uint32_t head_a = (uint32_t)(uintptr_t)(a3_lds_u8 *)smem3;
    const uint32_t turn_a = head_a + A3_HEAD_BYTES;
    const SpanDev span = spans[blockIdx.x];
    const SegDev seg = segs[span.seg];
    const uint8_t *d = in + seg.buf_off;
    uint16_t *lk = link + seg.buf_off;
    const int64_t warm0 = span.start - WSIZE > 0 ? span.start - WSIZE : 0;
    const uint64_t avail = in_total - seg.buf_off;

    {   // every bucket starts "far away": q - head > 32767 for every q of the span
        const uint32_t far = (uint32_t)(warm0 - 40000);
        uint4 *h4 = (uint4 *)smem3;
        for (int i = threadIdx.x; i < 32768 / 4; i += A_THREADS) h4[i] = make_uint4(far, far, far, far);
        if (threadIdx.x == 0) *(int *)(smem3 + A3_HEAD_BYTES) = 0;
    }
    __syncthreads();

    const uint64_t *b = bnds + seg.bnd_off;
    const int nb = (int)seg.bnd_cnt;
    int bi = 0;
    int64_t bcur = bi < nb ? (int64_t)b[bi] : INT64_MAX;
    const int64_t ngr = (span.end - warm0 + 64 * A3_U - 1) / (64 * A3_U);
    auto fetch = [&](int64_t qn) -> uint32_t { return (qn < span.end && (uint64_t)qn + 4 <= avail) ? load_u32_unaligned(d + qn) : 0u; };
    uint32_t wnext[A3_U];
#pragma unroll
    for (int u = 0; u < A3_U; u++) wnext[u] = fetch(warm0 + 64 * (A3_U * (int64_t)wave + u) + lane);
    for (int64_t t = wave; t < ngr; t += A_WAVES) {
        uint32_t haddr[A3_U], e_old[A3_U];
        uint64_t m_ins[A3_U];
#pragma unroll
        for (int u = 0; u < A3_U; u++) {
            const int64_t q0 = warm0 + 64 * (A3_U * t + u), q = q0 + lane;
            const uint32_t wcur = wnext[u];
            wnext[u] = fetch(q + 64 * A3_U * A_WAVES);
            while (bcur <= q0) { bi++; bcur = bi < nb ? (int64_t)b[bi] : INT64_MAX; } // uniform
            bool ins = q < span.end;
            if (bi >= nb) ins = false;
            else if (bcur < q0 + 64 + 2) { // a segment end is near: InsertString needs lookahead >= 3 (:780,:817)
                int j = bi;
                while (j < nb && (int64_t)b[j] <= q) j++;
                ins = ins && j < nb && (int64_t)b[j] - q >= 3;
            }
            if (hflags && ins && q < seg.seg_start) ins = (hflags[q >> 5] >> (q & 31)) & 1u; // history a DeflateFast level left: only inserted positions
            uint32_t h = 0;
            if (ins) {
                uint32_t w = wcur;
                if ((uint64_t)q + 4 > avail) w = (uint32_t)d[q] | ((uint32_t)d[q + 1] << 8) | ((uint32_t)d[q + 2] << 16);
                    const uint32_t b0 = w & 0xFF, b1 = (w >> 8) & 0xFF, b2 = (w >> 16) & 0xFF;
                h = ((b0 << 10) ^ (b1 << 5) ^ b2) & 0x7FFF; // :404,:420
            }
            haddr[u] = head_a + 4u * h; m_ins[u] = __ballot(ins);
            e_old[u] = 0;
        }
        // ---- the ticket: wait for it, exchange the four slices into the table in order, pass it on
        {
            static_assert(A3_U == 4, "the ticket section is written out for four slices");
            uint32_t c, sc;
            uint64_t sv;
            const uint32_t tt = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);   // wave-uniform (the compiler cannot see it)
            const uint32_t qb = (uint32_t)(warm0 + 64 * A3_U * t) + (uint32_t)lane;   // position of this lane in slice 0 (low 32 bits)
            const uint32_t q1 = qb + 64u, q2 = qb + 128u, q3 = qb + 192u, tn = tt + 1u;
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
                "s_mov_b64 exec, %[mi0]\n\t" "ds_wrxchg_rtn_b32 %[e0], %[a0], %[q0]\n\t"
                "s_mov_b64 exec, %[mi1]\n\t" "ds_wrxchg_rtn_b32 %[e1], %[a1], %[q1]\n\t"
                "s_mov_b64 exec, %[mi2]\n\t" "ds_wrxchg_rtn_b32 %[e2], %[a2], %[q2]\n\t"
                "s_mov_b64 exec, %[mi3]\n\t" "ds_wrxchg_rtn_b32 %[e3], %[a3], %[q3]\n\t"
                "s_mov_b64 exec, %[sv]\n\t"
                "ds_write_b32 %[ta], %[tn]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                : [e0] "+&v"(e_old[0]), [e1] "+&v"(e_old[1]), [e2] "+&v"(e_old[2]), [e3] "+&v"(e_old[3]),
                  [c] "=&v"(c), [sc] "=&s"(sc), [sv] "=&s"(sv)
                : [ta] "v"(turn_a), [tt] "s"(tt), [tn] "v"(tn),
                  [a0] "v"(haddr[0]), [a1] "v"(haddr[1]), [a2] "v"(haddr[2]), [a3] "v"(haddr[3]),
                  [q0] "v"(qb), [q1] "v"(q1), [q2] "v"(q2), [q3] "v"(q3),
                  [mi0] "s"(m_ins[0]), [mi1] "s"(m_ins[1]), [mi2] "s"(m_ins[2]), [mi3] "s"(m_ins[3])
                : "scc", "memory");
        }
        // ---- links of the four slices
        bool disorder = false;
#pragma unroll
        for (int u = 0; u < A3_U; u++) {
            const int64_t q = warm0 + 64 * (A3_U * t + u) + lane;
            uint32_t dist = 0;
            if ((m_ins[u] >> lane) & 1) {
                dist = (uint32_t)q - e_old[u];
                // EVERY exchange checks the assumption it rests on: what a lane receives is a position in front of its own (the table's
                // earlier content, or a lower lane of the same exchange).  If the LDS served two lanes with one bucket in descending
                // order, the lower lane — served second — would receive the HIGHER lane's position: a distance <= 0.  Any service order
                // other than ascending has such a pair, so this is a complete check of the order, on every position, not a sample.
                if ((int32_t)dist <= 0) disorder = true;
                if (dist > 32767u) dist = 0; // candidates farther than the window are never followed (:609)
            }
            if (q >= span.start && q < span.end) lk[q] = (uint16_t)dist;
        }
        if (order_flag && __any(disorder) && lane == 0) atomicAdd(order_flag, 1ull);
    }
}

// ============================================================================================
// Stage B: k_match.  One workgroup per 16384-position tile.  The tile's bytes plus 32512 bytes of
// history and the u16 links of both are staged in LDS (≈144 KiB of the CU's 160 KiB), so every chain
// step — prev[] hop, quick reject, byte compare — is an LDS access.  Lanes pull positions from an LDS
// counter and run a flattened state machine (one chain step or one 4-byte compare per iteration)
// so that wavefront lanes stay converged while walking chains of different lengths.
//   M2 = result of FindLongestMatch entered with matchLen 2 and the full max_chain budget,
//   Mq = the same walk's state after max_chain>>2 candidates (what the reference computes when it is
//        entered with matchLen >= goodLength, :495).
// Entry encoding: len | dist<<16 ; 0 = no match of length >= 3.
// ============================================================================================
enum : int { B_THREADS = 1024 };
enum : int { B_DATA_BYTES = B_HIST + B_TILE + B_TAIL + 8, B_LINKS = B_HIST + B_TILE };
enum : int { B_LDS_BYTES = B_DATA_BYTES + B_LINKS * 2 + 16 };

#if SZL_LAB   // the first form of the full search (lab library only; the product runs k_match4, szl_kernels_match2.hip)
template <bool DBG>
__global__ __launch_bounds__(B_THREADS) void k_match(const uint8_t *__restrict__ in, const SegDev *__restrict__ segs,
                                                     const TileDev *__restrict__ tiles, const uint16_t *__restrict__ link,
                                                     MTab mtab, LevelParams P, unsigned long long *dbg, int fth, int vth) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *sdata32 = (uint32_t *)smem;                          // B_DATA_BYTES
    uint16_t *slink = (uint16_t *)(smem + B_DATA_BYTES);           // B_LINKS entries
    int *s_counter = (int *)(smem + B_DATA_BYTES + B_LINKS * 2);

    const TileDev tile = tiles[blockIdx.x];
    const SegDev seg = segs[tile.seg];
    const uint8_t *d = in + seg.buf_off;
    const uint16_t *lk = link + seg.buf_off;
    uint32_t *__restrict__ mt2 = mtab.m2 + seg.buf_off;
    uint32_t *__restrict__ mtq = mtab.mq + seg.buf_off;
    const int64_t t0 = tile.start;
    const int tlen = tile.len;
    const int64_t dlo = t0 - B_HIST; // buffer position of LDS data byte 0 (may be negative)
    const int64_t seg_end = seg.look_end; // lookahead end

    // ---- stage the window into LDS
    for (int i = threadIdx.x; i < B_DATA_BYTES / 4; i += B_THREADS) {
        int64_t pos = dlo + 4 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos + 4 <= seg_end) w = load_u32_unaligned(d + pos);
        else {
            for (int k = 0; k < 4; k++) {
                int64_t pk = pos + k;
                if (pk >= 0 && pk < seg_end) w |= (uint32_t)d[pk] << (8 * k);
            }
        }
        sdata32[i] = w;
    }
    for (int i = threadIdx.x; i < B_LINKS / 2; i += B_THREADS) {
        int64_t pos = dlo + 2 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos + 2 <= t0 + tlen) {
            uint16_t a = lk[pos], c = lk[pos + 1];
            w = (uint32_t)a | ((uint32_t)c << 16);
        } else {
            if (pos >= 0 && pos < t0 + tlen) w |= lk[pos];
            if (pos + 1 >= 0 && pos + 1 < t0 + tlen) w |= (uint32_t)lk[pos + 1] << 16;
        }
        // "no previous position" (0) is staged as 0xFFFF: cl - 0xFFFF is below every limit, so the chain-end test
        // needs no separate zero check (real links are <= 32767)
        if ((w & 0xFFFFu) == 0) w |= 0xFFFFu;
        if ((w >> 16) == 0) w |= 0xFFFF0000u;
        ((uint32_t *)slink)[i] = w;
    }
    if (threadIdx.x == 0) *s_counter = 0;
    __syncthreads();

    // Unaligned 32-bit read at LDS data byte i from two aligned dwords.  (gfx950 LDS does accept unaligned
    // ds_read_b32 / ds_read_u16, but measured slower here: the kernel is co-limited by LDS bank conflicts.)
    auto ldsdw = [&](int i) -> uint32_t {
        uint32_t w0 = sdata32[i >> 2], w1 = sdata32[(i >> 2) + 1];
        return __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)(i & 3));
    };
    const uint8_t *sdata8 = smem;

    // Three phases per outer iteration, each a tight loop of its own so that the hot one (the chain step, ~80 % of
    // all lane-steps) is a couple of dozen instructions: FETCH hands out positions, QUICK walks chains — two bytes
    // (quick reject at offset `best`, :505) and one u16 (the candidate's link) from LDS per step — until too many
    // lanes have dropped out, VERIFY compares the candidates that passed the quick test dword by dword.
    // Window base (App. A.2) for the first and last position of the tile; inside a tile it changes at most once.
    const int64_t base_lo = base_of((int64_t)seg.abs0 + t0), base_hi = base_of((int64_t)seg.abs0 + t0 + tlen - 1);
    // first tile position that already uses base_hi: smallest s with s+1-base_lo > 65273
    const int64_t sw = base_lo == base_hi ? (int64_t)1 << 40 : (base_lo + 65273) - (int64_t)seg.abs0 - t0; // tile-relative
    const int basem_lo = (int)(base_lo - (int64_t)seg.abs0 - dlo), basem_hi = (int)(base_hi - (int64_t)seg.abs0 - dlo); // LDS index of window index 1... (clamped below)
    const int lane = threadIdx.x & 63;
    const uint64_t lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int SNAPLEFT = P.max_chain - (P.max_chain >> 2); // `left` value at which the quarter-budget walk would stop
    enum { NEED = 0, DONE = 1, QUICK = 2, VERIFY = 3 }; // idle lanes are those with mode < QUICK
    const int F_THRESH = fth, V_THRESH = vth;
    int wnext = 0, wend = 0;       // wave-uniform slice of tile positions being handed out
    bool exhausted = false;
    int mode = NEED;
    int pl = B_HIST, cl = B_HIST, off = 0, lnk = 0;
    int best = 2, cap = 4, nice = 4, mincl = 0, left = 1, p = 0;
    uint32_t pb = 0, res2 = 0, resq = 0;
    unsigned long long n_qs = 0, n_vs = 0; // DBG: steps this lane took part in

    // A small scheduler picks, per visit, the phase that has enough lanes waiting for it: the phases cost the same
    // for one lane as for 64, so each is run only when it is well occupied (or nothing else can make progress).
    for (;;) {
        const uint64_t idle = __ballot(mode < QUICK);
        const uint64_t vm = __ballot(mode == VERIFY);
        const int ni = __builtin_popcountll(idle), nv = __builtin_popcountll(vm);
        const int nq = 64 - ni - nv;
        if ((ni >= F_THRESH && !exhausted) || (nq == 0 && nv == 0)) {
            // ---------------- FETCH: retire finished positions and hand out new ones
            if (mode == DONE) {
                { const uint32_t e_ = mt_pack(res2, resq); mt2[t0 + p] = e_; if (e_ >> 25) mtq[t0 + p] = resq; }
                mode = NEED;
            }
            if (!exhausted) {
                if (wnext >= wend) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(s_counter, 256);
                    base = __builtin_amdgcn_readfirstlane(base);
                    wnext = base < tlen ? base : tlen;
                    wend = base + 256 < tlen ? base + 256 : tlen;
                    if (wnext >= wend) exhausted = true;
                }
                const int rank = __builtin_popcountll(idle & lanemask_lt);
                if (mode == NEED && wnext + rank < wend) {
                    p = wnext + rank;
                    const int64_t rem = seg_end - (t0 + p);
                    res2 = 0; resq = 0;
                    bool ok = rem >= MIN_MATCH && P.strategy != 2; // :780, HuffmanOnly :786
                    if (ok) {
                        pl = p + B_HIST;
                        const int l0 = (int)slink[pl];                           // hashHead (:782)
                        // LDS index of the buffer position whose window index is 1 (entries below it were clamped by a slide, :450-461)
                        const int basem = (int64_t)p >= sw ? basem_hi : basem_lo;
                        // first candidate: strstart - hashHead <= MAX_DIST (:788); chain: curMatch > limit (:609)
                        const int firstmin = pl - MAX_DIST > basem ? pl - MAX_DIST : basem;
                        cl = pl - l0;
                        ok = cl >= firstmin; // l0 == 0xFFFF (none) fails this too
                        if (ok) {
                            mincl = pl - (MAX_DIST - 1) > basem ? pl - (MAX_DIST - 1) : basem;
                            cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;         // scanMax :479
                            nice = rem < (int64_t)P.nice ? (int)rem : P.nice;       // :485 (64-bit compare: rem can exceed 2^31)
                            best = 2; left = P.max_chain;
                            pb = sdata8[pl + 2];
                            mode = QUICK;
                        }
                    }
                    if (!ok) mt2[t0 + p] = 0u;
                }
                wnext = wnext + ni < wend ? wnext + ni : wend;
            }
            if (exhausted && __all(mode == NEED)) break;
            continue;
        }
        if (nv >= V_THRESH || nq == 0) {
            // ---------------- VERIFY: dword-by-dword comparison of candidates that passed the quick test (2 steps per visit)
#pragma unroll
            for (int u = 0; u < 2; u++)
            if (mode == VERIFY) {
                if (DBG) n_vs++;
                const uint32_t x = ldsdw(cl + off) ^ ldsdw(pl + off);
                const bool eq = x == 0;
                const int l = off + (eq ? 4 : (__builtin_ctz(x) >> 3));
                const bool more = eq & (l < cap);
                off = l;
                if (!more) {
                    const int L = l < cap ? l : cap;
                    bool nicehit = false;
                    if (L > best) { // :593-607
                        best = L;
                        res2 = (uint32_t)L | ((uint32_t)(pl - cl) << 16);
                        if (left > SNAPLEFT) resq = res2; // this candidate is among the first max_chain>>2: the quarter walk sees it too
                        nicehit = L >= nice;
                        if (!nicehit) pb = sdata8[pl + L];
                    }
                    const int left1 = left - 1;
                    const int c2 = cl - lnk;
                    const bool end = (c2 < mincl) | (left1 == 0);
                    left = nicehit ? left : left1;
                    cl = (nicehit | end) ? cl : c2;
                    mode = (nicehit | end) ? DONE : QUICK;
                }
            }
            continue;
        }
        // ---------------- QUICK: chain steps (branch-free body: two u16 from LDS, a dozen VALU), 4 per visit
#pragma unroll
        for (int u = 0; u < 4; u++)
        if (mode == QUICK) {
            if (DBG) n_qs++;
            const uint32_t qb = sdata8[cl + best];   // a longer match must agree at offset `best` (scan_end, :505)
            lnk = (int)slink[cl];
            const bool pass = qb == pb;
            // next candidate of the chain, or the end of this position (:609)
            const int left1 = left - 1;
            const int c2 = cl - lnk;
            const bool end = (c2 < mincl) | (left1 == 0);
            left = pass ? left : left1;
            cl = (pass | end) ? cl : c2;
            off = 0;
            mode = pass ? VERIFY : (end ? DONE : QUICK);
        }
    }
    if (DBG && dbg) {
        // per-wave totals: n_qs/n_vs are per-lane participation counts, so reduce with max over the wave
        unsigned long long qs = n_qs, vs = n_vs;
        for (int o = 32; o > 0; o >>= 1) {
            unsigned long long a = __shfl_xor(qs, o), c = __shfl_xor(vs, o);
            qs = a > qs ? a : qs; vs = c > vs ? c : vs;
        }
        if (lane == 0) { atomicAdd(dbg + 2, qs); atomicAdd(dbg + 5, vs); }
        // lane-steps: every participating lane counted the ballot, so divide by participation via lane 0 only is wrong; sum 1 per lane-step instead
        unsigned long long ql = n_qs, vl = n_vs;
        for (int o = 32; o > 0; o >>= 1) { ql += __shfl_xor(ql, o); vl += __shfl_xor(vl, o); }
        if (lane == 0) { atomicAdd(dbg + 3, ql); atomicAdd(dbg + 4, vl); }
    }
}

#endif

// ============================================================================================
#if SZL_LAB   // (laboratory library only: the on-demand form of stage B — measured slower than the full search on every data class, profiles/r03/forms_by_class.log)
// Stage B, on-demand form: k_match_lazy.
// The parse of stage C only ever reads M2/Mq at the positions it visits — a clean iteration, or the lazy look at the
// position after a match start; everything inside an emitted match is skipped (C/DeflaterEngine.cs:802-828).  On
// repetitive data (logs: ≈10 % of the positions) searching every position is mostly wasted work.  Here the tile's lanes
// WALK instead: a lane takes a start (every STRIDE positions), evaluates FindLongestMatch there, applies the DeflateSlow
// step to learn which position is needed next, evaluates that, and so on, until it lands on a clean position another
// walker has already passed (from there on the two parses are identical) or leaves the tile.  Every path
// that starts at a stride position — in particular at every stage-C range start — is thereby fully evaluated inside the
// tile; entries nobody needed keep M_UNSET, and the handful stage C does touch (where the true path crosses a tile
// boundary) are evaluated there on demand (eval_global, szl_kernels_parse.hip).
// A walker is a dependent chain (evaluate -> step -> evaluate), so an evaluation costs 2.5-4x what it costs the
// independent dispenser of k_match: the engine runs a pilot on a sample of tiles and uses this kernel only when the
// evaluated fraction is small enough to pay for that (szl_engine.hip).
// PAIR: lanes work in pairs — the even lane is the walker and evaluates x, the odd lane is its scout and evaluates x+1
// at the same time (x+1 is always needed after a clean x, and after a lazy x whenever a longer match is found), which
// halves the length of the dependent chain; the step phase then consumes both positions at once.
// tile_step/tile_first: the launch handles tiles tile_first, tile_first + tile_step, ... (pilot: a sample).
// ============================================================================================
template <int STRIDE, bool PAIR>
__global__ __launch_bounds__(B_THREADS) void k_match_lazy(const uint8_t *__restrict__ in, const SegDev *__restrict__ segs,
                                                          const TileDev *__restrict__ tiles, const uint16_t *__restrict__ link,
                                                          MTab mtab, LevelParams P, unsigned long long *dbg, int fth, int vth,
                                                          int tile_first, int tile_step) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *sdata32 = (uint32_t *)smem;
    uint16_t *slink = (uint16_t *)(smem + B_DATA_BYTES);
    int *s_counter = (int *)(smem + B_DATA_BYTES + B_LINKS * 2);
    uint32_t *s_vis = (uint32_t *)(smem + B_DATA_BYTES + B_LINKS * 2 + 16); // B_TILE bits: clean iteration seen at this tile position

    const TileDev tile = tiles[tile_first + (int)blockIdx.x * tile_step];
    const SegDev seg = segs[tile.seg];
    const uint8_t *d = in + seg.buf_off;
    const uint16_t *lk = link + seg.buf_off;
    uint32_t *__restrict__ mt2 = mtab.m2 + seg.buf_off;
    uint32_t *__restrict__ mtq = mtab.mq + seg.buf_off;
    const int64_t t0 = tile.start;
    const int tlen = tile.len;
    const int64_t dlo = t0 - B_HIST;
    const int64_t seg_end = seg.look_end; // lookahead end

    for (int i = threadIdx.x; i < B_DATA_BYTES / 4; i += B_THREADS) {
        int64_t pos = dlo + 4 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos + 4 <= seg_end) w = load_u32_unaligned(d + pos);
        else {
            for (int k = 0; k < 4; k++) {
                int64_t pk = pos + k;
                if (pk >= 0 && pk < seg_end) w |= (uint32_t)d[pk] << (8 * k);
            }
        }
        sdata32[i] = w;
    }
    for (int i = threadIdx.x; i < B_LINKS / 2; i += B_THREADS) {
        int64_t pos = dlo + 2 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos + 2 <= t0 + tlen) {
            uint16_t a = lk[pos], c = lk[pos + 1];
            w = (uint32_t)a | ((uint32_t)c << 16);
        } else {
            if (pos >= 0 && pos < t0 + tlen) w |= lk[pos];
            if (pos + 1 >= 0 && pos + 1 < t0 + tlen) w |= (uint32_t)lk[pos + 1] << 16;
        }
        if ((w & 0xFFFFu) == 0) w |= 0xFFFFu;     // "no previous position": see k_match
        if ((w >> 16) == 0) w |= 0xFFFF0000u;
        ((uint32_t *)slink)[i] = w;
    }
    for (int i = threadIdx.x; i < B_TILE / 32; i += B_THREADS) s_vis[i] = 0;
    if (threadIdx.x == 0) *s_counter = 0;
    __syncthreads();

    auto ldsdw = [&](int i) -> uint32_t {
        uint32_t w0 = sdata32[i >> 2], w1 = sdata32[(i >> 2) + 1];
        return __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)(i & 3));
    };
    const uint8_t *sdata8 = smem;
    const int64_t base_lo = base_of((int64_t)seg.abs0 + t0), base_hi = base_of((int64_t)seg.abs0 + t0 + tlen - 1);
    const int64_t sw = base_lo == base_hi ? (int64_t)1 << 40 : (base_lo + 65273) - (int64_t)seg.abs0 - t0;
    const int basem_lo = (int)(base_lo - (int64_t)seg.abs0 - dlo), basem_hi = (int)(base_hi - (int64_t)seg.abs0 - dlo);
    const int lane = threadIdx.x & 63;
    const uint64_t lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int SNAPLEFT = P.max_chain - (P.max_chain >> 2);
    const int nstarts = (tlen + STRIDE - 1) / STRIDE;
    enum { NEED = 0, DONE = 1, QUICK = 2, VERIFY = 3 }; // NEED: no walker; DONE: evaluation of position p finished
    const int F_THRESH = fth, V_THRESH = vth;
    int wnext = 0, wend = 0;
    bool exhausted = false;
    int mode = NEED;
    int pl = B_HIST, cl = B_HIST, off = 0, lnk = 0;
    int best = 2, cap = 4, nice = 4, mincl = 0, left = 1, p = 0;
    uint32_t pb = 0, res2 = 0, resq = 0;
    int wL = 0;           // the walker's pending match length (0: clean)
    unsigned long long n_eval = 0;

    // The DeflateSlow step at tile position x given FindLongestMatch's results there (r2 full budget, rq quarter budget)
    // and the pending match length wL (0: clean iteration).  Returns the next position the parse needs; updates wL.
    auto consume = [&](int x, uint32_t r2, uint32_t rq, int &wl) -> int {
        if (wl == 0) { // clean iteration at x (:780-800)
            int len = (int)(r2 & 0xFFFF);
            const int dist = (int)(r2 >> 16);
            if (len != 0 && len <= 5 && (P.strategy == 1 || (len == MIN_MATCH && dist > TOO_FAR))) len = 0; // :794-797
            wl = len;
            return x + 1;
        }
        // lazy evaluation at x: is there a strictly longer match than the one found at x-1 ? (:802)
        const int64_t rem = seg_end - (t0 + x);
        uint32_t better = 0;
        if (rem >= MIN_MATCH) {
            const int capx = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
            if (wl < capx) {
                const int nicex = rem < (int64_t)P.nice ? (int)rem : P.nice;
                uint32_t cand;
                if (wl < P.good) cand = r2;
                else if (wl < nicex) cand = rq; // chainLength >>= 2 (:495)
                else { // entered with matchLen >= niceLength': first strictly longer candidate among max_chain>>2 wins (rare)
                    cand = 0;
                    const int c = x + B_HIST;
                    int budget = P.max_chain >> 2;
                    const int basem = (int64_t)x >= sw ? basem_hi : basem_lo;
                    const int firstmin = c - MAX_DIST > basem ? c - MAX_DIST : basem;
                    const int minc = c - (MAX_DIST - 1) > basem ? c - (MAX_DIST - 1) : basem;
                    int cc = c - (int)slink[c];
                    if (cc >= firstmin) {
                        for (;;) {
                            int l = 0;
                            if (sdata8[cc + wl] == sdata8[c + wl]) { while (l < capx && sdata8[cc + l] == sdata8[c + l]) l++; }
                            if (l > wl) { cand = (uint32_t)l | ((uint32_t)(c - cc) << 16); break; }
                            const int c2 = cc - (int)slink[cc];
                            if (c2 < minc) break;
                            if (--budget == 0) break;
                            cc = c2;
                        }
                    }
                }
                if ((int)(cand & 0xFFFF) > wl && !(P.strategy == 1 && (cand & 0xFFFF) <= 5)) better = cand;
            }
        }
        if (better) { wl = (int)(better & 0xFFFF); return x + 1; }
        const int nx = x - 1 + wl;
        wl = 0;
        return nx;
    };
    // start FindLongestMatch at tile position q on this lane
    auto launch = [&](int q) {
        p = q;
        n_eval++;
        const int64_t rem = seg_end - (t0 + p);
        res2 = 0; resq = 0;
        bool ok = rem >= MIN_MATCH && P.strategy != 2; // :780, HuffmanOnly :786
        if (ok) {
            pl = p + B_HIST;
            const int l0 = (int)slink[pl];
            const int basem = (int64_t)p >= sw ? basem_hi : basem_lo;
            const int firstmin = pl - MAX_DIST > basem ? pl - MAX_DIST : basem;
            cl = pl - l0;
            ok = cl >= firstmin;
            if (ok) {
                mincl = pl - (MAX_DIST - 1) > basem ? pl - (MAX_DIST - 1) : basem;
                cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;
                nice = rem < (int64_t)P.nice ? (int)rem : P.nice;
                best = 2; left = P.max_chain;
                pb = sdata8[pl + 2];
                mode = QUICK;
            } else cl = B_HIST;
        }
        if (!ok) mode = DONE; // nothing to search: consumed as (0, 0) on the next STEP visit
    };
    auto visit_clean = [&](int q) -> bool { // true if this walker is the first at clean position q
        const uint32_t bit = 1u << (q & 31);
        return (atomicOr(&s_vis[q >> 5], bit) & bit) == 0;
    };
    const bool head = !PAIR || (lane & 1) == 0;  // PAIR: even lane = walker (evaluates x), odd lane = its scout (evaluates x+1)
    const uint64_t EVEN = 0x5555555555555555ull;

    for (;;) {
        const uint64_t dm = __ballot(mode == DONE), nm = __ballot(mode == NEED);
        const int nv = __builtin_popcountll(__ballot(mode == VERIFY));
        const int nq = 64 - __builtin_popcountll(dm) - __builtin_popcountll(nm) - nv;
        // lanes (pairs) the STEP phase can move forward
        int actionable;
        if (PAIR) actionable = 2 * (__builtin_popcountll(dm & ((dm | nm) >> 1) & EVEN) + (exhausted ? 0 : __builtin_popcountll(nm & EVEN)));
        else actionable = __builtin_popcountll(dm) + (exhausted ? 0 : __builtin_popcountll(nm));
        if ((actionable >= F_THRESH) || (nq == 0 && nv == 0)) {
            // ---------------- STEP: consume finished evaluations (DeflateSlow step), start new walkers, launch the next evaluations
            int nx = -1; // tile position the walker needs next
            if (PAIR) {
                const int s_mode = __shfl_down(mode, 1);
                const uint32_t s_r2 = (uint32_t)__shfl_down((int)res2, 1), s_rq = (uint32_t)__shfl_down((int)resq, 1);
                const int s_p = __shfl_down(p, 1);
                const bool ready = head && mode == DONE && (s_mode == DONE || s_mode == NEED);
                const bool h_ready = __shfl((int)ready, lane & ~1) != 0;
                if (h_ready && mode == DONE) { const uint32_t e_ = mt_pack(res2, resq); mt2[t0 + p] = e_; if (e_ >> 25) mtq[t0 + p] = resq; }
                if (ready) {
                    const int x = p;
                    nx = consume(x, res2, resq, wL);
                    if (nx == x + 1 && s_mode == DONE && s_p == x + 1) { // the scout already has x+1
                        if (wL == 0 && !visit_clean(x + 1)) nx = -1;    // x+1 is a clean iteration another walker has passed: merge
                        else nx = consume(x + 1, s_r2, s_rq, wL);
                    }
                    if (nx >= tlen) { nx = -1; wL = 0; } // the walker leaves the tile
                    if (nx < 0) wL = 0;
                }
                if (h_ready) mode = NEED;
            } else if (mode == DONE) {
                const uint32_t e_ = mt_pack(res2, resq); mt2[t0 + p] = e_; if (e_ >> 25) mtq[t0 + p] = resq;
                nx = consume(p, res2, resq, wL);
                mode = NEED;
                if (nx >= tlen) { nx = -1; wL = 0; } // the walker leaves the tile (its continuation belongs to the next tile's walkers)
            }
            // walkers without a path take a start
            const uint64_t want = __ballot(head && mode == NEED && nx < 0);
            if (!exhausted && want) {
                if (wnext >= wend) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(s_counter, 64);
                    base = __builtin_amdgcn_readfirstlane(base);
                    wnext = base < nstarts ? base : nstarts;
                    wend = base + 64 < nstarts ? base + 64 : nstarts;
                    if (wnext >= wend) exhausted = true;
                }
                const int rank = __builtin_popcountll(want & lanemask_lt);
                if (head && mode == NEED && nx < 0 && wnext + rank < wend) { nx = (wnext + rank) * STRIDE; wL = 0; }
                const int nw = __builtin_popcountll(want);
                wnext = wnext + nw < wend ? wnext + nw : wend;
            }
            bool go = head && nx >= 0;
            if (go && wL == 0 && !visit_clean(nx)) go = false; // clean: merge with any walker that has been here
            int q = nx;
            if (PAIR) { // the scout follows its walker one position ahead (x+1 is needed after x whenever x is clean, usually otherwise)
                const bool h_go = __shfl((int)go, lane & ~1) != 0;
                const int h_q = __shfl(q, lane & ~1);
                if (!head) { go = h_go && h_q + 1 < tlen; q = h_q + 1; }
            }
            if (go) launch(q);
            if (exhausted && __all(mode == NEED)) break;
            continue;
        }
        if (nv >= V_THRESH || nq == 0) {
#pragma unroll
            for (int u = 0; u < 2; u++)
            if (mode == VERIFY) {
                const uint32_t x = ldsdw(cl + off) ^ ldsdw(pl + off);
                const bool eq = x == 0;
                const int l = off + (eq ? 4 : (__builtin_ctz(x) >> 3));
                const bool more = eq & (l < cap);
                off = l;
                if (!more) {
                    const int L = l < cap ? l : cap;
                    bool nicehit = false;
                    if (L > best) { // :593-607
                        best = L;
                        res2 = (uint32_t)L | ((uint32_t)(pl - cl) << 16);
                        if (left > SNAPLEFT) resq = res2;
                        nicehit = L >= nice;
                        if (!nicehit) pb = sdata8[pl + L];
                    }
                    const int left1 = left - 1;
                    const int c2 = cl - lnk;
                    const bool end = (c2 < mincl) | (left1 == 0);
                    left = nicehit ? left : left1;
                    cl = (nicehit | end) ? cl : c2;
                    mode = (nicehit | end) ? DONE : QUICK;
                }
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
        if (mode == QUICK) {
            const uint32_t qb = sdata8[cl + best];
            lnk = (int)slink[cl];
            const bool pass = qb == pb;
            const int left1 = left - 1;
            const int c2 = cl - lnk;
            const bool end = (c2 < mincl) | (left1 == 0);
            left = pass ? left : left1;
            cl = (pass | end) ? cl : c2;
            off = 0;
            mode = pass ? VERIFY : (end ? DONE : QUICK);
        }
    }
    if (dbg) {
        for (int o = 32; o > 0; o >>= 1) n_eval += __shfl_xor(n_eval, o);
        if (lane == 0) atomicAdd(dbg + 6, n_eval);
    }
}
#endif   // SZL_LAB

// hflags (optional, single streaming segment): bit q = buffer position q of the history was inserted into the hash chains
static std::atomic<int> g_links3_distrusted{0};   // the guard caught k_links3 storing a wrong link: this process uses k_links2 from then on
static std::atomic<uint32_t> g_links3_trips{0};
void links_distrust_ticket_form() { g_links3_distrusted.store(1, std::memory_order_release); g_links3_trips.fetch_add(1, std::memory_order_acq_rel); }
uint32_t links_guard_trips() { return g_links3_trips.load(std::memory_order_acquire); }
bool links_ticket_form_distrusted() { return g_links3_distrusted.load(std::memory_order_acquire) != 0; }

void launch_links(const uint8_t *in, uint64_t in_total, const SegDev *segs, const uint64_t *bnds, const SpanDev *spans,
                  int nspans, uint16_t *link, const uint32_t *hflags, unsigned long long *guard_flag, uint64_t span_bytes, hipStream_t st,
                  hipStream_t guard_st, hipEvent_t guard_ev) {
    if (nspans <= 0) return;
    int which = knob("SZL_LINKS", 3);   // 3 = pipelined (ticket) form, 2 = bucketed by owner wavefront, 1 = first form
    if (which == 3 && links_ticket_form_distrusted()) which = 2;
    static std::atomic<uint64_t> probe_done{0}, probe_ok{0};   // per device: does ds_wrxchg serve the lanes in ascending order?
    uint64_t dev_bit = 0;
    if (which == 3 && lds_attr_needed(probe_done, dev_bit)) {
        int *d_ok = nullptr, h_ok = 0;
        if (hipMalloc((void **)&d_ok, sizeof(int)) == hipSuccess) {
            hipLaunchKernelGGL(k_probe_xchg_order, dim3(1), dim3(A_THREADS), 0, st, 512, d_ok);
            if (hipMemcpyAsync(&h_ok, d_ok, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) h_ok = 0;
            (void)hipFree(d_ok);
        }
        if (h_ok && hipFuncSetAttribute((const void *)k_links3, hipFuncAttributeMaxDynamicSharedMemorySize, A3_LDS_BYTES) == hipSuccess)
            probe_ok.fetch_or(dev_bit, std::memory_order_release);
        else if (!h_ok) fprintf(stderr, "[szl] stage A: the LDS exchange order probe failed on this device; using the bucketed form (k_links2)\n");
        probe_done.fetch_or(dev_bit, std::memory_order_release);
    }
    if (which == 3 && (probe_ok.load(std::memory_order_acquire) & dev_bit)) {
        hipLaunchKernelGGL(k_links3, dim3(nspans), dim3(A_THREADS), A3_LDS_BYTES, st, in, in_total, segs, bnds, spans, link, hflags,
                           knob("SZL_LINKS_GUARD", 1) ? guard_flag : (unsigned long long *)nullptr);
        if (guard_flag && knob("SZL_LINKS_GUARD", 1)) {           // one sampled position per 4 MiB, 8..256 of them
            const int lab_break = knob("SZL_LINKS_GUARD_TEST", 0);   // (tests: pretend a mismatch, to exercise the fallback)
            uint64_t ns = span_bytes >> 22;
            ns = ns < 8 ? 8 : (ns > 256 ? 256 : ns);
            // the guard only reads (input, links) and raises a flag the caller looks at when the call is over: beside stage B on a side
            // stream when the caller has one (a 64 KiB call: 79 us of guard against 47 us of k_links3 in the critical path)
            hipStream_t gs = st;
            if (guard_st && guard_ev && hipEventRecord(guard_ev, st) == hipSuccess && hipStreamWaitEvent(guard_st, guard_ev, 0) == hipSuccess) gs = guard_st;
            hipLaunchKernelGGL(k_links_guard, dim3((unsigned)ns), dim3(64), 0, gs, in, in_total, segs, bnds, spans, nspans, (const uint16_t *)link, hflags, guard_flag);
            if (lab_break) (void)hipMemsetAsync(guard_flag, 1, 1, gs);
        }
        return;
    }
#if SZL_LAB
    if (which == 1 && !hflags) { hipLaunchKernelGGL(k_links, dim3(nspans), dim3(A_THREADS), 0, st, in, in_total, segs, bnds, spans, link); return; }
#endif
    hipLaunchKernelGGL(k_links2, dim3(nspans), dim3(A_THREADS), 0, st, in, in_total, segs, bnds, spans, link, hflags);
}

int match_lds_bytes() { return B_LDS_BYTES; }

hipError_t launch_match2(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int ntiles, const uint16_t *link,
                         MTab mtab, LevelParams P, unsigned long long *dbg, hipStream_t st);

#if SZL_LAB
hipError_t launch_match3(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int ntiles, const uint16_t *link, MTab mtab, LevelParams P,
                         unsigned long long *dbg, hipStream_t st);
#endif

hipError_t launch_match(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int ntiles, const uint16_t *link,
                        MTab mtab, LevelParams P, unsigned long long *dbg, hipStream_t st) {
#if !SZL_LAB
    return launch_match2(in, segs, tiles, ntiles, link, mtab, P, dbg, st);   // k_match4 (szl_kernels_match2.hip)
#else
    // SZL_MATCH_KERNEL (lab library): 5 = bucket order and 4 = ring-fed engine (own launches: the engine calls them), 3 = chain compression
    // (szl_kernels_match3.hip, the engine passes the four-byte links in mtab), 2 = k_match4 (default), 1 = k_match below
    const int which = SZL_LABKNOB("SZL_MATCH_KERNEL", 2);
    if (mtab.link4) return launch_match3(in, segs, tiles, ntiles, link, mtab, P, dbg, st);   // the engine set the call up for k_match6
    if (which >= 2) return launch_match2(in, segs, tiles, ntiles, link, mtab, P, dbg, st);
    static std::atomic<uint64_t> attr_mask{0};
    uint64_t attr_bit = 0;
    const bool want_dbg = knob("SZL_DEBUG", 0) != 0;
    const int fth = SZL_LABKNOB("SZL_FTH", 16), vth = SZL_LABKNOB("SZL_VTH", 20);
    if (lds_attr_needed(attr_mask, attr_bit)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_match<false>, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void *)k_match<true>, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(attr_bit, std::memory_order_release);
    }
    if (ntiles > 0) {
        if (want_dbg) hipLaunchKernelGGL(k_match<true>, dim3(ntiles), dim3(B_THREADS), B_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth);
        else hipLaunchKernelGGL(k_match<false>, dim3(ntiles), dim3(B_THREADS), B_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth);
    }
    return hipGetLastError();
#endif
}

// On-demand stage B over the tiles tile_first, tile_first + tile_step, ... (count of them = nblocks).

hipError_t launch_match_lazy(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int nblocks, int tile_first, int tile_step,
                             const uint16_t *link, MTab mtab, LevelParams P, unsigned long long *dbg, hipStream_t st) {
#if !SZL_LAB
    (void)in; (void)segs; (void)tiles; (void)nblocks; (void)tile_first; (void)tile_step;
    return hipErrorNotSupported;          // (never reached: the product's engine does not select this form)
#else
    static std::atomic<uint64_t> attr_mask{0};
    uint64_t attr_bit = 0;
    static const int fth = getenv("SZL_LAZY_FTH") ? atoi(getenv("SZL_LAZY_FTH")) : 4, vth = getenv("SZL_VTH") ? atoi(getenv("SZL_VTH")) : 20;
    static const int stride = getenv("SZL_STRIDE") ? atoi(getenv("SZL_STRIDE")) : 16;
    static const bool pair = !(getenv("SZL_PAIR") && atoi(getenv("SZL_PAIR")) == 0); // walker + scout pairs (measured best on the data this form is used for)
    const int lds = B_LDS_BYTES + B_TILE / 8;
    if (lds_attr_needed(attr_mask, attr_bit)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_match_lazy<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_match_lazy<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_match_lazy<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_match_lazy<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(attr_bit, std::memory_order_release);
    }
    if (nblocks <= 0) return hipSuccess;
    if (pair) {
        if (stride == 16) hipLaunchKernelGGL((k_match_lazy<16, true>), dim3(nblocks), dim3(B_THREADS), lds, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, tile_first, tile_step);
        else hipLaunchKernelGGL((k_match_lazy<32, true>), dim3(nblocks), dim3(B_THREADS), lds, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, tile_first, tile_step);
    } else {
        if (stride == 32) hipLaunchKernelGGL((k_match_lazy<32, false>), dim3(nblocks), dim3(B_THREADS), lds, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, tile_first, tile_step);
        else hipLaunchKernelGGL((k_match_lazy<16, false>), dim3(nblocks), dim3(B_THREADS), lds, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, tile_first, tile_step);
    }
    return hipGetLastError();
#endif
}

} // namespace szl
