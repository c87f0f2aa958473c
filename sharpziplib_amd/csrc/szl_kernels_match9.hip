// szl_kernels_match9.hip — stage B (FindLongestMatch for every position, C/DeflaterEngine.cs:474-612), the full search with the
// whole main loop in hand-written gfx950 assembly (text: szl_match9_asm.h; the reasons and the per-phase instruction counts of this
// engine against k_match4's are in that header and in DESIGN.md §4.2).  Same tiles, same tables, bit for bit, as k_match4.
//
// What this file adds around the text:
//   * staging with every global load of a thread in flight at once (k_match4's staging loop waited for each 4-byte load before it
//     issued the next: ~21 us of every ~240 us tile);
//   * the window at fixed LDS addresses (the text uses instruction offsets for the two array bases);
//   * (DBG) the tile's timeline: staged / first wavefront out of positions / last wavefront done, summed over the tiles.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "szl_internal.h"
#include "szl_match9_asm.h"

namespace szl {
int knob(const char *name, int dflt);

enum : int { B9_THREADS = 1024, B9_TILE = 21504, HOP_BLOCKS = 64 };
enum : int { B9_DATA_BYTES = B_HIST + B9_TILE + B_TAIL + 8, B9_LINKS = B_HIST + B9_TILE };
enum : int { B9_D = SZL9_D, B9_LB = B9_D + B9_DATA_BYTES, B9_DBG = B9_LB + 2 * B9_LINKS, B9_SCR = B9_DBG + 32 /* 64 bytes per wavefront: lane ids while walks move between contexts (the tail program) */,
             B9_LDS_BYTES = B9_SCR + B9_THREADS };
static_assert(B9_LB == SZL9_LB && B_HIST == SZL9_BHIST && B9_D % 16 == 0 && B9_LB % 16 == 0, "szl_match9_asm.h holds these numbers as literals");
static_assert(B9_LDS_BYTES <= 160 * 1024 && B9_DATA_BYTES % 16 == 0 && (2 * B9_LINKS) % 16 == 0, "the window must fit the CU's LDS");
static_assert(B_HIST - MAX_DIST == 6, "the fetch adds 6 / 7 for the two distance limits");

typedef __attribute__((address_space(3))) uint8_t lds9_u8;

__device__ __forceinline__ int64_t base_of9(int64_t s_abs) { // window base of an iteration starting at s (App. A.2; C/DeflaterEngine.cs:371,:771,:93)
    int64_t idx = s_abs + 1;
    if (idx <= 65273) return 0;
    return ((idx - 65273 + 32767) >> 15) << 15;
}
__device__ __forceinline__ uint4 ld16u(const void *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ uint32_t nolink(uint32_t w) {   // "no previous position" (0) is staged as 0xFFFF: the hop then leaves every window
    if ((w & 0xFFFFu) == 0) w |= 0xFFFFu;
    if ((w >> 16) == 0) w |= 0xFFFF0000u;
    return w;
}

__device__ __forceinline__ uint32_t short_hops(uint32_t w) { return ((w & 0xFF00u) == 0) + ((w & 0xFF000000u) == 0); }   // of two staged links: hops below 256

// Stage the window of one tile: bytes of history + tile + lookahead tail at smem + B9_D, links of history + tile at smem + B9_LB.
// 16-byte pieces; a thread first issues all its loads (4 of bytes, 7 of links), then stores.  Pieces that straddle the stream's
// start, the lookahead's end or the tile's end (links) are assembled byte by byte.
// Returns how many of the links this thread staged are hops below 256 (the tile picks the form of its text by their share).
__device__ __forceinline__ uint32_t b9_stage_window(uint8_t *smem, const uint8_t *d, const uint16_t *lk, int64_t dlo, int64_t seg_end, int64_t link_end) {
    enum : int { ND = B9_DATA_BYTES / 16, NL = 2 * B9_LINKS / 16, KD = (ND + B9_THREADS - 1) / B9_THREADS, KL = (NL + B9_THREADS - 1) / B9_THREADS };
    uint4 vd[KD], vl[KL];
    uint32_t nshort = 0;
#pragma unroll
    for (int k = 0; k < KD; k++) {
        const int i = threadIdx.x + k * B9_THREADS;
        const int64_t pos = dlo + 16 * (int64_t)i;
        vd[k] = make_uint4(0, 0, 0, 0);
        if (i < ND && pos >= 0 && pos + 16 <= seg_end) vd[k] = ld16u(d + pos);
    }
#pragma unroll
    for (int k = 0; k < KL; k++) {
        const int i = threadIdx.x + k * B9_THREADS;
        const int64_t pos = dlo + 8 * (int64_t)i;
        vl[k] = make_uint4(0, 0, 0, 0);
        if (i < NL && pos >= 0 && pos + 8 <= link_end) vl[k] = ld16u(lk + pos);
    }
#pragma unroll
    for (int k = 0; k < KD; k++) {
        const int i = threadIdx.x + k * B9_THREADS;
        const int64_t pos = dlo + 16 * (int64_t)i;
        if (i >= ND) continue;
        uint4 v = vd[k];
        if (!(pos >= 0 && pos + 16 <= seg_end)) {
            uint32_t w[4] = {0, 0, 0, 0};
            _Pragma("unroll 1") for (int b = 0; b < 16; b++) { const int64_t pk = pos + b; if (pk >= 0 && pk < seg_end) w[b >> 2] |= (uint32_t)d[pk] << (8 * (b & 3)); }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *(uint4 *)(smem + B9_D + 16 * i) = v;
    }
#pragma unroll
    for (int k = 0; k < KL; k++) {
        const int i = threadIdx.x + k * B9_THREADS;
        const int64_t pos = dlo + 8 * (int64_t)i;
        if (i >= NL) continue;
        uint4 v = vl[k];
        if (!(pos >= 0 && pos + 8 <= link_end)) {
            uint32_t w[4] = {0, 0, 0, 0};
            _Pragma("unroll 1") for (int b = 0; b < 8; b++) { const int64_t pk = pos + b; if (pk >= 0 && pk < link_end) w[b >> 1] |= (uint32_t)lk[pk] << (16 * (b & 1)); }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        v.x = nolink(v.x); v.y = nolink(v.y); v.z = nolink(v.z); v.w = nolink(v.w);
        *(uint4 *)(smem + B9_LB + 16 * i) = v;
        nshort += short_hops(v.x) + short_hops(v.y) + short_hops(v.z) + short_hops(v.w);
    }
    return nshort;
}

template <bool DBG, int FORMS>   // FORMS: 0 / 1 = the launch runs that form of the text (and holds no other), 2 = both, a tile picks
__global__ __launch_bounds__(B9_THREADS) void k_match9(const uint8_t *__restrict__ in, const SegDev *__restrict__ segs, const TileDev *__restrict__ tiles,
                                                       const uint16_t *__restrict__ link, MTab mtab, LevelParams P, unsigned long long *dbg,
                                                       int fth, int vth_in, int qkeep_in, int ktail, int slice, int vtht, int tailp, int mth, int ktail1, int vtht1, int guide, int form) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const TileDev tile = tiles[blockIdx.x];
    const SegDev seg = segs[tile.seg];
    const uint8_t *d = in + seg.buf_off;
    const uint16_t *lk = link + seg.buf_off;
    const int64_t t0 = tile.start;
    const int tlen = tile.len;
    const int64_t dlo = t0 - B_HIST;        // buffer position of LDS index 0 (may be negative)
    const int64_t seg_end = seg.look_end;   // lookahead end
    if ((uint32_t)(uintptr_t)(lds9_u8 *)smem != 0u) __builtin_trap();   // the text addresses the window through instruction offsets

    unsigned long long *s_t = (unsigned long long *)(smem + B9_DBG);     // (DBG) [0] first wavefront out of positions, [1] last wavefront done
    const unsigned long long t_start = DBG ? wall_clock64() : 0ull;
    uint32_t nshort = b9_stage_window(smem, d, lk, dlo, seg_end, t0 + tlen);
    if (FORMS == 2 && form == 2) {   // per wavefront: its threads' short hops (summed over the tile below)
#pragma unroll
        for (int o = 32; o; o >>= 1) nshort += __shfl_xor(nshort, o);
        if ((threadIdx.x & 63) == 0) *(uint32_t *)(smem + B9_SCR + (threadIdx.x & ~63u)) = nshort;
    }
    if (threadIdx.x == 0) { *(uint32_t *)smem = 0u; if (DBG) { s_t[0] = ~0ull; s_t[1] = 0ull; s_t[2] = 0ull; s_t[3] = 0ull; } }
    __syncthreads();
    const unsigned long long t_staged = DBG ? wall_clock64() : 0ull;
    // The form of the text this tile runs (szl_match9_asm.h, SZL9_V): form 1 pays one instruction per chain step for a filter byte that
    // follows the walk's failed compares, and wins where chains are dense — lines of a log, records of a table: most hops of prev[]
    // are short there (generated logs: 92 % below 256; text: 23 %) and most compares fail where the one before did.  Both forms
    // store the same tables; the choice is the tile's own (form == 2) or the caller's (SZL9_FORM 0 / 1, laboratory and tests).
    if (FORMS == 2 && form == 2) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < B9_THREADS / 64; w++) tot += *(const uint32_t *)(smem + B9_SCR + 64 * w);
        form = 2u * tot > (uint32_t)(B_HIST + tlen) ? 1 : 0;
        __syncthreads();   // (the slots are the tail program's scratch)
    }
    const int form_s = FORMS == 2 ? __builtin_amdgcn_readfirstlane(form) : FORMS;

    // window bases of the tile (see k_match4): positions from `sw` on belong to base_hi
    const int64_t base_lo = base_of9((int64_t)seg.abs0 + t0), base_hi = base_of9((int64_t)seg.abs0 + t0 + tlen - 1);
    const int64_t sw64 = base_lo == base_hi ? (int64_t)1 << 30 : (base_lo + 65273) - (int64_t)seg.abs0 - t0;
    const int sw = sw64 > (int64_t)B9_TILE ? B9_TILE : (int)sw64;
    const int bmlo = (int)(base_lo - (int64_t)seg.abs0 - dlo), bmhi = (int)(base_hi - (int64_t)seg.abs0 - dlo);
    const int64_t rem0_64 = seg_end - t0;
    const int rem0 = rem0_64 > (int64_t)(1 << 24) ? (1 << 24) : (int)rem0_64;
    const uint64_t stratm = P.strategy == 2 ? 0ull : ~0ull;             // HuffmanOnly: no search (:786)
    const int chainm2 = P.max_chain - 2, snapm1 = P.max_chain - (P.max_chain >> 2) - 1, nicel = P.nice;
    // results: entry of tile position p = LDS index pl - B_HIST is stored at byte offset 4 * pl
    const uint32_t *mt2b = mtab.m2 + seg.buf_off + t0 - B_HIST;
    const uint32_t *mtqb = mtab.mq + seg.buf_off + t0 - B_HIST;

    // the text's scalar operands must BE in SGPRs: the compiler only keeps what it can prove uniform there
    auto sgpr = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
    auto sgpr64 = [](uint64_t x) { return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32); };
    const int tlen_s = sgpr(tlen), slice_s = sgpr(slice), rem0_s = sgpr(rem0), sw_s = sgpr(sw), bmlo_s = sgpr(bmlo), bmhi_s = sgpr(bmhi), nicel_s = sgpr(nicel),
              chainm2_s = sgpr(chainm2), snapm1_s = sgpr(snapm1), vtht_s = sgpr(vtht), ktail_s = sgpr(ktail), qkeept_s = sgpr(128),
              tailp_s = sgpr(tailp), mth_s = sgpr(mth), ktail1_s = sgpr(ktail1), vtht1_s = sgpr(vtht1), wscr_s = sgpr(B9_SCR + (int)(threadIdx.x & ~63u)), guide_s = sgpr(tlen - guide);
    const uint64_t stratm_s = sgpr64(stratm), mt2b_s = sgpr64((uint64_t)(uintptr_t)mt2b), mtqb_s = sgpr64((uint64_t)(uintptr_t)mtqb);
    uint32_t vzero, vslice;   // (constants in VGPRs: 0 and the slice length; the text sets them)
#define SZL9_CTXV(X) uint32_t pl##X = 0, cb##X = 0, kk##X = 0, mincb##X = 0, left##X = 0, pb##X = 0, best##X = 2, off##X = 0, cap##X = MAX_MATCH, \
    nice##X = (uint32_t)nicel, res2##X = 0, resq##X = 0, p0##X = 0, p1##X = 0, p2##X = 0, p3##X = 0, hop##X = 0, kd##X = 0, t0##X, t1##X, t2##X, t3##X, t4##X, t5##X, t6##X, t7##X; \
    uint64_t q##X = 0, v##X = 0, w##X = 0, d##X = 0, m##X = 0, c##X = 0;
    SZL9_CTXV(A)
    SZL9_CTXV(B)
#undef SZL9_CTXV
    uint64_t sc, cm, sa, sv, texh = 0;
    uint32_t n0, n1, n2, f0, f1, f2, kt = 0;
    int wnext = 0, wend = 0, exh = 0;
    int bexit = sgpr(64 - fth), vth = sgpr(vth_in), qkeep = sgpr(qkeep_in);
#define SZL9_IO(X) [pl##X] "+&v"(pl##X), [cb##X] "+&v"(cb##X), [kk##X] "+&v"(kk##X), [mincb##X] "+&v"(mincb##X), [left##X] "+&v"(left##X), [pb##X] "+&v"(pb##X), \
    [best##X] "+&v"(best##X), [off##X] "+&v"(off##X), [cap##X] "+&v"(cap##X), [nice##X] "+&v"(nice##X), [res2##X] "+&v"(res2##X), [resq##X] "+&v"(resq##X), \
    [p0##X] "+&v"(p0##X), [p1##X] "+&v"(p1##X), [p2##X] "+&v"(p2##X), [p3##X] "+&v"(p3##X), [hop##X] "+&v"(hop##X), [kd##X] "+&v"(kd##X), \
    [t0##X] "=&v"(t0##X), [t1##X] "=&v"(t1##X), [t2##X] "=&v"(t2##X), [t3##X] "=&v"(t3##X), [t4##X] "=&v"(t4##X), [t5##X] "=&v"(t5##X), [t6##X] "=&v"(t6##X), [t7##X] "=&v"(t7##X), \
    [q##X] "+&s"(q##X), [v##X] "+&s"(v##X), [w##X] "+&s"(w##X), [d##X] "+&s"(d##X), [m##X] "+&s"(m##X), [c##X] "+&s"(c##X)
#define SZL9_RUN_TEXT() \
    asm volatile(SZL9_TEXT \
                 : SZL9_IO(A), SZL9_IO(B), \
                   [sc] "=&s"(sc), [cm] "=&s"(cm), [sa] "=&s"(sa), [sv] "=&s"(sv), [texh] "+&s"(texh), [n0] "=&s"(n0), [n1] "=&s"(n1), [n2] "=&s"(n2), \
                   [f0] "=&s"(f0), [f1] "=&s"(f1), [f2] "=&s"(f2), [kt] "+&s"(kt), [wnext] "+&s"(wnext), [wend] "+&s"(wend), [exh] "+&s"(exh), \
                   [bexit] "+&s"(bexit), [vth] "+&s"(vth), [qkeep] "+&s"(qkeep), \
                   [vzero] "=&v"(vzero), [vslice] "=&v"(vslice) \
                 : [tlen] "s"(tlen_s), [slice] "s"(slice_s), [rem0] "s"(rem0_s), [sw] "s"(sw_s), [bmlo] "s"(bmlo_s), \
                   [bmhi] "s"(bmhi_s), [nicel] "s"(nicel_s), [chainm2] "s"(chainm2_s), [snapm1] "s"(snapm1_s), [qkeept] "s"(qkeept_s), [vtht] "s"(vtht_s), \
                   [ktail] "s"(ktail_s), [stratm] "s"(stratm_s), [mt2b] "s"(mt2b_s), [mtqb] "s"(mtqb_s), \
                   [tailp] "s"(tailp_s), [mth] "s"(mth_s), [ktail1] "s"(ktail1_s), [wscr] "s"(wscr_s), [vtht1] "s"(vtht1_s), [guide] "s"(guide_s) \
                 : "vcc", "scc", "memory");
    if (FORMS == 0 || (FORMS == 2 && form_s == 0)) {
        SZL9_RUN_TEXT();
    } else {
#undef SZL9_V
#define SZL9_V 1
        SZL9_RUN_TEXT();
#undef SZL9_V
#define SZL9_V 0
    }
#undef SZL9_RUN_TEXT
#undef SZL9_IO
    if (DBG) {
        const unsigned long long t_end = wall_clock64();
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&s_t[0], (unsigned long long)texh); atomicMax(&s_t[1], t_end);
            atomicMax(&s_t[2], (unsigned long long)texh); atomicMax(&s_t[3], t_end - (unsigned long long)texh);   // last wavefront out of positions; longest tail of one wavefront
            if (dbg) atomicAdd(dbg + 44, t_end - (unsigned long long)texh);
        }
        __syncthreads();
        if (threadIdx.x == 0 && dbg) {
            atomicAdd(dbg + 40, t_staged - t_start); atomicAdd(dbg + 41, s_t[0] - t_staged); atomicAdd(dbg + 42, s_t[1] - s_t[0]); atomicAdd(dbg + 43, 1ull);
            atomicAdd(dbg + 45, s_t[2] - s_t[0]); atomicAdd(dbg + 46, s_t[3]);
        }
    }
}

// A sample of a call's prev[] hops for the choice of the text's form (Engine::pick_text_form): HOP_BLOCKS workgroups read 16 links per thread
// at evenly spaced places of link[lo, lo + n) and leave, per workgroup, how many links they saw (out[b]) and how many of them are hops
// below 256 (out[HOP_BLOCKS + b]) in mapped pinned memory.  ~5 us; the host adds them up behind the stream's synchronisation.
__global__ __launch_bounds__(256) void k_hop_stat(const uint16_t *__restrict__ link, int64_t lo, int64_t n, uint32_t *out) {
    __shared__ uint32_t s_c[2];
    if (threadIdx.x < 2) s_c[threadIdx.x] = 0;
    __syncthreads();
    const int64_t slots = (int64_t)HOP_BLOCKS * 256;
    const int64_t step = n / slots;                                   // (n >= 16 * slots: the caller's business)
    const int64_t at = (lo + ((int64_t)blockIdx.x * 256 + threadIdx.x) * step) & ~(int64_t)7;
    uint32_t seen = 0, sh = 0;
    if (at >= lo && at + 8 <= lo + n) {
        const uint4 v = ld16u(link + at);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t x = nolink(w[k]); sh += short_hops(x); }
        seen = 8;
    }
    for (int o = 32; o; o >>= 1) { seen += __shfl_xor(seen, o); sh += __shfl_xor(sh, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_c[0], seen); atomicAdd(&s_c[1], sh); }
    __syncthreads();
    if (threadIdx.x == 0) { out[blockIdx.x] = s_c[0]; out[HOP_BLOCKS + blockIdx.x] = s_c[1]; __threadfence_system(); }
}
void launch_hop_stat(const uint16_t *link, int64_t lo, int64_t n, uint32_t *out_pinned, hipStream_t st) {
    hipLaunchKernelGGL(k_hop_stat, dim3(HOP_BLOCKS), dim3(256), 0, st, link, lo, n, out_pinned);
}

static bool lds_attr_needed9(std::atomic<uint64_t> &mask, uint64_t &bit) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bit = 1ull << (dev & 63);
    return (mask.load(std::memory_order_acquire) & bit) == 0;
}

// tiles: at most B9_TILE positions each (the engine's work list for the full search, match2_tile())
hipError_t launch_match9(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int ntiles, const uint16_t *link, MTab mtab, LevelParams P,
                         unsigned long long *dbg, hipStream_t st) {
    static std::atomic<uint64_t> attr_mask{0};
    uint64_t attr_bit = 0;
    const bool want_dbg = knob("SZL_DEBUG", 0) != 0;
    // thresholds (tools/sim_match9.py counts instructions per position for any of them on the CPU)
    int fth = SZL_LABKNOB("SZL9_FTH", 16), vth = SZL_LABKNOB("SZL9_VTH", 2), qkeep = SZL_LABKNOB("SZL9_QKEEP", 64), ktail = SZL_LABKNOB("SZL9_KTAIL", 2), vtht = SZL_LABKNOB("SZL9_VTHT", 1);
    // the tail program (szl_match9_asm.h): 0 = the main loop to the end (laboratory); walks of both contexts move into one once they are
    // at most `mth` (<= 64; -1: never); iterations of the one-context walk between two looks at who left
    // (2 = the one-context loop with run-ahead and a third filter byte, round 6: ktail1 = 4 iterations of two steps between compare passes;
    // 1 = the plain one-context loop of round 4, where ktail1 = 1 measured best)
    int tailp = SZL_LABKNOB("SZL9_TAILP", 2), mth = SZL_LABKNOB("SZL9_MTH", 64), ktail1 = SZL_LABKNOB("SZL9_KTAIL1", 4), vtht1 = SZL_LABKNOB("SZL9_VTHT1", 1);
    mth = mth > 64 ? 64 : mth; ktail1 = ktail1 < 1 ? 1 : ktail1; vtht1 = vtht1 < 1 ? 1 : vtht1;
    int slice = SZL_LABKNOB("SZL_SLICE", 128);
    // hand-out near the tile's end: within `guide` positions of it a fetch takes as many positions as it has free lanes instead of a slice
    int guide = SZL_LABKNOB("SZL9_GUIDE", 8192);
    // form of the text: 0 / 1 = every tile runs that form, 2 = each tile picks (k_match9).  The engine says which (MTab::form, from a sample
    // of the call's prev[] hops: Engine::pick_text_form); SZL9_FORM (laboratory library, tests) overrides it.
    int form = SZL_LABKNOB("SZL9_FORM", -1);
    if (form < 0 || form > 2) form = mtab.form < 0 || mtab.form > 2 ? 0 : mtab.form;
    fth = fth < 1 ? 1 : (fth > 64 ? 64 : fth); vth = vth < 1 ? 1 : vth; qkeep = qkeep < 1 ? 1 : qkeep; ktail = ktail < 1 ? 1 : ktail; vtht = vtht < 1 ? 1 : vtht;
    slice = slice < 64 ? 64 : (slice > 4096 ? 4096 : slice);
    if (lds_attr_needed9(attr_mask, attr_bit)) {
        hipError_t e = hipSuccess;
        const void *fs[6] = {(const void *)k_match9<false, 0>, (const void *)k_match9<false, 1>, (const void *)k_match9<false, 2>,
                             (const void *)k_match9<true, 0>, (const void *)k_match9<true, 1>, (const void *)k_match9<true, 2>};
        for (const void *f : fs) if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, B9_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(attr_bit, std::memory_order_release);
    }
    if (ntiles > 0) {
        const dim3 g(ntiles), b(B9_THREADS);
#define SZL9_LAUNCH(D, F) hipLaunchKernelGGL((k_match9<D, F>), g, b, B9_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, qkeep, ktail, slice, vtht, tailp, mth, ktail1, vtht1, guide, 2)
        if (want_dbg) { if (form == 0) SZL9_LAUNCH(true, 0); else if (form == 1) SZL9_LAUNCH(true, 1); else SZL9_LAUNCH(true, 2); }
        else { if (form == 0) SZL9_LAUNCH(false, 0); else if (form == 1) SZL9_LAUNCH(false, 1); else SZL9_LAUNCH(false, 2); }
#undef SZL9_LAUNCH
    }
    return hipGetLastError();
}

int match9_tile() { return B9_TILE; }

} // namespace szl
