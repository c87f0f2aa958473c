// szl_kernels_match3.hip — stage B with chain compression (k_links4t + k_match6).  A LAB FORM, selected with SZL_MATCH_KERNEL=3 for
// levels 5-6: bit-exact (tests/test_gpu_stage_b_forms.py), but slower than k_match4 — see "Measured" below and DESIGN §4.2.
//
// Reference being restated: FindLongestMatch, C/DeflaterEngine.cs:474-612 — same M2 / Mq tables as k_match4.
//
// Once best_len >= 3 only a candidate that shares the position's first FOUR bytes can be strictly longer, and those candidates
// form a sub-chain of the 3-byte-hash chain.  k_links4 gives every position the distance to the previous position with the same
// four bytes (link4) and the number of hash-chain elements that hop passes (skip4, one byte: the form is used for max_chain <=
// 128, where a saturated count can only end a walk).  k_match6 walks link4 and charges skip4 against the chain budget, so the
// candidates it examines sit at the same chain indices as in the full walk: the limit test, max_chain and the quarter-budget
// snapshot (:495, :609) stay exact.  While best_len is still 2 a candidate changes the state of the walk only if its first THREE
// bytes equal the position's (a hash collision just uses up budget), so the first candidate compared is the first chain element
// with the same three bytes (e3: distance and chain index per position, found by k_links4 on its way); the sub-chain is entered
// at e3 if e3 is its first element, else from the position itself.  oracle/szl_model.c::flm_walk_k7 is this control flow on the
// CPU, checked against the plain walk on every data class.  34.1 -> 16.4 candidates per position on text.
// Price: a fourth byte per position in LDS, i.e. 8 Ki tiles where k_match4 has 21 Ki.
// Measured (profiles/r02/lab_s45_chain_compression_k_match6.log, 256 MiB of text, level 6): k_match6 58.7 ms per GiB + k_links4t 35 ms,
// against 53 ms for k_match4 with 16 Ki tiles.  Half the candidates is not half the time: a tile costs ~67 us beyond its walks
// (lab_s46_tile_length.log) and there are twice as many, and fetch / first compare / result stores per position stay.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "szl_internal.h"

namespace szl {
int knob(const char *name, int dflt);

__device__ __forceinline__ int64_t base_of3(int64_t s_abs) { // window base of an iteration starting at s (C/DeflaterEngine.cs:371,:771,:93)
    int64_t idx = s_abs + 1;
    if (idx <= 65273) return 0;
    return ((idx - 65273 + 32767) >> 15) << 15;
}
__device__ __forceinline__ uint32_t load_u32_unaligned3(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// ---- four-byte links ------------------------------------------------------------------------------------------------------
// One thread per position: follow the 3-byte-hash chain until a position with the same four bytes turns up; the first one with
// the same three bytes on the way is e3.  Arrays are addressed with arena indices g in [lo, hi); links exist from `lo` on.  A walk
// stops B_HIST positions back (stage B never follows a hop that far: its candidates are nearer than MAX_DIST < B_HIST) and after
// 255 hops (> max_chain: the element could not be reached within the chain budget).
__global__ __launch_bounds__(256) void k_links4(const uint8_t *__restrict__ d, const uint16_t *__restrict__ lk, int64_t lo, int64_t hi, int64_t n_end,
                                                uint16_t *__restrict__ link4, uint8_t *__restrict__ skip4, uint16_t *__restrict__ e3d,
                                                uint8_t *__restrict__ e3h) {
    const int64_t q = lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= hi) return;
    uint32_t l4 = 0, s4 = 0, ed = 0, eh = 0;
    if (q + 3 <= n_end && lk[q] != 0) {
        const bool has4 = q + 4 <= n_end;
        const uint32_t key = has4 ? load_u32_unaligned3(d + q) : ((uint32_t)d[q] | ((uint32_t)d[q + 1] << 8) | ((uint32_t)d[q + 2] << 16));
        int64_t c = q;
        for (uint32_t hops = 1; hops <= 255; hops++) {
            const uint32_t l = lk[c];
            if (l == 0) break;
            c -= l;
            if (q - c > B_HIST || c < lo) break;
            const uint32_t x = load_u32_unaligned3(d + c) ^ key;
            if (x & 0xFFFFFFu) continue;
            if (ed == 0) { ed = (uint32_t)(q - c); eh = hops; }
            if (!has4) break;
            if ((x >> 24) == 0) { l4 = (uint32_t)(q - c); s4 = hops; break; }
        }
    }
    link4[q] = (uint16_t)l4; skip4[q] = (uint8_t)s4; e3d[q] = (uint16_t)ed; e3h[q] = (uint8_t)eh;
}

// The same out of LDS: a workgroup stages the bytes and the 3-byte links of 16 Ki positions and of the B_HIST positions before
// them (the window of k_match4), and its lanes walk.  Walk lengths are skewed (6.7 hops on average on text, but the quarter of the
// positions without a four-byte predecessor walk to the end of their chain), so a lane that finishes takes the next position of
// its wave's slice instead of waiting for the longest walk of the wave.
enum : int { L4_THREADS = 1024, L4_TILE = 16384, L4_SLICE = 256 };
enum : int { L4_DATA_BYTES = B_HIST + L4_TILE + 16, L4_LINKS = B_HIST + L4_TILE };
enum : int { L4_LDS_BYTES = L4_DATA_BYTES + L4_LINKS * 2 + 16 };
static_assert(L4_LDS_BYTES <= 160 * 1024 && L4_DATA_BYTES % 4 == 0, "the window must fit the CU's LDS");

__global__ __launch_bounds__(L4_THREADS) void k_links4t(const uint8_t *__restrict__ d, const uint16_t *__restrict__ lk, int64_t lo, int64_t hi, int64_t n_end,
                                                       uint16_t *__restrict__ link4, uint8_t *__restrict__ skip4, uint16_t *__restrict__ e3d,
                                                       uint8_t *__restrict__ e3h, int refill) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem4[];
    uint32_t *sd32 = (uint32_t *)smem4;
    uint16_t *slink = (uint16_t *)(smem4 + L4_DATA_BYTES);
    int *s_counter = (int *)(smem4 + L4_DATA_BYTES + L4_LINKS * 2);
    const int64_t t0 = lo + (int64_t)blockIdx.x * L4_TILE;
    const int tlen = hi - t0 < (int64_t)L4_TILE ? (int)(hi - t0) : (int)L4_TILE;
    const int64_t dlo = t0 - B_HIST;
    for (int i = threadIdx.x; i < L4_DATA_BYTES / 4; i += L4_THREADS) {
        const int64_t pos = dlo + 4 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= lo && pos + 4 <= n_end) w = load_u32_unaligned3(d + pos);
        else for (int k = 0; k < 4; k++) { const int64_t pk = pos + k; if (pk >= lo && pk < n_end) w |= (uint32_t)d[pk] << (8 * k); }
        sd32[i] = w;
    }
    for (int i = threadIdx.x; i < L4_LINKS / 2; i += L4_THREADS) {
        const int64_t pos = dlo + 2 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= lo && pos < t0 + tlen) w |= lk[pos];
        if (pos + 1 >= lo && pos + 1 < t0 + tlen) w |= (uint32_t)lk[pos + 1] << 16;
        ((uint32_t *)slink)[i] = w;
    }
    if (threadIdx.x == 0) *s_counter = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const uint64_t lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int minc = lo - dlo > 0 ? (int)(lo - dlo) : 0;                       // LDS index of the first position that has a link
    int wnext = 0, wend = 0;
    bool exhausted = false, active = false;
    int p = 0, c = 0, climit = 0;
    uint32_t key = 0, hops = 0, ed = 0, eh = 0, l4 = 0, s4 = 0;
    bool has4 = false;
    for (;;) {
        const uint64_t act = __ballot(active);
        if (!exhausted && 64 - __builtin_popcountll(act) >= refill) {
            const uint64_t idle = ~act;
            if (wnext >= wend) {
                int base = 0;
                if (lane == 0) base = atomicAdd(s_counter, (int)L4_SLICE);
                base = __builtin_amdgcn_readfirstlane(base);
                wnext = base < tlen ? base : tlen;
                wend = base + L4_SLICE < tlen ? base + L4_SLICE : tlen;
                if (wnext >= wend) exhausted = true;
            }
            if (!exhausted) {
                const int rank = __builtin_popcountll(idle & lanemask_lt);
                if (!active && wnext + rank < wend) {
                    p = wnext + rank;
                    const int pl = p + B_HIST;
                    const int64_t q = t0 + p;
                    hops = 0; ed = 0; eh = 0; l4 = 0; s4 = 0;
                    c = pl;
                    climit = pl - B_HIST > minc ? pl - B_HIST : minc;
                    has4 = q + 4 <= n_end;
                    key = __builtin_amdgcn_alignbyte(sd32[(pl >> 2) + 1], sd32[pl >> 2], (uint32_t)pl & 3u);
                    if (!has4) key &= 0xFFFFFFu;
                    active = q + 3 <= n_end;
                    if (!active) { link4[q] = 0; skip4[q] = 0; e3d[q] = 0; e3h[q] = 0; }
                }
                const int ni = __builtin_popcountll(idle);
                wnext = wnext + ni < wend ? wnext + ni : wend;
            }
        } else if (act == 0) break;
#pragma unroll 1
        for (int it = 0; it < 4; it++) {
            if (active) {
                const uint32_t l = slink[c];
                bool fin = l == 0;
                c -= (int)l;
                fin = fin || c < climit;
                if (!fin) {
                    hops++;
                    const uint32_t x = __builtin_amdgcn_alignbyte(sd32[(c >> 2) + 1], sd32[c >> 2], (uint32_t)c & 3u) ^ key;
                    if ((x & 0xFFFFFFu) == 0) {
                        if (ed == 0) { ed = (uint32_t)(p + B_HIST - c); eh = hops; }
                        if (!has4) fin = true;
                        else if ((x >> 24) == 0) { l4 = (uint32_t)(p + B_HIST - c); s4 = hops; fin = true; }
                    }
                    fin = fin || hops >= 255;
                }
                if (fin) {
                    const int64_t q = t0 + p;
                    link4[q] = (uint16_t)l4; skip4[q] = (uint8_t)s4; e3d[q] = (uint16_t)ed; e3h[q] = (uint8_t)eh;
                    active = false;
                }
            }
        }
    }
}

enum : int { B6_THREADS = 1024, B6_TILE = 8192 };
enum : int { B6_DATA_BYTES = B_HIST + B6_TILE + B_TAIL + 8, B6_LINKS = B_HIST + B6_TILE };
enum : int { B6_LDS_BYTES = B6_DATA_BYTES + B6_LINKS * 2 + B6_LINKS + 16 };
static_assert(B6_LDS_BYTES <= 160 * 1024, "the window must fit the CU's LDS");
static_assert(B6_DATA_BYTES % 4 == 0 && B6_LINKS % 4 == 0, "dword staging");

typedef __attribute__((address_space(3))) uint8_t lds6_u8;

// ---- the engine (see szl_kernels_match2.hip for the scheme: two walks per lane, lane masks in SGPRs, exec-mask loops) -------
// QUICK step: hop (link4) and hop count (skip4) of the chain position, the candidate's two filter bytes
#define SZL6_Q_ISSUE(X) \
    "v_lshl_add_u32 %[t0" #X "], %[cl" #X "], 1, %[lbase]\n\t" \
    "v_add_u32 %[t3" #X "], %[sbase], %[cl" #X "]\n\t" \
    "v_add3_u32 %[t1" #X "], %[cl" #X "], %[best" #X "], %[dbm1]\n\t" \
    "ds_read_u16 %[t0" #X "], %[t0" #X "]\n\t" \
    "ds_read_u8 %[t3" #X "], %[t3" #X "]\n\t" \
    "ds_read_u8 %[t2" #X "], %[t1" #X "]\n\t" \
    "ds_read_u8 %[t1" #X "], %[t1" #X "] offset:1\n\t"
#define SZL6_Q_FINISH_(X, TAIL) \
    "v_lshl_or_b32 %[t1" #X "], %[t1" #X "], 8, %[t2" #X "]\n\t" \
    "v_cmpx_ne_u32 vcc, %[pb" #X "], %[t1" #X "]\n\t"                       /* scan_end / scan_end1 differ: stay in the walk */ \
    "v_sub_u32 %[cl" #X "], %[cl" #X "], %[t0" #X "]\n\t"                  /* next element of the sub-chain, in place */ \
    "v_cmpx_ge_i32 vcc, %[cl" #X "], %[mincl" #X "]\n\t"                    /* curMatch > limit (:609); "no link" is 0xFFFF */ \
    "v_sub_co_u32 %[left" #X "], vcc, %[left" #X "], %[t3" #X "]\n\t"       /* the hash-chain elements the hop passed */ \
    TAIL
#define SZL6_Q_FINISH(X) SZL6_Q_FINISH_(X, "s_andn2_b64 exec, exec, vcc\n\t" "s_mov_b64 %[m" #X "], exec\n\t")
#define SZL6_Q_FINISH_LAST(X) SZL6_Q_FINISH_(X, "s_andn2_b64 %[m" #X "], exec, vcc\n\t")
// leavers: filter passed -> VERIFY of that candidate (vcl), else the walk is over
#define SZL6_Q_CLASSIFY(X) \
    "s_andn2_b64 exec, %[q" #X "], %[m" #X "]\n\t" \
    "v_cmp_eq_u32 vcc, %[pb" #X "], %[t1" #X "]\n\t" \
    "v_mov_b32 %[off" #X "], 0\n\t" \
    "v_mov_b32 %[vcl" #X "], %[cl" #X "]\n\t" \
    "s_or_b64 %[v" #X "], %[v" #X "], vcc\n\t" \
    "s_andn2_b64 %[sc], exec, vcc\n\t" \
    "s_or_b64 %[d" #X "], %[d" #X "], %[sc]\n\t" \
    "s_mov_b64 %[q" #X "], %[m" #X "]\n\t"
#define SZL6_V_ISSUE(X) \
    "v_add3_u32 %[t0" #X "], %[vcl" #X "], %[off" #X "], %[dbase]\n\t" \
    "v_add3_u32 %[t1" #X "], %[p" #X "], %[off" #X "], %[pbase]\n\t" \
    "v_and_b32 %[t4" #X "], -4, %[t0" #X "]\n\t" \
    "v_and_b32 %[t5" #X "], -4, %[t1" #X "]\n\t" \
    "ds_read_b32 %[t2" #X "], %[t4" #X "]\n\t" \
    "ds_read_b32 %[t3" #X "], %[t4" #X "] offset:4\n\t" \
    "ds_read_b32 %[t4" #X "], %[t4" #X "] offset:8\n\t" \
    "ds_read_b32 %[t6" #X "], %[t5" #X "]\n\t" \
    "ds_read_b32 %[t7" #X "], %[t5" #X "] offset:4\n\t" \
    "ds_read_b32 %[t5" #X "], %[t5" #X "] offset:8\n\t"
#define SZL6_V_FINISH(X) \
    "v_alignbyte_b32 %[t2" #X "], %[t3" #X "], %[t2" #X "], %[t0" #X "]\n\t" \
    "v_alignbyte_b32 %[t3" #X "], %[t4" #X "], %[t3" #X "], %[t0" #X "]\n\t" \
    "v_alignbyte_b32 %[t6" #X "], %[t7" #X "], %[t6" #X "], %[t1" #X "]\n\t" \
    "v_alignbyte_b32 %[t7" #X "], %[t5" #X "], %[t7" #X "], %[t1" #X "]\n\t" \
    "v_xor_b32 %[t2" #X "], %[t2" #X "], %[t6" #X "]\n\t" \
    "v_xor_b32 %[t0" #X "], %[t3" #X "], %[t7" #X "]\n\t" \
    "v_ffbl_b32 %[t2" #X "], %[t2" #X "]\n\t" \
    "v_ffbl_b32 %[t0" #X "], %[t0" #X "]\n\t" \
    "v_or_b32 %[t0" #X "], 32, %[t0" #X "]\n\t" \
    "v_min_u32 %[t2" #X "], %[t2" #X "], %[t0" #X "]\n\t" \
    "v_lshrrev_b32 %[t2" #X "], 3, %[t2" #X "]\n\t" \
    "v_min_u32 %[t2" #X "], 8, %[t2" #X "]\n\t" \
    "v_add_u32 %[off" #X "], %[off" #X "], %[t2" #X "]\n\t" \
    "v_cmpx_eq_u32 vcc, 8, %[t2" #X "]\n\t" \
    "v_cmpx_lt_i32 vcc, %[off" #X "], %[cap" #X "]\n\t" \
    "s_mov_b64 %[m" #X "], exec\n\t"
// :593-609 for the compares that finished (v & ~m); the others hop along the sub-chain from their chain position cl.  After the
// first compare of a walk that is the position itself when e3 is not on the sub-chain: the hop then counts from the position, so
// the budget e3 was charged (kadj) goes back first.
#define SZL6_V_COMPLETE(X) \
    "s_andn2_b64 %[cm], %[v" #X "], %[m" #X "]\n\t" \
    "s_mov_b64 exec, %[cm]\n\t" \
    "v_lshl_add_u32 %[t0" #X "], %[cl" #X "], 1, %[lbase]\n\t" \
    "v_add_u32 %[t4" #X "], %[sbase], %[cl" #X "]\n\t" \
    "ds_read_u16 %[t0" #X "], %[t0" #X "]\n\t" \
    "ds_read_u8 %[t4" #X "], %[t4" #X "]\n\t" \
    "v_min_i32 %[t2" #X "], %[off" #X "], %[cap" #X "]\n\t"                /* L */ \
    "v_cmp_gt_i32 %[sc], %[t2" #X "], %[best" #X "]\n\t" \
    "s_mov_b64 exec, %[sc]\n\t" \
    "v_mov_b32 %[best" #X "], %[t2" #X "]\n\t" \
    "v_sub_u32 %[t1" #X "], %[p" #X "], %[vcl" #X "]\n\t" \
    "v_add_u32 %[t1" #X "], %[bhist], %[t1" #X "]\n\t"                       /* distance = (p + B_HIST) - vcl */ \
    "v_lshl_or_b32 %[res2" #X "], %[t1" #X "], 16, %[t2" #X "]\n\t" \
    "v_cmp_ge_i32 vcc, %[left" #X "], %[snap]\n\t"                           /* seen by the quarter-budget walk too (:495) */ \
    "v_cndmask_b32 %[resq" #X "], %[resq" #X "], %[res2" #X "], vcc\n\t" \
    "v_add3_u32 %[t1" #X "], %[p" #X "], %[t2" #X "], %[pbm1]\n\t" \
    "ds_read_u8 %[t3" #X "], %[t1" #X "]\n\t" \
    "ds_read_u8 %[t1" #X "], %[t1" #X "] offset:1\n\t" \
    "v_cmp_ge_i32 %[sc], %[t2" #X "], %[nice" #X "]\n\t"                   /* >= niceLength: stop (:603) */ \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_lshl_or_b32 %[pb" #X "], %[t1" #X "], 8, %[t3" #X "]\n\t" \
    "s_mov_b64 exec, %[cm]\n\t" \
    "s_or_b64 %[d" #X "], %[d" #X "], %[sc]\n\t" \
    "s_andn2_b64 exec, %[cm], %[sc]\n\t" \
    "v_sub_u32 %[t0" #X "], %[cl" #X "], %[t0" #X "]\n\t"                  /* next element of the sub-chain */ \
    "v_cmp_lt_i32 vcc, %[t0" #X "], %[mincl" #X "]\n\t" \
    "s_or_b64 %[d" #X "], %[d" #X "], vcc\n\t" \
    "s_andn2_b64 exec, exec, vcc\n\t" \
    "v_add_u32 %[left" #X "], %[left" #X "], %[kadj" #X "]\n\t" \
    "v_mov_b32 %[kadj" #X "], 0\n\t" \
    "v_sub_co_u32 %[left" #X "], vcc, %[left" #X "], %[t4" #X "]\n\t" \
    "s_or_b64 %[d" #X "], %[d" #X "], vcc\n\t" \
    "s_andn2_b64 exec, exec, vcc\n\t" \
    "v_mov_b32 %[cl" #X "], %[t0" #X "]\n\t" \
    "s_or_b64 %[q" #X "], %[q" #X "], exec\n\t" \
    "s_mov_b64 %[v" #X "], %[m" #X "]\n\t"

struct WalkCtx6 {
    int p;            // tile position being searched
    int cl;           // chain position: LDS index of the sub-chain element the walk stands on (or of p itself, see FETCH)
    int vcl;          // candidate being / last compared
    int best, left, off, mincl, cap, nice;   // as WalkCtx; left = max_chain - (hash-chain index of the last candidate)
    int kadj;         // chain index of e3 while cl is still the position itself, else 0
    uint32_t pb, res2, resq;
};

template <bool DBG>
__global__ __launch_bounds__(B6_THREADS) void k_match6(const uint8_t *__restrict__ in, const SegDev *__restrict__ segs,
                                                       const TileDev *__restrict__ tiles, const uint16_t *__restrict__ link,
                                                       MTab mtab, LevelParams P, unsigned long long *dbg, int fth, int vth, int qkeep, int vkeep, int slice) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem6[];
    const TileDev tile = tiles[blockIdx.x];
    const SegDev seg = segs[tile.seg];
    uint32_t *sdata32 = (uint32_t *)smem6;
    uint16_t *slink4 = (uint16_t *)(smem6 + B6_DATA_BYTES);
    uint8_t *sskip = smem6 + B6_DATA_BYTES + B6_LINKS * 2;
    int *s_counter = (int *)(smem6 + B6_DATA_BYTES + B6_LINKS * 3);
    const uint8_t *d = in + seg.buf_off;
    const uint16_t *lk = link + seg.buf_off;
    const uint16_t *lk4 = mtab.link4 + seg.buf_off;
    const uint8_t *sk4 = mtab.skip4 + seg.buf_off;
    const uint16_t *e3dg = mtab.e3d + seg.buf_off;
    const uint8_t *e3hg = mtab.e3h + seg.buf_off;
    uint32_t *__restrict__ mt2 = mtab.m2 + seg.buf_off;
    uint32_t *__restrict__ mtq = mtab.mq + seg.buf_off;
    const int64_t t0 = tile.start;
    const int tlen = tile.len;
    const int64_t dlo = t0 - B_HIST;
    const int64_t seg_end = seg.look_end;

    // ---- stage the window: bytes, four-byte links ("none" = 0xFFFF) and hop counts of history + tile
    for (int i = threadIdx.x; i < B6_DATA_BYTES / 4; i += B6_THREADS) {
        const int64_t pos = dlo + 4 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos + 4 <= seg_end) w = load_u32_unaligned3(d + pos);
        else for (int k = 0; k < 4; k++) { const int64_t pk = pos + k; if (pk >= 0 && pk < seg_end) w |= (uint32_t)d[pk] << (8 * k); }
        sdata32[i] = w;
    }
    for (int i = threadIdx.x; i < B6_LINKS / 2; i += B6_THREADS) {
        const int64_t pos = dlo + 2 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos < t0 + tlen) w |= lk4[pos];
        if (pos + 1 >= 0 && pos + 1 < t0 + tlen) w |= (uint32_t)lk4[pos + 1] << 16;
        if ((w & 0xFFFFu) == 0) w |= 0xFFFFu;
        if ((w >> 16) == 0) w |= 0xFFFF0000u;
        ((uint32_t *)slink4)[i] = w;
    }
    for (int i = threadIdx.x; i < B6_LINKS / 4; i += B6_THREADS) {
        const int64_t pos = dlo + 4 * (int64_t)i;
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) { const int64_t pk = pos + k; if (pk >= 0 && pk < t0 + tlen) w |= (uint32_t)sk4[pk] << (8 * k); }
        ((uint32_t *)sskip)[i] = w;
    }
    if (threadIdx.x == 0) *s_counter = 0;
    __syncthreads();

    const uint8_t *sdata8 = smem6;
    const uint32_t dbase = (uint32_t)(uintptr_t)(lds6_u8 *)smem6;
    const uint32_t lbase = dbase + (uint32_t)B6_DATA_BYTES;
    const uint32_t sbase = lbase + (uint32_t)B6_LINKS * 2u;
    const uint32_t pbase = dbase + (uint32_t)B_HIST;

    const int64_t base_lo = base_of3((int64_t)seg.abs0 + t0), base_hi = base_of3((int64_t)seg.abs0 + t0 + tlen - 1);
    const int64_t sw64 = base_lo == base_hi ? (int64_t)1 << 30 : (base_lo + 65273) - (int64_t)seg.abs0 - t0;
    const int sw = sw64 > (int64_t)B6_TILE ? B6_TILE : (int)sw64;
    const int basem_lo = (int)(base_lo - (int64_t)seg.abs0 - dlo), basem_hi = (int)(base_hi - (int64_t)seg.abs0 - dlo);
    const int64_t rem0_64 = seg_end - t0;
    const int rem0 = rem0_64 > (int64_t)(1 << 24) ? (1 << 24) : (int)rem0_64;
    const int lane = threadIdx.x & 63;
    const uint64_t lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int SNAPLEFT = P.max_chain - (P.max_chain >> 2);
    const int bhist = B_HIST;
    int wnext = 0, wend = 0;
    bool exhausted = false;

    WalkCtx6 A, B;
    A.p = 0; A.cl = B_HIST; A.vcl = B_HIST; A.best = 2; A.left = 0; A.off = 0; A.mincl = 0; A.cap = MAX_MATCH; A.nice = P.nice; A.pb = 0; A.res2 = 0; A.resq = 0; A.kadj = 0;
    B = A;
    uint64_t qA = 0, vA = 0, dA = 0, qB = 0, vB = 0, dB = 0;

    // FETCH: the first candidate compared is e3 (see the head of the file); hashHead itself only decides whether there is a search.
    auto fetch = [&](WalkCtx6 &C, uint64_t &q, uint64_t &v, uint64_t &dm) {
        if (__builtin_amdgcn_inverse_ballot_w64(dm)) { const uint32_t e = mt_pack(C.res2, C.resq); mt2[t0 + C.p] = e; if (e >> 25) mtq[t0 + C.p] = C.resq; }
        dm = 0;
        if (exhausted) return;
        const uint64_t idle = ~(q | v);
        const int ni = __builtin_popcountll(idle);
        if (ni == 0) return;
        if (wnext >= wend) {
            int base = 0;
            if (lane == 0) base = atomicAdd(s_counter, slice);
            base = __builtin_amdgcn_readfirstlane(base);
            wnext = base < tlen ? base : tlen;
            wend = base + slice < tlen ? base + slice : tlen;
            if (wnext >= wend) { exhausted = true; return; }
        }
        const int rank = __builtin_popcountll(idle & lanemask_lt);
        bool toverify = false;
        if (__builtin_amdgcn_inverse_ballot_w64(idle) && wnext + rank < wend) {
            const int p = wnext + rank;
            C.p = p;
            const int rem = rem0 - p;
            C.res2 = 0; C.resq = 0;
            bool ok = rem >= MIN_MATCH && P.strategy != 2;              // :780, HuffmanOnly :786
            if (ok) {
                const int pl = p + B_HIST;
                const int l3 = (int)lk[t0 + p];                          // hashHead (:782)
                const int ed = (int)e3dg[t0 + p], eh = (int)e3hg[t0 + p];
                const int basem = p >= sw ? basem_hi : basem_lo;
                const int firstmin = pl - MAX_DIST > basem ? pl - MAX_DIST : basem; // strstart - hashHead <= MAX_DIST (:788)
                const int mincl = pl - (MAX_DIST - 1) > basem ? pl - (MAX_DIST - 1) : basem; // curMatch > limit (:609)
                const int c = pl - ed;
                ok = l3 != 0 && pl - l3 >= firstmin && ed != 0 && eh <= P.max_chain && (ed == l3 || c >= mincl);
                if (ok) {
                    C.vcl = c;
                    C.mincl = mincl;
                    C.cap = rem < MAX_MATCH ? rem : MAX_MATCH;
                    C.nice = rem < P.nice ? rem : P.nice;
                    C.best = 2;
                    C.left = P.max_chain - eh;
                    if ((int)slink4[pl] == ed) { C.cl = c; C.kadj = 0; } else { C.cl = pl; C.kadj = eh; }
                    C.pb = ((uint32_t)sdata8[pl + 2] << 8) | sdata8[pl + 1];
                    C.off = 0;
                    toverify = true;
                }
            }
            if (!ok) mt2[t0 + p] = 0u;
        }
        v |= __ballot(toverify);
        wnext = wnext + ni < wend ? wnext + ni : wend;
    };

    for (;;) {
        fetch(A, qA, vA, dA);
        fetch(B, qB, vB, dB);
        if ((qA | vA | qB | vB) == 0) { if (exhausted) break; else continue; }
        const uint32_t busy_exit = exhausted ? 0u : (uint32_t)(128 - fth);
        uint32_t t0A, t1A, t2A, t3A, t4A, t5A, t6A, t7A, t0B, t1B, t2B, t3B, t4B, t5B, t6B, t7B;
        uint64_t mA, mB, sc, cm, sv;
        uint32_t n0, n1, n2;
        asm volatile(
            "s_mov_b64 %[sv], exec\n"
            "10:\n\t"                                           // ---- census
            "s_or_b64 %[sc], %[qA], %[vA]\n\t"
            "s_bcnt1_i32_b64 %[n0], %[sc]\n\t"
            "s_or_b64 %[sc], %[qB], %[vB]\n\t"
            "s_bcnt1_i32_b64 %[n1], %[sc]\n\t"
            "s_add_u32 %[n0], %[n0], %[n1]\n\t"               // contexts walking or comparing
            "s_cmp_eq_u32 %[n0], 0\n\t"
            "s_cbranch_scc1 19f\n\t"
            "s_cmp_le_u32 %[n0], %[bexit]\n\t"
            "s_cbranch_scc1 19f\n\t"
            "s_bcnt1_i32_b64 %[n1], %[vA]\n\t"
            "s_bcnt1_i32_b64 %[n2], %[vB]\n\t"
            "s_add_u32 %[n1], %[n1], %[n2]\n\t"               // contexts waiting for VERIFY
            "s_cmp_ge_u32 %[n1], %[vth]\n\t"
            "s_cbranch_scc1 14f\n\t"
            "s_cmp_eq_u32 %[n0], %[n1]\n\t"                    // nothing in QUICK
            "s_cbranch_scc1 14f\n"
            // ---- QUICK phase
            "s_mov_b64 %[mA], %[qA]\n\t"
            "s_mov_b64 %[mB], %[qB]\n"
            "11:\n\t"
            "s_mov_b64 exec, %[mA]\n\t"
            SZL6_Q_ISSUE(A)
            "s_mov_b64 exec, %[mB]\n\t"
            SZL6_Q_ISSUE(B)
            "s_mov_b64 exec, %[mA]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            SZL6_Q_FINISH(A)
            SZL6_Q_ISSUE(A)
            "s_mov_b64 exec, %[mB]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            SZL6_Q_FINISH(B)
            SZL6_Q_ISSUE(B)
            "s_mov_b64 exec, %[mA]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            SZL6_Q_FINISH_LAST(A)
            "s_mov_b64 exec, %[mB]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            SZL6_Q_FINISH_LAST(B)
            "s_bcnt1_i32_b64 %[n0], %[mA]\n\t"
            "s_bcnt1_i32_b64 %[n1], %[mB]\n\t"
            "s_add_u32 %[n0], %[n0], %[n1]\n\t"
            "s_cmp_ge_u32 %[n0], %[qkeep]\n\t"
            "s_cbranch_scc1 11b\n\t"
            SZL6_Q_CLASSIFY(A)
            SZL6_Q_CLASSIFY(B)
            "s_branch 10b\n"
            // ---- VERIFY phase
            "14:\n\t"
            "s_mov_b64 %[mA], %[vA]\n\t"
            "s_mov_b64 %[mB], %[vB]\n\t"
            "s_cmp_eq_u64 %[vB], 0\n\t"
            "s_cbranch_scc1 16f\n\t"
            "s_cmp_eq_u64 %[vA], 0\n\t"
            "s_cbranch_scc1 17f\n"
            "15:\n\t"
            "s_mov_b64 exec, %[mA]\n\t"
            SZL6_V_ISSUE(A)
            "s_mov_b64 exec, %[mB]\n\t"
            SZL6_V_ISSUE(B)
            "s_mov_b64 exec, %[mA]\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            SZL6_V_FINISH(A)
            "s_mov_b64 exec, %[mB]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            SZL6_V_FINISH(B)
            "s_bcnt1_i32_b64 %[n0], %[mA]\n\t"
            "s_bcnt1_i32_b64 %[n1], %[mB]\n\t"
            "s_add_u32 %[n0], %[n0], %[n1]\n\t"
            "s_cmp_ge_u32 %[n0], %[vkeep]\n\t"
            "s_cbranch_scc1 15b\n\t"
            SZL6_V_COMPLETE(A)
            SZL6_V_COMPLETE(B)
            "s_branch 10b\n"
            "16:\n\t"
            "s_mov_b64 exec, %[mA]\n\t"
            SZL6_V_ISSUE(A)
            "s_waitcnt lgkmcnt(0)\n\t"
            SZL6_V_FINISH(A)
            "s_bcnt1_i32_b64 %[n0], %[mA]\n\t"
            "s_cmp_ge_u32 %[n0], %[vkeep]\n\t"
            "s_cbranch_scc1 16b\n\t"
            SZL6_V_COMPLETE(A)
            "s_branch 10b\n"
            "17:\n\t"
            "s_mov_b64 exec, %[mB]\n\t"
            SZL6_V_ISSUE(B)
            "s_waitcnt lgkmcnt(0)\n\t"
            SZL6_V_FINISH(B)
            "s_bcnt1_i32_b64 %[n0], %[mB]\n\t"
            "s_cmp_ge_u32 %[n0], %[vkeep]\n\t"
            "s_cbranch_scc1 17b\n\t"
            SZL6_V_COMPLETE(B)
            "s_branch 10b\n"
            "19:\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            : [pA] "+&v"(A.p), [clA] "+&v"(A.cl), [vclA] "+&v"(A.vcl), [bestA] "+&v"(A.best), [leftA] "+&v"(A.left), [kadjA] "+&v"(A.kadj), [offA] "+&v"(A.off), [pbA] "+&v"(A.pb),
              [res2A] "+&v"(A.res2), [resqA] "+&v"(A.resq),
              [pB] "+&v"(B.p), [clB] "+&v"(B.cl), [vclB] "+&v"(B.vcl), [bestB] "+&v"(B.best), [leftB] "+&v"(B.left), [kadjB] "+&v"(B.kadj), [offB] "+&v"(B.off), [pbB] "+&v"(B.pb),
              [res2B] "+&v"(B.res2), [resqB] "+&v"(B.resq),
              [qA] "+&s"(qA), [vA] "+&s"(vA), [dA] "+&s"(dA), [qB] "+&s"(qB), [vB] "+&s"(vB), [dB] "+&s"(dB),
              [t0A] "=&v"(t0A), [t1A] "=&v"(t1A), [t2A] "=&v"(t2A), [t3A] "=&v"(t3A), [t4A] "=&v"(t4A), [t5A] "=&v"(t5A), [t6A] "=&v"(t6A), [t7A] "=&v"(t7A),
              [t0B] "=&v"(t0B), [t1B] "=&v"(t1B), [t2B] "=&v"(t2B), [t3B] "=&v"(t3B), [t4B] "=&v"(t4B), [t5B] "=&v"(t5B), [t6B] "=&v"(t6B), [t7B] "=&v"(t7B),
              [mA] "=&s"(mA), [mB] "=&s"(mB), [sc] "=&s"(sc), [cm] "=&s"(cm), [sv] "=&s"(sv), [n0] "=&s"(n0), [n1] "=&s"(n1), [n2] "=&s"(n2)
            : [minclA] "v"(A.mincl), [capA] "v"(A.cap), [niceA] "v"(A.nice), [minclB] "v"(B.mincl), [capB] "v"(B.cap), [niceB] "v"(B.nice),
              [lbase] "s"(lbase), [sbase] "s"(sbase), [dbase] "s"(dbase), [pbase] "s"(pbase), [dbm1] "s"(dbase - 1u), [pbm1] "s"(pbase - 1u), [bhist] "s"(bhist),
              [snap] "s"(SNAPLEFT), [bexit] "s"(busy_exit), [vth] "s"(vth), [qkeep] "s"(qkeep), [vkeep] "s"(vkeep)
            : "vcc", "scc", "memory");
    }
}

static bool lds_attr_needed3(std::atomic<uint64_t> &mask, uint64_t &bit) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bit = 1ull << (dev & 63);
    return (mask.load(std::memory_order_acquire) & bit) == 0;
}

int match3_tile() { return B6_TILE; }

void launch_links4(const uint8_t *in, const uint16_t *link, int64_t lo, int64_t hi, int64_t n_end, uint16_t *link4, uint8_t *skip4, uint16_t *e3d,
                   uint8_t *e3h, hipStream_t st) {
    if (hi <= lo) return;
    if (SZL_LABKNOB("SZL_LINKS4", 1) == 0) {   // (lab) the walk out of global memory: 90 ms per GiB on text
        hipLaunchKernelGGL(k_links4, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, st, in, link, lo, hi, n_end, link4, skip4, e3d, e3h);
        return;
    }
    static std::atomic<uint64_t> attr_mask{0};
    uint64_t attr_bit = 0;
    if (lds_attr_needed3(attr_mask, attr_bit)) {
        if (hipFuncSetAttribute((const void *)k_links4t, hipFuncAttributeMaxDynamicSharedMemorySize, L4_LDS_BYTES) != hipSuccess) return;
        attr_mask.fetch_or(attr_bit, std::memory_order_release);
    }
    int refill = SZL_LABKNOB("SZL_L4_REFILL", 16);
    refill = refill < 1 ? 1 : (refill > 64 ? 64 : refill);
    hipLaunchKernelGGL(k_links4t, dim3((unsigned)((hi - lo + L4_TILE - 1) / L4_TILE)), dim3(L4_THREADS), L4_LDS_BYTES, st, in, link, lo, hi, n_end, link4, skip4,
                       e3d, e3h, refill);
}

hipError_t launch_match3(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int ntiles, const uint16_t *link, MTab mtab, LevelParams P,
                         unsigned long long *dbg, hipStream_t st) {
    static std::atomic<uint64_t> attr_mask{0};
    uint64_t attr_bit = 0;
    const bool want_dbg = knob("SZL_DEBUG", 0) != 0;
    int fth = SZL_LABKNOB("SZL_FTH6", 32), vth = SZL_LABKNOB("SZL_VTH6", 2), qkeep = SZL_LABKNOB("SZL_QKEEP6", 64), vkeep = SZL_LABKNOB("SZL_VKEEP6", 2), slice = SZL_LABKNOB("SZL_SLICE6", 128);
    if (lds_attr_needed3(attr_mask, attr_bit)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_match6<false>, hipFuncAttributeMaxDynamicSharedMemorySize, B6_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_match6<true>, hipFuncAttributeMaxDynamicSharedMemorySize, B6_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(attr_bit, std::memory_order_release);
    }
    fth = fth < 1 ? 1 : (fth > 128 ? 128 : fth); vth = vth < 1 ? 1 : vth; qkeep = qkeep < 1 ? 1 : qkeep; vkeep = vkeep < 1 ? 1 : vkeep;
    slice = slice < 64 ? 64 : (slice > 4096 ? 4096 : slice);
    if (ntiles > 0) {
        const dim3 g(ntiles), b(B6_THREADS);
        if (want_dbg) hipLaunchKernelGGL((k_match6<true>), g, b, B6_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, qkeep, vkeep, slice);
        else hipLaunchKernelGGL((k_match6<false>), g, b, B6_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, qkeep, vkeep, slice);
    }
    return hipGetLastError();
}

} // namespace szl
