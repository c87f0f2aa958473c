// szl_kernels_match.hip — stage A (hash links) and stage B (match tables) for gfx950.
//
// Reference being restated (C/ = /root/reference/src/ICSharpCode.SharpZipLib/Zip/Compression/):
//   stage A  InsertString / UpdateHash / SlideWindow      C/DeflaterEngine.cs:402-462
//   stage B  FindLongestMatch                              C/DeflaterEngine.cs:474-612
// Both are *parse-independent* at levels 5-9 (DeflateSlow inserts every position),
// so they are computed for every position in parallel; the lazy parse (stage C) only looks results up.
#include <hip/hip_runtime.h>
#include "szl_internal.h"

namespace szl {

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
// handles ONLY positions that hash into them, so table updates need no inter-wave ordering and no
// barriers; every wavefront reads the whole span (L1/L2-served after the first).  Inside a
// 64-position batch, "previous lane with the same bucket" is found with a ballot loop over the
// distinct buckets present (≈4 per batch per wave).  Stale entries are aged out every 16384
// positions exactly like SlideWindow's clamp (:450-461), which keeps 16-bit entries unambiguous.
// ============================================================================================
enum : int { A_THREADS = 1024, A_WAVES = 16 };

__global__ __launch_bounds__(A_THREADS) void k_links(const uint8_t *__restrict__ in, uint64_t in_total,
                                                     const SegDev *__restrict__ segs, const uint64_t *__restrict__ bnds,
                                                     const SpanDev *__restrict__ spans, uint16_t *__restrict__ link) {
    __shared__ uint16_t head[32768];
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

    for (int64_t q0 = warm0; q0 < span.end; q0 += 64) {
        if (q0 != warm0 && ((q0 - warm0) & 16383) == 0) {
            for (int i = lane; i < 2048; i += 64) {
                uint32_t dist = (uint32_t)(q0 - myhead[i]) & 0xFFFF;
                if (dist >= 32768u || dist == 0u) myhead[i] = (uint16_t)((q0 - 40000) & 0xFFFF);
            }
        }
        const int64_t q = q0 + lane;
        while (bi < nb && (int64_t)b[bi] <= q0) bi++; // wave-uniform
        bool ins = q < span.end;
        if (bi < nb && (int64_t)b[bi] < q0 + 66) {     // a segment end is near: InsertString needs lookahead >= 3 (:780,:817)
            int j = bi;
            while (j < nb && (int64_t)b[j] <= q) j++;
            ins = ins && j < nb && (int64_t)b[j] - q >= 3;
        } else if (bi >= nb) {
            ins = false;
        }
        uint32_t idx = 0;
        bool owned = false;
        if (ins) {
            uint32_t b0, b1, b2;
            if ((uint64_t)q + 4 <= avail) {
                uint32_t w = load_u32_unaligned(d + q);
                b0 = w & 0xFF; b1 = (w >> 8) & 0xFF; b2 = (w >> 16) & 0xFF;
            } else { b0 = d[q]; b1 = d[q + 1]; b2 = d[q + 2]; }
            uint32_t h = ((b0 << 10) ^ (b1 << 5) ^ b2) & 0x7FFF; // :404,:420
            uint32_t o = (h ^ (h >> 5) ^ (h >> 10)) & 15;         // owner wavefront (bijective with h>>4)
            idx = (o << 11) | (h >> 4);
            owned = (int)o == wave;
        }
        uint32_t e_old = owned ? head[idx] : 0;
        uint64_t mm = __ballot(owned);
        int predlane = -1;
        bool islast = false;
        while (mm) {
            int l = __builtin_ctzll(mm);
            uint32_t k = __builtin_amdgcn_readlane(idx, l);
            bool mine = owned && idx == k;
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
            if (islast) head[idx] = (uint16_t)(q & 0xFFFF);
        } else if (!ins && wave == 0 && q >= span.start && q < span.end) {
            lk[q] = 0; // position never inserted (tail of a segment)
        }
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

__global__ __launch_bounds__(B_THREADS) void k_match(const uint8_t *__restrict__ in, const SegDev *__restrict__ segs,
                                                     const TileDev *__restrict__ tiles, const uint16_t *__restrict__ link,
                                                     uint2 *__restrict__ mtab, LevelParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *sdata32 = (uint32_t *)smem;                          // B_DATA_BYTES
    uint16_t *slink = (uint16_t *)(smem + B_DATA_BYTES);           // B_LINKS entries
    int *s_counter = (int *)(smem + B_DATA_BYTES + B_LINKS * 2);

    const TileDev tile = tiles[blockIdx.x];
    const SegDev seg = segs[tile.seg];
    const uint8_t *d = in + seg.buf_off;
    const uint16_t *lk = link + seg.buf_off;
    uint2 *mt = mtab + seg.buf_off;
    const int64_t t0 = tile.start;
    const int tlen = tile.len;
    const int64_t dlo = t0 - B_HIST; // buffer position of LDS data byte 0 (may be negative)
    const int64_t seg_end = seg.seg_end;

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
        ((uint32_t *)slink)[i] = w;
    }
    if (threadIdx.x == 0) *s_counter = 0;
    __syncthreads();

    auto ldsdw = [&](int i) -> uint32_t { // unaligned 32-bit read at LDS data byte i
        uint32_t w0 = sdata32[i >> 2], w1 = sdata32[(i >> 2) + 1];
        return __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)(i & 3));
    };

    const int SNAP = P.max_chain >> 2;
    enum { M_FETCH = 0, M_CHAIN = 1, M_EXT = 2, M_DONE = 3 };
    int mode = M_FETCH;
    int p = 0;        // tile-relative position
    int pl = 0;       // LDS data index of p
    int cl = 0;       // LDS data index of the current candidate
    int mincl = 0;    // lowest admissible candidate (LDS index) for chain continuation
    int best = 2, cap = 0, nice = 0, budget = 0, cnt = 0, off = 0, qoff = 0;
    uint32_t pq = 0, res2 = 0, resq = 0;

    for (;;) {
        if (mode == M_FETCH) {
            int k = atomicAdd(s_counter, 1);
            if (k >= tlen) mode = M_DONE;
            else {
                p = k;
                pl = p + B_HIST;
                const int64_t Pp = t0 + p;
                const int64_t rem = seg_end - Pp;
                res2 = 0; resq = 0;
                bool ok = rem >= MIN_MATCH && P.strategy != 2;  // :780, HuffmanOnly :786
                uint32_t l0 = ok ? slink[pl] : 0;               // hashHead (:782)
                ok = ok && l0 != 0;
                if (ok) {
                    const int64_t pabs = (int64_t)seg.abs0 + Pp;
                    const int64_t basem = base_of(pabs) - (int64_t)seg.abs0; // buffer position of window index 1 is basem
                    // first candidate: strstart - hashHead <= MAX_DIST (:788) and entry not clamped by a slide (index >= 1)
                    int64_t c = Pp - l0;
                    int64_t firstmin = Pp - MAX_DIST > basem ? Pp - MAX_DIST : basem;
                    ok = c >= firstmin;
                    if (ok) {
                        cap = rem < MAX_MATCH ? (int)rem : MAX_MATCH;         // scanMax :479
                        nice = P.nice < (int)rem ? P.nice : (int)rem;        // :485
                        int64_t minc = Pp - (MAX_DIST - 1) > basem ? Pp - (MAX_DIST - 1) : basem; // curMatch > limit (:609)
                        mincl = (int)(minc - dlo);
                        cl = (int)(c - dlo);
                        best = 2; budget = P.max_chain; cnt = 0; qoff = 0;
                        pq = ldsdw(pl);
                        mode = M_CHAIN;
                    }
                }
                if (!ok) mt[t0 + p] = make_uint2(0u, 0u);
            }
        } else if (mode != M_DONE) {
            int L = -1; // >=0: candidate fully compared with common prefix L
            if (mode == M_CHAIN) {
                uint32_t cd = ldsdw(cl + qoff);
                if (best == 2) {
                    uint32_t x = cd ^ pq;
                    int l4 = x ? (__builtin_ctz(x) >> 3) : 4;
                    if (l4 >= 3) {
                        if (l4 == 4 && cap > 4) { mode = M_EXT; off = 4; }
                        else L = l4 < cap ? l4 : cap;
                    } else L = 0;
                } else {
                    if (cd == pq) { mode = M_EXT; off = 0; } // bytes best-3..best agree: compare from the start
                    else L = 0;
                }
            } else { // M_EXT
                uint32_t x = ldsdw(cl + off) ^ ldsdw(pl + off);
                if (x == 0) {
                    off += 4;
                    if (off >= cap) L = cap;
                } else {
                    int l = off + (__builtin_ctz(x) >> 3);
                    L = l < cap ? l : cap;
                }
            }
            if (L >= 0) {
                bool finish = false;
                mode = M_CHAIN;
                if (L > best) { // :593-607
                    best = L;
                    res2 = (uint32_t)L | ((uint32_t)(pl - cl) << 16);
                    if (best >= nice) {
                        finish = true;
                        if (cnt < SNAP) resq = res2;
                    } else {
                        qoff = best - 3;
                        pq = ldsdw(pl + qoff);
                    }
                }
                if (!finish) {
                    cnt++;
                    if (cnt == SNAP) resq = res2;
                    uint32_t l = slink[cl];
                    int c2 = cl - (int)l;
                    if (l == 0 || c2 < mincl || --budget == 0) {
                        finish = true;
                        if (cnt < SNAP) resq = res2;
                    } else cl = c2;
                }
                if (finish) {
                    mt[t0 + p] = make_uint2(res2, resq);
                    mode = M_FETCH;
                }
            }
        }
        if (__all(mode == M_DONE)) break;
    }
}

void launch_links(const uint8_t *in, uint64_t in_total, const SegDev *segs, const uint64_t *bnds, const SpanDev *spans,
                  int nspans, uint16_t *link, hipStream_t st) {
    if (nspans > 0) hipLaunchKernelGGL(k_links, dim3(nspans), dim3(A_THREADS), 0, st, in, in_total, segs, bnds, spans, link);
}

int match_lds_bytes() { return B_LDS_BYTES; }

hipError_t launch_match(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int ntiles, const uint16_t *link,
                        uint2 *mtab, LevelParams P, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_match, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (ntiles > 0) hipLaunchKernelGGL(k_match, dim3(ntiles), dim3(B_THREADS), B_LDS_BYTES, st, in, segs, tiles, link, mtab, P);
    return hipGetLastError();
}

} // namespace szl
