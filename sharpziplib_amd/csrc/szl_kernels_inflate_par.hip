// szl_kernels_inflate_par.hip — helpers that let ONE deflate member be inflated by thousands of wavefronts
// (SURVEY §7.5 stage 1 / §8e "single-member gzip": the pugz / rapidgzip two-pass shape).
//
// Reference being restated: the decode itself is C/Inflater.cs:283-552 as in szl_kernels_inflate.hip (k_inflate's chunk modes);
// what is new here has no counterpart in the reference, which decodes a member strictly front to back:
//   k_find_blocks   for every chunk of compressed bytes, the first bit offset at which a complete, valid DYNAMIC block header
//                   parses (C/InflaterDynHeader.cs:42-120 with strict completeness checks) — a place to start decoding from;
//   k_resolve_wins  front to back over the chunks: the last 32 KiB of output of each chunk with its "byte i of the preceding
//                   32 KiB" symbols replaced — the window (CS/OutputWindow.cs) the next chunk's back-references read;
//   k_convert       all symbols -> bytes.
// A wrong guess of k_find_blocks cannot produce wrong output: the count pass must end every chunk exactly on the next chunk's
// start bit, otherwise the member is decoded by the ordinary single-wavefront path.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <atomic>
#include "szl_internal.h"
#include "szl_inflate.h"

namespace szl {

__constant__ uint8_t c_meta_order2[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15}; // C/InflaterDynHeader.cs:23-24

// 64 stream bits starting at bit position bp (LSB-first), zero beyond the end of the input
__device__ __forceinline__ uint64_t bits_at(const uint8_t *in, uint64_t in_len, uint64_t bp) {
    const uint64_t b = bp >> 3;
    uint64_t v = 0;
    if (b + 9 <= in_len) {
        uint64_t lo; uint8_t hi;
        __builtin_memcpy(&lo, in + b, 8);
        hi = in[b + 8];
        const uint32_t sh = (uint32_t)(bp & 7);
        v = sh ? (lo >> sh) | ((uint64_t)hi << (64 - sh)) : lo;
    } else {
        for (int k = 0; k < 9; k++) {
            const uint64_t by = b + k < in_len ? in[b + k] : 0;
            const int pos = 8 * k - (int)(bp & 7);
            if (pos >= 0 && pos < 64) v |= by << pos;
            else if (pos < 0) v |= by >> (-pos);
        }
    }
    return v;
}

// Full parse of a dynamic block header, ONE LANE PER CANDIDATE (round 5).  True iff the header at bit position p of the member is
// complete and consistent.  Through round 4 one lane parsed one candidate at a time out of an LDS stage the whole wavefront filled for
// it (260 dwords per candidate), and a chunk sees ~170 candidates that pass the cheap tests before its first block header: the parses,
// not the scan, were most of the finder's 7.5 ms for a 1 GiB member.  The parse needs no array but the 128-entry table of the
// code-length code (one byte per entry, this lane's 128 bytes of LDS): a repeat code needs the PREVIOUS length only, completeness is two
// running Kraft sums, the end-of-block symbol's length is noted as the run passes index 256 — so up to 64 candidates are parsed at once,
// each lane reading its own header through bits_at (a header is at most 563 bytes: L1 / L2 resident after the first touch).
__device__ bool header_ok_lane(const uint8_t *in, uint64_t in_len, uint64_t p, uint8_t *mlut /* 128, this lane's */) {
    const uint64_t left = in_len * 8 - p;                         // input bits from the header's first bit to the end of the member
    uint32_t w = (uint32_t)bits_at(in, in_len, p);
    if ((w & 7) != 4) return false;                               // BFINAL = 0, BTYPE = 2 (the final block is left to its predecessor's decode)
    const uint32_t nl = ((w >> 3) & 31) + 257, nd = ((w >> 8) & 31) + 1, nm = ((w >> 13) & 15) + 4;
    if (nl > 286 || nd > 30) return false;                        // C/InflaterDynHeader.cs:50-52
    uint32_t rel = 17;                                            // bits consumed
    const uint64_t mw = bits_at(in, in_len, p + 17);
    constexpr int ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};   // C/InflaterDynHeader.cs:23-24
    uint32_t ml[19];
#pragma unroll
    for (int i = 0; i < 19; i++) ml[i] = 0;
    int kraft = 0;
#pragma unroll
    for (int i = 0; i < 19; i++) {                                // (unrolled: ml[] is indexed by constants only and stays in registers)
        const uint32_t l = (uint32_t)i < nm ? (uint32_t)(mw >> (3 * i)) & 7u : 0u;
        ml[ORDER[i]] = l;
        if (l) kraft += 128 >> l;
    }
    if (kraft != 128) return false;                               // a complete code-length code (what every encoder writes)
    rel += 3 * nm;
    if ((uint64_t)rel + 64 > left) return false;
    for (int i = 0; i < 128; i += 4) *(uint32_t *)(mlut + i) = 0;
    int code = 0;
    for (int l = 1; l < 8; l++) {
#pragma unroll
        for (int i = 0; i < 19; i++) {
            if (ml[i] != (uint32_t)l) continue;
            const uint32_t rev = (__builtin_bitreverse32((uint32_t)code++) >> (32 - l)) & ((1u << l) - 1);
            for (uint32_t j = rev; j < 128; j += (1u << l)) mlut[j] = (uint8_t)((i << 3) | l);      // (symbol < 19, length 1..7: never 0)
        }
        code <<= 1;
    }
    uint32_t idx = 0, prev = 0, len256 = 0, lastlit = 0;
    uint32_t pk = 0, prep = 0, eq = 0, zl = 0;                       // the symbol before: 0 none, 1 a length, 2 / 3 / 4 = 16 / 17 / 18 and its count; equal lengths / zeros written out in a row
    const uint32_t total = nl + nd;
    int kl = 0, kd = 0, ndist = 0;                                // Kraft sums in units of 2^-15
    // (the symbols come out of a 64-bit window that is fetched again when fewer than 16 bits of it are left — a symbol takes at most
    // 7 + 7 — so the parse of a true header, 316 symbols, is ~35 dependent global loads instead of 316)
    uint64_t buf = 0;
    int have = 0;
    while (idx < total) {
        if ((uint64_t)rel + 16 > left) return false;
        if (have < 16) { buf = bits_at(in, in_len, p + rel); have = 64; }
        w = (uint32_t)buf;
        const uint32_t e = mlut[w & 127];
        if (e == 0) return false;
        const uint32_t sl = e & 7, sym = e >> 3;
        w >>= sl;
        uint32_t used = sl;
        uint32_t rep = 1, val = sym;
        if (sym == 16) { if (idx == 0) return false; val = prev; rep = 3 + (w & 3); used += 2; }               // :83
        else if (sym == 17) { val = 0; rep = 3 + (w & 7); used += 3; }
        else if (sym == 18) { val = 0; rep = 11 + (w & 127); used += 7; }
        rel += used; buf >>= used; have -= (int)used;
        if (idx + rep > total) return false;                       // :106
        // The code lengths as an encoder writes them (round 6, last third: three false headers in thirty calls of the laboratory scans —
        // complete codes, trimmed counts and all, each costing the job in front of it a second chunk, i.e. the pass half its time again).
        // DeflaterHuffman.CalcBLFreq / WriteTree (C/DeflaterHuffman.cs:349-473) and zlib's scan_tree / send_tree emit a run of equal lengths
        // greedily and each of the two code sets on its own: a zero run is [18 x 138]* and then nothing, one or two 0s, one 17 or one 18;
        // a run of v != 0 is v, [16 x 6]*, and then nothing, one or two v, or one 16.  So: no repeat across the literal/distance border and
        // no 16 at the start of either set; nothing zero-valued behind a 17, behind an 18 shorter than 138 or behind two 0s, and no 17 / 18
        // behind a 0; behind a 16 shorter than 6 nothing of the same length; a 16 only behind the run's first length or a full 16; at most
        // three equal lengths written out.  Accidents break these within a few symbols; an encoder that writes its runs otherwise (zopfli,
        // 7-zip's optimiser) merely loses its blocks as starts.
        if (idx < nl && idx + rep > nl) return false;
        if (idx == 0 || idx == nl) { pk = 0; eq = 0; zl = 0; if (sym == 16) return false; }
        if (sym == 16) {
            if (prev == 0) return false;
            if (pk == 2 && prep < 6) return false;
            if (pk == 1 && eq >= 2) return false;
            pk = 2; prep = rep;
        } else if (sym >= 17) {
            if (pk != 0 && prev == 0 && !(pk == 4 && prep == 138)) return false;
            pk = sym == 17 ? 3 : 4; prep = rep;
        } else if (sym == 0) {
            if (pk == 3 || (pk == 4 && prep < 138)) return false;
            zl = (pk == 1 && prev == 0) ? zl + 1 : 1;
            if (zl > 2) return false;
            pk = 1;
        } else {
            if (pk == 2 && prep < 6 && sym == prev) return false;
            if (pk == 1 && sym == prev) eq++;
            else eq = (pk == 2 && sym == prev) ? 2 : 1;                 // (behind full 16s at most two more of the same length)
            if (eq > 3) return false;
            pk = 1;
        }
        if (val) {
            const uint32_t in_lit = idx >= nl ? 0u : (idx + rep <= nl ? rep : nl - idx);       // how many of the run are literal/length codes
            kl += (int)in_lit * (32768 >> val);
            kd += (int)(rep - in_lit) * (32768 >> val);
            ndist += (int)(rep - in_lit);
        }
        if (idx <= 256u && 256u < idx + rep) len256 = val;
        if (idx < nl && nl <= idx + rep) lastlit = val;             // (the length of symbol nl - 1)
        idx += rep;
        prev = val;
        if (kl > 32768 || kd > 32768) return false;               // over-subscribed
    }
    if (len256 == 0) return false;                                 // :113
    // HLIT / HDIST / HCLEN exist to drop trailing zero lengths, and every encoder uses them so (DeflaterHuffman.BuildCodes: numCodes =
    // maxCode + 1, :803-810 for the code-length code; zlib's max_code): a header whose last literal/length, distance or code-length entry
    // is zero is one of the accidents — complete codes in random bits, about one per gigabyte, each of which costs the job in front of it a
    // second chunk (a zlib-made 1 GiB member: 52 instead of 31 ms).  An encoder that does not trim merely loses that block as a start.
    if (nl > 257 && lastlit == 0) return false;
    if (nd > 1 && prev == 0) return false;
    if (nm > 4 && ((uint32_t)(mw >> (3 * (nm - 1))) & 7u) == 0) return false;
    if (kl != 32768) return false;                                 // complete literal/length code
    if (!(kd == 32768 || ndist <= 1)) return false;               // complete distance code, or the one-code / no-code tree of zlib
    return true;
}

// A candidate start of the other kind (round 5): a non-final STORED block whose header is followed, LEN bytes on, by another block
// header that holds — a stored one with its own LEN / NLEN pair and room for its bytes, or a dynamic one that parses.  Data that does not
// compress is runs of such blocks (16 KiB each from a level 5-9 encoder, 64 KiB from level 0) with no dynamic header for megabytes: a
// member of them had no candidate starts at all and was decoded by ONE wavefront (256 MiB: 123 ms, after 907), a stretch of them inside a
// member was one job however long.  The cheap test is in the scan (three zero type bits, zero padding to the byte, NLEN == ~LEN: one bit
// position in 2^19 + padding of random data passes); this is the second look, a lane per survivor.
__device__ bool stored_ok_lane(const uint8_t *in, uint64_t in_len, uint64_t p, uint8_t *mlut) {
    const uint64_t b = (p + 3 + 7) >> 3;                           // LEN, NLEN at bytes b .. b+3 (C/Inflater.cs:490-512)
    if (b + 4 > in_len) return false;
    const uint32_t len = (uint32_t)in[b] | ((uint32_t)in[b + 1] << 8);
    const uint64_t nb = b + 4 + len;                               // the next block's header byte
    if (nb + 5 > in_len) return false;                             // (the member's last blocks are left to their predecessor's decode)
    const uint32_t t = in[nb] & 7u;
    if ((t >> 1) == 0) {
        const uint32_t len2 = (uint32_t)in[nb + 1] | ((uint32_t)in[nb + 2] << 8), nlen2 = (uint32_t)in[nb + 3] | ((uint32_t)in[nb + 4] << 8);
        return nlen2 == (len2 ^ 0xFFFFu) && nb + 5 + len2 <= in_len;
    }
    if (t == 4) return header_ok_lane(in, in_len, 8 * nb, mlut);
    return false;
}

// One wavefront per finder job: first valid dynamic header at a bit offset in [lo_bit, hi_bit) of the job's member.  The scan tests
// 64 bit positions per step with the cheap tests and notes the positions that pass; every FIND_FLUSH steps (or with 64 of them
// noted) they are parsed, a lane each, and the lowest one that holds is the answer.
enum : int { FIND_FLUSH = 1536, FIND_CAP = 256 /* 63 left by the step before + a Kraft round + a step's stored candidates + what is noted, drained before a flush */, FIND_STAGE_BYTES = 256, FIND_STAGE_DW = FIND_STAGE_BYTES / 4 + 6 };
__global__ __launch_bounds__(64) void k_find_blocks(const uint8_t *__restrict__ in_base, const FindJob *__restrict__ fjobs, uint32_t njobs,
                                                    uint64_t *__restrict__ start_bit) {
    __shared__ __attribute__((aligned(16))) uint8_t s_mlut[64][128];
    __shared__ uint64_t s_cand[FIND_CAP];
    __shared__ uint16_t s_pre[128];                                // stage offsets that passed the first test, in order
    __shared__ uint32_t s_stage[FIND_STAGE_DW];
    __shared__ uint8_t s_k9[512];                                  // Kraft weight (units of 2^-7) of three 3-bit code lengths at once
    if (blockIdx.x >= njobs) return;
    for (int i = threadIdx.x; i < 512; i += 64) {
        int k = 0;
        for (int f = 0; f < 3; f++) { const int l = (i >> (3 * f)) & 7; if (l) k += 128 >> l; }
        s_k9[i] = (uint8_t)k;
    }
    const FindJob fj = fjobs[blockIdx.x];
    const uint8_t *in = in_base + fj.in_off;
    const uint64_t in_len = fj.in_len;
    const int lane = threadIdx.x;
    const uint64_t lo = fj.lo_bit;
    uint64_t hi = fj.hi_bit;
    if (hi + 128 > in_len * 8) hi = in_len * 8 > 128 ? in_len * 8 - 128 : 0;
    uint64_t found = ~0ull;
    int ncand = 0, since = 0, npre = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // (s_k9)
    auto flush = [&]() {                                           // parse what has been noted; the lowest position that holds is the answer
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint64_t POS = ~(1ull << 63);
        for (int off = 0; off < ncand; off += 64) {                // (all of them: the stored-block candidates are noted ahead of the scan)
            const bool mine = off + lane < ncand;
            const uint64_t p = mine ? s_cand[off + lane] : 0;
            const uint64_t pp = p & POS;                           // (bit 63: a stored-block candidate)
            const bool ok = mine && ((p >> 63) ? stored_ok_lane(in, in_len, pp, s_mlut[lane]) : header_ok_lane(in, in_len, pp, s_mlut[lane]));
            for (uint64_t okm = __ballot(ok); okm; okm &= okm - 1) {
                const uint64_t v = s_cand[off + __builtin_ctzll(okm)];
                if (found == ~0ull || (v & POS) < (found & POS)) found = v;      // (bit 63 stays: the host wants to know the kind)
            }
        }
        ncand = 0; since = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    // The scan reads the member ONCE, 256 bytes at a time with one coalesced dword per lane into LDS, and every bit position's cheap
    // tests take their 17 + 57 bits out of that stage (round 5: each lane used to fetch its own two overlapping 9-byte windows from
    // global memory per step, and a step waited out a memory round trip: 4096 steps of a 32 KiB chunk without a header took 4.6 ms,
    // as much as the symbol pass of a small piece).
    for (uint64_t blk = lo >> 3; blk * 8 < hi && found == ~0ull; blk += FIND_STAGE_BYTES) {
        for (int i = lane; i < FIND_STAGE_DW; i += 64) {
            const uint64_t q = blk + 4ull * (uint64_t)i;
            uint32_t v = 0;
            if (q + 4 <= in_len) __builtin_memcpy(&v, in + q, 4);
            else for (int kb = 0; kb < 4; kb++) if (q + kb < in_len) v |= (uint32_t)in[q + kb] << (8 * kb);
            s_stage[i] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint64_t blk_bit = blk * 8;
        // Round 6: the two tests of a bit position are run apart.  One position in nine passes the first (type bits, HLIT, HDIST) and
        // some lane did in nearly every step, so every step paid for the second — the Kraft sum of the code-length code's lengths, 64-bit
        // shifts and seven table look-ups, most of the scan's instructions — with seven lanes in it.  Now a step notes the offsets that pass
        // the first test (s_pre, in order), and the second runs on 64 noted offsets at a time: once in nine steps.
        auto kraft_round = [&](int take) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            bool cand = false;
            uint32_t off = 0;
            if (lane < take) {
                off = s_pre[lane];
                const uint32_t dw = off >> 5, sh = off & 31u;
                const uint32_t w = __builtin_amdgcn_alignbit(s_stage[dw + 1], s_stage[dw], sh);
                // the code-length code's own lengths (nm x 3 bits) must be a complete code: their Kraft sum, three lengths per table look-up
                const uint32_t nm = ((w >> 13) & 15) + 4;
                const uint32_t o2 = off + 17u, d2 = o2 >> 5, s2 = o2 & 31u;          // the 57 bits of the lengths
                uint64_t m = (uint64_t)__builtin_amdgcn_alignbit(s_stage[d2 + 1], s_stage[d2], s2) |
                             ((uint64_t)__builtin_amdgcn_alignbit(s_stage[d2 + 2], s_stage[d2 + 1], s2) << 32);
                m &= (1ull << (3 * nm)) - 1ull;                                      // (nm <= 19: at most 57 bits)
                const int kraft = (int)s_k9[(uint32_t)m & 511u] + (int)s_k9[(uint32_t)(m >> 9) & 511u] + (int)s_k9[(uint32_t)(m >> 18) & 511u] +
                                  (int)s_k9[(uint32_t)(m >> 27) & 511u] + (int)s_k9[(uint32_t)(m >> 36) & 511u] + (int)s_k9[(uint32_t)(m >> 45) & 511u] +
                                  (int)s_k9[(uint32_t)(m >> 54) & 511u];
                cand = kraft == 128;
            }
            const uint64_t mm = __ballot(cand);
            if (mm) {
                if (cand) s_cand[ncand + __builtin_popcountll(mm & ((1ull << lane) - 1ull))] = blk_bit + (uint64_t)off;
                ncand += __builtin_popcountll(mm);
            }
            const int rest = npre - take;                                            // (< 64: a round runs as soon as 64 are noted)
            const uint16_t mv = lane < rest ? s_pre[lane + take] : (uint16_t)0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < rest) s_pre[lane] = mv;
            npre = rest;
        };
        for (uint32_t step = 0; step < FIND_STAGE_BYTES * 8 / 64 && found == ~0ull; step++) {
            const uint64_t p = blk_bit + 64ull * step + (uint64_t)lane;
            const uint32_t off = 64u * step + (uint32_t)lane;                      // bit offset inside the stage
            bool pre = false;
            if (p >= lo && p < hi) {
                const uint32_t dw = off >> 5, sh = off & 31u;
                const uint32_t w = __builtin_amdgcn_alignbit(s_stage[dw + 1], s_stage[dw], sh);
                pre = (w & 7) == 4 && ((w >> 3) & 31) <= 29 && ((w >> 8) & 31) <= 29;
            }
            const uint64_t pm = __ballot(pre);
            if (pm) {
                if (pre) s_pre[npre + __builtin_popcountll(pm & ((1ull << lane) - 1ull))] = (uint16_t)off;
                npre += __builtin_popcountll(pm);
                if (npre >= 64) kraft_round(64);
            }
            // ... or a stored block's (stored_ok_lane).  Its LEN / NLEN pair is byte aligned, so the cheap test goes by BYTE, a lane each,
            // once per eight steps (per bit position it was 16 of the scan's 127 instructions per step, +20 % on a member of text): lane l
            // looks at the byte hb that would hold the end of the header's zero bits — NLEN == ~LEN behind it (one byte in 2^16 of random
            // data), its top three bits or more zero — and names the first bit of that run of zeros, reaching up to two bits into the
            // byte before: what a scan by bit position finds first.
            if ((step & 7u) == 0) {
                const uint32_t bo = 8u * step + (uint32_t)lane;                         // byte offset inside the stage (<= 255)
                const uint32_t o1 = bo + 1u, d1 = o1 >> 2, s1 = (o1 & 3u) << 3;
                const uint32_t ln = __builtin_amdgcn_alignbit(s_stage[d1 + 1], s_stage[d1], s1);
                bool sc = (ln >> 16) == ((ln & 0xFFFFu) ^ 0xFFFFu);
                uint64_t sp = 0;
                if (sc) {
                    const uint32_t hbyte = (s_stage[bo >> 2] >> ((bo & 3u) << 3)) & 0xFFu;
                    uint32_t z = hbyte ? (uint32_t)__builtin_clz(hbyte) - 24u : 8u;    // zero bits at the top of the byte
                    sc = z >= 3u;
                    if (z == 8u && bo > 0) {
                        const uint32_t pb = (s_stage[(bo - 1u) >> 2] >> (((bo - 1u) & 3u) << 3)) & 0xFFu;
                        z += (pb & 0x80u) ? 0u : ((pb & 0x40u) ? 1u : 2u);
                    }
                    sp = blk_bit + 8ull * (uint64_t)bo + 8ull - (uint64_t)z;
                    sc = sc && sp >= lo && sp < hi;
                }
                const uint64_t sm = __ballot(sc);
                if (sm) {
                    if (sc) s_cand[ncand + __builtin_popcountll(sm & ((1ull << lane) - 1ull))] = sp | (1ull << 63);
                    ncand += __builtin_popcountll(sm);
                }
            }
            if (++since >= FIND_FLUSH || ncand >= 64) {
                while (npre) kraft_round(npre < 64 ? npre : 64);   // (what is noted lies in front of a stored candidate of this step)
                flush();
            }
        }
        while (npre) kraft_round(npre < 64 ? npre : 64);           // (offsets are the stage's)
        if (ncand >= 64) flush();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (found == ~0ull && ncand) flush();
    if (lane == 0) start_bit[blockIdx.x] = found;
}

// The windows W_j = the 32 KiB of output that end where job j's output ends, as bytes: wins[(j + 1) * 32768 ..] = W_j ;
// wins[0 .. 32768) = W_-1 = zeros (a fresh OutputWindow, CS/OutputWindow.cs:22).  W_j follows from job j's symbols and W_j-1 — a
// chain over the jobs.  One workgroup walking thousands of jobs front to back (a global round trip and a barrier each) was 10 of
// the 77 ms of a 1 GiB member with the rest of the device idle, so the chain is cut into groups of RES_GROUP jobs:
//   k_resolve_rel    every group walks its jobs with 16-bit entries — a byte, or 0x8000 | i = "byte i of the window in front of
//                    the group" — and leaves the map M_g of its last job;
//   k_resolve_chain  one workgroup per member: E_0 = zeros, E_g+1 = M_g applied to E_g (a few dozen steps);
//   k_resolve_wins   every group walks its jobs again from its true entry window E_g and writes the W_j.
// Members of at most RES_GROUP jobs are one group: k_resolve_wins alone, from zeros.
template <typename T>
__device__ __forceinline__ void resolve_step(const uint16_t *__restrict__ sym, uint64_t jb, uint64_t len, const T *prev, T *next, int tid) {
    if (len >= 32768) {   // the usual job: its last 32 KiB are all its own symbols — 32 independent loads per thread in flight
        const uint16_t *src = sym + jb + (len - 32768) + tid;
        uint32_t sv[32];
#pragma unroll
        for (int k = 0; k < 32; k++) sv[k] = src[1024 * k];
#pragma unroll
        for (int k = 0; k < 32; k++) next[tid + 1024 * k] = sv[k] < 0x8000u ? (T)sv[k] : prev[sv[k] & 0x7FFF];
    } else {
        for (int t = tid; t < 32768; t += 1024) {
            T b;
            if ((uint64_t)(32768 - t) <= len) {                // position o1 - 32768 + t lies inside the job's output
                const uint32_t sv = sym[jb + (len - 32768 + (uint64_t)t)];
                b = sv < 0x8000u ? (T)sv : prev[sv & 0x7FFF];
            } else b = prev[(uint64_t)t + len];                // still the previous window, shifted by this job's output
            next[t] = b;
        }
    }
}

__global__ __launch_bounds__(1024) void k_resolve_rel(const uint16_t *__restrict__ sym, const uint64_t *__restrict__ ooff_all,
                                                      const uint64_t *__restrict__ jbase_all, const ParMember *__restrict__ mem,
                                                      const ResGroup *__restrict__ groups, uint16_t *__restrict__ gmaps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_rel[];
    uint16_t *const wa = (uint16_t *)smem_rel, *const wb = (uint16_t *)smem_rel + 32768;
    const ResGroup g = groups[blockIdx.x];
    const ParMember m = mem[g.mem];
    const uint64_t *out_off = ooff_all + m.ooff_off;
    const uint64_t *jbase = jbase_all + m.ooff_off;
    const int tid = threadIdx.x;
    for (int i = tid; i < 32768; i += 1024) wa[i] = g.first ? (uint16_t)(m.win0 ? m.win0[i] : 0) : (uint16_t)(0x8000u | (uint32_t)i);
    __syncthreads();
    int cur = 0;
    for (uint32_t j = g.j0; j < g.j1; j++) {
        resolve_step<uint16_t>(sym, jbase[j], out_off[j + 1] - out_off[j], cur ? wb : wa, cur ? wa : wb, tid);
        __syncthreads();
        cur ^= 1;
    }
    uint4 *d4 = (uint4 *)(gmaps + g.slot * 32768);
    const uint4 *w4 = (const uint4 *)(cur ? wb : wa);
    for (int i = tid; i < 4096; i += 1024) d4[i] = w4[i];
}

// grid = members; groups of member blockIdx.x = [gfirst[b], gfirst[b + 1]); ewins[slot] = entry window of the group
__global__ __launch_bounds__(1024) void k_resolve_chain(const ResGroup *__restrict__ groups, const uint32_t *__restrict__ gfirst,
                                                        const uint16_t *__restrict__ gmaps, uint8_t *__restrict__ ewins, const ParMember *__restrict__ mem) {
    __shared__ __attribute__((aligned(16))) uint8_t s_e[2][32768];
    const uint32_t g0 = gfirst[blockIdx.x], g1 = gfirst[blockIdx.x + 1];
    const int tid = threadIdx.x;
    const uint8_t *win0 = mem[blockIdx.x].win0;    // what lies in front of the member's first job (a member's start: zeros)
    for (int i = tid; i < 32768; i += 1024) s_e[0][i] = win0 ? win0[i] : (uint8_t)0;
    __syncthreads();
    int cur = 0;
    for (uint32_t g = g0; g < g1; g++) {
        const uint64_t slot = groups[g].slot;
        {
            const uint4 *w4 = (const uint4 *)s_e[cur];
            uint4 *d4 = (uint4 *)(ewins + slot * 32768);
            d4[tid] = w4[tid]; d4[tid + 1024] = w4[tid + 1024];
        }
        if (g + 1 == g1) break;
        const uint16_t *mp = gmaps + slot * 32768;
        uint32_t mv[32];
#pragma unroll
        for (int k = 0; k < 32; k++) mv[k] = mp[tid + 1024 * k];
#pragma unroll
        for (int k = 0; k < 32; k++) s_e[cur ^ 1][tid + 1024 * k] = mv[k] < 0x8000u ? (uint8_t)mv[k] : s_e[cur][mv[k] & 0x7FFF];
        __syncthreads();
        cur ^= 1;
    }
}

// grid = groups; ewins == nullptr: every member is one group (entry window = zeros)
__global__ __launch_bounds__(1024) void k_resolve_wins(const uint16_t *__restrict__ sym, const uint64_t *__restrict__ ooff_all,
                                                       const uint64_t *__restrict__ jbase_all, uint8_t *__restrict__ wins_all,
                                                       const ParMember *__restrict__ mem, const ResGroup *__restrict__ groups,
                                                       const uint8_t *__restrict__ ewins) {
    __shared__ __attribute__((aligned(16))) uint8_t s_w[2][32768];
    const ResGroup g = groups[blockIdx.x];
    const ParMember m = mem[g.mem];
    const uint64_t *out_off = ooff_all + m.ooff_off;
    const uint64_t *jbase = jbase_all + m.ooff_off;   // first symbol of job j in the staging (the jobs' regions need not be adjacent)
    uint8_t *wins = wins_all + m.win_off;
    const int tid = threadIdx.x;
    if (ewins && !g.first) {
        const uint4 *e4 = (const uint4 *)(ewins + g.slot * 32768);
        uint4 *w4 = (uint4 *)s_w[0];
        w4[tid] = e4[tid]; w4[tid + 1024] = e4[tid + 1024];
    } else for (int i = tid; i < 32768; i += 1024) s_w[0][i] = m.win0 ? m.win0[i] : (uint8_t)0;
    if (g.first) for (int i = tid; i < 32768; i += 1024) wins[i] = m.win0 ? m.win0[i] : (uint8_t)0;
    __syncthreads();
    int cur = 0;
    for (uint32_t j = g.j0; j < g.j1; j++) {
        uint8_t *next = s_w[cur ^ 1];
        resolve_step<uint8_t>(sym, jbase[j], out_off[j + 1] - out_off[j], s_w[cur], next, tid);
        __syncthreads();
        {   // the finished window goes out 32 bytes per thread (the next job only reads it: no second barrier needed)
            const uint4 *w4 = (const uint4 *)next;
            uint4 *d4 = (uint4 *)(wins + (uint64_t)(j + 1) * 32768);
            d4[tid] = w4[tid];
            d4[tid + 1024] = w4[tid + 1024];
        }
        cur ^= 1;
    }
}

// symbols -> bytes; one workgroup per CONV_BLOCK (64 KiB) of a member's output (blk0 = the member's first workgroup), 16 KiB at a time.
// A thread takes 16 consecutive bytes per step: two 16-byte loads of symbols (their address is only 2-byte aligned: the hardware's
// unaligned access mode), the low bytes packed (v_perm_b32), one 16-byte store — it used to be one 2-byte load and one 1-byte store per
// thread and step, 0.86 TB/s of the chip's 8 (3.75 ms per GiB of output).  Nearly every 16 KiB lie inside ONE job (a job's output is
// hundreds of KiB): a thread then issues all eight of its loads before it looks at any of them.
// Symbols that name the window in front of their job (0x8000 | i) — a tenth of a text member's, but some in most groups of sixteen —
// were looked up with a byte load from global memory each: 1.0 of the kernel's 1.6 ms per GiB went there, whether the loads waited for
// each other or were issued together (round 6, profiles/r06/inflate_finder_convert.log: the texture path takes a wave's 64 byte addresses as 64
// accesses).  Now the first 16 KiB of a workgroup that meet such a symbol copy the job's 32 KiB window into LDS, and the look-ups are
// ds_read_u8.  The 16 KiB that straddle two jobs or end the member go symbol by symbol as before.
__global__ __launch_bounds__(256) void k_convert(const uint16_t *__restrict__ sym, const uint64_t *__restrict__ ooff_all,
                                                 const uint64_t *__restrict__ jbase_all, const uint8_t *__restrict__ wins_all,
                                                 uint8_t *__restrict__ out_base, const ParMember *__restrict__ mem, uint32_t nmem) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[32768];
    uint32_t a = 0, z = nmem;                                      // last member with blk0 <= blockIdx.x
    while (z - a > 1) { const uint32_t mid = (a + z) >> 1; if (mem[mid].blk0 <= blockIdx.x) a = mid; else z = mid; }
    const ParMember m = mem[a];
    const uint64_t *out_off = ooff_all + m.ooff_off;
    const uint64_t *jbase = jbase_all + m.ooff_off;
    const uint8_t *wins = wins_all + m.win_off;
    uint8_t *out = out_base + m.out_off;
    const uint32_t njobs = m.njobs;
    const uint64_t total = m.total;
    const uint64_t c0 = (uint64_t)(blockIdx.x - m.blk0) * CONV_BLOCK;
    if (c0 >= total) return;
    const uint64_t c1 = c0 + CONV_BLOCK < total ? c0 + CONV_BLOCK : total;
    uint32_t lo = 0, hi = njobs;                                   // last job with out_off <= c0
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (out_off[mid] <= c0) lo = mid; else hi = mid; }
    uint32_t staged = ~0u;                                         // the job whose window is in s_win
    for (uint64_t b0 = c0; b0 < c1; b0 += 16384) {                 // (every condition on b0, b1, lo is the workgroup's: the barriers below are met by all)
        const uint64_t b1 = b0 + 16384 < c1 ? b0 + 16384 : c1;
        while (lo + 1 < njobs && out_off[lo + 1] <= b0) lo++;
        if (b1 - b0 == 16384 && (lo + 1 >= njobs || out_off[lo + 1] >= b1)) {
            const uint16_t *sp0 = sym + jbase[lo] + (b0 - out_off[lo]) + 16u * threadIdx.x;
            uint4 s[8];
#pragma unroll
            for (int k = 0; k < 4; k++) { __builtin_memcpy(&s[2 * k], sp0 + 4096 * k, 16); __builtin_memcpy(&s[2 * k + 1], sp0 + 4096 * k + 8, 16); }
            uint32_t fl = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) fl |= s[k].x | s[k].y | s[k].z | s[k].w;
            fl &= 0x80008000u;
            if (__syncthreads_or((int)fl) && staged != lo) {   // (the vote is a barrier: nobody still reads the window staged before)
                const uint4 *src = (const uint4 *)(wins + (uint64_t)lo * 32768);
                uint4 v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = src[threadIdx.x + 256 * k];
#pragma unroll
                for (int k = 0; k < 8; k++) ((uint4 *)s_win)[threadIdx.x + 256 * k] = v[k];
                staged = lo;
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t w[8] = {s[2 * k].x, s[2 * k].y, s[2 * k].z, s[2 * k].w, s[2 * k + 1].x, s[2 * k + 1].y, s[2 * k + 1].z, s[2 * k + 1].w};
                if (((w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) & 0x80008000u) != 0) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        uint32_t sa = w[i] & 0xFFFFu, sb = w[i] >> 16;
                        if (sa & 0x8000u) sa = s_win[sa & 0x7FFFu];
                        if (sb & 0x8000u) sb = s_win[sb & 0x7FFFu];
                        w[i] = sa | (sb << 16);
                    }
                }
                uint4 o;   // the low bytes of eight pairs of symbols (v_perm_b32: bytes 0 and 2 of the second operand, then of the first)
                o.x = __builtin_amdgcn_perm(w[1], w[0], 0x06040200u); o.y = __builtin_amdgcn_perm(w[3], w[2], 0x06040200u);
                o.z = __builtin_amdgcn_perm(w[5], w[4], 0x06040200u); o.w = __builtin_amdgcn_perm(w[7], w[6], 0x06040200u);
                __builtin_memcpy(out + b0 + 16u * threadIdx.x + 4096u * k, &o, 16);
            }
            continue;
        }
        uint32_t j = lo;
        for (uint64_t q = b0 + 16ull * threadIdx.x; q < b1; q += 16ull * 256) {
            for (uint64_t r = q; r < (q + 16 < b1 ? q + 16 : b1); r++) {
                while (j + 1 < njobs && out_off[j + 1] <= r) j++;
                const uint32_t sv = sym[jbase[j] + (r - out_off[j])];
                out[r] = sv < 0x8000u ? (uint8_t)sv : wins[(uint64_t)j * 32768 + (sv & 0x7FFF)];
            }
        }
    }
}

void launch_find_blocks(const uint8_t *in_base, const FindJob *fjobs, uint32_t njobs, uint64_t *start_bit, hipStream_t st) {
    if (njobs) hipLaunchKernelGGL(k_find_blocks, dim3(njobs), dim3(64), 0, st, in_base, fjobs, njobs, start_bit);
}
// groups / gfirst: device copies of the group table (ngroups entries; nmem + 1 first-group indices); gmaps (ngroups x 64 KiB) and
// ewins (ngroups x 32 KiB) are only needed when some member has more than one group (chained == true)
int launch_resolve_wins(const uint16_t *sym, const uint64_t *ooff, const uint64_t *jbase, uint8_t *wins, const ParMember *mem, uint32_t nmem,
                        const ResGroup *groups, uint32_t ngroups, const uint32_t *gfirst, bool chained, uint16_t *gmaps, uint8_t *ewins, hipStream_t st) {
    if (!ngroups) return 0;
    if (chained) {
        static std::atomic<uint64_t> attr_mask{0};
        int dev = 0; (void)hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_mask.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void *)k_resolve_rel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) return SZL_E_DEVICE;
            attr_mask.fetch_or(bit, std::memory_order_acq_rel);
        }
        hipLaunchKernelGGL(k_resolve_rel, dim3(ngroups), dim3(1024), 131072, st, sym, ooff, jbase, mem, groups, gmaps);
        hipLaunchKernelGGL(k_resolve_chain, dim3(nmem), dim3(1024), 0, st, groups, gfirst, (const uint16_t *)gmaps, ewins, mem);
    }
    hipLaunchKernelGGL(k_resolve_wins, dim3(ngroups), dim3(1024), 0, st, sym, ooff, jbase, wins, mem, groups, chained ? (const uint8_t *)ewins : nullptr);
    return 0;
}
void launch_convert(const uint16_t *sym, const uint64_t *ooff, const uint64_t *jbase, const uint8_t *wins, uint8_t *out_base, const ParMember *mem,
                    uint32_t nmem, uint32_t nblocks, hipStream_t st) {
    if (nblocks) hipLaunchKernelGGL(k_convert, dim3(nblocks), dim3(256), 0, st, sym, ooff, jbase, wins, out_base, mem, nmem);
}

} // namespace szl
