// szl_kernels_inflate.hip — Inflater (C/Inflater.cs) on the device: one wavefront per deflate stream.
//
// Reference being restated (C/ = /root/reference/src/ICSharpCode.SharpZipLib/Zip/Compression/):
//   block headers / stored / static / dynamic         C/Inflater.cs:429-552 (Decode), :283-386 (DecodeHuffman)
//   length/distance base+extra tables                  C/Inflater.cs:39-68
//   dynamic header (HLIT/HDIST/HCLEN, RLE 16/17/18)    C/InflaterDynHeader.cs:42-120
//   code construction                                  C/InflaterHuffmanTree.cs:87-169 (canonical, LSB-first lookup)
//   32 KiB output window with overlap-safe repeat       CS/OutputWindow.cs:63-92
// The decoder is resumable at any token (NEED_INPUT / OUTPUT_FULL) like the reference's 13-mode state machine;
// its persistent state lives in InfState.  Inside a Huffman block the wavefront decodes speculatively: every lane
// decodes the token that would start at its bit offset of a 128-bit span and a readlane walk picks the real token
// starts; block headers, long codes, stored blocks and stream ends go through lane 0's careful path.  All lanes
// do the data movement (match copies inside the LDS window, window flushes to HBM, input staging, table
// construction); independent streams (zip entries, gzip members) run on other wavefronts — 4 (10 in the short-window form) per CU, bounded
// by the 32 KiB window each keeps in LDS.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "szl_internal.h"
#include "szl_inflate.h"
#include "szl_inflate_reftree.h"

namespace szl {

#ifndef SZL_INF_NPO
#define SZL_INF_NPO 4   // bit offsets every lane decodes speculatively per round (64 * NPO bits of input per round; 6 and 8 measured: +-1 %)
#endif
enum : int { I_WIN = 32768, I_STAGE = 1024, I_LPB = 10, I_DPB = 9, MAX_MATCH_I = 258 };
// One-shot jobs (batch API: the whole output of a stream is one contiguous region) use the SHORT-window form: only the
// last 8 KiB of output live in LDS and matches that reach farther back read the output region itself, which lets a CU keep
// 10 streams in flight instead of 4.  Streaming jobs (window saved between calls, preset dictionary) keep all 32 KiB in LDS.
enum : int { I_WIN_SHORT = 8192 };

__constant__ uint16_t c_cplens[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_cplext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_cpdist[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_cpdext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_meta_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15}; // C/InflaterDynHeader.cs:23-24

struct HuffTab {            // canonical code of up to 288 symbols; LSB-first primary table + canonical second level
    uint16_t first[16];     // first canonical code of each length (MSB-first value)
    uint16_t count[16];
    uint16_t offs[16];      // index into sorted[] of the first symbol of each length
    uint16_t sorted[288];   // symbols ordered by (length, symbol)
};

template <int WIN, typename WT>
struct InfLds {
    WT win[WIN];
    uint32_t stage[I_STAGE / 4 + 4];
    uint16_t llut[1 << I_LPB];
    uint16_t dlut[1 << I_DPB];
    HuffTab lt, dt;
    uint8_t lens[320];
    uint16_t codes[320];
    uint32_t queue[64];
    uint16_t toff[64 + 2];  // output offset of each queued token inside the round (apply step)
};

// Build decode tables from code lengths lens[0..n) (all lanes). pb = primary bits.
// Returns false for an over-subscribed set (wave-uniform): the reference's BuildTree then throws IndexOutOfRangeException
// out of DeflaterHuffman.BitReverse as soon as a canonical code reaches 65536 (C/InflaterHuffmanTree.cs:133-166,
// C/DeflaterHuffman.cs:924-930) — always, because the last code of the longest length is >= 65536 exactly when the
// lengths' Kraft sum exceeds 1.  Incomplete sets are accepted like there (:116-121 commented out).
// Returns 2 instead of 1 for a set on which the reference's table is not a canonical decoder (incomplete, with codes of 10+
// bits: szl_inflate_reftree.h) — the caller then leaves the block to k_inflate_exact.
__device__ int build_tab(const uint8_t *lens, int n, HuffTab *T, uint16_t *lut, int pb, uint16_t *codes, int lane) {
    for (int i = lane; i < (1 << pb); i += 64) lut[i] = 0;
    // Counts per length, first canonical code and offset of each length, and every symbol's code: all lanes, registers only.
    // (The obvious form — lane 0 counting into cnt[lens[i]] — indexes private arrays dynamically, which puts them in scratch
    // memory: ~1 us per dependent access, three passes over up to 286 symbols, ~1 ms per block — a third of the whole decode.)
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; l++) cnt[l] = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        const int l = i < n ? (int)lens[i] : 0;
#pragma unroll
        for (int L = 1; L < 16; L++) cnt[L] += (uint32_t)__builtin_popcountll(__ballot(l == L));
    }
    uint32_t first[16], offs[16];
    int over, quirk;
    {
        uint32_t code = 0, off = 0, kraft = 0, longc = 0;
#pragma unroll
        for (int L = 1; L < 16; L++) {
            first[L] = code & 0xFFFFu; offs[L] = off;
            code = (code + cnt[L]) << 1;
            off += cnt[L];
            kraft += cnt[L] << (16 - L);
            if (L >= 10) longc += cnt[L];
        }
        over = kraft > 65536;
        quirk = kraft < 65536 && longc != 0;       // rt_is_quirk_set
    }
    if (lane == 0) {
#pragma unroll
        for (int L = 1; L < 16; L++) { T->first[L] = (uint16_t)first[L]; T->count[L] = (uint16_t)cnt[L]; T->offs[L] = (uint16_t)offs[L]; }
    }
    if (!over) {
        uint32_t run[16];
#pragma unroll
        for (int L = 0; L < 16; L++) run[L] = 0;
        const uint64_t below = (1ull << lane) - 1ull;
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int i = c0 + lane;
            const int l = i < n ? (int)lens[i] : 0;
            uint32_t mycode = 0, mypos = 0;
#pragma unroll
            for (int L = 1; L < 16; L++) {
                const uint64_t m = __ballot(l == L);
                const uint32_t r = run[L] + (uint32_t)__builtin_popcountll(m & below);
                if (l == L) { mycode = first[L] + r; mypos = offs[L] + r; }
                run[L] += (uint32_t)__builtin_popcountll(m);
            }
            if (l) { codes[i] = (uint16_t)mycode; T->sorted[mypos] = (uint16_t)i; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (__builtin_amdgcn_readfirstlane(over)) return 0;
    for (int i = lane; i < n; i += 64) {
        int l = lens[i];
        if (l == 0) continue;
        uint32_t rev = (__builtin_bitreverse32((uint32_t)codes[i]) >> (32 - l)) & ((1u << l) - 1);
        if (l <= pb) {
            uint16_t e = (uint16_t)((i << 4) | l);
            for (uint32_t j = rev; j < (1u << pb); j += (1u << l)) lut[j] = e;
        } else {
            lut[rev & ((1u << pb) - 1)] = 0xFFFE; // longer than the primary table: canonical second level
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    return __builtin_amdgcn_readfirstlane(quirk) ? 2 : 1;
}

// inclusive prefix sum over the wavefront in registers (Hillis-Steele inside each row of 16 lanes, then the row totals):
// __shfl_up is a ds_bpermute — six dependent LDS crossbar trips per scan, twice per round
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15: total of row 0 -> row 1, of row 2 -> row 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31: total of rows 0-1 -> rows 2 and 3
    return x;
}
// OR over the wavefront without the LDS (DPP row shifts and broadcasts); the result is wave-uniform
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t x) {
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8  (lane 15 of every row: the row's OR)
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// decode one symbol from the low bits of `bits` (LSB-first); returns sym | len<<16, or -1 invalid
__device__ __forceinline__ int decode_sym(const HuffTab *T, const uint16_t *lut, int pb, uint32_t bits) {
    uint32_t e = lut[bits & ((1u << pb) - 1)];
    if (e != 0xFFFE) {
        if (e == 0) return -1;
        return (int)(e >> 4) | ((int)(e & 15) << 16);
    }
    uint32_t rev15 = __builtin_bitreverse32(bits) >> 17; // first 15 stream bits, MSB-first
    for (int l = pb + 1; l <= 15; l++) {
        uint32_t c = rev15 >> (15 - l);
        uint32_t idx = c - T->first[l];
        if (idx < T->count[l]) return (int)T->sorted[T->offs[l] + idx] | (l << 16);
    }
    return -1;
}

// (second launch-bound: waves per SIMD the register allocation must allow — LDS admits 10 streams per CU in the short-window form)
// PMODE: chunked decode of ONE member (szl_api_inflate.hip, inflate_member_parallel).  0 = ordinary job.  1 = count pass: decode
// the blocks from job.start_bit up to the block boundary job.stop_bit, produce no bytes, report the output length and the bit
// position reached.  2 = symbol pass: the same decode, writing 16-bit symbols to job.sym_out — a byte, or 0x8000 | i for
// "byte i of the 32 KiB in front of this chunk" (what a back-reference reaching before the chunk reads; resolved afterwards).
// DENSE (symbol pass only, behind SZL_INF_DENSE=1; SZL_INF_SLOTS_PER_CU sizes the chunks for the jobs a CU then holds): the register allocation for 3 wavefronts per SIMD (168 registers instead
// of 210, no scratch), so that the LDS (15 KiB per job) admits 10 chunk jobs per CU instead of 8.  Round 4 measured a 512 MiB text member
// 28.5 -> 26.2 ms with it (profiles/r04/inflate_slots.log in the history of 9e0c386) but could not finish the A/B of the whole bench for
// want of GPU minutes; the round's last GPU seconds then showed 64 x 4 MiB members THREE TIMES SLOWER with both knobs set
// (profiles/r04/r5_dense_on_64x4mib_members.log): NOT the default; the same source, another register budget (checked on the interpreter: tools/gfxsim suite inflate_parallel).
template <bool SHORTWIN, int PMODE, bool DENSE = false>
__global__ __launch_bounds__(64, (SHORTWIN && PMODE != 2) ? 3 : (SHORTWIN ? (DENSE ? 3 : 2) : 1)) void k_inflate(const uint8_t *__restrict__ in_base, uint8_t *__restrict__ out_base,
                                                InfJob *jobs, InfState *states, uint32_t njobs) {
    // (round 2 measured the 4096-entry window slower for one long member — 74 -> 93 ms per 256 MiB — when every far read was an
    // agent-scope load that missed the L2; round 3, with workgroup-scope far reads, it wins)
#ifndef SZL_INF_SYMWIN
#define SZL_INF_SYMWIN 4096   // entries of the symbol pass's LDS window: 8 KiB of 16-bit symbols, 8 chunk jobs per CU instead of 6 (1 GiB member 61.0 -> 58.1 ms, now that far reads are workgroup-scope)
#endif
    constexpr int WIN = SHORTWIN ? (PMODE == 2 ? SZL_INF_SYMWIN : I_WIN_SHORT) : I_WIN;
    constexpr uint64_t I_WMASK = WIN - 1;
    // SHORTWIN: bytes older than FAR_DIST are read from the output region; everything older than ROOM is flushed there first
    constexpr uint32_t FAR_DIST = WIN - 512;
    constexpr uint64_t ROOM = SHORTWIN ? (uint64_t)WIN / 2 : (uint64_t)(I_WIN - 300);
    constexpr uint64_t FLUSH_AT = SHORTWIN ? 1024 : 16384;
    constexpr uint32_t ROUND_MAX = SHORTWIN ? (uint32_t)WIN / 4 : 64u * MAX_MATCH_I; // output bytes one parallel round may queue
    using WT = typename std::conditional<PMODE == 2, uint16_t, uint8_t>::type;
    __shared__ InfLds<WIN, WT> S;
    const uint32_t ji = blockIdx.x;
    if (ji >= njobs) return;
    const int lane = threadIdx.x;
    InfJob job = jobs[ji];
    InfState *st = &states[ji];
    const uint8_t *in = in_base + job.in_off;
    uint8_t *out = out_base + job.out_off;
    const uint64_t in_bits = job.in_len * 8ull;

    // ---- load persistent state (wave-uniform scalars via lane 0 reads + broadcast is unnecessary: all lanes read)
    uint64_t bitpos = PMODE ? job.start_bit : st->bitpos, outpos = PMODE ? 0 : st->outpos;
    uint32_t mode = PMODE ? (uint32_t)INF_M_HEADER : st->mode, lastblk = PMODE ? 0u : st->last, stored_left = st->stored_left, btype = st->btype;
    uint32_t lnum = st->lnum, dnum = st->dnum;
    uint64_t stop_bit = PMODE ? job.stop_bit : ~0ull;   // chunk jobs: where this job ends (moves on past false candidates: InfJob.starts)
    uint32_t sidx = 0, moves = 0;
    const uint32_t pend_len = 0, pend_dist = 0; // a token that does not fit is simply not consumed
    const uint64_t out_start = outpos;             // stream position of out[0] for this call
    uint64_t out_limit = PMODE == 1 ? ~0ull >> 1 : outpos + job.out_cap;         // (symbol pass: the job's staging region — which may grow: EV_STOP)
    uint64_t flushed = outpos;
    int status = INF_RUNNING;

    // window: restore the last 32 KiB of output
    if (PMODE == 2) {
        // the bytes in front of the chunk are not known yet: window position -j (ring index WIN - j) holds the symbol "byte
        // 32768 - j of the preceding 32 KiB"
        for (int i = lane; i < WIN; i += 64) S.win[i] = (WT)(0x8000u | (uint32_t)(32768 - WIN + i));
    } else if (PMODE == 1) {
    } else if (!SHORTWIN && (outpos > 0 || job.load_window) && job.window) {
        for (int i = lane * 16; i < I_WIN; i += 64 * 16) *(uint4 *)&S.win[i] = *(const uint4 *)&job.window[i];
    } else {
        // A fresh stream starts on the reference's zero-initialised window (CS/OutputWindow.cs:22): a match whose distance
        // reaches before the first output byte yields zeros there, never stale LDS of another workgroup.
        for (int i = lane * 16; i < WIN; i += 64 * 16) *(uint4 *)&S.win[i] = make_uint4(0, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // zlib header (C/Inflater.cs:211-249)
    if (mode == INF_M_ZHEADER) {
        if (in_bits < 16) status = INF_NEED_INPUT;
        else {
            uint32_t h = ((uint32_t)in[0] << 8) | in[1];
            if (h % 31 != 0) status = SZL_E_HEADER_CHECKSUM;
            else if ((h & 0x0f00) != (8u << 8)) status = SZL_E_METHOD_UNKNOWN;
            else if (h & 0x0020) { // FDICT: DECODE_DICT (:254-270) — the caller must SetDictionary; DICTID is bytes 2..5
                if (in_bits < 48) status = INF_NEED_INPUT;
                else {
                    if (lane == 0) st->adler_read = ((uint32_t)in[2] << 24) | ((uint32_t)in[3] << 16) | ((uint32_t)in[4] << 8) | in[5];
                    bitpos = 48; mode = INF_M_HEADER; status = INF_NEED_DICT;
                }
            } else { bitpos = 16; mode = INF_M_HEADER; }
        }
    }
    // 0: over-subscribed set (the reference's BuildTree throws); 2: the block's literal/length or distance set is one the reference's
    // lookup table decodes differently from a canonical decoder (incomplete, with codes of 10+ bits: szl_inflate_reftree.h) — such a
    // block, and the bit buffer's misbehaviour it can set off, belong to k_inflate_exact (szl_kernels_inflate_exact.hip)
    auto rebuild_tables = [&]() -> int {
        if (btype == 1) {
            for (int i = lane; i < 288; i += 64) S.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)); // C/InflaterHuffmanTree.cs:34-70
            if (lane < 32) S.lens[288 + lane] = 5;
            __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            build_tab(S.lens, 288, &S.lt, S.llut, I_LPB, S.codes, lane);
            build_tab(S.lens + 288, 32, &S.dt, S.dlut, I_DPB, S.codes, lane);
            return 1;
        }
        // literal/length tree first, then the distance tree (C/InflaterDynHeader.cs:126-134 via C/Inflater.cs DECODE_DYN_HEADER)
        const int rl = build_tab(S.lens, (int)lnum, &S.lt, S.llut, I_LPB, S.codes, lane);
        if (!rl) return 0;
        const int rd = build_tab(S.lens + lnum, (int)dnum, &S.dt, S.dlut, I_DPB, S.codes, lane);
        if (!rd) return 0;
        return (rl == 2 || rd == 2) ? 2 : 1;
    };
    if (PMODE == 0 && status == INF_RUNNING && mode == INF_M_HUFF) {
        if (btype == 2) for (int i = lane; i < 320; i += 64) S.lens[i] = st->lens[i];
        __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        (void)rebuild_tables(); // a resumed block: these lengths were accepted when its header was read (never a set of k_inflate_exact's)
    }
    // second level of the literal/length code for the wave-parallel decode.  The codes of a canonical set, left-aligned to 15 bits, lie in
    // consecutive intervals by length: those of length l in [limit(l - 1), limit(l)), limit(l) = (first[l] + count[l]) << (15 - l).  Lane l
    // holds limit(l) and offs[l] - first[l]: a code's length is I_LPB + 1 + the number of limits it has reached, its place in `sorted` its
    // top l bits plus that difference (round 6: four compares and an add where five compare-and-select steps per offset were 11 % of the
    // symbol pass's instructions, tools/gfxsim/srcprof.py)
    uint32_t l2a = 0, l2b = 0;
    auto load_second_level = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        l2a = lane < 16 && lane >= 1 ? ((uint32_t)S.lt.first[lane] + (uint32_t)S.lt.count[lane]) << (15 - lane) : 0u;
        l2b = lane < 16 ? (uint32_t)S.lt.offs[lane] - (uint32_t)S.lt.first[lane] : 0u;
    };
    load_second_level();

    // ---- input staging: S.stage holds input bytes [sbase, sbase + I_STAGE)
    uint64_t sbase = ~0ull;
    auto restage = [&](uint64_t bytepos) {
        sbase = bytepos & ~3ull;
        for (int i = lane; i < I_STAGE / 4 + 4; i += 64) {
            uint64_t p = sbase + 4ull * i;
            uint32_t w = 0;
            if (p + 4 <= job.in_len) __builtin_memcpy(&w, in + p, 4);
            else for (int k = 0; k < 4; k++) if (p + k < job.in_len) w |= (uint32_t)in[p + k] << (8 * k);
            S.stage[i] = w;
        }
        __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    };
    auto flush = [&](uint64_t upto) { // copy window bytes [flushed, upto) to HBM
        uint64_t n = upto - flushed;
        if (PMODE == 2) { for (uint64_t i = lane; i < n; i += 64) job.sym_out[flushed + i] = (uint16_t)S.win[(flushed + i) & I_WMASK]; }
        else if (PMODE == 0) { for (uint64_t i = lane; i < n; i += 64) out[flushed - out_start + i] = (uint8_t)S.win[(flushed + i) & I_WMASK]; }
        flushed = upto;
        // later far matches of THIS wavefront read these bytes back: workgroup scope orders its own stores and loads (both go through the
        // CU's L1, which its stores write through); agent scope would write the XCD's L2 back on every KiB and make every far read miss it
        if (SHORTWIN) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    };

    // bit reader (lane 0 owns bb/nb; bitpos is the stream position of bit 0 of bb)
    uint64_t bb = 0; int nb = 0;
    auto stage_ok = [&](uint64_t bytepos) { return sbase != ~0ull && bytepos >= sbase && bytepos + 8 <= sbase + I_STAGE; };
    auto refill = [&]() { // lane 0 only; requires stage_ok((bitpos+nb)>>3)
        while (nb <= 32) {
            uint32_t o = (uint32_t)(((bitpos + nb) >> 3) - sbase);
            uint32_t w0 = S.stage[o >> 2], w1 = S.stage[(o >> 2) + 1];
            uint32_t x = __builtin_amdgcn_alignbyte(w1, w0, o & 3);
            bb |= (uint64_t)x << nb;
            nb += 32;
        }
    };

    enum { EV_NONE = 0, EV_RESTAGE, EV_TABLES, EV_STORED, EV_STOP };
    enum { QN = 64, NPO = SZL_INF_NPO, PAR_W = 64 * NPO, PAR_BYTES = PAR_W / 8 + 24 }; // offsets per lane; bits per round; bytes the parallel round may touch from its first byte on
    // Decode state (bit buffer, block mode, table sizes) is private to lane 0.  Each round lane 0 decodes up to 64
    // tokens into an LDS queue; then the whole wavefront applies them (prefix sum of lengths, literals in parallel,
    // matches in order with all lanes copying), flushes the window when half full, and services lane 0's request.
    uint32_t dbg_rounds = 0, dbg_par = 0, dbg_partok = 0;
    while (status == INF_RUNNING) {
        int ev = EV_NONE, ea = 0, eb = 0, ntok = 0;
        dbg_rounds++;
        // ---------------- wave-parallel round (the common case inside a Huffman block)
        // Every lane decodes the complete token that WOULD start at NPO bit offsets of a 64*NPO-bit span (offset = lane,
        // lane+64, ...): literal/length code through the primary table, length extra bits, distance code, distance extra bits
        // (C/Inflater.cs:283-386).  Which offsets really are token starts is then a walk from offset 0 over the decoded
        // bit counts (scalar, one readlane per token, no memory traffic); the tokens at the starts are queued by all lanes
        // at once: one round yields ~19 tokens on text instead of lane 0 crawling through them.  Anything unusual at a real
        // token start — code longer than the primary table, end of block, an invalid code — stops the walk there and is left
        // to the careful single-token path below.
        bool par_ok = false;   // preconditions of the parallel round held (then the careful path handles one token only)
        uint32_t rmax = ROUND_MAX;
        uint64_t par_bitpos = 0;
        {
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mode);
            const uint64_t P = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bitpos >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bitpos);
            if (m0 == INF_M_HUFF && sbase != ~0ull) {
                const uint64_t byte0 = P >> 3;
                const uint64_t availb = in_bits > P ? in_bits - P : 0;
                const uint64_t room_lim = flushed + ROOM;
                const uint64_t olim = out_limit < room_lim ? out_limit : room_lim;
                par_ok = byte0 >= sbase && byte0 + PAR_BYTES <= sbase + I_STAGE && availb >= (uint64_t)(PAR_W + 64) &&
                         outpos + (uint64_t)MAX_MATCH_I <= olim;
                // bytes this round may queue before its last token (up to the end of the output region, not ROUND_MAX short of it:
                // the last 2 KiB of every 64 KiB zip entry used to crawl through lane 0)
                if (par_ok) { const uint64_t r = olim - outpos - MAX_MATCH_I; rmax = r < (uint64_t)ROUND_MAX ? (uint32_t)r : ROUND_MAX; }
                if (!par_ok && byte0 >= sbase + 256 && byte0 + PAR_BYTES > sbase + I_STAGE && byte0 + PAR_BYTES <= job.in_len) {
                    // the staged input is nearly used up: slide it (cooperative) instead of crawling through the careful loop
                    ev = EV_RESTAGE; ea = (int)(uint32_t)byte0; eb = (int)(uint32_t)(byte0 >> 32);
                }
            }
            if (par_ok) {
                // Branch-free on purpose: with exec-mask regions around the rare cases the compiler keeps the NPO decodes apart, each
                // waiting out its own three LDS round trips (170 instructions per 64 offsets, two thirds of a round's issue slots);
                // as straight-line code they interleave.  32-bit windows throughout (v_alignbit instead of 64-bit shifts): a token
                // is at most 15+5+15+13 = 48 bits, the length part at most 20, so the distance part fits one register.
                uint32_t tokv[NPO];
                const uint32_t p7 = (uint32_t)P & 7u, so0 = (uint32_t)((P >> 3) - sbase);
                uint32_t fa2[16], of2[16];                                   // second level of the literal/length code (uniform)
#pragma unroll
                for (int l = I_LPB; l <= 15; l++) { fa2[l] = (uint32_t)__builtin_amdgcn_readlane((int)l2a, l); of2[l] = (uint32_t)__builtin_amdgcn_readlane((int)l2b, l); }
                // (written as phases over all NPO offsets so that each phase's LDS reads are in flight together)
                uint32_t lo[NPO], hi[NPO], ev_[NPO], si_[NPO], sl2_[NPO], sr_[NPO], t2_[NPO], de_[NPO];
                {
                    uint32_t w0[NPO], w1[NPO], w2[NPO], sh[NPO];
#pragma unroll
                    for (int jj = 0; jj < NPO; jj++) {
                        const uint32_t kk = p7 + (uint32_t)lane + 64u * (uint32_t)jj;     // bit offset from the byte of P
                        const uint32_t so = so0 + (kk >> 3);
                        sh[jj] = ((so & 3u) << 3) | (kk & 7u);
                        w0[jj] = S.stage[so >> 2]; w1[jj] = S.stage[(so >> 2) + 1]; w2[jj] = S.stage[(so >> 2) + 2];
                    }
#pragma unroll
                    for (int jj = 0; jj < NPO; jj++) {
                        lo[jj] = __builtin_amdgcn_alignbit(w1[jj], w0[jj], sh[jj]); hi[jj] = __builtin_amdgcn_alignbit(w2[jj], w1[jj], sh[jj]); // 64 stream bits from the offset on
                        ev_[jj] = S.llut[lo[jj] & ((1u << I_LPB) - 1)];
                    }
                }
#pragma unroll
                for (int jj = 0; jj < NPO; jj++) {
                    // a literal/length code longer than the primary table (every ~86th token on text; each used to cost a round of its
                    // own through lane 0): the canonical second level (decode_sym) — first code, count and offset of the lengths
                    // I_LPB+1..15 come out of two registers (l2a, l2b), the symbol is one LDS read (made by every lane: no branch)
                    const uint32_t rev15 = __builtin_bitreverse32(lo[jj]) >> 17;
                    uint32_t len2 = I_LPB + 1, base2 = of2[I_LPB + 1];
#pragma unroll
                    for (int l = I_LPB + 1; l < 15; l++) {                // (the limits ascend: the code's length is the first whose limit it is below)
                        const bool ge = rev15 >= fa2[l];
                        len2 += ge ? 1u : 0u; base2 = ge ? of2[l + 1] : base2;
                    }
                    const bool long_code = rev15 >= fa2[I_LPB] && rev15 < fa2[15];   // (a code of I_LPB + 1 .. 15 bits at all: `sorted` is read by every lane)
                    const uint32_t si = long_code ? (rev15 >> (15u - len2)) + base2 : 0u, sl2 = long_code ? len2 : 0u;
                    si_[jj] = si; sl2_[jj] = sl2;
                    uint32_t sr = S.lt.sorted[si];
                    asm volatile("" : "+v"(sr));                           // (keeps the read out of a conditional region)
                    sr_[jj] = sr;
                }
                uint32_t sl_[NPO], xl_[NPO], len_[NPO], sym_[NPO];
                bool evalid_[NPO], islit_[NPO], islen_[NPO];
#pragma unroll
                for (int jj = 0; jj < NPO; jj++) {
                    const uint32_t e2 = sl2_[jj] ? ((sr_[jj] << 4) | sl2_[jj]) : 0u;
                    const uint32_t e = ev_[jj] == 0xFFFEu ? e2 : ev_[jj];
                    const uint32_t sl = e & 15u, sym = e >> 4;
                    evalid_[jj] = e - 1u < 0xFFFDu;                                                   // neither invalid (0) nor a long code
                    islit_[jj] = sym < 256u; islen_[jj] = sym - 257u < 29u;
                    const uint32_t ls = islen_[jj] ? sym - 257u : 0u;
                    const uint32_t xl = (ls < 8u || ls == 28u) ? 0u : ((ls - 4u) >> 2);                 // CPLEXT :44-48
                    const uint32_t lbase = ls < 8u ? 3u + ls : (ls == 28u ? 258u : 3u + ((4u + (ls & 3u)) << xl)); // CPLENS :39-43
                    const uint32_t t1 = __builtin_amdgcn_alignbit(hi[jj], lo[jj], sl);                  // the bits behind the code
                    len_[jj] = lbase + __builtin_amdgcn_ubfe(t1, 0u, xl);
                    t2_[jj] = __builtin_amdgcn_alignbit(hi[jj], lo[jj], sl + xl);                       // ... behind the length's extra bits (<= 20)
                    de_[jj] = S.dlut[t2_[jj] & ((1u << I_DPB) - 1)];
                    sl_[jj] = sl; xl_[jj] = xl; sym_[jj] = sym;
                }
#pragma unroll
                for (int jj = 0; jj < NPO; jj++) {
                    const uint32_t de = de_[jj];
                    const bool dvalid = de - 1u < 0xFFFDu && (de >> 4) < 30u;
                    const uint32_t dl = de & 15u, dsym = dvalid ? de >> 4 : 0u;
                    const uint32_t xd = dsym < 4u ? 0u : ((dsym >> 1) - 1u);                            // CPDEXT :62-68
                    const uint32_t dbase = dsym < 4u ? 1u + dsym : 1u + ((2u + (dsym & 1u)) << xd);     // CPDIST :50-60
                    const uint32_t dist = dbase + __builtin_amdgcn_ubfe(t2_[jj] >> dl, 0u, xd);
                    const bool lit = evalid_[jj] && islit_[jj], ismatch = evalid_[jj] && islen_[jj] && dvalid;
                    const uint32_t nbv = lit ? sl_[jj] : (ismatch ? sl_[jj] + xl_[jj] + dl + xd : 0u);
                    const uint32_t tk = lit ? sym_[jj] : (ismatch ? (len_[jj] | (dist << 16)) : 0u);
                    tokv[jj] = tk | (nbv << 10); // bits 10..15 (free: literal/length use 9 bits): bit count of the token, 0 = stop
                }
                // Which offsets are token starts: a walk over the bit counts alone (NPO x 6 bits packed per lane, one readlane and a
                // handful of scalar instructions per token — this serial loop was 38 % of the kernel when it also carried the tokens),
                // recorded as one lane mask per 64 offsets.  The tokens are then queued by all lanes at once (rank = tokens before).
                uint32_t pk[(NPO + 4) / 5] = {};                  // five 6-bit counts per register
#pragma unroll
                for (int jj = 0; jj < NPO; jj++) pk[jj / 5] |= ((tokv[jj] >> 10) & 63u) << (6 * (jj % 5));
                uint64_t smask[NPO];
                uint32_t o = 0;
                bool stopped = false;
#pragma unroll
                for (int jj = 0; jj < NPO; jj++) {
                    uint64_t m = 0;
                    if (!stopped && o < 64u * (uint32_t)(jj + 1)) {
                        // while (o < limit) { nx = (readlane(pk, o & 63) >> shift) & 63; if (!nx) break; m |= 1 << (o & 63); o += nx; }
                        // — eight scalar instructions and one taken branch per token (readlane and s_bitset1 use bits 5:0 of o)
                        uint32_t nx = 1, t;
                        asm volatile("1%=:\n\t"
                                     "v_readlane_b32 %[t], %[pk], %[o]\n\t"
                                     "s_bfe_u32 %[n], %[t], %[bf]\n\t"
                                     "s_cmp_eq_u32 %[n], 0\n\t"
                                     "s_cbranch_scc1 2%=f\n\t"
                                     "s_bitset1_b64 %[m], %[o]\n\t"
                                     "s_add_u32 %[o], %[o], %[n]\n\t"
                                     "s_cmp_lt_u32 %[o], %[lim]\n\t"
                                     "s_cbranch_scc1 1%=b\n"
                                     "2%=:"
                                     : [o] "+s"(o), [m] "+s"(m), [n] "+s"(nx), [t] "=&s"(t)
                                     : [pk] "v"(pk[jj / 5]), [bf] "s"((6u << 16) | (uint32_t)(6 * (jj % 5))), [lim] "s"(64u * (uint32_t)(jj + 1))
                                     : "scc");
                        stopped = nx == 0;
                    }
                    smask[jj] = m;
                }
                uint32_t before = 0;
#pragma unroll
                for (int jj = 0; jj < NPO; jj++) {
                    const uint64_t m = smask[jj];
                    if ((m >> lane) & 1ull) {
                        const uint32_t rank = before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                        if (rank < (uint32_t)QN) {
                            S.queue[rank] = tokv[jj] & 0xFFFF03FFu;
                            S.toff[rank] = (uint16_t)((uint32_t)lane + 64u * (uint32_t)jj + ((tokv[jj] >> 10) & 63u));   // bit offset behind the token (apply rewrites toff)
                        }
                    }
                    before += (uint32_t)__builtin_popcountll(m);
                }
                ntok = (int)(before < (uint32_t)QN ? before : (uint32_t)QN);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (ntok > 0) {
                    // a round queues at most rmax bytes (+ one token): cut behind the first token that starts past that
                    if ((uint32_t)ntok * (uint32_t)MAX_MATCH_I > rmax) {
                        const uint32_t tq = lane < ntok ? S.queue[lane] : 0;
                        const uint32_t ml = lane < ntok ? ((tq >> 16) ? (tq & 0xFFFF) : 1u) : 0u;
                        const uint32_t inc = wave_scan_add_u32(ml);
                        ntok = __builtin_popcountll(__ballot(lane < ntok && inc - ml <= rmax));
                    }
                    o = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.toff[ntok - 1]);
                }
                par_bitpos = P + o;
            }
        }
        const int npar = ntok;
        if (npar) { dbg_par++; dbg_partok += (uint32_t)npar; }
        if (npar > 0) { bitpos = par_bitpos; bb = 0; nb = -1; }   // adopt the parallel round's result (all lanes: only lane 0's copy counts);
                                                                   // the bit buffer is primed when the careful path next needs it
        if (npar == 0 && lane == 0) {                              // (after a parallel round: nothing more this round)
            uint64_t opos = outpos;                       // position after the queued tokens
            const uint64_t room_lim = flushed + ROOM;
            const int careful_cap = par_ok ? 1 : QN;
            for (;;) {
                if (ntok >= careful_cap || ev != EV_NONE) break;
                if (nb < 0) { // a wave-parallel round moved bitpos: prime the bit buffer there (possibly mid-byte)
                    if (!stage_ok(bitpos >> 3)) { ev = EV_RESTAGE; ea = (int)(uint32_t)(bitpos >> 3); eb = (int)(uint32_t)((bitpos >> 3) >> 32); break; }
                    const uint32_t o2 = (uint32_t)((bitpos >> 3) - sbase);
                    const uint32_t w0 = S.stage[o2 >> 2], w1 = S.stage[(o2 >> 2) + 1];
                    const uint32_t sh = (uint32_t)(bitpos & 7);
                    bb = (uint64_t)(__builtin_amdgcn_alignbyte(w1, w0, o2 & 3) >> sh); nb = 32 - (int)sh;
                }
                const uint64_t bytepos = (bitpos + nb) >> 3;
                if (!stage_ok(bytepos)) { ev = EV_RESTAGE; ea = (int)(uint32_t)(bitpos >> 3); eb = (int)(uint32_t)((bitpos >> 3) >> 32); break; }
                refill();
                const uint64_t avail = in_bits > bitpos ? in_bits - bitpos : 0; // valid bits from bitpos on
                if (mode == INF_M_HUFF) {
                    if (opos + MAX_MATCH_I > room_lim) break;              // apply + flush first
                    int r = decode_sym(&S.lt, S.llut, I_LPB, (uint32_t)bb);
                    // no code for these bits (an incomplete set, every code of which is at most 9 bits long — sets with longer codes are
                    // k_inflate_exact's): GetSymbol throws when it can peek 9 bits and reads the empty slot as "symbol 0, 0 bits" when it
                    // cannot (C/InflaterHuffmanTree.cs:184-193 vs :224-233).  (A chunk job of the parallel path has no output bound in its
                    // count pass: there anything odd is an error, and the member goes to the one-wavefront decoder.)
                    if (r < 0) { if (avail >= 9 || PMODE) { ev = EV_STOP; ea = SZL_E_CODELEN_ZERO; break; } if (job.in_more) { ev = EV_STOP; ea = INF_NEED_INPUT; break; } r = 0; }
                    const uint32_t sl = (uint32_t)r >> 16, sym = (uint32_t)r & 0xFFFF;
                    if (avail < sl) { ev = EV_STOP; ea = INF_NEED_INPUT; break; }
                    if (sym < 256) {
                        if (opos + 1 > out_limit) { ev = EV_STOP; ea = INF_OUTPUT_FULL; break; }
                        bb >>= sl; nb -= (int)sl; bitpos += sl;
                        S.queue[ntok++] = sym; opos++;
                        continue;
                    }
                    if (sym == 256) { bb >>= sl; nb -= (int)sl; bitpos += sl; mode = INF_M_HEADER; continue; }
                    if (sym - 257 >= 29) { ev = EV_STOP; ea = SZL_E_ILLEGAL_LEN_CODE; break; } // :323-326
                    uint64_t tb = bb; int tn = nb; uint64_t used = sl;
                    tb >>= sl; tn -= (int)sl;
                    // CPLENS/CPLEXT (C/Inflater.cs:39-48) in closed form: constant-memory tables indexed per lane are a
                    // vector load from HBM on this hardware (≈1 µs each) — arithmetic keeps the token loop in registers/LDS
                    const uint32_t ls = sym - 257;
                    const uint32_t xl = (ls < 8 || ls == 28) ? 0u : ((ls - 4) >> 2);
                    const uint32_t lbase = ls < 8 ? 3 + ls : (ls == 28 ? 258u : 3 + ((4 + (ls & 3)) << xl));
                    const uint32_t len = lbase + ((uint32_t)tb & ((1u << xl) - 1));
                    tb >>= xl; tn -= (int)xl; used += xl;
                    if (tn < 28) { // top up (the staged window always has >= 8 readable bytes past bytepos)
                        uint32_t o = (uint32_t)(((bitpos + used + tn) >> 3) - sbase);
                        uint32_t w0 = S.stage[o >> 2], w1 = S.stage[(o >> 2) + 1];
                        tb |= (uint64_t)__builtin_amdgcn_alignbyte(w1, w0, o & 3) << tn;
                        tn += 32;
                    }
                    int rd = decode_sym(&S.dt, S.dlut, I_DPB, (uint32_t)tb);
                    if (rd < 0) {                                      // (see the literal/length code above)
                        if (avail < used) { ev = EV_STOP; ea = INF_NEED_INPUT; break; }
                        if (avail - used >= 9 || PMODE) { ev = EV_STOP; ea = SZL_E_CODELEN_ZERO; break; }
                        if (job.in_more) { ev = EV_STOP; ea = INF_NEED_INPUT; break; }   // (only a prefix of the caller's input is here)
                        rd = 0;
                    }
                    const uint32_t dl = (uint32_t)rd >> 16, dsym = (uint32_t)rd & 0xFFFF;
                    if (dsym >= 30) { ev = EV_STOP; ea = avail < used + dl ? INF_NEED_INPUT : SZL_E_ILLEGAL_DIST_CODE; break; } // :356-359
                    tb >>= dl; tn -= (int)dl; used += dl;
                    const uint32_t xd = dsym < 4 ? 0u : ((dsym >> 1) - 1);                        // CPDIST/CPDEXT :50-68
                    const uint32_t dbase = dsym < 4 ? 1 + dsym : 1 + ((2 + (dsym & 1)) << xd);
                    const uint32_t dist = dbase + ((uint32_t)tb & ((1u << xd) - 1));
                    tb >>= xd; tn -= (int)xd; used += xd;
                    if (avail < used) { ev = EV_STOP; ea = INF_NEED_INPUT; break; }
                    if (opos + len > out_limit) { ev = EV_STOP; ea = INF_OUTPUT_FULL; break; } // not consumed: decoded again next call
                    bb = tb; nb = tn; bitpos += used;
                    S.queue[ntok++] = len | (dist << 16); opos += len;
                    continue;
                }
                if (mode == INF_M_STORED) {
                    if (stored_left == 0) { mode = INF_M_HEADER; continue; }
                    if (ntok) break; // apply what is queued first
                    // byte aligned here (SkipToByteBoundary :490): copy bytes input -> window with the whole wavefront
                    const uint64_t bp = bitpos >> 3;
                    const uint64_t can_in = job.in_len > bp ? job.in_len - bp : 0;
                    const uint64_t can_out = out_limit - outpos;
                    uint64_t n = stored_left;
                    if (n > can_in) n = can_in;
                    if (n > can_out) n = can_out;
                    if (n == 0) { ev = EV_STOP; ea = can_in == 0 ? INF_NEED_INPUT : INF_OUTPUT_FULL; break; }
                    ev = EV_STORED; ea = (int)n; eb = 0;     // (not bounded by the window's room: the bytes go from the input to the output, see EV_STORED)
                    stored_left -= (uint32_t)n;
                    break;
                }
                // ---------------- INF_M_HEADER
                if (lastblk) { mode = INF_M_DONE; ev = EV_STOP; ea = INF_FINISHED; break; }
                if (PMODE && bitpos >= stop_bit) {                  // the next chunk's block starts here —
                    bool ends = true;
                    // — or the candidate lay inside the block that ended here: on to the next one.  (Not for a candidate within ten bits of the
                    // boundary — a stored header named early, the host's chain walk knows it for the same start — and not more than twice: a
                    // job that itself began at a false candidate may be following a consistent chain of its own, the blocks of a deflate
                    // stream the member carries as payload, and must not follow it to its end.)
                    if (bitpos - stop_bit > 10 && job.starts && moves < 2u) {
                        moves++;
                        uint32_t k = sidx;
                        while (k < job.nstarts && job.starts[k] < bitpos) k++;
                        sidx = k; stop_bit = k < job.nstarts ? job.starts[k] : job.stop_last;
                        ends = bitpos >= stop_bit;
                    }
                    if (ends) { ev = EV_STOP; ea = INF_CHUNK_END; break; }
                }
                if (PMODE == 0 && job.stop_at_header) { ev = EV_STOP; ea = INF_CHUNK_END; break; } // (streaming object: a block boundary was asked for)
                if (avail < 3) { ev = EV_STOP; ea = INF_NEED_INPUT; break; }
                const uint32_t t = (uint32_t)bb & 7;
                const uint32_t type = t >> 1;
                if (type == 3) { ev = EV_STOP; ea = SZL_E_UNKNOWN_BLOCK; break; }
                if (type == 0) {
                    const uint32_t skip = 3 + (uint32_t)((0 - (bitpos + 3)) & 7);
                    if (avail < skip + 32) { ev = EV_STOP; ea = INF_NEED_INPUT; break; }
                    bb >>= skip; nb -= (int)skip; bitpos += skip;
                    refill();
                    const uint32_t len = (uint32_t)bb & 0xFFFF, nlen = (uint32_t)(bb >> 16) & 0xFFFF;
                    if (nlen != (len ^ 0xFFFF)) { ev = EV_STOP; ea = SZL_E_BROKEN_STORED; break; } // :509-512
                    bb >>= 32; nb -= 32; bitpos += 32;
                    lastblk |= t & 1;
                    stored_left = len; mode = INF_M_STORED;
                    continue;
                }
                if (type == 1) {
                    bb >>= 3; nb -= 3; bitpos += 3;
                    lastblk |= t & 1; btype = 1; mode = INF_M_HUFF;
                    ev = EV_TABLES; ea = 1; break;
                }
                // dynamic: parse the whole header here; if input runs out, nothing is consumed (restart at the block header)
                {
                    uint64_t hb = bb; int hn = nb; uint64_t hp = bitpos; // local cursor
                    auto need = [&](int k) -> bool { // ensure k bits in hb; false = out of staged window
                        while (hn < k) {
                            uint64_t bp = (hp + hn) >> 3;
                            if (!(bp >= sbase && bp + 8 <= sbase + I_STAGE)) return false;
                            uint32_t o = (uint32_t)(bp - sbase);
                            uint32_t w0 = S.stage[o >> 2], w1 = S.stage[(o >> 2) + 1];
                            hb |= (uint64_t)__builtin_amdgcn_alignbyte(w1, w0, o & 3) << hn;
                            hn += 32;
                        }
                        return true;
                    };
                    auto take = [&](int k) -> uint32_t { uint32_t v = (uint32_t)hb & ((1u << k) - 1); hb >>= k; hn -= k; hp += k; return v; };
                    int fail = 0; // 1 need input, 2 need restage, <0 error
                    auto want = [&](int k) -> bool {
                        if ((in_bits > hp ? in_bits - hp : 0) < (uint64_t)k) { fail = 1; return false; }
                        if (!need(k)) { fail = 2; return false; }
                        return true;
                    };
                    uint32_t nl = 0, nd = 0, nm = 0;
                    do {
                        if (!want(17)) break;
                        take(3);
                        nl = take(5) + 257; nd = take(5) + 1; nm = take(4) + 4;
                        if (nl > 286 || nd > 30) { fail = SZL_E_DYN_HEADER; break; } // :50-52
                        // code lengths of the code-length alphabet, kept in LDS (S.codes[128..147]) to stay out of scratch
                        uint16_t *ml = S.codes + 128;
                        for (int i = 0; i < 19; i++) ml[i] = 0;
                        for (uint32_t i = 0; i < nm; i++) { if (!want(3)) break; ml[c_meta_order[i]] = (uint16_t)take(3); }
                        if (fail) break;
                        uint16_t *mlut = S.codes; // 128 entries
                        for (int i = 0; i < 128; i++) mlut[i] = 0;
                        { // new InflaterHuffmanTree(codeLengths) of the code-length alphabet (C/InflaterDynHeader.cs:64): over-subscribed => throws
                            int kraft = 0;
                            for (int i = 0; i < 19; i++) if (ml[i]) kraft += 1 << (16 - ml[i]);
                            if (kraft > 65536) { fail = SZL_E_CODE_OVERSUBSCRIBED; break; }
                        }
                        int code = 0;
                        for (int l = 1; l < 8; l++) { // canonical codes, length by length, symbols in order
                            for (int i = 0; i < 19; i++) {
                                if (ml[i] != l) continue;
                                uint32_t rev = (__builtin_bitreverse32((uint32_t)code++) >> (32 - l)) & ((1u << l) - 1);
                                for (uint32_t j = rev; j < 128; j += (1u << l)) mlut[j] = (uint16_t)((i << 4) | l);
                            }
                            code <<= 1;
                        }
                        uint32_t idx = 0, total = nl + nd;
                        while (idx < total) {
                            const uint64_t rem = in_bits > hp ? in_bits - hp : 0;
                            if (!need((int)(rem < 14 ? rem : 14))) { fail = 2; break; } // code (<=7) + extra bits (<=7)
                            uint32_t e = mlut[(uint32_t)hb & 127];
                            // no code for these bits: GetSymbol throws when it can peek 9 bits (C/InflaterHuffmanTree.cs:191-193) and reads the
                            // empty slot as "symbol 0, 0 bits" when it cannot (:224-233) — a code length of 0 that consumes nothing
                            if (e == 0 && (rem >= 9 || PMODE)) { fail = SZL_E_CODELEN_ZERO; break; }
                            if (e == 0 && job.in_more) { fail = 1; break; }   // (the rule below is for the end of the caller's input, not of the uploaded prefix)
                            const uint32_t sl = e & 15, sym = e >> 4;
                            const uint32_t xb = sym < 16 ? 0 : (sym == 16 ? 2 : (sym == 17 ? 3 : 7));
                            if (rem < sl + xb) { fail = 1; break; }
                            take((int)sl);
                            if (sym < 16) { S.lens[idx++] = (uint8_t)sym; continue; }
                            uint32_t rep, val = 0;
                            if (sym == 16) {
                                if (idx == 0) { fail = SZL_E_DYN_HEADER; break; } // :83
                                val = S.lens[idx - 1]; rep = 3 + take(2);
                            } else if (sym == 17) rep = 3 + take(3);
                            else rep = 11 + take(7);
                            if (idx + rep > total) { fail = SZL_E_DYN_HEADER; break; } // :106
                            while (rep--) S.lens[idx++] = (uint8_t)val;
                        }
                        if (fail) break;
                        if (S.lens[256] == 0) { fail = SZL_E_DYN_HEADER; break; } // :113
                    } while (0);
                    if (fail == 1) { ev = EV_STOP; ea = INF_NEED_INPUT; break; }
                    if (fail == 2) { // header straddles the staged window: restage at the block header and retry
                        if (((bitpos >> 3) & ~3ull) == sbase) { ev = EV_STOP; ea = SZL_E_DYN_HEADER; break; } // cannot happen: a header is < 1 KiB
                        ev = EV_RESTAGE; ea = (int)(uint32_t)(bitpos >> 3); eb = (int)(uint32_t)((bitpos >> 3) >> 32); break;
                    }
                    if (fail < 0) { ev = EV_STOP; ea = fail; break; }
                    eb = (int)(uint32_t)(hp - bitpos);                 // the header's bits: k_inflate_exact starts in front of them
                    bb = hb; nb = hn; bitpos = hp;
                    lastblk |= t & 1; btype = 2; lnum = nl; dnum = nd; mode = INF_M_HUFF;
                    ev = EV_TABLES; ea = 2 | (int)(nl << 8) | (int)(nd << 20); break;
                }
            }
        }
        // ---------------- the whole wavefront: apply the queue
        ev = __builtin_amdgcn_readfirstlane(ev);
        ea = __builtin_amdgcn_readfirstlane(ea);
        eb = __builtin_amdgcn_readfirstlane(eb);
        ntok = __builtin_amdgcn_readfirstlane(ntok);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (ntok) {
            const uint32_t tok = lane < ntok ? S.queue[lane] : 0;
            const uint32_t dist = tok >> 16;
            const uint32_t mylen = lane < ntok ? (dist ? (tok & 0xFFFF) : 1u) : 0u;
            const uint32_t incl = wave_scan_add_u32(mylen);
            const uint64_t mypos = outpos + (incl - mylen);
            // Tokens are applied in stream order: the window is circular, so a literal written early could overwrite
            // history (32768 positions back) that an earlier far-distance match of the same round still has to read.
            uint64_t mm = __ballot(lane < ntok && dist != 0);
            int done_upto = 0; // lanes < done_upto have been applied
            if (PMODE == 1) mm = 0;   // count pass: only the lengths matter
            // Fast form for the usual round (text: ~9 tokens, ~50 bytes): if no match of the round reads a byte the round itself
            // produces (distance >= offset in the round + length; that also rules out overlapping copies), every output byte
            // is a pure function of the window BEFORE the round, so the lanes take one byte each instead of one match at a time:
            // byte b -> its token (binary search over the tokens' offsets) -> literal, window[p - dist + k] or, behind the short
            // window, the flushed output.  Chunks of 64 bytes go in ascending order, so a ring slot is overwritten only after
            // every reader of the position it held (at most WIN - 512 back) is through.
            if (mm && SHORTWIN && !__any(lane < ntok && dist != 0 && dist < (incl - mylen) + mylen)) {
                const uint32_t rtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (lane < ntok) S.toff[lane] = (uint16_t)(incl - mylen);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                for (uint32_t b0 = 0; b0 < rtot; b0 += 64) {
                    const uint32_t b = b0 + lane;
                    const bool act = b < rtot;
                    // last token with toff <= b: the token starts inside this chunk as a 64-bit mask (an OR over the wavefront, no
                    // LDS round trips — a binary search over toff[] was five dependent ones) + the tokens that start before it
                    const uint32_t rel = (incl - mylen) - b0;
                    const bool inchunk = lane < ntok && rel < 64u;
                    const uint32_t mlo = wave_or_u32(inchunk && rel < 32u ? 1u << rel : 0u), mhi = wave_or_u32(inchunk && rel >= 32u ? 1u << (rel - 32u) : 0u);
                    const uint64_t starts = ((uint64_t)mhi << 32) | mlo;
                    const uint32_t nbefore = (uint32_t)__builtin_popcountll(__ballot(lane < ntok && (incl - mylen) < b0));
                    const uint32_t lo = nbefore + (uint32_t)__builtin_popcountll(starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull))) - 1u;
                    const uint32_t tv = S.queue[lo];
                    const uint32_t k = b - (uint32_t)S.toff[lo];
                    const uint32_t d2 = tv >> 16;
                    WT val = (WT)(uint8_t)tv;
                    if (act && d2 != 0) {
                        const uint64_t p = outpos + (uint32_t)S.toff[lo];
                        if (d2 > FAR_DIST) {
                            const int64_t sp = (int64_t)(p - out_start) - (int64_t)d2 + (int64_t)k;
                            if (PMODE == 2) val = sp >= 0 ? (WT)__hip_atomic_load(job.sym_out + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : (WT)(0x8000u | (uint32_t)(32768 + sp));
                            else val = sp >= 0 ? (WT)__hip_atomic_load(out + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : (WT)0;
                        } else val = S.win[(p - d2 + k) & I_WMASK];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    if (act) S.win[(outpos + b) & I_WMASK] = val;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
                mm = 0;
                done_upto = 64;   // literals are written too
            }
            while (mm) { // matches in stream order; CS/OutputWindow.cs:63-92: out[p+k] = out[p-dist+(k mod dist)]
                const int l = __builtin_ctzll(mm);
                mm &= mm - 1;
                if (lane >= done_upto && lane < l && dist == 0) S.win[mypos & I_WMASK] = (WT)(uint8_t)tok; // literals before this match
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const uint32_t t2 = __builtin_amdgcn_readlane(tok, l);
                const uint32_t off = __builtin_amdgcn_readlane(incl - mylen, l);
                const uint32_t len = t2 & 0xFFFF, d2 = t2 >> 16;
                const uint64_t p = outpos + off;
                if (SHORTWIN && d2 > FAR_DIST) { // older than the LDS window: already flushed to the output region (ROOM < FAR_DIST - 258)
                    // (one-shot jobs: out_start == 0.)  Bytes before the start of the stream are the zeros of a fresh
                    // OutputWindow — never another stream's output region in front of this one.
                    const int64_t s0 = (int64_t)(p - out_start) - (int64_t)d2;
                    if (PMODE == 2) { // symbols already flushed to sym_out, or — before the chunk — "byte 32768 + s of the preceding 32 KiB"
                        for (uint32_t k = lane; k < len; k += 64) {
                            const int64_t sp = s0 + (int64_t)k;
                            S.win[(p + k) & I_WMASK] = sp >= 0 ? (WT)__hip_atomic_load(job.sym_out + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : (WT)(0x8000u | (uint32_t)(32768 + sp));
                        }
                    } else
                    for (uint32_t k = lane; k < len; k += 64)
                        S.win[(p + k) & I_WMASK] = s0 + (int64_t)k >= 0 ? (WT)__hip_atomic_load(out + (s0 + (int64_t)k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : (WT)0;
                } else if (d2 >= len) { // no overlap (wave-uniform test): plain copy
                    for (uint32_t k = lane; k < len; k += 64) S.win[(p + k) & I_WMASK] = S.win[(p - d2 + k) & I_WMASK];
                } else {
                    for (uint32_t k = lane; k < len; k += 64) S.win[(p + k) & I_WMASK] = S.win[(p - d2 + k % d2) & I_WMASK];
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                done_upto = l + 1;
            }
            if (PMODE != 1 && lane >= done_upto && lane < ntok && dist == 0) S.win[mypos & I_WMASK] = (WT)(uint8_t)tok; // trailing literals
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            outpos += __builtin_amdgcn_readlane(incl, 63);
        }
        if (outpos - flushed >= FLUSH_AT) flush(outpos);
        switch (ev) {
        case EV_RESTAGE: {
            const uint64_t bp = ((uint64_t)(uint32_t)eb << 32) | (uint32_t)ea;
            restage(bp);
            if (lane == 0) { // re-prime lane 0's bit buffer at bitpos (possibly mid-byte)
                uint32_t o = (uint32_t)(bp - sbase);
                uint32_t w0 = S.stage[o >> 2], w1 = S.stage[(o >> 2) + 1];
                uint32_t x = __builtin_amdgcn_alignbyte(w1, w0, o & 3);
                uint32_t sh = (uint32_t)(bitpos & 7);
                bb = (uint64_t)(x >> sh); nb = 32 - (int)sh;
            }
        } break;
        case EV_TABLES:
            btype = (uint32_t)ea & 3; lnum = ((uint32_t)ea >> 8) & 0xFFF; dnum = ((uint32_t)ea >> 20) & 0xFF;
            {
                const int rt = rebuild_tables();
                if (rt == 0) status = SZL_E_CODE_OVERSUBSCRIBED;
                else if (rt == 2) {
                    // a chunk job of a member: any error makes the parallel path step aside; the one-wavefront decoder then stops here
                    // with the stream back at the block's header (nothing of the block is consumed) and k_inflate_exact goes on
                    if (PMODE) status = SZL_E_CODELEN_ZERO;
                    else { status = INF_EXACT; bitpos -= (uint64_t)(uint32_t)eb; bb = 0; nb = -1; mode = INF_M_HEADER; lastblk = 0; }
                }
            }
            load_second_level();
            break;
        case EV_STORED: {
            uint64_t n = (uint32_t)ea;
            uint64_t bp = 0;
            { // lane 0 holds bitpos: broadcast the byte position of the stored data
                uint64_t v = bitpos >> 3;
                bp = ((uint64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            }
            uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)stored_left);   // of THIS block, behind the n bytes taken now
            uint32_t lastb = (uint32_t)__builtin_amdgcn_readfirstlane((int)lastblk);
            uint64_t r_stop = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(stop_bit >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)stop_bit);
            uint32_t r_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)sidx), r_moves = (uint32_t)__builtin_amdgcn_readfirstlane((int)moves);
            // Stored bytes go straight from the input to the output, 16 per lane and step (an incompressible member is nothing but such
            // blocks: through the window, a byte per lane, one wavefront moved 290 MiB/s — 3.6 s per GiB); the window then takes the
            // block's last WIN bytes, which is all a later match can reach (CS/OutputWindow.cs:98-110 CopyStored).  And a stored block
            // that is followed by another one (data that does not compress comes as runs of them: 16 KiB each from a level 5-9 encoder)
            // is followed at once — its header is five bytes at a known place — instead of three rounds through lane 0's careful path
            // per block (restage, header, event): 15 us per 16 KiB block, 1 GiB/s.
            const uint64_t run_start = outpos;
            for (;;) {
                const bool look = left == 0 && !lastb && bp + n + 5 <= job.in_len;   // the next header is all there
                uint32_t hb[5] = {0, 0, 0, 0, 0};
                if (look) {                                                 // (asked for before the copy: its latency hides behind it)
#pragma unroll
                    for (int k = 0; k < 5; k++) hb[k] = in[bp + n + k];
                }
                if (PMODE != 1) {
                    flush(outpos);                                         // what the window still holds comes first
                    if (PMODE == 2) {
                        uint16_t *dst = job.sym_out + outpos;
                        uint64_t i = 16ull * lane;
                        for (; i + 16 <= n; i += 1024) {
                            uint4 v; __builtin_memcpy(&v, in + bp + i, 16);
                            uint4 a, b;
                            a.x = (v.x & 0xFFu) | ((v.x & 0xFF00u) << 8); a.y = ((v.x >> 16) & 0xFFu) | ((v.x >> 8) & 0xFF0000u);
                            a.z = (v.y & 0xFFu) | ((v.y & 0xFF00u) << 8); a.w = ((v.y >> 16) & 0xFFu) | ((v.y >> 8) & 0xFF0000u);
                            b.x = (v.z & 0xFFu) | ((v.z & 0xFF00u) << 8); b.y = ((v.z >> 16) & 0xFFu) | ((v.z >> 8) & 0xFF0000u);
                            b.z = (v.w & 0xFFu) | ((v.w & 0xFF00u) << 8); b.w = ((v.w >> 16) & 0xFFu) | ((v.w >> 8) & 0xFF0000u);
                            __builtin_memcpy(dst + i, &a, 16); __builtin_memcpy(dst + i + 8, &b, 16);
                        }
                        for (uint64_t k = (n & ~15ull) + lane; k < n; k += 64) dst[k] = (uint16_t)in[bp + k];
                    } else {
                        uint8_t *dst = out + (outpos - out_start);
                        uint64_t i = 16ull * lane;
                        for (; i + 16 + 7168 <= n; i += 8192) {            // eight loads in flight per lane (one wavefront: latency is all there is)
                            uint4 v[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) __builtin_memcpy(&v[u], in + bp + i + 1024 * u, 16);
#pragma unroll
                            for (int u = 0; u < 8; u++) __builtin_memcpy(dst + i + 1024 * u, &v[u], 16);
                        }
                        for (; i + 16 <= n; i += 1024) { uint4 v; __builtin_memcpy(&v, in + bp + i, 16); __builtin_memcpy(dst + i, &v, 16); }
                        for (uint64_t k = (n & ~15ull) + lane; k < n; k += 64) dst[k] = in[bp + k];
                    }
                }
                flushed = outpos + n;
                outpos += n;
                bp += n;
                if (!look) break;
                // the header behind the block (byte aligned: C/Inflater.cs:490 SkipToByteBoundary ... :509-512): taken here only if it is
                // another stored block that is all there and fits; anything else is the careful path's
                const uint32_t t = hb[0] & 7u;
                if ((t >> 1) != 0) break;
                if (PMODE && 8 * bp >= r_stop) {                          // (a chunk job ends at its stop — or moves it, as in the careful path)
                    if (8 * bp - r_stop <= 10 || !job.starts || r_moves >= 2u) break;
                    r_moves++;
                    uint32_t k = r_idx;
                    while (k < job.nstarts && job.starts[k] < 8 * bp) k++;
                    r_idx = k; r_stop = k < job.nstarts ? job.starts[k] : job.stop_last;
                    if (8 * bp >= r_stop) break;
                }
                if (PMODE == 0 && job.stop_at_header) break;
                const uint32_t len = hb[1] | (hb[2] << 8), nlen = hb[3] | (hb[4] << 8);
                if (nlen != (len ^ 0xFFFFu)) break;
                if (bp + 5 + len > job.in_len || outpos + len > out_limit) break;
                lastb |= t & 1u;
                bp += 5; n = len; left = 0;
            }
            if (PMODE != 1) {
                // the window: the last WIN bytes of the run, read back from where they went (workgroup scope orders this wavefront's own
                // stores and loads, as for the far reads behind flush())
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                const uint64_t run = outpos - run_start, m = run < (uint64_t)WIN ? run : (uint64_t)WIN;
                const uint64_t p0 = outpos - m;
#pragma unroll 8
                for (uint64_t k = lane; k < m; k += 64) {
                    WT v;
                    if (PMODE == 2) v = (WT)__hip_atomic_load(job.sym_out + p0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else v = (WT)__hip_atomic_load(out + (p0 - out_start) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    S.win[(p0 + k) & I_WMASK] = v;
                }
            }
            // (all lanes compute the same; lane 0's copy is the decoder's)
            bitpos = 8 * bp; bb = 0; nb = 0; stored_left = left; lastblk = lastb; stop_bit = r_stop; sidx = r_idx; moves = r_moves;
            sbase = ~0ull; // force a restage at the new position
        } break;
        case EV_STOP:
            if (PMODE == 2 && ea == INF_OUTPUT_FULL && job.spill_cursor) {
                // the staging region is full: a larger one from the spill area (InfJob.spill_cursor), the symbols so far moved there, and on
                const uint64_t want = 4 * job.out_cap;
                unsigned long long at = 0;
                if (lane == 0) at = atomicAdd((unsigned long long *)job.spill_cursor, (unsigned long long)want);
                at = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
                if (at + want <= job.spill_end) {
                    flush(outpos);                                          // (what the window holds fits the old region: the token that did not was not queued)
                    uint16_t *nd = job.sym_base + at;
                    const uint64_t have = outpos - out_start;
                    for (uint64_t i = lane; i < have; i += 64) nd[i] = job.sym_out[i];
                    job.sym_out = nd; job.out_cap = want; out_limit = out_start + want;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (far reads of this wavefront come from the new region, as behind flush())
                    break;
                }
            }
            status = ea;
            break;
        default: break;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    // ---- epilogue: flush, save state
    flush(outpos);
    // lane 0 owns the decode state: make the scalars the other lanes use uniform again
    bitpos = ((uint64_t)__builtin_amdgcn_readfirstlane((int)(bitpos >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)bitpos);
    mode = (uint32_t)__builtin_amdgcn_readfirstlane((int)mode);
    btype = (uint32_t)__builtin_amdgcn_readfirstlane((int)btype);
    lastblk = (uint32_t)__builtin_amdgcn_readfirstlane((int)lastblk);
    stored_left = (uint32_t)__builtin_amdgcn_readfirstlane((int)stored_left);
    lnum = (uint32_t)__builtin_amdgcn_readfirstlane((int)lnum);
    dnum = (uint32_t)__builtin_amdgcn_readfirstlane((int)dnum);
    if (PMODE == 0 && status == INF_FINISHED && job.zlib) { // Adler-32 trailer (C/Inflater.cs:397-418): align, 4 bytes big-endian
        uint64_t bytepos = (bitpos + 7) >> 3;
        if (bytepos + 4 > job.in_len) { status = INF_NEED_INPUT; mode = INF_M_HEADER; /* lastblk stays set: resumes straight to DONE */ }
        else {
            if (lane == 0) st->adler_read = ((uint32_t)in[bytepos] << 24) | ((uint32_t)in[bytepos + 1] << 16) | ((uint32_t)in[bytepos + 2] << 8) | in[bytepos + 3];
            bitpos = (bytepos + 4) * 8;
        }
    }
    if (PMODE == 0 && !SHORTWIN && job.window && (status != INF_FINISHED || job.keep_window)) {
        for (int i = lane * 16; i < I_WIN; i += 64 * 16) *(uint4 *)&job.window[i] = *(const uint4 *)&S.win[i];
    }
    if (PMODE == 0 && btype == 2 && mode == INF_M_HUFF) for (int i = lane; i < 320; i += 64) st->lens[i] = S.lens[i];
    if (lane == 0) {
        st->bitpos = bitpos; st->outpos = outpos; st->mode = mode; st->last = lastblk; st->stored_left = stored_left;
        st->btype = btype; st->lnum = lnum; st->dnum = dnum; st->pend_len = pend_len; st->pend_dist = pend_dist;
        st->status = status;
        jobs[ji].out_written = outpos - out_start;
        jobs[ji].status = status;
        jobs[ji].consumed = (bitpos + 7) >> 3 > job.in_len ? job.in_len : (bitpos + 7) >> 3;   // (exact-table mode can drop bits past the input's end, like the reference)
        jobs[ji].end_bit = bitpos;
        if (PMODE == 2) { jobs[ji].sym_out = job.sym_out; jobs[ji].out_cap = job.out_cap; }   // (the region may have moved: EV_STOP)
        jobs[ji].dbg_rounds = dbg_rounds; jobs[ji].dbg_par = dbg_par; jobs[ji].dbg_partok = dbg_partok;
    }
}

// one_shot: every job starts at outpos 0, has no saved window / dictionary and owns one contiguous output region
void launch_inflate(const uint8_t *in, uint8_t *out, InfJob *jobs, InfState *states, uint32_t njobs, bool one_shot, hipStream_t st) {
#if SZL_LAB
    static const bool allow_short = !(getenv("SZL_INF_SHORT") && atoi(getenv("SZL_INF_SHORT")) == 0);
#else
    const bool allow_short = true;
#endif
    if (!njobs) return;
    if (one_shot && allow_short) hipLaunchKernelGGL((k_inflate<true, 0>), dim3(njobs), dim3(64), 0, st, in, out, jobs, states, njobs);
    else hipLaunchKernelGGL((k_inflate<false, 0>), dim3(njobs), dim3(64), 0, st, in, out, jobs, states, njobs);
}
// chunk jobs of one member: pass 1 (count) or 2 (symbols); `in` = first byte of the member
int knob(const char *name, int dflt);
// chunk jobs of the symbol pass per CU the chunks are sized for in inflate_members_parallel: 8, or 10 with the DENSE register budget
// (three wavefronts per SIMD).  Round 5 measured the dense build with chunks cut for 10 jobs per CU on every shape
// (profiles/r05/dense_ab.log): one 1 GiB member 46.7 -> 43.3 ms, 64 x 4 MiB members 17.6 -> 26.9 ms, everything else within 3 % — so it is
// the form of ONE long member and of nothing else (round 6: `dense` below; inflate_members_parallel decides).
int inflate_slots_per_cu(bool dense) {
    const int v = SZL_LABKNOB("SZL_INF_SLOTS_PER_CU", 0);
    return v >= 1 ? v : (dense ? 10 : 8);
}
void launch_inflate_chunks(const uint8_t *in, InfJob *jobs, InfState *states, uint32_t njobs, int pass, hipStream_t st, bool dense) {
    if (!njobs) return;
    if (pass == 1) hipLaunchKernelGGL((k_inflate<true, 1>), dim3(njobs), dim3(64), 0, st, in, (uint8_t *)nullptr, jobs, states, njobs);
    else if (dense || SZL_LABKNOB("SZL_INF_DENSE", 0) != 0) hipLaunchKernelGGL((k_inflate<true, 2, true>), dim3(njobs), dim3(64), 0, st, in, (uint8_t *)nullptr, jobs, states, njobs);
    else hipLaunchKernelGGL((k_inflate<true, 2>), dim3(njobs), dim3(64), 0, st, in, (uint8_t *)nullptr, jobs, states, njobs);
}

} // namespace szl
