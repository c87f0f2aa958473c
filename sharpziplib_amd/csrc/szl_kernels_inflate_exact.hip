// szl_kernels_inflate_exact.hip — the reference's Inflater step for step on ONE lane, bit buffer included: the decoder for the blocks
// on which "the same bytes as a canonical decoder" is not what the reference produces.
//
// When a dynamic block's literal/length or distance set is incomplete AND holds codes of 10+ bits, InflaterHuffmanTree's lookup table
// is not a decoder of the canonical code (csrc/szl_inflate_reftree.h builds that very table): unassigned second-level slots read as
// "symbol 0, 0 bits", and the long codes of the last, partial 9-bit prefix sit in the PRIMARY table with bit counts of 10-15.  GetSymbol
// drops such a count after peeking only 9 bits (C/InflaterHuffmanTree.cs:184-196) — and StreamManipulator.DropBits does not look
// (CS/StreamManipulator.cs:86-90): bitsInBuffer_ goes negative, the next PeekBits shifts the 16 bits it loads by a negative count
// (C# masks the count to 5 bits, :37-44), and for the next few tokens the reference decodes bits that are not in the stream.  Garbage in,
// garbage out — but the same garbage.  k_inflate (szl_kernels_inflate.hip) therefore stops in front of such a block (INF_EXACT) and
// this kernel takes over: StreamManipulator (32-bit buffer, 16-bit loads, the odd first byte of SetInput :244-262), InflaterDynHeader
// (C/InflaterDynHeader.cs:42-120), InflaterHuffmanTree, Inflater.Decode / DecodeHuffman (C/Inflater.cs:283-552) and OutputWindow's
// copies (CS/OutputWindow.cs) as the reference runs them.  It hands the stream back (INF_RUNNING at a block header) once the bit buffer
// holds nothing but stream bits again; corrupt input only, so speed is not a goal: one lane decodes, 64 lanes fill and copy.
//
// One thing the device cannot know: whether the reference's buffer held 16 more bits than the position implies when the block began
// (it loads 16 bits whenever a peek finds too few, so that depends on the peeks of the tokens before).  The state starts "lazy": the
// first PeekBits that needs the load settles it (the dynamic header peeks dozens of times), and a drop that would underflow before
// that is taken as not underflowing.
#include <hip/hip_runtime.h>
#include "szl_internal.h"
#include "szl_inflate.h"
#include "szl_inflate_reftree.h"

namespace szl {

enum : int { EX_BLOCKS = 2, EX_STORED_LEN1, EX_STORED_LEN2, EX_STORED, EX_DYN_HEADER, EX_HUFFMAN, EX_HUFFMAN_LENBITS, EX_HUFFMAN_DIST,
             EX_HUFFMAN_DISTBITS, EX_CHKSUM, EX_FINISHED };   // C/Inflater.cs:84-96 (DECODE_HEADER / DECODE_DICT stay with k_inflate)

__constant__ uint8_t c_meta_order_x[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};   // C/InflaterDynHeader.cs:23-24

struct ExLds {
    int16_t meta[512], litlen[RT_CAP_LITLEN], dist[RT_CAP_DIST];
    uint8_t lens[320];
    uint32_t wk[32];
};

// One wavefront per job.  `which[b]` = index of the job (and of its InfState / ExState).
__global__ __launch_bounds__(64) void k_inflate_exact(const uint8_t *__restrict__ in_base, uint8_t *__restrict__ out_base, InfJob *jobs, InfState *states,
                                                      ExState *exs, const uint32_t *which, uint32_t n) {
    __shared__ ExLds S;
    if (blockIdx.x >= n) return;
    const uint32_t ji = which[blockIdx.x];
    const int lane = threadIdx.x;
    InfJob job = jobs[ji];
    InfState *st = &states[ji];
    ExState *ex = &exs[blockIdx.x];
    uint8_t *out = out_base + job.out_off;
    const uint64_t out_start = st->outpos, out_limit = st->outpos + job.out_cap;
    uint64_t outpos = st->outpos;
    uint8_t *win = job.window;                       // streaming object: the 32 KiB ring (index = position & 32767); one-shot: the output region is the history
    ExSM sm;
    sm.in = in_base + job.in_off; sm.we = job.in_len;
    int mode, neededBits, repLength, repDist, uncomprLen, isLastBlock, trees, blocks_done = 0;
    uint32_t readAdler;
    int dh_step, dh_ll, dh_d, dh_m, dh_n, dh_i, dh_index, dh_symbol, dh_len;
    if (ex->init == 0) {   // fresh: the stream stands at a block header at st->bitpos
        const uint64_t bp = st->bitpos;
        const uint32_t r = (uint32_t)((8u * (uint32_t)(job.in_len & 1) - (uint32_t)bp) & 15u);
        sm.ws = (bp + r) >> 3; sm.bits = (int32_t)r; sm.lazy = 1; sm.dirty = 0;
        uint32_t b = 0;                               // the r stream bits in front of the next 16-bit load
        for (uint32_t k = 0; k < r; k++) { const uint64_t q = bp + k; if ((q >> 3) < job.in_len && ((sm.in[q >> 3] >> (q & 7)) & 1)) b |= 1u << k; }
        sm.buffer = b;
        mode = EX_BLOCKS; neededBits = repLength = repDist = uncomprLen = 0; isLastBlock = (int)st->last; trees = 0; readAdler = 0;
        dh_step = dh_ll = dh_d = dh_m = dh_n = dh_i = dh_index = dh_symbol = dh_len = 0;
    } else {
        sm.ws = ex->ws; sm.bits = ex->bits; sm.buffer = ex->buffer; sm.lazy = ex->lazy; sm.dirty = ex->dirty;
        mode = ex->mode; neededBits = ex->neededBits; repLength = ex->repLength; repDist = ex->repDist; uncomprLen = ex->uncomprLen;
        isLastBlock = ex->isLastBlock; trees = ex->trees; readAdler = ex->readAdler;
        dh_step = ex->dh_step; dh_ll = ex->dh_ll; dh_d = ex->dh_d; dh_m = ex->dh_m; dh_n = ex->dh_n; dh_i = ex->dh_i; dh_index = ex->dh_index;
        dh_symbol = ex->dh_symbol; dh_len = ex->dh_len;
        for (int i = lane; i < 320; i += 64) S.lens[i] = ex->lens[i];
        for (int i = lane; i < 512; i += 64) S.meta[i] = ex->meta[i];
        for (int i = lane; i < RT_CAP_LITLEN; i += 64) S.litlen[i] = ex->litlen[i];
        for (int i = lane; i < RT_CAP_DIST; i += 64) S.dist[i] = ex->dist[i];
    }
    __syncthreads();
    auto hist = [&](uint64_t p) -> uint8_t {          // output byte at stream position p < outpos (zeros in front of the stream: a fresh OutputWindow)
        if (win) return win[p & 32767];
        return p >= out_start ? out[p - out_start] : (uint8_t)0;   // (one-shot: out_start == 0)
    };
    auto put = [&](uint64_t p, uint8_t v) { out[p - out_start] = v; if (win) win[p & 32767] = v; };
    int status = INF_RUNNING;
    // requests of lane 0 to the wavefront: 1 = fill `fill_n` zeros at outpos, 2 = copy `fill_n` input bytes from `copy_from`
    int req = 0; uint64_t fill_n = 0, copy_from = 0;
    uint32_t budget = 1u << 22;                      // tokens per launch (the host launches again: no call runs for minutes)
    while (status == INF_RUNNING) {
        req = 0;
        if (lane == 0) {
            for (;;) {
                if (budget == 0) { status = INF_EXACT; break; }
                budget--;
                if (mode == EX_BLOCKS) {             // C/Inflater.cs:438-489
                    if (blocks_done && sm.dirty <= 0 && sm.bits >= 0 && !sm.lazy) { status = INF_CHUNK_END; break; }   // clean again: back to k_inflate
                    if (isLastBlock) {
                        if (!job.zlib) { mode = EX_FINISHED; status = INF_FINISHED; break; }
                        sm.buffer >>= (sm.bits & 7); sm.bits &= ~7;                       // SkipToByteBoundary :142-146
                        neededBits = 32; mode = EX_CHKSUM; continue;
                    }
                    const int type = ex_peek(sm, 3);
                    if (type < 0) { status = INF_NEED_INPUT; break; }
                    ex_drop(sm, 3);
                    isLastBlock |= type & 1;
                    blocks_done++;
                    if ((type >> 1) == 0) { sm.buffer >>= (sm.bits & 7); sm.bits &= ~7; mode = EX_STORED_LEN1; }
                    else if ((type >> 1) == 1) {
                        for (int i = 0; i < 288; i++) S.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));   // C/InflaterHuffmanTree.cs:34-70
                        rt_build(S.lens, 288, S.litlen, RT_CAP_LITLEN, S.wk, S.wk + 16);
                        for (int i = 0; i < 32; i++) S.lens[i] = 5;
                        rt_build(S.lens, 32, S.dist, RT_CAP_DIST, S.wk, S.wk + 16);
                        trees = 1; mode = EX_HUFFMAN;
                    } else if ((type >> 1) == 2) {
                        for (int i = 0; i < 320; i++) S.lens[i] = 0;
                        dh_step = 0; mode = EX_DYN_HEADER;
                    } else { status = SZL_E_UNKNOWN_BLOCK; break; }
                    continue;
                }
                if (mode == EX_STORED_LEN1) {
                    if ((uncomprLen = ex_peek(sm, 16)) < 0) { status = INF_NEED_INPUT; break; }
                    ex_drop(sm, 16); mode = EX_STORED_LEN2; continue;
                }
                if (mode == EX_STORED_LEN2) {
                    const int nlen = ex_peek(sm, 16);
                    if (nlen < 0) { status = INF_NEED_INPUT; break; }
                    ex_drop(sm, 16);
                    if (nlen != (uncomprLen ^ 0xffff)) { status = SZL_E_BROKEN_STORED; break; }
                    mode = EX_STORED; continue;
                }
                if (mode == EX_STORED) {             // OutputWindow.CopyStored :100-128 + StreamManipulator.CopyBytes :183-225
                    if (uncomprLen == 0) { mode = EX_BLOCKS; continue; }
                    if (outpos >= out_limit) { status = INF_OUTPUT_FULL; break; }
                    int64_t length = uncomprLen;
                    const int64_t avail = ex_available_bytes(sm), room = (int64_t)(out_limit - outpos);
                    if (length > avail) length = avail;
                    if (length > room) length = room;
                    if (length <= 0) { status = sm.ws >= sm.we ? INF_NEED_INPUT : INF_OUTPUT_FULL; break; }
                    int64_t done = 0;
                    while (sm.bits > 0 && done < length) { put(outpos + (uint64_t)done, (uint8_t)sm.buffer); sm.buffer >>= 8; sm.bits -= 8; done++; if (sm.dirty > 0) sm.dirty -= 8; }
                    outpos += (uint64_t)done; uncomprLen -= (int)done;
                    int64_t rest = length - done;
                    const int64_t inwin = (int64_t)sm.we - (int64_t)sm.ws;
                    if (rest > inwin) rest = inwin;
                    if (rest > 0) { req = 2; fill_n = (uint64_t)rest; copy_from = sm.ws; sm.ws += (uint64_t)rest; uncomprLen -= (int)rest; }
                    else if (uncomprLen == 0) { mode = EX_BLOCKS; continue; }
                    else { status = sm.ws >= sm.we ? INF_NEED_INPUT : INF_OUTPUT_FULL; }
                    break;                           // (the odd byte goes into the buffer once the copy is made, below)
                }
                if (mode == EX_DYN_HEADER) {         // C/InflaterDynHeader.cs:42-120
                    int bits;
                    if (dh_step == 0) { if ((bits = ex_peek(sm, 5)) < 0) { status = INF_NEED_INPUT; break; } ex_drop(sm, 5); dh_ll = bits + 257; dh_step = 1; continue; }
                    if (dh_step == 1) { if ((bits = ex_peek(sm, 5)) < 0) { status = INF_NEED_INPUT; break; } ex_drop(sm, 5); dh_d = bits + 1; dh_step = 2; continue; }
                    if (dh_step == 2) {
                        if ((bits = ex_peek(sm, 4)) < 0) { status = INF_NEED_INPUT; break; }
                        ex_drop(sm, 4); dh_m = bits + 4; dh_n = dh_ll + dh_d;
                        if (dh_ll > 286 || dh_d > 30 || dh_m > 19) { status = SZL_E_DYN_HEADER; break; }
                        dh_i = 0; dh_step = 3; continue;
                    }
                    if (dh_step == 3) {
                        bool need = false;
                        while (dh_i < dh_m) {
                            if ((bits = ex_peek(sm, 3)) < 0) { need = true; break; }
                            ex_drop(sm, 3);
                            S.lens[c_meta_order_x[dh_i]] = (uint8_t)bits; dh_i++;
                        }
                        if (need) { status = INF_NEED_INPUT; break; }
                        {   // new InflaterHuffmanTree(codeLengths): over-subscribed => IndexOutOfRange out of BitReverse
                            uint32_t kraft = 0;
                            for (int i = 0; i < 19; i++) if (S.lens[i]) kraft += 1u << (16 - S.lens[i]);
                            if (kraft > 65536u || rt_build(S.lens, 19, S.meta, 512, S.wk, S.wk + 16) < 0) { status = SZL_E_CODE_OVERSUBSCRIBED; break; }
                        }
                        for (int i = 0; i < 19; i++) S.lens[i] = 0;   // (the code lengths proper are written over the same array from index 0 on)
                        dh_index = 0; dh_step = 4; continue;
                    }
                    if (dh_step == 4) {
                        if (dh_index >= dh_n) { dh_step = 6; continue; }
                        const int symbol = ex_get_symbol(S.meta, sm);
                        if (symbol == -1) { status = INF_NEED_INPUT; break; }
                        if (symbol == -2) { status = SZL_E_CODELEN_ZERO; break; }
                        if (symbol < 16) { S.lens[dh_index++] = (uint8_t)symbol; continue; }
                        dh_symbol = symbol;
                        if (symbol == 16) { if (dh_index == 0) { status = SZL_E_DYN_HEADER; break; } dh_len = S.lens[dh_index - 1]; } else dh_len = 0;
                        dh_step = 5; continue;
                    }
                    if (dh_step == 5) {
                        const int nb = dh_symbol == 16 ? 2 : (dh_symbol == 17 ? 3 : 7), base = dh_symbol == 18 ? 11 : 3;
                        if ((bits = ex_peek(sm, nb)) < 0) { status = INF_NEED_INPUT; break; }
                        ex_drop(sm, nb);
                        int rep = bits + base;
                        if (dh_index + rep > dh_n) { status = SZL_E_DYN_HEADER; break; }
                        while (rep-- > 0) S.lens[dh_index++] = (uint8_t)dh_len;
                        dh_step = 4; continue;
                    }
                    // dh_step == 6
                    if (S.lens[256] == 0) { status = SZL_E_DYN_HEADER; break; }
                    {
                        uint32_t k1 = 0, k2 = 0;
                        for (int i = 0; i < dh_ll; i++) if (S.lens[i]) k1 += 1u << (16 - S.lens[i]);
                        for (int i = 0; i < dh_d; i++) if (S.lens[dh_ll + i]) k2 += 1u << (16 - S.lens[dh_ll + i]);
                        if (k1 > 65536u || rt_build(S.lens, dh_ll, S.litlen, RT_CAP_LITLEN, S.wk, S.wk + 16) < 0 ||
                            k2 > 65536u || rt_build(S.lens + dh_ll, dh_d, S.dist, RT_CAP_DIST, S.wk, S.wk + 16) < 0) { status = SZL_E_CODE_OVERSUBSCRIBED; break; }
                    }
                    trees = 2; mode = EX_HUFFMAN; continue;
                }
                if (mode == EX_CHKSUM) {             // :397-418
                    bool need = false;
                    while (neededBits > 0) {
                        const int b = ex_peek(sm, 8);
                        if (b < 0) { need = true; break; }
                        ex_drop(sm, 8);
                        readAdler = (readAdler << 8) | (uint32_t)b; neededBits -= 8;
                    }
                    if (need) { status = INF_NEED_INPUT; break; }
                    mode = EX_FINISHED; status = INF_FINISHED; break;   // (the host compares with the Adler-32 of what was decoded, as for k_inflate)
                }
                if (mode == EX_FINISHED) { status = INF_FINISHED; break; }
                // ---- DecodeHuffman :283-386
                if (mode == EX_HUFFMAN) {
                    if (outpos >= out_limit) { status = INF_OUTPUT_FULL; break; }       // (every token writes at least one byte)
                    const uint32_t b0 = sm.buffer; const int32_t n0 = sm.bits; const uint64_t w0 = sm.ws;
                    const int symbol = ex_get_symbol(S.litlen, sm);
                    if (symbol == -1) { status = INF_NEED_INPUT; break; }
                    if (symbol == -2) { status = SZL_E_CODELEN_ZERO; break; }
                    if (symbol < 256) {
                        if (symbol == 0 && sm.buffer == b0 && sm.bits == n0 && sm.ws == w0) {   // "symbol 0, 0 bits": nothing changes any more —
                            req = 1; fill_n = out_limit - outpos; break;                           // zeros to the end of the room, all lanes
                        }
                        put(outpos, (uint8_t)symbol); outpos++;
                        continue;
                    }
                    if (symbol == 256) { trees = 0; mode = EX_BLOCKS; continue; }
                    if (symbol - 257 >= 29) { status = SZL_E_ILLEGAL_LEN_CODE; break; }
                    const int ls = symbol - 257;
                    const int xl = (ls < 8 || ls == 28) ? 0 : ((ls - 4) >> 2);
                    repLength = ls < 8 ? 3 + ls : (ls == 28 ? 258 : 3 + ((4 + (ls & 3)) << xl));   // CPLENS / CPLEXT :39-48
                    neededBits = xl; mode = EX_HUFFMAN_LENBITS;
                }
                if (mode == EX_HUFFMAN_LENBITS) {
                    if (neededBits > 0) {
                        const int i = ex_peek(sm, neededBits);
                        if (i < 0) { status = INF_NEED_INPUT; break; }
                        ex_drop(sm, neededBits); repLength += i;
                    }
                    mode = EX_HUFFMAN_DIST;
                }
                if (mode == EX_HUFFMAN_DIST) {
                    const int symbol = ex_get_symbol(S.dist, sm);
                    if (symbol == -1) { status = INF_NEED_INPUT; break; }
                    if (symbol == -2) { status = SZL_E_CODELEN_ZERO; break; }
                    if (symbol >= 30) { status = SZL_E_ILLEGAL_DIST_CODE; break; }
                    const int xd = symbol < 4 ? 0 : ((symbol >> 1) - 1);
                    repDist = symbol < 4 ? 1 + symbol : 1 + ((2 + (symbol & 1)) << xd);            // CPDIST / CPDEXT :50-68
                    neededBits = xd; mode = EX_HUFFMAN_DISTBITS;
                }
                if (mode == EX_HUFFMAN_DISTBITS) {
                    if (neededBits > 0) {
                        const int i = ex_peek(sm, neededBits);
                        if (i < 0) { status = INF_NEED_INPUT; break; }
                        ex_drop(sm, neededBits); repDist += i; neededBits = 0;
                    }
                    if (outpos + (uint64_t)repLength > out_limit) { status = INF_OUTPUT_FULL; break; }   // (stays in this mode: the copy is made by the next call)
                    for (int k = 0; k < repLength; k++) put(outpos + (uint64_t)k, hist(outpos + (uint64_t)k - (uint64_t)repDist));   // OutputWindow.Repeat :63-92
                    outpos += (uint64_t)repLength;
                    mode = EX_HUFFMAN;
                    continue;
                }
            }
        }
        // ---- the wavefront: lane 0's request
        req = __builtin_amdgcn_readfirstlane(req);
        status = __builtin_amdgcn_readfirstlane(status);
        if (req) {
            const uint64_t nlo = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)fill_n), nhi = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(fill_n >> 32));
            const uint64_t cnt = nlo | (nhi << 32);
            const uint64_t plo = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)outpos), phi = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(outpos >> 32));
            const uint64_t p0 = plo | (phi << 32);
            const uint64_t flo = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)copy_from), fhi = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(copy_from >> 32));
            const uint64_t from = flo | (fhi << 32);
            for (uint64_t i = lane; i < cnt; i += 64) {
                const uint8_t v = req == 2 ? sm.in[from + i] : (uint8_t)0;
                out[p0 + i - out_start] = v;
                if (win) win[(p0 + i) & 32767] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (lane == 0) {
                outpos += cnt;
                if (req == 1) status = INF_OUTPUT_FULL;
                else {
                    if (((sm.ws - sm.we) & 1) != 0) { sm.buffer = sm.ws < sm.we ? sm.in[sm.ws] : 0u; sm.ws++; sm.bits = 8; }   // CopyBytes :216-222
                    if (uncomprLen == 0) mode = EX_BLOCKS;
                    else if (sm.ws >= sm.we) status = INF_NEED_INPUT;     // Decode() returns !IsNeedingInput (:527)
                }
            }
            status = __builtin_amdgcn_readfirstlane(status);
        }
    }
    // ---- save: the exact state, and what k_inflate / the host read
    if (lane == 0) {
        ex->init = 1; ex->ws = sm.ws; ex->bits = sm.bits; ex->buffer = sm.buffer; ex->lazy = sm.lazy; ex->dirty = sm.dirty;
        ex->mode = mode; ex->neededBits = neededBits; ex->repLength = repLength; ex->repDist = repDist; ex->uncomprLen = uncomprLen;
        ex->isLastBlock = isLastBlock; ex->trees = trees; ex->readAdler = readAdler;
        ex->dh_step = dh_step; ex->dh_ll = dh_ll; ex->dh_d = dh_d; ex->dh_m = dh_m; ex->dh_n = dh_n; ex->dh_i = dh_i; ex->dh_index = dh_index;
        ex->dh_symbol = dh_symbol; ex->dh_len = dh_len;
        const int64_t consumed = (int64_t)sm.ws - (int64_t)(sm.bits >> 3);           // TotalIn = given - AvailableBytes (:131, C/Inflater.cs:862-884)
        st->outpos = outpos; st->last = (uint32_t)isLastBlock; st->status = status; st->adler_read = readAdler;
        if (status == INF_CHUNK_END) {                                               // clean at a block header: k_inflate goes on from here
            st->bitpos = 8 * sm.ws - (uint64_t)sm.bits; st->mode = INF_M_HEADER; st->stored_left = 0;
            ex->init = 0;
        } else {
            st->bitpos = consumed > 0 ? 8 * (uint64_t)consumed : 0;
            st->mode = status == INF_FINISHED ? (uint32_t)INF_M_DONE : (uint32_t)INF_M_HEADER;
        }
        jobs[ji].out_written = outpos - out_start;
        jobs[ji].status = status;
        jobs[ji].consumed = consumed < 0 ? 0 : ((uint64_t)consumed > job.in_len ? job.in_len : (uint64_t)consumed);
        jobs[ji].end_bit = st->bitpos;
    }
    __syncthreads();
    for (int i = lane; i < 320; i += 64) ex->lens[i] = S.lens[i];
    for (int i = lane; i < 512; i += 64) ex->meta[i] = S.meta[i];
    for (int i = lane; i < RT_CAP_LITLEN; i += 64) ex->litlen[i] = S.litlen[i];
    for (int i = lane; i < RT_CAP_DIST; i += 64) ex->dist[i] = S.dist[i];
}

void launch_inflate_exact(const uint8_t *in, uint8_t *out, InfJob *jobs, InfState *states, ExState *exs, const uint32_t *which, uint32_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_inflate_exact, dim3(n), dim3(64), 0, st, in, out, jobs, states, exs, which, n);
}

} // namespace szl
