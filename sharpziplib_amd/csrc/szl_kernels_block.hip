// szl_kernels_block.hip — stage D: DeflaterHuffman (C/DeflaterHuffman.cs) on the device.
//   k_seg_blocks   block count per segment            C/DeflaterEngine.cs:841-852, :750-768
//   k_block_build  per block: tally (TallyLit/TallyDist :873,:894), BuildTree x3 (:196), BuildLength (:475),
//                  CalcBLFreq (:349), opt_len/static_len/stored decision (FlushBlock :788-857),
//                  BuildCodes (:151), header rendering (SendAllTrees :676, WriteTree :411)
//   k_block_scan   bit offset of every block inside its segment's output (PendingBuffer.WriteBits/AlignToByte
//                  C/PendingBuffer.cs:168,:143 are a running bit position)
//   k_block_encode CompressBlock (:701) / FlushStoredBlock (:766): tokens -> bits at their final position
//   k_seg_finish   Deflater.Deflate's FLUSHING/FINISHING tails (C/Deflater.cs:486-517)
// The Huffman construction is a sequence of heap operations whose tie-breaking IS the specification (value<<8|depth keys, strict
// comparisons, forced second symbol, overflow repair in childs[] order); since round 5 a wavefront evaluates each operation level-
// parallel and leaves the reference's heap array behind after every one of them (build_tree below).
#include <hip/hip_runtime.h>
#include "szl_internal.h"

namespace szl {

__device__ __forceinline__ int64_t base_of_b(int64_t s_abs) {
    int64_t idx = s_abs + 1;
    if (idx <= 65273) return 0;
    return ((idx - 65273 + 32767) >> 15) << 15;
}

__device__ __forceinline__ int lcode_of(int l /* len-3 */) { // Lcode :932
    if (l == 255) return 285;
    if (l < 8) return 257 + l;
    int k = 31 - __builtin_clz((unsigned)l);
    return 257 + 4 * (k - 2) + (l >> (k - 2));
}
__device__ __forceinline__ int dcode_of(int dd /* dist-1 */) { // Dcode :948
    if (dd < 4) return dd;
    int k = 31 - __builtin_clz((unsigned)dd);
    return 2 * (k - 1) + (dd >> (k - 1));
}
__device__ __forceinline__ uint32_t bitrev16(uint32_t v) { return __builtin_bitreverse32(v) >> 16; } // BitReverse :924

__device__ __forceinline__ void static_lit(int sym, uint32_t &code, int &len) { // static ctor :602-631
    if (sym < 144) { code = bitrev16((0x030 + sym) << 8); len = 8; }
    else if (sym < 256) { code = bitrev16((0x190 - 144 + sym) << 7); len = 9; }
    else if (sym < 280) { code = bitrev16((0x000 - 256 + sym) << 9); len = 7; }
    else { code = bitrev16((0x0c0 - 280 + sym) << 8); len = 8; }
}

__global__ void k_seg_blocks(const SegDev *segs, uint32_t nseg, const uint32_t *tokens, const uint64_t *blk_off, SegOut *so, int fast) {
    uint32_t si = blockIdx.x * blockDim.x + threadIdx.x;
    if (si >= nseg) return;
    const uint64_t T = so[si].tok_count;
    uint64_t full = T / BLOCK_TOKENS, rem = T % BLOCK_TOKENS;
    uint64_t nb = full;
    if (segs[si].flags & SEG_SWITCH_CUT) { if (rem > 0) nb++; }   // FlushBlock only `if (strstart > blockStart)` (C/DeflaterEngine.cs:336,:345): no empty block, no sync block
    else if (rem > 0 || T == 0) nb++;
    else if (fast) { if (!segs[si].finish) nb++; } // DeflateFast: the full block was the last one iff finishing (:729); a flush adds an empty block (:664)
    else if ((tokens[so[si].tok_first + T - 1] >> 16) != 0 && !segs[si].finish) nb++; // sync flush right after a full block
    so[si].blk_first = (uint32_t)blk_off[si];
    so[si].blk_count = (uint32_t)nb;
}

// ------------------------------------------------------------------------------------------------
// Tree construction in LDS by ONE WAVEFRONT per tree (round 5; through round 4 a single lane ran the reference's loops verbatim).
//
// Tree.BuildTree (C/DeflaterHuffman.cs:196-329) is a binary heap keyed by value << 8 | depth with strict comparisons; ties are broken
// by heap POSITION, so the tree — and with equal frequencies the code length of each symbol — can only be reproduced by reproducing
// the heap array after every operation.  What can be done differently is how ONE operation is evaluated.  The reference's sift
// (:264-291, :299-323) moves the hole at the root down to a leaf along the smaller children (ties: the left one), then lets the
// inserted value rise from that leaf while its parent is strictly greater: two dependent LDS round trips per level, 572 sifts per
// literal tree, 0.37 ms per block on one lane.  Here (tools/heap_model.c checks the form against the reference's on 200 000 random
// trees with heavy ties, after every sift):
//   1. the root-to-leaf path of smaller children is read off one "winner" bit per internal node (right child exists and is strictly
//      smaller) — the bits of nodes 32 L .. 32 L + 31 live in a register of lane L and a step of the path is a v_readlane and three
//      scalar instructions; lane k notes the path's node at level k;
//   2. lanes 1 .. m read their node's value and owner, its sibling's value and the value below it, all in ONE LDS round trip;
//   3. the path's values are non-decreasing downwards, so the inserted value lands at level j = popcount(ballot(value_k <= inserted));
//      values 1 .. j move up one level (lane k writes its parent's slot), lane j writes the inserted value;
//   4. the winner bits of levels 0 .. j-1 follow from what the lanes hold (new path child against the sibling).
// The heap array after a sift is the reference's, operation by operation; the leaf insertion (:205-227), BuildLength's repair of
// over-long codes (:519-571) and everything around stay the reference's loops on lane 0.  BuildLength's depth walk (:491-510) —
// "a node's children are one deeper than the node", in reverse creation order — is evaluated for all nodes at once from parent
// links (the counts it produces do not depend on the order).
template <int N> // N = number of symbols of the largest tree built in this scratch
struct TreeScratchT {
    int heap[N];
    int hval[N];             // values[heap[i]] cached next to heap[i]
    short childs[4 * N];
    int values[2 * N];
    unsigned char lengths[2 * N];
    short parent[2 * N];     // BuildLength: the node a node hangs under
};
using TreeScratch = TreeScratchT<LIT_NUM>;       // literal/length tree (286 symbols)
using TreeScratchSmall = TreeScratchT<DIST_NUM>; // distance tree (30) and code-length tree (19)

#if SZL_LAB   // (laboratory library: where one block's k_block_build spends its time — 100 MHz stamps of workgroup 0, tools/lab/block_build_phases.py)
__device__ unsigned long long g_bb_stamps[16];
#define BB_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (k) >= 0) g_bb_stamps[(k)] = wall_clock64(); } while (0)
#else
#define BB_STAMP(k) do { } while (0)
#endif
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// one sift: the hole is at the root of heap[0 .. heapLen), (last, lastVal) is inserted.  `w`: this lane's 32 winner bits.
__device__ __forceinline__ void sift_wave(int *heap, int *hval, int heapLen, int last, int lastVal, uint32_t &w, int lane) {
    // 1. the path, in scalar registers: node + 1 at level k is the binary number "1 b0 b1 .. b(k-1)" of the winner bits met on the way,
    //    so the deepest node's number holds the whole path and lane k reads its own node off it with a shift
    int code = 1, m = 0;                                            // code = node + 1
    while (2 * code - 1 < heapLen) {                                // (node has a left child: 2 * node + 1 < heapLen)
        const int n = code - 1;
        const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)w, n >> 5);
        code = 2 * code + (int)((word >> (n & 31)) & 1u);
        ++m;
    }
    const bool on_path = lane >= 1 && lane <= m;
    const int mypos = lane <= m ? (code >> (m - lane)) - 1 : 0;     // the path's node at level `lane`
    const int below = lane < m ? (code >> (m - lane - 1)) - 1 : 0;  // ... and at the level below
    int v = 0, id = 0, sv = 0, vbelow = 0;
    bool has_sib = false;
    if (on_path) {                                                  // 2. one round trip
        const int sib = (mypos & 1) ? mypos + 1 : mypos - 1;
        has_sib = sib < heapLen;
        v = hval[mypos]; id = heap[mypos];
        if (has_sib) sv = hval[sib];
        if (lane < m) vbelow = hval[below];
    }
    const int j = __popcll(__ballot(on_path && v <= lastVal));      // 3. where the inserted value lands
    // 4. winner bit of my parent (levels 0 .. j-1 change): its path child now holds the value below me (k < j) or the inserted one (k == j)
    const int newchild = lane < j ? vbelow : lastVal;
    const bool bit = (mypos & 1) ? (has_sib && newchild > sv) : (sv > newchild);   // left child = path child : left child = sibling
    if (on_path && lane <= j) { const int pp = (mypos - 1) >> 1; heap[pp] = id; hval[pp] = v; }
    if (lane == j) { heap[mypos] = last; hval[mypos] = lastVal; }   // (lane 0 notes level 0: the root)
    const uint32_t bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__ballot(bit));   // (lane k's bit belongs to the path's node at level k - 1 — a shift of `code`; the path is at most 9 levels deep)
    for (int k = 1; k <= j; k++) {
        const int nd = __builtin_amdgcn_readfirstlane((code >> (m - k + 1)) - 1);
        const uint32_t b = (bits >> k) & 1u;
        if (lane == (nd >> 5)) w = (w & ~(1u << (nd & 31))) | (b << (nd & 31));
    }
    wave_sync();
}

// all 64 lanes of ONE wavefront call this with the same arguments; `lane` = lane id
template <typename SCR>
__device__ void build_tree(const int *freqs, int numSymbols, int minNumCodes, int maxLength, SCR *S,
                           unsigned char *length /*[numSymbols]*/, int *bl_counts /*[15]*/, int *numCodesOut, int lane, int sb = -1) {
    int *heap = S->heap, *hval = S->hval, *values = S->values;
    short *childs = S->childs;
    int heapLen = 0, maxCode = 0;
    // :205-241 — the leaves enter the heap one after the other, in symbol order, each sifted UP from the end of the array.  The nodes it may
    // pass are its ancestors, whose positions follow from its own: lane k reads the ancestor k levels up (one round trip for all of them),
    // the ones with a larger frequency — the lower end of the path, by the heap's own order — each move down a level and the leaf takes the
    // place of the topmost.  (One lane walking up level by level, two dependent LDS reads a level, was 90 of a literal tree's 300 us.)
    for (int base = 0; base < numSymbols; base += 64) {
        const int myf = base + lane < numSymbols ? freqs[base + lane] : 0;
        for (unsigned long long nz = __ballot(myf != 0); nz; nz &= nz - 1) {
            const int b = __builtin_ctzll(nz);
            const int freq = __builtin_amdgcn_readlane(myf, b);
            const int code = heapLen + 1;                           // position + 1: the ancestors' are its right shifts
            heapLen++;
            const bool valid = lane >= 1 && lane < 31 && (code >> lane) >= 1;
            const int anc = valid ? (code >> lane) - 1 : 0;
            const int av = valid ? hval[anc] : 0, an = valid ? heap[anc] : 0;
            const bool gt = valid && av > freq;                     // :219 "freqs[heap[ppos]] > freq"
            const int cnt = __popcll(__ballot(gt));
            if (gt) { const int to = (code >> (lane - 1)) - 1; heap[to] = an; hval[to] = av; }
            if (lane == 0) { const int at = (code >> cnt) - 1; heap[at] = base + b; hval[at] = freq; }
            maxCode = base + b;
            wave_sync();
        }
    }
    BB_STAMP(sb);
    if (lane == 0) {
        while (heapLen < 2) {
            int node = maxCode < 2 ? ++maxCode : 0;
            heap[heapLen++] = node;
        }
        *numCodesOut = (maxCode + 1 > minNumCodes) ? maxCode + 1 : minNumCodes;
    }
    wave_sync();
    heapLen = __builtin_amdgcn_readfirstlane(heapLen);
    const int numLeafs = heapLen;
    const int childsLen = 4 * heapLen - 2;
    int numNodes = numLeafs;
    for (int i = lane; i < heapLen; i += 64) {                       // :243-252
        const int node = heap[i];
        childs[2 * i] = (short)node;
        childs[2 * i + 1] = -1;
        const int v = freqs[node] << 8;
        values[i] = v;
        hval[i] = v;
    }
    wave_sync();
    for (int i = lane; i < heapLen; i += 64) heap[i] = i;
    wave_sync();
    uint32_t w = 0;                                                  // winner bits of nodes 32 * lane ..
    for (int c = 0; 64 * c < heapLen; c++) {
        const int nd = 64 * c + lane, l = 2 * nd + 1, r = l + 1;
        const bool bit = r < heapLen && hval[l] > hval[r];
        const unsigned long long mask = __ballot(bit);
        if (lane == 2 * c) w = (uint32_t)mask;
        if (lane == 2 * c + 1) w = (uint32_t)(mask >> 32);
    }
    do {                                                             // :258-327
        const int first = heap[0], firstVal = hval[0];
        --heapLen;
        int last = heap[heapLen], lastVal = hval[heapLen];
        {   // the slot that left the heap: its parent has lost its right child, or its only one
            const int p = (heapLen - 1) >> 1;
            if (heapLen > 0 && lane == (p >> 5)) w &= ~(1u << (p & 31));
        }
        wave_sync();
        sift_wave(heap, hval, heapLen, last, lastVal, w, lane);
        const int second = heap[0], secondVal = hval[0];
        last = numNodes++;
        const int d1 = firstVal & 0xff, d2 = secondVal & 0xff;
        const int mindepth = d1 < d2 ? d1 : d2;
        lastVal = firstVal + secondVal - mindepth + 1;
        if (lane == 0) { childs[2 * last] = (short)first; childs[2 * last + 1] = (short)second; values[last] = lastVal; }
        wave_sync();
        sift_wave(heap, hval, heapLen, last, lastVal, w, lane);
    } while (heapLen > 1);
    BB_STAMP(sb < 0 ? sb : sb + 1);

    // ---- BuildLength :475
    const int nNodes = childsLen / 2;
    unsigned char *lengths = S->lengths;
    short *parent = S->parent;
    for (int i = lane; i < numSymbols; i += 64) length[i] = 0;
    if (lane < maxLength) bl_counts[lane] = 0;
    for (int i = lane; i < nNodes; i += 64) if (childs[2 * i + 1] != -1) { parent[childs[2 * i]] = (short)i; parent[childs[2 * i + 1]] = (short)i; }
    if (lane == 0) parent[nNodes - 1] = -1;
    wave_sync();
    // depth of every node (:491-510: lengths[child] = min(lengths[node] + 1, maxLength)): walk up the parent links, all nodes at once
    int overflow = 0;
    for (int i = lane; i < nNodes; i += 64) {
        int d = 0;
        for (int q = parent[i]; q >= 0; q = parent[q]) d++;
        const int bitLength = d > maxLength ? maxLength : d;
        lengths[i] = (unsigned char)bitLength;
        if (childs[2 * i + 1] != -1) { if (d + 1 > maxLength) overflow++; }                       // an inner node whose children are clamped
        else { atomicAdd(&bl_counts[bitLength - 1], 1); length[childs[2 * i]] = (unsigned char)bitLength; }
    }
    for (int o = 32; o > 0; o >>= 1) overflow += __shfl_xor(overflow, o);
    wave_sync();
    BB_STAMP(sb < 0 ? sb : sb + 2);
    if (overflow == 0) return;
    if (lane == 0) {                                                 // :519-571, the reference's loops
        int incrBitLen = maxLength - 1;
        do {
            while (bl_counts[--incrBitLen] == 0) { }
            do {
                bl_counts[incrBitLen]--;
                bl_counts[++incrBitLen]++;
                overflow -= 1 << (maxLength - 1 - incrBitLen);
            } while (overflow > 0 && incrBitLen < maxLength - 1);
        } while (overflow > 0);
        bl_counts[maxLength - 1] += overflow;
        bl_counts[maxLength - 2] -= overflow;
        int nodePtr = 2 * numLeafs;
        for (int bits = maxLength; bits != 0; bits--) {
            int n = bl_counts[bits - 1];
            while (n > 0) {
                int childPtr = 2 * childs[nodePtr++];
                if (childs[childPtr + 1] == -1) {
                    length[childs[childPtr]] = (unsigned char)bits;
                    n--;
                }
            }
        }
    }
    wave_sync();
}

// CalcBLFreq :349 / WriteTree :411 share this scanner; `emit(sym, extra_val, extra_bits)` is called per bl symbol.
template <typename F>
__device__ void scan_code_lengths(const unsigned char *length, int numCodes, F emit) {
    int max_count, min_count, count, curlen = -1;
    int i = 0;
    while (i < numCodes) {
        count = 1;
        int nextlen = length[i];
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else {
            max_count = 6; min_count = 3;
            if (curlen != nextlen) { emit(nextlen, 0, 0); count = 0; }
        }
        curlen = nextlen;
        i++;
        while (i < numCodes && curlen == length[i]) {
            i++;
            if (++count >= max_count) break;
        }
        if (count < min_count) { while (count-- > 0) emit(curlen, 0, 0); }
        else if (curlen != 0) emit(16, count - 3, 2);
        else if (count <= 10) emit(17, count - 3, 3);
        else emit(18, count - 11, 7);
    }
}

struct BitW { // LSB-first bit writer into an LDS byte array
    unsigned char *p;
    unsigned long long acc;
    int nacc;
    int total;
    __device__ void put(unsigned v, int n) {
        acc |= (unsigned long long)v << nacc;
        nacc += n; total += n;
        while (nacc >= 8) { *p++ = (unsigned char)acc; acc >>= 8; nacc -= 8; }
    }
    __device__ void flush() { if (nacc > 0) { *p++ = (unsigned char)acc; acc = 0; nacc = 0; } }
};

// ---- CalcBLFreq / WriteTree by runs (round 6).  scan_code_lengths above is the reference's loop: one code length per step, each an
// LDS read the next step waits for, four passes over ~316 lengths on one thread — 68 of the 340 us k_block_build takes when the call is
// ONE block.  What the loop emits for a run of `len` equal lengths `v` depends on nothing but (v, len) (curlen differs from the run's
// value when the run begins: runs are maximal, and each of the two scans starts at curlen = -1):
//   v == 0: the run is cut into pieces of 138 (max_count), each one symbol 18 with count - 11; the rest r: r < 3 -> r symbols 0,
//           r <= 10 -> symbol 17 (r - 3), else symbol 18 (r - 11);
//   v != 0: symbol v, then the other len - 1 in pieces of 6, each symbol 16 with count - 3; the rest r: r < 3 -> r symbols v, else 16 (r - 3).
// So a wavefront finds the runs with a ballot, a lane takes a run, and the bit offsets of the header are a prefix sum over the runs.
template <typename F>
__device__ __forceinline__ void run_emit(int v, int len, F emit) {
    if (v == 0) {
        const int full = len / 138, r = len - full * 138;
        for (int k = 0; k < full; k++) emit(18, 127, 7);
        if (r >= 11) emit(18, r - 11, 7);
        else if (r >= 3) emit(17, r - 3, 3);
        else for (int k = 0; k < r; k++) emit(0, 0, 0);
    } else {
        emit(v, 0, 0);
        const int rest = len - 1, full = rest / 6, r = rest - full * 6;
        for (int k = 0; k < full; k++) emit(16, 3, 2);
        if (r >= 3) emit(16, r - 3, 2);
        else for (int k = 0; k < r; k++) emit(v, 0, 0);
    }
}
// all 64 lanes of one wavefront: rs[0 .. R) = first index of every run of length[0 .. n), rs[R] = n; returns R
__device__ __forceinline__ int find_runs(const unsigned char *length, int n, unsigned short *rs, int lane) {
    int R = 0;
    for (int base = 0; base < n; base += 64) {
        const int p = base + lane;
        const bool valid = p < n;
        const int v = valid ? (int)length[p] : -1, pv = (valid && p > 0) ? (int)length[p - 1] : -2;
        const bool st = valid && v != pv;
        const unsigned long long m = __ballot(st);
        if (st) rs[R + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)p;
        R += __popcll(m);
    }
    if (lane == 0) rs[R] = (unsigned short)n;
    wave_sync();
    return R;
}
__device__ __forceinline__ void count_runs(const unsigned char *length, const unsigned short *rs, int R, int *blfreq, int lane) {
    for (int r = lane; r < R; r += 64) {
        const int s = rs[r], e = rs[r + 1];
        run_emit((int)length[s], e - s, [&](int sym, int, int) { atomicAdd(&blfreq[sym], 1); });
    }
}
// n bits of v at bit `pos` of a zeroed LSB-first bit string held in 32-bit words (n <= 16)
__device__ __forceinline__ void or_bits(uint32_t *words, int pos, uint32_t v, int n) {
    if (n == 0) return;
    const unsigned long long x = (unsigned long long)v << (pos & 31);
    atomicOr(&words[pos >> 5], (uint32_t)x);
    if ((pos & 31) + n > 32) atomicOr(&words[(pos >> 5) + 1], (uint32_t)(x >> 32));
}
// the runs' symbols as bits from `bitpos` on; returns the position behind them (all 64 lanes, the same value)
__device__ __forceinline__ int write_runs(const unsigned char *length, const unsigned short *rs, int R, const unsigned short *blcode,
                                          const unsigned char *bllen, uint32_t *words, int bitpos, int lane) {
    for (int rb = 0; rb < R; rb += 64) {
        const int r = rb + lane;
        const bool valid = r < R;
        const int s = valid ? (int)rs[r] : 0, len = valid ? (int)rs[r + 1] - s : 0, v = valid ? (int)length[s] : 0;
        int nb = 0;
        if (valid) run_emit(v, len, [&](int sym, int, int xb) { nb += bllen[sym] + xb; });
        int inc = nb;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        int off = bitpos + inc - nb;
        if (valid) run_emit(v, len, [&](int sym, int xv, int xb) {
            const int cl = bllen[sym];
            or_bits(words, off, (uint32_t)blcode[sym] | ((uint32_t)xv << cl), cl + xb);
            off += cl + xb;
        });
        bitpos += __shfl(inc, 63);
    }
    return bitpos;
}

__constant__ int c_bl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15}; // :37

enum : int { D_THREADS = 128 }; // two wavefronts: one per tree (literal/length, distance); small workgroups keep more blocks in flight per CU

__global__ __launch_bounds__(D_THREADS) void k_block_build(const SegDev *__restrict__ segs, uint32_t nseg,
                                                           const SegOut *__restrict__ so, const uint64_t *__restrict__ blk_off,
                                                           const uint32_t *__restrict__ tokens,
                                                           const int64_t *__restrict__ blk_start_pos,
                                                           const int64_t *__restrict__ blk_lasttok_pos, BlockDesc *descs,
                                                           uint32_t nblk_slots, int fast) {
    __shared__ int lfreq[LIT_NUM + 2], dfreq[DIST_NUM + 2], blfreq[BL_NUM + 1];
    __shared__ unsigned char llen[LIT_NUM + 2], dlen[DIST_NUM + 2], bllen[BL_NUM + 1];
    __shared__ int lblc[15], dblc[15], blblc[15];
    __shared__ int lnum, dnum, blnum, extra_bits, s_sums[4];
    __shared__ unsigned short blcode[BL_NUM + 1];
    __shared__ TreeScratch scrL;
    __shared__ TreeScratchSmall scrD;
    __shared__ uint32_t hdrw[164];                // the rendered header, LSB-first (640 bytes of BlockDesc.hdr and a word to spill into)
    __shared__ unsigned short rsL[LIT_NUM + 4], rsD[DIST_NUM + 4];   // first index of every run of equal code lengths
    __shared__ int s_type, s_hdrbits, s_RL, s_RD, s_bltc;

    const uint32_t gb = blockIdx.x;
    if (gb >= nblk_slots) return;
    // which segment owns block slot gb
    uint32_t lo = 0, hi = nseg - 1;
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (blk_off[mid] <= gb) lo = mid; else hi = mid - 1; }
    const uint32_t si = lo;
    const uint32_t lb = gb - (uint32_t)blk_off[si];
    BlockDesc *bd = &descs[gb];
    const int tid = threadIdx.x;
    if (lb >= so[si].blk_count) { if (tid == 0) bd->type = 0xFFu; return; }
    const SegDev s = segs[si];
    const uint64_t T = so[si].tok_count;
    const uint64_t tfirst = so[si].tok_first + (uint64_t)lb * BLOCK_TOKENS;
    const uint64_t tdone = (uint64_t)lb * BLOCK_TOKENS;
    const int ntok = (int)(T > tdone ? (T - tdone < BLOCK_TOKENS ? T - tdone : BLOCK_TOKENS) : 0);
    const int last = (s.finish && lb == so[si].blk_count - 1) ? 1 : 0;

    BB_STAMP(0);
    for (int i = tid; i < LIT_NUM + 2; i += D_THREADS) lfreq[i] = 0;
    if (tid < DIST_NUM + 2) dfreq[tid] = 0;
    if (tid < BL_NUM + 1) blfreq[tid] = 0;
    if (tid == 0) extra_bits = 0;
    __syncthreads();
    int ex = 0;
    // (eight loads in flight per thread: one load and its wait per step was 128 round trips to memory for a full block — a quarter of the
    // kernel's 340 us when the call is ONE block, round 6)
    constexpr int HB = 8;
    for (int i0 = tid; i0 < ntok; i0 += D_THREADS * HB) {
        uint32_t tk[HB];
#pragma unroll
        for (int u = 0; u < HB; u++) { const int i = i0 + u * D_THREADS; tk[u] = i < ntok ? tokens[tfirst + i] : 0u; }
#pragma unroll
        for (int u = 0; u < HB; u++) {
            if (i0 + u * D_THREADS >= ntok) break;
            const uint32_t t = tk[u];
            const uint32_t dist = t >> 16;
            if (dist == 0) atomicAdd(&lfreq[t & 0xFF], 1);
            else {
                int lc = lcode_of((int)(t & 0xFFFF) - 3), dc = dcode_of((int)dist - 1);
                atomicAdd(&lfreq[lc], 1);
                atomicAdd(&dfreq[dc], 1);
                if (lc >= 265 && lc < 285) ex += (lc - 261) / 4; // :903-906
                if (dc >= 4) ex += dc / 2 - 1;                   // :910-913
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) ex += __shfl_xor(ex, o);
    if ((tid & 63) == 0 && ex) atomicAdd(&extra_bits, ex);
    __syncthreads();
    if (tid == 0) lfreq[256]++; // EOF_SYMBOL :790
    __syncthreads();
    // the two wavefronts of the workgroup build the two trees side by side, a wavefront per tree (build_tree)
    BB_STAMP(1);
    if (tid < 64) build_tree(lfreq, LIT_NUM, 257, 15, &scrL, llen, lblc, &lnum, tid, 8);
    else build_tree(dfreq, DIST_NUM, 1, 15, &scrD, dlen, dblc, &dnum, tid - 64);
    __syncthreads();
    BB_STAMP(2);
    {   // CalcBLFreq :349 for both trees, a wavefront each, a lane per run (run_emit)
        const int lane = tid & 63;
        if (tid < 64) { const int R = find_runs(llen, lnum, rsL, lane); count_runs(llen, rsL, R, blfreq, lane); if (lane == 0) s_RL = R; }
        else { const int R = find_runs(dlen, dnum, rsD, lane); count_runs(dlen, rsD, R, blfreq, lane); if (lane == 0) s_RD = R; }
        for (int i = tid; i < 164; i += D_THREADS) hdrw[i] = 0u;
    }
    __syncthreads();
    BB_STAMP(3);
    if (tid < 64) build_tree(blfreq, BL_NUM, 4, 7, &scrD, bllen, blblc, &blnum, tid);
    __syncthreads();
    BB_STAMP(4);
    // encoded lengths (GetEncodedLength :331) and static_len (:815-823) in parallel
    int a = 0, st = 0, abl = 0;
    for (int i = tid; i < LIT_NUM; i += D_THREADS) {
        uint32_t c; int l;
        static_lit(i, c, l);
        a += lfreq[i] * llen[i];
        st += lfreq[i] * l;
    }
    if (tid < DIST_NUM) { a += dfreq[tid] * dlen[tid]; st += dfreq[tid] * 5; }
    if (tid < BL_NUM) abl = blfreq[tid] * bllen[tid];
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); st += __shfl_xor(st, o); abl += __shfl_xor(abl, o); }
    if (tid < 4) s_sums[tid] = 0;
    __syncthreads();
    if ((tid & 63) == 0) { atomicAdd(&s_sums[0], a); atomicAdd(&s_sums[1], st); atomicAdd(&s_sums[2], abl); }
    __syncthreads();

    const int64_t seg_stop = (s.flags & SEG_SWITCH_CUT) ? so[si].cut_x : s.seg_end;   // (a cut segment's last block ends where the engine stands)
    const int64_t in_start = ntok > 0 ? blk_start_pos[gb] : seg_stop;
    int64_t in_next = seg_stop;
    if (lb + 1 < so[si].blk_count && T > tdone + BLOCK_TOKENS) in_next = blk_start_pos[gb + 1];
    const int in_len = (int)(in_next - in_start);

    if (tid == 0) {
        int blTreeCodes = 4;
        for (int i = 18; i > blTreeCodes; i--) if (bllen[c_bl_order[i]] > 0) blTreeCodes = i + 1; // :803-810
        int opt_len = 14 + blTreeCodes * 3 + s_sums[2] + s_sums[0] + extra_bits; // :811-813 (NB: omits the repeat codes' extra bits)
        int static_len = extra_bits + s_sums[1];
        if (opt_len >= static_len) opt_len = static_len;             // :824-828
        // storedOffset = window index of the block start when FlushBlock runs (:830): base of the iteration that
        // emitted the block's last token, s = min(u+1, n-1)
        int storedOffsetOk = 0;
        if (ntok > 0) {
            int64_t u = blk_lasttok_pos[gb];
            int64_t sIter = u + 1 < s.seg_end - 1 ? u + 1 : s.seg_end - 1;
            int64_t base = base_of_b((int64_t)s.abs0 + sIter);
            if (fast) base = u; // DeflateFast: k_fast recorded the window base at FlushBlock time in this table
            storedOffsetOk = ((int64_t)s.abs0 + in_start + 1 - base) >= 0;
        }
        int type;
        if (storedOffsetOk && in_len + 4 < (opt_len >> 3)) type = 0;
        else if (opt_len == static_len) type = 1;
        else type = 2;
        s_type = type; s_bltc = blTreeCodes;
        if (type == 2) { // bl codes (BuildCodes :151 for the 19-symbol tree)
            int nextCode[7], code = 0;
            for (int bits = 0; bits < 7; bits++) { nextCode[bits] = code; code += blblc[bits] << (15 - bits); }
            for (int i = 0; i < blnum; i++) {
                int bits = bllen[i];
                if (bits > 0) { blcode[i] = (unsigned short)bitrev16((uint32_t)nextCode[bits - 1]); nextCode[bits - 1] += 1 << (16 - bits); }
            }
        }
        bd->seg = si; bd->type = (uint32_t)type; bd->last = (uint32_t)last; bd->ntok = (uint32_t)ntok;
        bd->tok_first = tfirst; bd->in_start = in_start; bd->in_len = (uint32_t)in_len;
        bd->opt_len = (uint32_t)opt_len; bd->static_len = (uint32_t)static_len;
    }
    __syncthreads();
    BB_STAMP(5);
    if (tid < 64) {   // the header's bits: :773,:843,:852, SendAllTrees :676 — WriteTree :411 a lane per run (write_runs)
        const int type = s_type;
        int total = 3;
        if (tid == 0) or_bits(hdrw, 0, (uint32_t)((type << 1) + last), 3);
        if (type == 2) {
            const int bltc = s_bltc;
            if (tid == 0) { or_bits(hdrw, 3, (uint32_t)(lnum - 257), 5); or_bits(hdrw, 8, (uint32_t)(dnum - 1), 5); or_bits(hdrw, 13, (uint32_t)(bltc - 4), 4); }
            if (tid < bltc) or_bits(hdrw, 17 + 3 * tid, bllen[c_bl_order[tid]], 3);
            total = write_runs(llen, rsL, s_RL, blcode, bllen, hdrw, 17 + 3 * bltc, tid);
            total = write_runs(dlen, rsD, s_RD, blcode, bllen, hdrw, total, tid);
        }
        if (tid == 0) {
            s_hdrbits = total;
            bd->hdr_bits = (uint32_t)total;
            // body_bits = bits actually written.  opt_len is only the reference's *estimate*: GetEncodedLength (:331)
            // does not count the 2/3/7 extra bits of bl symbols 16/17/18, so a dynamic block is longer than 3+opt_len.
            if (type == 0) bd->body_bits = 3;                                  // + alignment + 32 + 8*len, added by the scan
            else if (type == 1) bd->body_bits = 3 + (uint64_t)(extra_bits + s_sums[1]);
            else bd->body_bits = (uint64_t)total + (uint64_t)s_sums[0] + (uint64_t)extra_bits;
        }
    }
    __syncthreads();
    BB_STAMP(6);
    const int type = s_type;
    // code tables (BuildCodes :151) — canonical code of symbol i = nextCode[len] + (#earlier symbols of equal length)
    for (int i = tid; i < LIT_NUM; i += D_THREADS) {
        uint32_t code = 0; int len = 0;
        if (type == 1) static_lit(i, code, len);
        else if (type == 2) {
            len = i < lnum ? llen[i] : 0;
            if (len > 0) {
                int base = 0;
                for (int b = 0; b < len - 1; b++) base += lblc[b] << (15 - b);
                int rank = 0;
                for (int j = 0; j < i; j++) rank += (llen[j] == len);
                code = bitrev16((uint32_t)(base + (rank << (16 - len))));
            }
        }
        bd->lcode[i] = (uint16_t)code; bd->llen[i] = (uint8_t)len;
    }
    if (tid < DIST_NUM) {
        uint32_t code = 0; int len = 0;
        if (type == 1) { code = bitrev16((uint32_t)tid << 11); len = 5; } // :633-642
        else if (type == 2) {
            len = tid < dnum ? dlen[tid] : 0;
            if (len > 0) {
                int base = 0;
                for (int b = 0; b < len - 1; b++) base += dblc[b] << (15 - b);
                int rank = 0;
                for (int j = 0; j < tid; j++) rank += (dlen[j] == len);
                code = bitrev16((uint32_t)(base + (rank << (16 - len))));
            }
        }
        bd->dcode[tid] = (uint16_t)code; bd->dlen[tid] = (uint8_t)len;
    }
    const int hb = (s_hdrbits + 7) >> 3;
    for (int i = tid; i < hb; i += D_THREADS) bd->hdr[i] = ((const unsigned char *)hdrw)[i];
    BB_STAMP(7);
}

// Bit position of every block: one wavefront per segment, 64 blocks per step.  Stored blocks align to a byte
// after their 3 header bits (FlushStoredBlock :766-779), so the running position is a scan with a per-block
// function of (position mod 8).  Blocks are few (<= n/16384+1 per segment): a serial carry between
// 64-wide steps is enough.
__global__ __launch_bounds__(64) void k_block_scan(const SegDev *segs, uint32_t nseg, SegOut *so, BlockDesc *descs) {
    uint32_t si = blockIdx.x;
    if (si >= nseg) return;
    const int lane = threadIdx.x;
    const uint32_t nb = so[si].blk_count, b0 = so[si].blk_first;
    uint64_t pos = segs[si].start_bit;
    for (uint32_t k0 = 0; k0 < nb; k0 += 64) {
        uint32_t k = k0 + lane;
        // delta[r] for r = start mod 8
        uint64_t body = 0; uint32_t type = 1, inlen = 0;
        if (k < nb) { body = descs[b0 + k].body_bits; type = descs[b0 + k].type; inlen = descs[b0 + k].in_len; }
        uint64_t start = pos;
        if (__ballot(k < nb && type == 0) == 0) { // no stored block among these 64 (the rule): positions are a plain prefix sum
            const uint64_t mine = k < nb ? body : 0;
            uint64_t incl = mine;
            for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(incl, o); if (lane >= o) incl += u; }
            start = pos + (incl - mine);
            pos += __shfl(incl, 63);
            if (k < nb) descs[b0 + k].bit_start = start;
            continue;
        }
        // a stored block aligns to a byte (FlushStoredBlock :766-779): serial within the 64-wide step, lane i needs the end of lane i-1
        for (int i = 0; i < 64; i++) {
            uint64_t st_i = __shfl(start, i);
            uint64_t b_i = __shfl(body, i);
            uint32_t t_i = __shfl(type, i);
            uint32_t l_i = __shfl(inlen, i);
            uint64_t end_i;
            if (t_i == 0) end_i = ((st_i + 3 + 7) & ~7ull) + 32 + 8ull * l_i;
            else end_i = st_i + b_i;
            if (k0 + i >= nb) end_i = st_i;
            if (lane == i + 1) start = end_i;
            if (i == 63) pos = end_i;
        }
        pos = __shfl(pos, 0);
        if (k < nb) descs[b0 + k].bit_start = start;
    }
    if (lane == 0) so[si].end_bit = pos;
}

// ------------------------------------------------------------------------------------------------
// Encode.  One workgroup per block; 256 tokens per step.  Bits are assembled in an LDS staging area
// with LDS atomic OR (each token code is <= 48 bits at an arbitrary bit offset) and flushed with plain
// dword stores; only the first and last dword of a step can be shared with a neighbour step/block and
// use a global atomic OR into the pre-zeroed output.
enum : int { E_THREADS = 256, E_TPT = 4, E_STAGE_DW = (31 + E_THREADS * E_TPT * 48) / 32 + 3 };   // E_TPT tokens of at most 48 bits per thread and round

__device__ __forceinline__ void lds_or_bits(uint32_t *stage, uint32_t bitoff, unsigned long long v, int nbits) {
    if (nbits == 0) return;
    uint32_t w = bitoff >> 5, sh = bitoff & 31;
    // v has <= 48 significant bits; shifted it spans <= 3 dwords
    unsigned long long lo = v << sh;
    atomicOr(&stage[w], (uint32_t)lo);
    uint32_t mid = (uint32_t)(lo >> 32);
    if (sh + nbits > 32) atomicOr(&stage[w + 1], mid);
    if (sh + nbits > 64) atomicOr(&stage[w + 2], (uint32_t)(v >> (64 - sh)));
}

__device__ void flush_stage(uint32_t *stage, uint32_t *out32, uint64_t bit_lo, uint64_t bit_hi /*exclusive*/, int tid) {
    // stage[0] corresponds to dword (bit_lo>>5)
    if (bit_hi <= bit_lo) return;
    uint64_t w0 = bit_lo >> 5, w1 = (bit_hi - 1) >> 5;
    int n = (int)(w1 - w0 + 1);
    for (int i = tid; i < n; i += E_THREADS) {
        uint32_t v = stage[i];
        if (i == 0 || i == n - 1) { if (v) atomicOr(&out32[w0 + i], v); }
        else out32[w0 + i] = v;
    }
}

// A barrier that orders LDS only.  __syncthreads() is also a fence for global memory, i.e. a wait for every store in flight, and
// k_block_encode stores a round's words to the output between two barriers: 64 rounds of a full block each waited out their stores'
// trip to memory (71 us for ONE block, round 6).  Nothing a round writes to global memory is read again by this kernel, and two rounds
// share at most the word between them, which both OR into atomically.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__global__ __launch_bounds__(E_THREADS) void k_block_encode(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                            const SegDev *__restrict__ segs, const BlockDesc *__restrict__ descs,
                                                            const uint32_t *__restrict__ tokens, uint32_t nblk_slots) {
    __shared__ uint32_t stage[E_STAGE_DW + 4];
    __shared__ uint32_t wsum[E_THREADS / 64];
    __shared__ uint16_t s_lcode[LIT_NUM]; __shared__ uint8_t s_llen[LIT_NUM];
    __shared__ uint16_t s_dcode[DIST_NUM]; __shared__ uint8_t s_dlen[DIST_NUM];
    const uint32_t gb = blockIdx.x;
    if (gb >= nblk_slots) return;
    const BlockDesc *bd = &descs[gb];
    if (bd->type == 0xFFu) return;
    const SegDev s = segs[bd->seg];
    const int tid = threadIdx.x;
    uint8_t *obase = out + s.out_off;          // out_off is 4-byte aligned (engine guarantees)
    uint32_t *out32 = (uint32_t *)obase;
    const uint64_t bit_start = bd->bit_start;

    if (bd->type == 0) { // FlushStoredBlock :766
        const uint64_t hdr_end = bit_start + 3;
        const uint64_t byte0 = (hdr_end + 7) >> 3;
        if (tid == 0) {
            uint32_t v = (uint32_t)((0 << 1) + bd->last);
            if (v) atomicOr(&out32[bit_start >> 5], v << (bit_start & 31)); // 3 bits never straddle... may straddle a dword:
            if (v && ((bit_start & 31) + 3 > 32)) atomicOr(&out32[(bit_start >> 5) + 1], v >> (32 - (bit_start & 31)));
            uint32_t len = bd->in_len;
            obase[byte0 + 0] = (uint8_t)len; obase[byte0 + 1] = (uint8_t)(len >> 8);
            obase[byte0 + 2] = (uint8_t)~len; obase[byte0 + 3] = (uint8_t)((~len) >> 8);
        }
        const uint8_t *src = in + s.buf_off + bd->in_start;
        uint8_t *dst = obase + byte0 + 4;
        for (uint32_t i = tid; i < bd->in_len; i += E_THREADS) dst[i] = src[i];
        return;
    }
    for (int i = tid; i < LIT_NUM; i += E_THREADS) { s_lcode[i] = bd->lcode[i]; s_llen[i] = bd->llen[i]; }
    if (tid < DIST_NUM) { s_dcode[tid] = bd->dcode[tid]; s_dlen[tid] = bd->dlen[tid]; }
    // ---- header bits (pre-rendered, LSB-first bytes): 8 bits per thread per pass
    uint64_t pos = bit_start;
    {
        const uint32_t hb = bd->hdr_bits;
        for (uint32_t base = 0; base < hb; base += E_THREADS * 8) {
            for (int i = tid; i < E_STAGE_DW + 4; i += E_THREADS) stage[i] = 0;
            lds_barrier();
            uint32_t chunk = hb - base < (uint32_t)E_THREADS * 8 ? hb - base : (uint32_t)E_THREADS * 8;
            uint32_t mybit = (uint32_t)tid * 8;
            if (mybit < chunk) {
                int nb = chunk - mybit < 8 ? (int)(chunk - mybit) : 8;
                unsigned v = bd->hdr[(base >> 3) + tid] & ((1u << nb) - 1);
                lds_or_bits(stage, (uint32_t)(pos & 31) + mybit, v, nb);
            }
            lds_barrier();
            flush_stage(stage, out32, pos, pos + chunk, tid);
            lds_barrier();
            pos += chunk;
        }
    }
    // ---- tokens (CompressBlock :701)
    const uint32_t ntok = bd->ntok;
    const uint64_t tfirst = bd->tok_first;
    // (E_TPT consecutive tokens per thread and round: a round ends with stores that the next round's loads queue behind — one counter
    // for both on this chip — so a round costs a trip to memory whatever it holds; 64 rounds of 256 tokens were 71-84 us for ONE full
    // block, round 6)
    for (uint32_t t0 = 0; t0 < ntok + 1; t0 += E_THREADS * E_TPT) { // +1: the EOB symbol rides as a pseudo token
        for (int i = tid; i < E_STAGE_DW + 4; i += E_THREADS) stage[i] = 0;
        uint32_t tk[E_TPT];
#pragma unroll
        for (int u = 0; u < E_TPT; u++) { const uint32_t ti = t0 + (uint32_t)tid * E_TPT + u; tk[u] = ti < ntok ? tokens[tfirst + ti] : 0u; }
        unsigned long long v[E_TPT]; int nb[E_TPT];
        int mine = 0;
#pragma unroll
        for (int u = 0; u < E_TPT; u++) {
            const uint32_t ti = t0 + (uint32_t)tid * E_TPT + u;
            v[u] = 0; nb[u] = 0;
            if (ti < ntok) {
                const uint32_t t = tk[u];
                const uint32_t dist = t >> 16;
                if (dist == 0) { v[u] = s_lcode[t & 0xFF]; nb[u] = s_llen[t & 0xFF]; }
                else {
                    int l = (int)(t & 0xFFFF) - 3;
                    int lc = lcode_of(l);
                    v[u] = s_lcode[lc]; nb[u] = s_llen[lc];
                    int bits = (lc - 261) / 4;                       // :716
                    if (bits > 0 && bits <= 5) { v[u] |= (unsigned long long)(l & ((1 << bits) - 1)) << nb[u]; nb[u] += bits; }
                    int dd = (int)dist - 1;
                    int dc = dcode_of(dd);
                    v[u] |= (unsigned long long)s_dcode[dc] << nb[u]; nb[u] += s_dlen[dc];
                    bits = dc / 2 - 1;                               // :725
                    if (bits > 0) { v[u] |= (unsigned long long)(dd & ((1 << bits) - 1)) << nb[u]; nb[u] += bits; }
                }
            } else if (ti == ntok) { v[u] = s_lcode[256]; nb[u] = s_llen[256]; } // EOF_SYMBOL :749
            mine += nb[u];
        }
        // exclusive scan of the threads' bits over the workgroup
        int incl = mine;
        for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(incl, o); if ((tid & 63) >= o) incl += y; }
        if ((tid & 63) == 63) wsum[tid >> 6] = (uint32_t)incl;
        lds_barrier();
        uint32_t woff = 0, total = 0;
        for (int w = 0; w < E_THREADS / 64; w++) { uint32_t x = wsum[w]; if (w < (tid >> 6)) woff += x; total += x; }
        uint32_t off = (uint32_t)(pos & 31) + woff + (uint32_t)(incl - mine);
#pragma unroll
        for (int u = 0; u < E_TPT; u++) { lds_or_bits(stage, off, v[u], nb[u]); off += (uint32_t)nb[u]; }
        lds_barrier();
        flush_stage(stage, out32, pos, pos + total, tid);
        lds_barrier();
        pos += total;
    }
}

// Deflater.Deflate tails (C/Deflater.cs:486-517): sync-flush padding blocks, the empty final block of a
// Flush()+Finish() pair, byte alignment, optional zlib Adler-32 trailer.
__global__ void k_seg_finish(const SegDev *segs, uint32_t nseg, SegOut *so, uint8_t *out) {
    uint32_t si = blockIdx.x * blockDim.x + threadIdx.x;
    if (si >= nseg) return;
    const SegDev s = segs[si];
    uint32_t *out32 = (uint32_t *)(out + s.out_off);
    uint64_t E = so[si].end_bit;
    auto put10 = [&](uint32_t v) {
        atomicOr(&out32[E >> 5], v << (E & 31));
        if ((E & 31) + 10 > 32) atomicOr(&out32[(E >> 5) + 1], v >> (32 - (E & 31)));
        E += 10;
    };
    if (s.flags & SEG_SYNC_PAD) { // :491-501
        int neededbits = 8 + (int)((0 - E) & 7);
        while (neededbits > 0) { put10(2); neededbits -= 10; }
    }
    if (s.flags & SEG_EXTRA_FINAL_EMPTY) put10(3); // FlushBlock of an empty final block = static header + EOB
    uint64_t bytes = (E + 7) >> 3;
    if (s.flags & SEG_ZLIB_TRAILER) { // :510-515
        uint32_t a = so[si].adler32;
        uint8_t *o = out + s.out_off + bytes;
        o[0] = (uint8_t)(a >> 24); o[1] = (uint8_t)(a >> 16); o[2] = (uint8_t)(a >> 8); o[3] = (uint8_t)a;
        bytes += 4; E = bytes * 8;
    }
    if (s.flags & SEG_GZIP) { // RFC 1952 member trailer: CRC32 | ISIZE, little endian (S/GZip/GzipOutputStream.cs:315-337)
        uint32_t c = so[si].crc32, n = (uint32_t)((uint64_t)(s.seg_end - s.seg_start) & 0xffffffffu);
        uint8_t *o = out + s.out_off + bytes;
        o[0] = (uint8_t)c; o[1] = (uint8_t)(c >> 8); o[2] = (uint8_t)(c >> 16); o[3] = (uint8_t)(c >> 24);
        o[4] = (uint8_t)n; o[5] = (uint8_t)(n >> 8); o[6] = (uint8_t)(n >> 16); o[7] = (uint8_t)(n >> 24);
        bytes += 8; E = bytes * 8;
        uint8_t *h = out + s.out_off; // header (S/GZip/GzipOutputStream.cs:339-375): ID1 ID2 CM FLG MTIME XFL OS
        uint32_t t = s.hdr_word;
        h[0] = 0x1F; h[1] = 0x8B; h[2] = 8; h[3] = 0; h[4] = (uint8_t)t; h[5] = (uint8_t)(t >> 8); h[6] = (uint8_t)(t >> 16);
        h[7] = (uint8_t)(t >> 24); h[8] = 0; h[9] = 255;
    }
    if (s.flags & SEG_ZLIB_HEADER) { uint8_t *h = out + s.out_off; h[0] = (uint8_t)(s.hdr_word >> 8); h[1] = (uint8_t)s.hdr_word; } // C/Deflater.cs:436-461
    so[si].end_bit = E;
    so[si].out_bytes = bytes;
}

// Level 0 (DeflateStored, C/DeflaterEngine.cs:614-649 + FlushStoredBlock C/DeflaterHuffman.cs:766-779): the block list is
// pure arithmetic on lengths (host); the device moves the bytes.  Every block of a level-0 stream starts byte aligned.
__global__ __launch_bounds__(256) void k_stored(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const StoredBlk *blks, uint32_t n) {
    const uint32_t b = blockIdx.x;
    if (b >= n) return;
    const StoredBlk k = blks[b];
    uint8_t *o = out + k.out_off;
    if (threadIdx.x == 0) {
        o[0] = (uint8_t)k.last;                       // 3 header bits (BFINAL, BTYPE=00) + 5 alignment bits
        o[1] = (uint8_t)k.len; o[2] = (uint8_t)(k.len >> 8);
        o[3] = (uint8_t)~k.len; o[4] = (uint8_t)((~k.len) >> 8);
    }
    const uint8_t *src = in + k.in_off;
    for (uint32_t i = threadIdx.x; i < k.len; i += 256) o[5 + i] = src[i];
}
void launch_stored(const uint8_t *in, uint8_t *out, const StoredBlk *blks, uint32_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_stored, dim3(n), dim3(256), 0, st, in, out, blks, n);
}


// The encoder ORs bits into its output, so every segment's region [out_off, out_off + out_cap) starts as zeros — and ONLY the
// regions: bytes of the caller's buffer before, between and after them (zip local headers in the passthrough layout,
// INTEGRATION.md §3) are never touched.  zoff[i] = index of segment i's first 64 KiB piece (host-built prefix sum).
enum : int { Z_PIECE = 65536 };
__global__ __launch_bounds__(256) void k_zero_regions(const SegDev *__restrict__ segs, uint32_t nseg, const uint64_t *__restrict__ zoff,
                                                      uint8_t *__restrict__ out) {
    const uint64_t b = blockIdx.x;
    uint32_t lo = 0, hi = nseg; // last segment with zoff[seg] <= b
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (zoff[mid] <= b) lo = mid; else hi = mid; }
    const SegDev s = segs[lo];
    const uint64_t a = (b - zoff[lo]) * (uint64_t)Z_PIECE;
    if (a >= s.out_cap) return;
    const uint64_t len = s.out_cap - a < (uint64_t)Z_PIECE ? s.out_cap - a : (uint64_t)Z_PIECE;
    uint8_t *o = out + s.out_off + a;          // out_off is 4-byte aligned, a is a multiple of 64 KiB
    const uint64_t ndw = len >> 2;
    uint32_t *o32 = (uint32_t *)o;
    if ((((uintptr_t)o) & 15) == 0) {
        uint4 *o128 = (uint4 *)o;
        const uint64_t n16 = ndw >> 2;
        for (uint64_t i = threadIdx.x; i < n16; i += 256) o128[i] = make_uint4(0, 0, 0, 0);
        for (uint64_t i = (n16 << 2) + threadIdx.x; i < ndw; i += 256) o32[i] = 0;
    } else {
        for (uint64_t i = threadIdx.x; i < ndw; i += 256) o32[i] = 0;
    }
    for (uint64_t i = (ndw << 2) + threadIdx.x; i < len; i += 256) o[i] = 0;
}
// Several small arrays zeroed by ONE launch (a call of the streaming object clears five: each memset of its own is a kernel and a
// gap of ~6 us on the stream; a 64 KiB entry's call spent more time between kernels than in stage B).  Sizes in 4-byte words.
struct ZeroMany { uint32_t *p[6]; uint32_t words[6]; int n; };
// A few bytes between mapped pinned host memory and device memory, either way, WITHOUT the copy engine (szl_engine.hip, copy_small): the
// engine's queue is one per direction for the whole process, and a table of a streaming Deflater's part waited there behind every 16 MiB
// upload the caller had in flight — three host round trips of a 3.9 ms part took 35 ms when the caller wrote faster than PCIe carries.
__global__ __launch_bounds__(256) void k_copy_small(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint32_t nwords) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nwords) dst[i] = src[i];
    __threadfence_system();
}
void launch_copy_small(void *dst, const void *src, size_t bytes, hipStream_t st) {   // 4-byte aligned, bytes % 4 == 0
    const uint32_t nw = (uint32_t)(bytes / 4);
    if (nw) hipLaunchKernelGGL(k_copy_small, dim3((nw + 255) / 256), dim3(256), 0, st, (uint32_t *)dst, (const uint32_t *)src, nw);
}

__global__ __launch_bounds__(256) void k_zero_many(ZeroMany z) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    for (int k = 0; k < z.n; k++) {
        if (i < z.words[k]) { z.p[k][i] = 0u; return; }
        i -= z.words[k];
    }
}
void launch_zero_many(void *const *ptrs, const size_t *bytes, int n, hipStream_t st) {
    ZeroMany z{};
    uint64_t total = 0;
    for (int k = 0; k < n && z.n < 6; k++) {
        if (!ptrs[k] || !bytes[k]) continue;
        if (bytes[k] > (4u << 20) || (bytes[k] & 3) || ((uintptr_t)ptrs[k] & 3)) { (void)hipMemsetAsync(ptrs[k], 0, bytes[k], st); continue; }   // a long array: the runtime's own fill
        z.p[z.n] = (uint32_t *)ptrs[k]; z.words[z.n] = (uint32_t)(bytes[k] >> 2); total += z.words[z.n]; z.n++;
    }
    if (total) hipLaunchKernelGGL(k_zero_many, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z);
}
void launch_zero_regions(const SegDev *segs, uint32_t nseg, const uint64_t *zoff, uint64_t npieces, uint8_t *out, hipStream_t st) {
    if (npieces) hipLaunchKernelGGL(k_zero_regions, dim3((unsigned)npieces), dim3(256), 0, st, segs, nseg, zoff, out);
}
int zero_piece_bytes() { return Z_PIECE; }

void launch_seg_blocks(const SegDev *segs, uint32_t nseg, const uint32_t *tokens, const uint64_t *blk_off, SegOut *so,
                       int fast, hipStream_t st) {
    hipLaunchKernelGGL(k_seg_blocks, dim3((nseg + 255) / 256), dim3(256), 0, st, segs, nseg, tokens, blk_off, so, fast);
}
// test tap (szl_debug_tree_lengths): the code lengths of n frequency vectors, one wavefront each — build_tree against the oracle's
// Tree.BuildTree on histograms no token stream would produce (ties everywhere, over-long codes)
__global__ __launch_bounds__(64) void k_tree_probe(const int *freqs, int n, int numSymbols, int minCodes, int maxLength, unsigned char *len_out, int *ncodes_out) {
    __shared__ int f[LIT_NUM + 2];
    __shared__ unsigned char len[LIT_NUM + 2];
    __shared__ int blc[15];
    __shared__ int ncodes;
    __shared__ TreeScratch scr;
    if ((int)blockIdx.x >= n) return;
    const int lane = threadIdx.x;
    for (int i = lane; i < LIT_NUM + 2; i += 64) f[i] = i < numSymbols ? freqs[(size_t)blockIdx.x * numSymbols + i] : 0;
    __syncthreads();
    build_tree(f, numSymbols, minCodes, maxLength, &scr, len, blc, &ncodes, lane);
    __syncthreads();
    for (int i = lane; i < numSymbols; i += 64) len_out[(size_t)blockIdx.x * numSymbols + i] = len[i];
    if (lane == 0) ncodes_out[blockIdx.x] = ncodes;
}
void launch_tree_probe(const int *freqs, int n, int numSymbols, int minCodes, int maxLength, unsigned char *len_out, int *ncodes_out, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_tree_probe, dim3(n), dim3(64), 0, st, freqs, n, numSymbols, minCodes, maxLength, len_out, ncodes_out);
}

void launch_block_build(const SegDev *segs, uint32_t nseg, const SegOut *so, const uint64_t *blk_off, const uint32_t *tokens,
                        const int64_t *bsp, const int64_t *blp, BlockDesc *descs, uint32_t nslots, int fast, hipStream_t st) {
    if (nslots) hipLaunchKernelGGL(k_block_build, dim3(nslots), dim3(D_THREADS), 0, st, segs, nseg, so, blk_off, tokens, bsp, blp, descs, nslots, fast);
}
void launch_block_scan(const SegDev *segs, uint32_t nseg, SegOut *so, BlockDesc *descs, hipStream_t st) {
    hipLaunchKernelGGL(k_block_scan, dim3(nseg), dim3(64), 0, st, segs, nseg, so, descs);
}
void launch_block_encode(const uint8_t *in, uint8_t *out, const SegDev *segs, const BlockDesc *descs, const uint32_t *tokens,
                         uint32_t nslots, hipStream_t st) {
    if (nslots) hipLaunchKernelGGL(k_block_encode, dim3(nslots), dim3(E_THREADS), 0, st, in, out, segs, descs, tokens, nslots);
}
void launch_seg_finish(const SegDev *segs, uint32_t nseg, SegOut *so, uint8_t *out, hipStream_t st) {
    hipLaunchKernelGGL(k_seg_finish, dim3((nseg + 255) / 256), dim3(256), 0, st, segs, nseg, so, out);
}

// ------------------------------------------------------------------------------------------------
// Block positions from the tokens alone (one stream assembled from the tokens of several engines, Engine::finish_tokens): the
// parse kernels normally note where every 16384-token block starts and where its last token starts while they emit; both are
// prefix sums of token lengths (literal 1, match len).
__global__ __launch_bounds__(256) void k_block_tok_sums(const uint32_t *__restrict__ tokens, uint64_t ntok, uint64_t *__restrict__ sums,
                                                        uint32_t *__restrict__ lastlen) {
    __shared__ uint64_t part[256];
    const uint64_t b = blockIdx.x;
    const uint64_t t0 = b * BLOCK_TOKENS, t1 = t0 + BLOCK_TOKENS < ntok ? t0 + BLOCK_TOKENS : ntok;
    uint64_t acc = 0;
    for (uint64_t i = t0 + threadIdx.x; i < t1; i += 256) { const uint32_t t = tokens[i]; acc += (t >> 16) ? (t & 0xFFFFu) : 1u; }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) {
        sums[b] = part[0];
        const uint32_t t = t1 > t0 ? tokens[t1 - 1] : 0u;
        lastlen[b] = t1 > t0 ? ((t >> 16) ? (t & 0xFFFFu) : 1u) : 0u;
    }
}
// (one workgroup: a thread sums a run of consecutive blocks, the runs' totals are scanned in LDS — one thread walking the 14000 blocks of
// a 1 GiB stream with a global round trip each was 1.6 ms of every Finish())
__global__ __launch_bounds__(1024) void k_block_positions(const uint64_t *__restrict__ sums, const uint32_t *__restrict__ lastlen, uint64_t nblk, int64_t seg_start,
                                                          int64_t *__restrict__ bsp, int64_t *__restrict__ blp) {
    __shared__ int64_t s_tot[1024];
    const int tid = threadIdx.x;
    const uint64_t per = (nblk + 1023) / 1024;
    const uint64_t b0 = (uint64_t)tid * per < nblk ? (uint64_t)tid * per : nblk, b1 = b0 + per < nblk ? b0 + per : nblk;
    int64_t acc = 0;
    for (uint64_t b = b0; b < b1; b++) acc += (int64_t)sums[b];
    s_tot[tid] = acc;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {          // inclusive scan
        const int64_t v = tid >= o ? s_tot[tid - o] : 0;
        __syncthreads();
        s_tot[tid] += v;
        __syncthreads();
    }
    int64_t pos = seg_start + s_tot[tid] - acc;
    for (uint64_t b = b0; b < b1; b++) { bsp[b] = pos; blp[b] = pos + (int64_t)sums[b] - (int64_t)lastlen[b]; pos += (int64_t)sums[b]; }
}
void launch_block_positions(const uint32_t *tokens, uint64_t ntok, int64_t seg_start, uint64_t *sums, uint32_t *lastlen, int64_t *bsp, int64_t *blp,
                            hipStream_t st) {
    const uint64_t nblk = (ntok + BLOCK_TOKENS - 1) / BLOCK_TOKENS;
    if (!nblk) return;
    hipLaunchKernelGGL(k_block_tok_sums, dim3((unsigned)nblk), dim3(256), 0, st, tokens, ntok, sums, lastlen);
    hipLaunchKernelGGL(k_block_positions, dim3(1), dim3(1024), 0, st, sums, lastlen, nblk, seg_start, bsp, blp);
}

#if SZL_LAB
__global__ void k_bb_stamps_out(unsigned long long *out) { out[threadIdx.x] = g_bb_stamps[threadIdx.x]; }
} // namespace szl
extern "C" int szl_lab_bb_stamps(unsigned long long *out) {   // (a copy kernel, not hipMemcpyFromSymbol: the fake runtime of tools/gfxsim has no symbols)
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, 16 * 8) != hipSuccess) return -1;
    hipLaunchKernelGGL(szl::k_bb_stamps_out, dim3(1), dim3(16), 0, 0, d);
    const bool ok = hipMemcpy(out, d, 16 * 8, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? 0 : -1;
}
namespace szl {
#endif
} // namespace szl
