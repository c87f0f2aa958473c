// szl_inflate.h — device inflater job/state structures (shared by kernel and host API; not part of the C ABI).
#pragma once
#include <stdint.h>
#include "../../include/szl.h"

namespace szl {

// status values written by k_inflate (>= 0); negative values are szl_status error codes
enum : int { INF_RUNNING = 0, INF_FINISHED = 1, INF_NEED_INPUT = 2, INF_OUTPUT_FULL = 3, INF_NEED_DICT = 4,
             INF_CHUNK_END = 5 /* chunked decode of one member: reached the block boundary at which the next chunk starts */,
             INF_EXACT = 6 /* k_inflate stands in front of a block whose code set the reference's lookup table decodes differently from a
                              canonical decoder: k_inflate_exact (szl_kernels_inflate_exact.hip) takes the stream from here */ };
// decoder modes (the reference's 13 modes collapse to these because a token is decoded atomically)
enum : uint32_t { INF_M_HEADER = 0, INF_M_STORED = 1, INF_M_HUFF = 2, INF_M_DONE = 3, INF_M_ZHEADER = 4 };

struct InfJob {
    uint64_t in_off;     // arena offset of the first byte of the stream (zlib header included)
    uint64_t in_len;     // bytes of the stream available so far
    uint64_t out_off;    // arena offset where this call's output goes
    uint64_t out_cap;    // bytes that may be produced by this call
    uint8_t *window;     // 32 KiB device buffer holding the last 32 KiB of output between calls (NULL: one-shot)
    uint32_t zlib;       // 1: zlib framing (header at mode INF_M_ZHEADER, Adler-32 trailer)
    uint32_t keep_window;
    uint32_t load_window; // 1: restore the window even at outpos 0 (preset dictionary)
    uint32_t dbg_rounds;  // [out] decode rounds of this call (SZL_DEBUG prints the sums)
    // chunked decode of ONE member (szl_api_inflate.hip, inflate_member_parallel): this job decodes the blocks from bit `start_bit`
    // (a block header) up to the block boundary `stop_bit` (~0: to the end of the stream); `sym_out` receives 16-bit symbols —
    // a byte, or 0x8000 | i for "byte i of the 32 KiB of output in front of this chunk", unknown until the chunks before are done
    uint64_t start_bit, stop_bit;
    uint16_t *sym_out;
    uint64_t end_bit;    // [out] bit position the decoder stopped at
    // results
    uint64_t out_written;
    uint64_t consumed;   // ceil(bits consumed / 8)  == Inflater.TotalIn at this point
    int32_t status;
    uint32_t dbg_par;     // [out] rounds decoded by the whole wavefront (the rest: lane 0's careful path, headers, restaging)
    uint32_t dbg_partok;  // [out] tokens of those rounds
    uint32_t stop_at_header; // 1: stop with INF_CHUNK_END as soon as the decoder stands at a block header (the streaming object brings the
                             // stream to a block boundary before it hands a long input to the chunk-parallel decoder)
    // chunk jobs: the member's candidate starts behind stop_bit, ascending (starts[0] == stop_bit), and where the last job stops.  A job
    // that stands at a block boundary PAST its stop — the candidate lay inside a block: a false one, e.g. a block header of deflate data
    // that the member merely carries as its payload — goes on to the next candidate at or behind that boundary instead of ending where
    // nobody starts (each such end used to cost a repair pass of its own, and a member with more than five of them the whole parallel path)
    const uint64_t *starts; uint64_t stop_last; uint32_t nstarts; uint32_t pad0;
    // symbol pass, single-pass form: a job whose output outgrows its staging region takes one four times as long from the call's spill area —
    // symbols [*spill_cursor, spill_end) of sym_base, handed out by atomic add —, moves what it has there and goes on; sym_out / out_cap
    // come back changed.  (The host used to run such a job again in a pass of its own: one job's 30-50 ms with the device idle.)
    uint64_t *spill_cursor; uint64_t spill_end; uint16_t *sym_base;
    uint32_t in_more;        // 1: the caller holds more input behind in_len (the streaming object uploads a bounded prefix per step): where the
                             // reference's GetSymbol would look at bits past in_len the decoder stops with INF_NEED_INPUT instead of applying the
                             // "fewer than 9 bits left: an empty slot reads as symbol 0, 0 bits" rule, which is for the END of the input only
};

struct InfState {
    uint64_t bitpos, outpos;
    uint32_t mode, last, stored_left, btype, lnum, dnum, pend_len, pend_dist;
    int32_t status;
    uint32_t adler_read;
    uint8_t lens[320];
};

// k_inflate_exact's persistent state (szl_kernels_inflate_exact.hip): the reference's StreamManipulator, Inflater and InflaterDynHeader fields
struct ExState {
    uint32_t init;        // 0: derive the bit buffer from InfState.bitpos at a block header; 1: everything below is live
    uint32_t buffer; int32_t bits; uint32_t lazy; int32_t dirty; uint64_t ws;
    int32_t mode, neededBits, repLength, repDist, uncomprLen, isLastBlock, trees; uint32_t readAdler;
    int32_t dh_step, dh_ll, dh_d, dh_m, dh_n, dh_i, dh_index, dh_symbol, dh_len;
    uint8_t lens[320];
    int16_t meta[512], litlen[1024], dist[1184];
};

// chunk-parallel decode of whole members (szl_kernels_inflate_par.hip)
struct FindJob { uint64_t in_off, in_len, lo_bit, hi_bit; };     // look for a block header of the member at in_off in [lo_bit, hi_bit)
enum : uint32_t { CONV_BLOCK = 65536 };   // bytes of a member's output to a workgroup of k_convert
struct ParMember {                                               // one member being assembled from its chunk jobs
    uint64_t ooff_off;  // first of its njobs + 1 output offsets (and of its jobs' staging offsets)
    uint64_t win_off;   // its (njobs + 1) windows of 32 KiB (bytes)
    uint64_t out_off;   // arena offset of its output region
    uint64_t total;     // output bytes
    uint32_t njobs;
    uint32_t blk0;      // first workgroup of k_convert that belongs to it
    const uint8_t *win0; // the 32 KiB of output in front of its first job, oldest byte first (nullptr: zeros — a member's start).  Set when a
                         // streaming Inflater hands the decoder a long piece of input in mid-stream (szl_api_inflate.hip, inflater_bulk)
};

// k_resolve_*: the jobs of a member in groups of RES_GROUP consecutive ones (a member of thousands of chunks is not resolved
// front to back by ONE workgroup: every group first resolves relative to its unknown entry window, a short chain fixes the
// entry windows, then every group resolves for real)
enum : uint32_t { RES_GROUP = 32 };
struct ResGroup { uint32_t mem; uint32_t j0, j1; uint32_t first; uint64_t slot; };   // member, its jobs [j0, j1), first group of the member?, index of its map / entry window

} // namespace szl
