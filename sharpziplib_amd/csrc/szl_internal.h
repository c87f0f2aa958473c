// szl_internal.h — structures shared by the HIP kernels and the host engine (not part of the C ABI).
//
// Data layout in HBM.  One engine call works on an *input arena* (all streams'
// bytes, device memory) and per-position side arrays indexed by the same byte offset g:
//     in[g]       u8     input bytes
//     link[g]     u16    stage A: distance to the previous inserted position with the same 3-byte hash
//     m2[g],mq[g] u32    stage B: FindLongestMatch(p) from matchLen 2 with the full / quarter chain budget (M_UNSET = not evaluated)
//     spec_tok[g] u32    stage C: tokens of each range's speculative path, stored at the range's first positions
//     visited     1 bit  stage C: "a speculative parse of this position's range had a clean iteration here"
//     tokens[]    u32    stage C: dense token stream (literal byte | dist<<16 | len)
// A *segment* is the span of one stream between two Flush()/Finish() calls; all kernels are driven
// by small per-segment / per-tile tables built on the host.
#pragma once
#include <stdint.h>

namespace szl { int knob(const char *name, int dflt); }
// A tuning value whose sweep is over (DESIGN_HISTORY.md names the logs): the product library compiles the measured best in, the laboratory
// library (-DSZL_LAB=1, libszl_amd_lab.so: tools/gpu_matchlab.py, tests/test_gpu_stage_b_forms.py) still reads the knob.
#if SZL_LAB
#define SZL_LABKNOB(name, dflt) (szl::knob(name, dflt))
#else
#define SZL_LABKNOB(name, dflt) (dflt)
#endif

namespace szl {

enum : int { WSIZE = 32768, MAX_DIST = 32506, MAX_MATCH = 258, MIN_MATCH = 3, TOO_FAR = 4096, BLOCK_TOKENS = 16384,
              MIN_LOOKAHEAD = MAX_MATCH + MIN_MATCH + 1 /* C/DeflaterConstants.cs:104 */ };
enum : int { LIT_NUM = 286, DIST_NUM = 30, BL_NUM = 19 };

// fast != 0: DeflateFast (levels 1-4): max_lazy is the longest match whose interior is still inserted (:697)
struct LevelParams { int good, nice, max_chain, strategy, max_lazy, fast; };

// Stage-B output, indexed like the input buffer.  The parse reads ~4 bytes per position of it and is HBM-bound on exactly that
// (k_spec_win, round-2 VERDICT), so an entry is ONE packed word in `m2`:
//   [8:0] len of M2 (0 = no match, else 3..258), [23:9] its distance (<= 32506), [25:24] what Mq is:
//   0 = the same as M2, 1 = empty, 2 = something else — only then does `mq` hold an entry for the position (len | dist<<16;
//   ~10 % of the positions on text), and only a lazy look after a match of goodLength or more reads it.
// M_UNSET (all ones) = not evaluated (on-demand form).  Kernels exchange results as len | dist<<16 ("legacy" below).
struct MTab {
    uint32_t *m2;
    uint32_t *mq;
    const uint16_t *link4 = nullptr;   // chain compression (szl_kernels_match3.hip): set => stage B may use k_match6
    const uint8_t *skip4 = nullptr;
    const uint16_t *e3d = nullptr;     // first chain element with the same three bytes: distance, chain index
    const uint8_t *e3h = nullptr;
    int form = 0;                      // which form of k_match9's text the launch runs: 0 / 1, 2 = every tile picks its own (launch_match9)
};

enum : uint32_t { M_UNSET = 0xFFFFFFFFu }; // M2 entry of a position no stage-B walker evaluated (valid entries have len <= 258)
#if defined(__HIPCC__)
#define SZL_HD __host__ __device__ __forceinline__
#else
#define SZL_HD inline
#endif
SZL_HD uint32_t mt_pack(uint32_t r2, uint32_t rq) {     // r2, rq: len | dist<<16
    const uint32_t code = rq == r2 ? 0u : (rq == 0u ? 1u : 2u);
    return (r2 & 0x1FFu) | ((r2 >> 16) << 9) | (code << 24);
}
SZL_HD uint32_t mt_m2(uint32_t e) { return (e & 0x1FFu) | (((e >> 9) & 0x7FFFu) << 16); }   // packed -> len | dist<<16
SZL_HD uint32_t mt_code(uint32_t e) { return (e >> 24) & 3u; }

struct SegDev {
    uint64_t buf_off;    // arena offset of buffer position 0 of this segment's stream window
    uint64_t abs0;       // absolute stream position of buffer position 0 (window base arithmetic)
    int64_t seg_start;   // buffer position of the first byte to parse (bytes before it are history)
    int64_t seg_end;     // buffer position one past the last byte
    uint32_t bnd_off;    // boundaries (segment ends of this stream inside the buffer, ascending, last == seg_end)
    uint32_t bnd_cnt;
    uint32_t finish;     // segment closed by Finish() (else by Flush())
    uint32_t flags;      // SEG_* below
    uint64_t range_off;  // first stage-C range of this segment (DeflateFast: index of the segment's first token)
    uint32_t range_cnt;
    uint32_t hdr_word;   // SEG_ZLIB_HEADER: the 16-bit zlib header; SEG_GZIP: MTIME of the member header
    uint64_t out_off;    // output arena offset of this segment's output region
    uint64_t out_cap;    // bytes
    uint32_t start_bit;  // bit offset inside the region at which this segment's first block starts
    uint32_t stream_idx; // caller's stream index (results)
    uint64_t vis_word_off; // first 32-bit word of this segment's `visited` bitmap (bit i = position seg_start+i);
                           // DeflateFast: of its "inserted" bitmap (bit q = buffer position q)
    uint32_t adler_init;   // running Adler32.Value before this segment's bytes (zlib framing)
    uint32_t crc_init;     // running Crc32.Value before this segment's bytes
    // SetLevel / SetStrategy while input is pending (C/DeflaterEngine.cs:304-361, DEFLATE_SLOW -> DEFLATE_SLOW): an iteration of
    // DeflateSlow that STARTS at buffer position x >= sw_pos[k] runs with sw_P[k] (the last such k) instead of the call's
    // LevelParams — the search at x (stage B) and the lazy decision at x (stage C) alike.  The engine stops at the first
    // iteration start within MIN_LOOKAHEAD - 1 of the input it has, so "every iteration start >= that threshold" is exactly
    // "every iteration after the call".
    uint32_t sw_cnt;     // entries of sw_pos / sw_P (device arrays of the call, any number; the engine uploads them: Engine::sw_pos_in / sw_P_in)
    uint32_t range_len;  // positions per stage-C range (C_RANGE; shorter for small calls: the ranges are walked serially, latency counts there)
    const int64_t *sw_pos;
    const LevelParams *sw_P;
    // SEG_SWITCH_CUT (SetLevel across compression functions with bytes pending, C/DeflaterEngine.cs:319-359): the reference flushes a
    // block where its engine STANDS and continues with the other function.  The engine stands at the first iteration start at or
    // behind cut_pos (it stopped for want of lookahead: `while (lookahead >= MIN_LOOKAHEAD || flush)`, :681,:759); the segment's
    // tokens end there (a DeflateSlow iteration with a match pending drops the match and tallies its first byte, :338-341), the
    // last block is flushed without sync padding, and SegOut.cut_x says where the next function starts.
    int64_t cut_pos;
    int64_t look_end;      // buffer position one past the last byte the ENGINE HAS SEEN (lookahead, FillWindow :379-394).  Equals
                           // seg_end except for the windows of a long stream (Engine::deflate_windowed): there seg_end only ends
                           // the window's parse ranges, while matches and the insert rule look on to the true end of the input.
};
enum : uint32_t { SEG_SYNC_PAD = 1, SEG_EXTRA_FINAL_EMPTY = 2, SEG_ZLIB_TRAILER = 4, SEG_ZLIB_HEADER = 8, SEG_GZIP = 16, SEG_SWITCH_CUT = 32 };

struct SpanDev { uint32_t seg; uint32_t pad; int64_t start, end; };          // stage A: emit links for [start,end)
struct TileDev { uint32_t seg; uint32_t pad; int64_t start; int32_t len; int32_t pad2; }; // stage B tile (pad2: index + 1 into the segment's sw_P, 0 = the call's parameters; host side only)

enum : int { B_TILE = 16384, B_HIST = 32512, B_TAIL = 264 };
enum : int { C_RANGE = 4096 };
enum : int { C_WIN_HALO = 2048 }; // match-table / link entries kept past a window's parse end (a walk crosses it by < 513 positions)

// Per-range results of stage C
struct RangeDev {
    int64_t exit_spec;   // first clean position >= range end reached by the speculative walk from range start
    int64_t exit_true;   // same for the true parse (after fix-up / resolve)
    int64_t entry;       // clean position at which the true parse enters the range
    uint32_t spec_count; // tokens owned by spec nodes in the range
    uint32_t true_count; // tokens owned by true nodes in [entry, range end)
    uint32_t merged;     // 1 if the fix-up walk landed on the speculative path inside the range
    uint32_t pad;
};

struct SegOut {          // per-segment results written by the device
    uint64_t tok_first;  // index of the first token of the segment in tokens[]
    uint64_t tok_count;
    uint32_t blk_first;  // index of first block descriptor
    uint32_t blk_count;
    uint64_t end_bit;    // bit offset (inside out region) after the last block / trailer
    uint64_t out_bytes;  // bytes of output (incl. a trailing partial byte)
    uint32_t crc32, adler32;
    int64_t cut_x;       // SEG_SWITCH_CUT: buffer position the segment's tokens end at (the next function's first iteration)
};

struct StoredBlk { uint64_t in_off; uint64_t out_off; uint32_t len; uint32_t last; }; // level 0: one stored block

struct BlockDesc {
    uint32_t seg;
    uint32_t type;        // 0 stored, 1 static, 2 dynamic
    uint32_t last;
    uint32_t ntok;
    uint64_t tok_first;   // global token index
    int64_t in_start;     // buffer position of first input byte covered
    uint32_t in_len;      // bytes covered (stored length)
    uint32_t hdr_bits;    // dynamic header bits (after the 3-bit block header)
    uint64_t body_bits;   // bits of header(3) + trees + tokens + EOB (non-stored) ; for stored: 3 (before alignment)
    uint64_t bit_start;   // absolute bit position inside the segment's out region
    uint16_t lcode[LIT_NUM];
    uint8_t llen[LIT_NUM];
    uint16_t dcode[DIST_NUM];
    uint8_t dlen[DIST_NUM];
    uint32_t opt_len, static_len; // for the parity taps
    uint8_t hdr[640];     // pre-rendered dynamic header bitstream, LSB-first
};

} // namespace szl
