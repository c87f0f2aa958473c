// szl_engine.h — host-side engine object behind the C ABI (include/szl.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>
#include "../../include/szl.h"
#include "szl_internal.h"

namespace szl {

void set_error(const char *fmt, ...);
const char *last_error();
int level_params(int level, int strategy, LevelParams *P);
int knob(const char *name, int dflt);   // szl_debug_set() value, else environment variable, else dflt
int knob_set(const char *name, int value);

enum : int { SZL_MATCH_KERNEL_DEFAULT = 2 };   // form of stage B's full search (SZL_MATCH_KERNEL): 2 k_match4 (prev[] walks), 5 k_match5 (bucket order)

bool host_is_pinned(const void *p, size_t n);   // [p, p+n) lies in memory handed out by szl_host_alloc / named by szl_host_register

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n);   // grow-only; contents are NOT preserved
    int ensure_keep(size_t n, size_t keep, hipStream_t st);   // grow-only; the first `keep` bytes are preserved (copied on `st`, which is synchronised)
    void release();
};

// Bytes in pinned host memory with the few members of std::vector<uint8_t> the streaming objects use (szl_api.hip): what the caller
// gives is copied here once and goes to the device by DMA while he is still writing; what the device produces comes back the same
// way.  No zero fill on growth, no page faults on reuse.  Allocation failure throws std::bad_alloc like the vector it replaces (the
// C entry points catch it).
// Pinned host memory is expensive to get (the pages are locked and mapped into the device's address space: a gigabyte costs hundreds of
// milliseconds) and the streaming objects are short-lived (GZipOutputStream makes a new Deflater per stream, S/GZip/GzipOutputStream.cs:87;
// InflaterPool resets and reuses).  Blocks a buffer outgrows or leaves behind go to a process-wide pool (SZL_PIN_POOL_MIB, default 4096:
// what it may hold; 0 = none) and are handed out again, best fit, to whoever asks next.
uint8_t *pin_alloc(size_t want, size_t *cap_out, bool growing = false);    // nullptr: no pinned memory of that size
void pin_free(uint8_t *p, size_t cap);
size_t pin_pool_trim(size_t keep);   // frees pooled pinned blocks (largest first) until at most `keep` bytes stay pooled

void host_copy(void *dst, const void *src, size_t k);   // memcpy, on several cores when long (szl_engine.hip)
struct PinVec {
    uint8_t *p = nullptr; size_t n = 0, cap = 0;
    hipStream_t busy = nullptr;                       // a stream with copies out of this memory in flight (synchronised before it moves)
    PinVec() = default;
    PinVec(const PinVec &) = delete;
    PinVec &operator=(const PinVec &) = delete;
    ~PinVec() { release(); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    uint8_t *begin() { return p; }
    uint8_t *end() { return p + n; }
    uint8_t &operator[](size_t i) { return p[i]; }
    const uint8_t &operator[](size_t i) const { return p[i]; }
    void clear() { n = 0; }
    void reserve(size_t want);
    void resize(size_t k) { reserve(k); n = k; }   // (new bytes are uninitialised)
    void push_back(uint8_t b) { reserve(n + 1); p[n++] = b; }
    void append(const uint8_t *src, size_t k);
    void erase_front(size_t k);
    void release();
};

class Engine {
  public:
    Engine();
    ~Engine();
    // Compress `segs` (see SegDev) found in the input arena d_in[0..in_total) into d_out[0..out_total).
    // d_out and every seg.out_off must be 4-byte aligned.  Synchronous.
    int deflate(const uint8_t *d_in, uint64_t in_total, uint8_t *d_out, uint64_t out_total, std::vector<SegDev> &segs,
                const std::vector<uint64_t> &bnds, LevelParams P, unsigned want_ck, std::vector<SegOut> &results, hipStream_t st);

    static bool uses_window_pipeline(size_t n_segments, bool deflate_slow, uint64_t stream_len, uint64_t *window_out);   // (see szl_engine.hip)
    // One long stream (a single segment, levels 5-9) as a software pipeline of windows: stages A-C run window by window on
    // side arrays sized for ONE window — the parse of a window starts at the clean iteration the previous one ended on — and
    // stage D runs once over the whole token stream.  Output is bit-identical to deflate()'s (tests force tiny windows).
    int deflate_windowed(const uint8_t *d_in, uint64_t in_total, uint8_t *d_out, uint64_t out_total, SegDev seg,
                         const std::vector<uint64_t> &bnds, LevelParams P, unsigned want_ck, std::vector<SegOut> &results, hipStream_t st,
                         uint64_t window);
    // One stream over several engines (szl_api.hip, stream_multi_run): an engine runs stages A-C of a *part* [first, parse_end) of
    // the stream with deflate_windowed and keeps the tokens; stage D then runs once, on one engine, over all parts' tokens
    // (finish_tokens).  A part that does not start the stream does not know the iteration the true parse enters it on: it
    // parses a warm-up stretch [warm_from, first) from an assumed clean state first and drops those tokens — two parses that are
    // clean at the same position are identical from there on, so the entry it arrives at (`entry`) is the true one iff the
    // previous part's parse leaves on it (`exit`); the caller checks that and re-runs a part with force_entry otherwise.
    int deflate_impl(const uint8_t *d_in, uint64_t in_total, uint8_t *d_out, uint64_t out_total, std::vector<SegDev> &segs,
                     const std::vector<uint64_t> &bnds, LevelParams P, unsigned want_ck, std::vector<SegOut> &results, hipStream_t st);
    int deflate_windowed_impl(const uint8_t *d_in, uint64_t in_total, uint8_t *d_out, uint64_t out_total, SegDev seg,
                              const std::vector<uint64_t> &bnds, LevelParams P, unsigned want_ck, std::vector<SegOut> &results, hipStream_t st,
                              uint64_t window);
    struct PartRun {
        bool active = false;
        int64_t first = 0, parse_end = 0;      // buffer positions
        int64_t warm_from = -1;                // >= 0: warm-up from here
        int64_t force_entry = -1;              // >= 0: start exactly here (no warm-up)
        int64_t entry = 0, exit = 0;           // [out] the part's tokens cover [entry, exit)
        uint64_t tok_start = 0;                // the part's tokens go to tokens[tok_start ..) (what is in front of them is kept)
        uint64_t tok_count = 0;                // [out]
    } part;
    // Stage D (and the checksums) of one stream whose tokens[0 .. tok_total) are in `tokens`: block positions are recomputed
    // from the tokens themselves.
    int finish_tokens(const uint8_t *d_in, uint64_t in_total, uint8_t *d_out, SegDev seg, uint64_t tok_total, unsigned want_ck,
                      std::vector<SegOut> &results, hipStream_t st);
    // progress of an overlapped host->device copy of the input arena (szl_deflate_batch_host): the engine waits until the bytes a
    // window needs have arrived.  nullptr: everything is resident.
    const volatile uint64_t *in_ready = nullptr;
    DevBuf tabs;                       // the small host-built tables of a call (segments, tiles, spans, ...) in ONE upload: tab_pin -> tabs
    uint8_t *tab_pin = nullptr; size_t tab_pin_cap = 0;
    uint8_t *pin = nullptr;            // 256 bytes of pinned host memory: few-byte read-backs into pageable memory cost ~1 ms each
    uint8_t *ring = nullptr; size_t ring_at = 0; hipStream_t ring_st = nullptr;   // mapped pinned staging of h2d_small (szl_engine.hip)
    int h2d_small(void *dst, const void *src, size_t n, hipStream_t st);
    uint32_t *stat_pin = nullptr;      // 512 bytes of mapped pinned memory: k_hop_stat's per-workgroup counts
    int pick_text_form(const uint16_t *lk, int64_t lo, int64_t n, hipStream_t st);   // which form of k_match9's text a launch over link[lo, lo + n) runs (MTab::form)
    int last_text_form = 0;            // (parity tap / debug line)
    int d2h_small(void *pin_dst, const void *src, size_t n, hipStream_t st);
    uint32_t last_par_jobs = 0;        // chunk jobs of the last parallel single-member inflate
    uint64_t last_workspace_bytes = 0; // device bytes held by the side arrays after the last call (parity tap / DESIGN §3)

    // Level 0: write the stored blocks `blks` (host-built) and, if want_ck, the checksums of d_in[ck_off, ck_off+ck_len).
    int deflate_stored(const uint8_t *d_in, uint8_t *d_out, const std::vector<StoredBlk> &blks, unsigned want_ck, uint64_t ck_off,
                       uint64_t ck_len, uint32_t crc_init, uint32_t adler_init, uint32_t *crc_out, uint32_t *adler_out, hipStream_t st);
    szl_timing timing{};
    uint64_t last_nranges = 0, last_in_total = 0, last_blk_slots = 0;
    size_t last_mt_stride = 0;
    // stage B on demand: positions evaluated by walkers / by the parse itself, pilot result, form used by the last call
    uint64_t last_evaluated = 0, last_eval_fallbacks = 0;
    double last_pilot_frac = -1.0;
    bool last_lazy = false;
    int match_mode_override = -1; // debug tap: -1 = SZL_MATCH_MODE / default (2 = pilot), 0 full, 1 on demand
    // DeflateFast, single-segment calls (streaming Deflater): "inserted" bits of the buffer's history in (bit q = buffer
    // position q), and of the last 32 Ki positions out (bit 0 of fast_tail_bits = position fast_tail_start).
    // SetLevel / SetStrategy inside the (single) segment of a call: an iteration that starts at buffer position >= sw_pos_in[k] runs with
    // sw_P_in[k] (the last such k).  Set by the caller before deflate(), any number of entries; cleared by the caller.
    std::vector<int64_t> sw_pos_in;
    std::vector<LevelParams> sw_P_in;
    std::vector<uint32_t> fast_hist_in, fast_tail_bits;
    bool fast_want_tail = false;
    int64_t fast_tail_start = 0;
    DevBuf link, link4, skip4, e3dist, e3hops, mtab, tokens, visited, ranges, counts, range_tok, descs, d_segs, d_bnds, d_spans, d_tiles, d_stripes, d_so, blk_counts, blk_off,
        bsp, blp, counters, ckparts, ckoff, cubtmp, stage_in, stage_out, bad_slot, bad_range, exmap, cnmap, chain_buf, d_stored, spec_tok, d_zoff,
        inf_sym, inf_wins, inf_jobs, inf_states, inf_misc, inf_groups, hist_flags_dev, m5_scratch, d_sw_pos, d_sw_P;   // parallel decode of one member (szl_api_inflate.hip)
    hipEvent_t ev[8];
    // checksum kernels run beside stages A-C on this stream (they only share the input bytes)
    hipStream_t side = nullptr;
    std::vector<DevBuf *> all_bufs();
    size_t device_bytes();
    void trim(size_t keep);          // releases the largest side arrays until at most `keep` bytes of them are left (an idle engine)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_guard = nullptr, ev_gjoin = nullptr, ev_zfork = nullptr, ev_zjoin = nullptr;
};

} // namespace szl

// The handle behind szl_engine_create (include/szl.h): an Engine, the device it lives on, and — for the streaming objects, which take their
// engine from a pool (szl_api.hip: engine_take / engine_give) — the object's long device buffers while no object owns it.
struct szl_engine {
    szl::Engine e;
    int device = 0;
    szl::DevBuf io_a, io_b, io_c, io_d;   // Deflater: d_in, d_out; Inflater: d_bulk_in, d_bulk_out
    hipStream_t st_a = nullptr, st_b = nullptr;   // Deflater: its upload stream and its parts' stream, kept with the engine between objects (a stream costs ~5 ms to make:
                                                  // the first two SetInput calls of every GZipOutputStream paid that)
};
namespace szl {
// A streaming Deflater / Inflater is often short-lived — GZipOutputStream makes a new Deflater per stream (S/GZip/GzipOutputStream.cs:87),
// GZipInputStream a new Inflater (S/GZip/GzipInputStream.cs:86) — and its engine's work space is not: ~19 bytes of device memory per input
// byte of the longest stream it has compressed, whose allocation cost a fresh object 20-800 ms of its first Finish()
// (profiles/r05/finish_breakdown.log).  Engines the streaming objects give back are kept, work space and all — SZL_ENGINE_POOL of them per
// process (2; 0: none); szl_multi_release() frees them.
szl_engine *engine_take();
void engine_give(szl_engine *e);
void engine_pool_release();
void object_born();                  // a streaming object (szl_deflater / szl_inflater) was created / destroyed: when the last one goes the
void object_gone();                  // pools shrink to SZL_IDLE_KEEP_MIB (szl_api.hip)
}

