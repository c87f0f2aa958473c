// szl_api.hip — the C ABI of include/szl.h: Deflater state machine + batch entry points.
// Mirrors C/Deflater.cs (state constants :126-140, Deflate loop :427-522) above the device engine.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <deque>
#include <chrono>
#include <condition_variable>
#include <atomic>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "szl_engine.h"

using namespace szl;

namespace szl { uint32_t links_guard_trips(); }

static thread_local int g_device = 0;

namespace { std::mutex g_idle_mu; std::vector<szl_engine *> g_idle; }
namespace szl {
szl_engine *engine_take() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_idle_mu);
        for (size_t i = g_idle.size(); i-- > 0;)
            if (g_idle[i]->device == dev) { szl_engine *e = g_idle[i]; g_idle.erase(g_idle.begin() + (ptrdiff_t)i); return e; }
    }
    return szl_engine_create();
}
void engine_give(szl_engine *e) {
    if (!e) return;
    Engine &E = e->e;                                    // what a call may have left set (the callers clear these themselves; belt and braces)
    E.in_ready = nullptr; E.part = Engine::PartRun{}; E.sw_pos_in.clear(); E.sw_P_in.clear(); E.fast_hist_in.clear(); E.fast_tail_bits.clear();
    E.fast_want_tail = false; E.match_mode_override = -1;
    const int keep = knob("SZL_ENGINE_POOL", 2);
    {
        std::lock_guard<std::mutex> lk(g_idle_mu);
        if ((int)g_idle.size() < keep) { g_idle.push_back(e); return; }
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    const int dev = e->device;                           // read before the engine is deleted
    if (cur != dev) (void)hipSetDevice(dev);
    szl_engine_destroy(e);
    if (cur != dev) (void)hipSetDevice(cur);
}
// What the library keeps when NO streaming object is alive (round 6).  The pools exist for the caller who makes one object after
// another — a GZipOutputStream makes a new Deflater per stream (S/GZip/GzipOutputStream.cs:87), and the ~19 bytes of device memory per
// input byte cost its first Finish() 20-800 ms — but a host that compressed one gigabyte an hour ago should not sit on twenty.  When the
// last szl_deflater / szl_inflater has been gone for SZL_IDLE_TRIM_MS (below) the idle engines' side arrays and the pinned pool shrink to
// SZL_IDLE_KEEP_MIB each (1024; largest buffers first: what stays is what many small streams need); szl_trim() gives everything back.
static std::atomic<long> g_live_objects{0};
void object_born() { g_live_objects.fetch_add(1, std::memory_order_acq_rel); }
static void engine_pool_trim(size_t keep) {
    std::lock_guard<std::mutex> lk(g_idle_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    size_t total = 0;
    for (szl_engine *e : g_idle) total += e->e.device_bytes() + e->io_a.cap + e->io_b.cap + e->io_c.cap + e->io_d.cap;
    for (szl_engine *e : g_idle) {
        if (total <= keep) break;
        (void)hipSetDevice(e->device);
        for (szl::DevBuf *b : {&e->io_a, &e->io_b, &e->io_c, &e->io_d}) if (total > keep) { total -= b->cap; b->release(); }
        const size_t had = e->e.device_bytes(), want = total > keep ? (had > total - keep ? had - (total - keep) : 0) : had;
        e->e.trim(want);
        total -= had - e->e.device_bytes();
    }
    (void)hipSetDevice(cur);
}
static void idle_trim_now() {
    const size_t keep = (size_t)std::max(0, knob("SZL_IDLE_KEEP_MIB", 1024)) << 20;
    engine_pool_trim(keep);
    (void)pin_pool_trim(keep);
}
// ... but not at once: a caller who makes one object after another (a Deflater per stream) destroys its last object many times a
// second, and a gigabyte of pinned memory costs hundreds of milliseconds to get back (one GZipOutputStream per GiB: 20 -> 200 ms of
// Write() with an immediate trim).  The pools shrink when no object has existed for SZL_IDLE_TRIM_MS (2000; 0 = at once), on a thread
// that sleeps until then and is joined when the library is unloaded.
namespace {
struct IdleTrimmer {
    std::thread th; std::mutex mu; std::condition_variable cv;
    bool quit = false, armed = false; uint64_t gen = 0;
    void arm() {
        std::lock_guard<std::mutex> lk(mu);
        gen++; armed = true;
        if (!th.joinable()) th = std::thread([this]() { run(); });
        cv.notify_all();
    }
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&]() { return quit || armed; });
            if (quit) return;
            const uint64_t g = gen;
            armed = false;
            const int ms = std::max(1, knob("SZL_IDLE_TRIM_MS", 2000));
            if (cv.wait_for(lk, std::chrono::milliseconds(ms), [&]() { return quit || gen != g; })) continue;   // (somebody came: a new deadline when they leave)
            if (g_live_objects.load(std::memory_order_acquire) != 0) continue;
            lk.unlock();
            idle_trim_now();
            lk.lock();
        }
    }
    ~IdleTrimmer() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
} g_idle_trimmer;
}
void object_gone() {
    if (g_live_objects.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    if (knob("SZL_IDLE_TRIM_MS", 2000) <= 0) idle_trim_now();
    else g_idle_trimmer.arm();
}
void engine_pool_release() {
    std::vector<szl_engine *> idle;
    { std::lock_guard<std::mutex> lk(g_idle_mu); idle.swap(g_idle); }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (szl_engine *e : idle) { (void)hipSetDevice(e->device); szl_engine_destroy(e); }
    (void)hipSetDevice(cur);
}
}

extern "C" {

const char *szl_strerror(int st) {
    switch (st) {
    case SZL_OK: return "ok";
    case SZL_E_ARG: return "argument out of range";
    case SZL_E_STATE: return "invalid operation for the current state";
    case SZL_E_DEVICE: return "HIP device error";
    case SZL_E_NOMEM: return "out of device memory";
    case SZL_E_UNSUPPORTED: return "not supported on the device path";
    case SZL_E_OUTPUT_TOO_SMALL: return "output region too small";
    case SZL_E_HEADER_CHECKSUM: return "Header checksum illegal";
    case SZL_E_METHOD_UNKNOWN: return "Compression Method unknown";
    case SZL_E_ILLEGAL_LEN_CODE: return "Illegal rep length code";
    case SZL_E_ILLEGAL_DIST_CODE: return "Illegal rep dist code";
    case SZL_E_ADLER_MISMATCH: return "Adler chksum doesn't match";
    case SZL_E_UNKNOWN_BLOCK: return "Unknown block type";
    case SZL_E_BROKEN_STORED: return "broken uncompressed block";
    case SZL_E_CODELEN_ZERO: return "Encountered invalid codelength 0";
    case SZL_E_DYN_HEADER: return "invalid dynamic block header";
    case SZL_E_UNEXPECTED_EOF: return "Unexpected EOF";
    case SZL_E_WINDOW_FULL: return "Window full";
    case SZL_E_CODE_OVERSUBSCRIBED: return "Index was outside the bounds of the array (over-subscribed code lengths)";
    case SZL_E_INDEX: return "Index was outside the bounds of the array";
    default: return "unknown status";
    }
}
const char *szl_last_error(void) { return szl::last_error(); }

int szl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}
int szl_set_device(int ordinal) {
    if (hipSetDevice(ordinal) != hipSuccess) { set_error("hipSetDevice(%d) failed", ordinal); return SZL_E_DEVICE; }
    g_device = ordinal;
    return 0;
}

uint64_t szl_deflate_bound(uint64_t n) {
    // Non-stored blocks cost <= 9 bits per literal / <= 31 bits per >=3-byte match (opt_len <= static_len,
    // C/DeflaterHuffman.cs:824-828); stored blocks add <= 5 bytes per 16384 tokens; plus sync padding / trailer.
    return n + n / 3 + (n >> 12) + 256;
}

szl_engine *szl_engine_create(void) {
    if (szl_device_count() <= 0) { set_error("no gfx950 device available"); return nullptr; }
    szl_engine *e = new (std::nothrow) szl_engine();
    if (e) (void)hipGetDevice(&e->device);
    return e;
}
void szl_engine_destroy(szl_engine *e) {
    if (!e) return;
    e->io_a.release(); e->io_b.release(); e->io_c.release(); e->io_d.release();
    if (e->st_a) (void)hipStreamDestroy(e->st_a);
    if (e->st_b) (void)hipStreamDestroy(e->st_b);
    delete e;
}

int szl_engine_last_timing(const szl_engine *e, szl_timing *t) {
    if (!e || !t) return SZL_E_ARG;
    *t = e->e.timing;
    t->links_guard_trips = szl::links_guard_trips();
    return 0;
}

static int zlib_header(int level, bool preset_dict = false);
static int build_batch(szl_stream *streams, size_t n, unsigned flags, int level, std::vector<SegDev> &segs, std::vector<uint64_t> &bnds,
                       uint64_t *in_total, uint64_t *out_total) {
    segs.clear(); bnds.clear();
    uint64_t it = 0, ot = 0;
    const bool gzip = flags & SZL_F_GZIP;
    const bool nowrap = (flags & SZL_F_NOWRAP) || gzip;
    for (size_t i = 0; i < n; i++) {
        szl_stream &s = streams[i];
        s.status = 0; s.out_len = 0; s.crc32 = 0; s.adler32 = 1;
        if (s.out_off & 3) { set_error("stream %zu: out_off must be a multiple of 4", i); return SZL_E_ARG; }
        if (s.out_cap < szl_deflate_bound(s.in_len) + (gzip ? 18 : (nowrap ? 0 : 6))) { set_error("stream %zu: out_cap %llu < bound", i, (unsigned long long)s.out_cap); return SZL_E_OUTPUT_TOO_SMALL; }
        SegDev d{};
        d.buf_off = s.in_off; d.abs0 = 0; d.seg_start = 0; d.seg_end = (int64_t)s.in_len;
        d.bnd_off = (uint32_t)bnds.size(); d.bnd_cnt = 1; bnds.push_back(s.in_len);
        d.out_off = s.out_off; d.out_cap = s.out_cap; d.stream_idx = (uint32_t)i;
        d.start_bit = gzip ? 80 : (nowrap ? 0 : 16); d.adler_init = 1; d.crc_init = 0;
        d.hdr_word = gzip ? s.reserved : (uint32_t)zlib_header(level);
        if (flags & SZL_F_SYNC_FLUSH_BEFORE_FINISH) { d.finish = 0; d.flags = SEG_SYNC_PAD | SEG_EXTRA_FINAL_EMPTY; }
        else { d.finish = 1; d.flags = 0; }
        if (!nowrap) d.flags |= SEG_ZLIB_TRAILER | SEG_ZLIB_HEADER;
        if (gzip) d.flags |= SEG_GZIP;
        segs.push_back(d);
        it = std::max(it, s.in_off + s.in_len);
        ot = std::max(ot, s.out_off + s.out_cap);
    }
    *in_total = it; *out_total = ot;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Level 0: DeflaterEngine.DeflateStored (C/DeflaterEngine.cs:614-649) driven by FillWindow (:366-400).  Block cuts depend
// only on how many bytes each engine call sees, i.e. on the SetInput chunk sizes — pure arithmetic, replayed here.
struct L0State { int64_t base = 0; int strstart = 1, blockStart = 1, lookahead = 0; uint64_t fed = 0; /* bytes FillWindow has copied (== engine.TotalIn) */ };
struct L0Blk { uint64_t abs_off; uint32_t len; uint32_t last; };
static bool l0_engine_deflate(L0State &s, uint64_t &avail, bool flush, bool finish, std::vector<L0Blk> &out) { // Deflate :104
    enum { MIN_LOOKAHEAD = 262, MAX_BLOCK_SIZE = 65531 };
    for (;;) {
        if (s.strstart >= WSIZE + MAX_DIST) { s.strstart -= WSIZE; s.blockStart -= WSIZE; s.base += WSIZE; } // SlideWindow :441
        if (s.lookahead < MIN_LOOKAHEAD && avail > 0) {
            int64_t more = 2 * WSIZE - s.lookahead - s.strstart;
            if ((uint64_t)more > avail) more = (int64_t)avail;
            avail -= (uint64_t)more; s.lookahead += (int)more; s.fed += (uint64_t)more;
        }
        const bool canFlush = flush && avail == 0;
        bool progress, emitted = false;
        if (!canFlush && s.lookahead == 0) progress = false;                                           // :616-619
        else {
            s.strstart += s.lookahead; s.lookahead = 0;
            int storedLength = s.strstart - s.blockStart;
            if (storedLength >= MAX_BLOCK_SIZE || (s.blockStart < WSIZE && storedLength >= MAX_DIST) || canFlush) { // :626-628
                bool last = finish;
                if (storedLength > MAX_BLOCK_SIZE) { storedLength = MAX_BLOCK_SIZE; last = false; }
                out.push_back(L0Blk{(uint64_t)(s.blockStart - 1 + s.base), (uint32_t)storedLength, last ? 1u : 0u});
                s.blockStart += storedLength;
                emitted = true;
                progress = !(last || storedLength == 0);
            } else progress = true;
        }
        if (emitted || !progress) return progress; // `while (pending.IsFlushed && progress)` :135 — a block makes pending non-empty
    }
}
// Replays Write(chunk)... then Flush()/Finish() (CS/DeflaterOutputStream.cs:506,388,100)
// The first `ndrained` chunks were followed by Deflate() calls before Flush()/Finish() (the DeflaterOutputStream.Write pattern),
// so the engine saw them with flush = finish = false; later chunks are first seen with the final flags — which matters at
// level 0: DeflateStored marks a block final as soon as `finish` is set, even if input remains (:630-631).
static void l0_replay(L0State &s, const std::vector<uint64_t> &chunks, size_t ndrained, bool flush, bool finish, std::vector<L0Blk> &out) {
    uint64_t tail = 0;
    for (size_t i = 0; i < chunks.size(); i++) {
        if (i >= ndrained) { tail += chunks[i]; continue; }
        uint64_t avail = chunks[i];
        while (l0_engine_deflate(s, avail, false, false, out)) { } // Deflater.Deflate keeps calling the engine until it returns false
    }
    if (flush || finish) {
        uint64_t avail = tail;
        while (l0_engine_deflate(s, avail, true, finish, out)) { }
    }
}
int szl_debug_host_copy(void *dst, const void *src, size_t n) {
    if ((!dst || !src) && n) return SZL_E_ARG;
    try { host_copy(dst, src, n); } catch (...) { return SZL_E_NOMEM; }
    return 0;
}
// Parity tap (host arithmetic only, no device): block list of a level-0 stream. rows: abs_off, len, last.
int szl_debug_stored_layout(const uint64_t *chunks, size_t nchunks, int flush_before_finish, uint64_t *rows, size_t cap_rows, size_t *n_rows) {
    // flush_before_finish bit 0: Flush() before Finish(); bits 8..: number of chunks NOT followed by a Deflate() call (counted
    // from the end); bits 32.. are not available in an int, so the dictionary length rides in rows[0] on entry.
    L0State s; std::vector<L0Blk> out;
    std::vector<uint64_t> cs(chunks, chunks + nchunks);
    const size_t undrained = (size_t)((unsigned)flush_before_finish >> 8);
    const size_t nd = cs.size() >= undrained ? cs.size() - undrained : 0;
    const int dict_len = (rows && cap_rows) ? (int)rows[0] : 0;
    s.strstart = s.blockStart = 1 + dict_len;
    if (flush_before_finish & 1) { l0_replay(s, cs, nd, true, false, out); l0_replay(s, {}, 0, false, true, out); }
    else l0_replay(s, cs, nd, false, true, out);
    for (size_t i = 0; i < out.size() && i < cap_rows; i++) { rows[3 * i] = out[i].abs_off; rows[3 * i + 1] = out[i].len; rows[3 * i + 2] = out[i].last; }
    if (n_rows) *n_rows = out.size();
    return 0;
}

static int zlib_header(int level, bool preset_dict) { // C/Deflater.cs:436-461
    int header = (8 + ((15 - 8) << 4)) << 8;
    int level_flags = (level - 1) >> 1;
    if (level_flags < 0 || level_flags > 3) level_flags = 3;
    header |= level_flags << 6;
    if (preset_dict) header |= 0x20; // PRESET_DICT :451-455
    header += 31 - (header % 31);
    return header;
}

extern "C++" {
namespace szl {
int region_checksums(const uint8_t *base, const std::vector<std::pair<uint64_t, uint64_t>> &regs, unsigned want,
                     std::vector<std::pair<uint32_t, uint32_t>> &out, const std::vector<std::pair<uint32_t, uint32_t>> *init, hipStream_t st);
}
}
// Level 0 batch: every stream == new Deflater(0, nowrap); SetInput(all, in <= 1 GiB pieces); [Flush();] Finish()
static int deflate_batch_stored(szl_engine *e, const uint8_t *d_in, uint8_t *d_out, szl_stream *streams, size_t n, unsigned flags, hipStream_t st) {
    const bool gzip = flags & SZL_F_GZIP;              // GZipOutputStream: a raw deflate stream inside the RFC 1952 member (:315-375)
    const bool nowrap = (flags & SZL_F_NOWRAP) || gzip;
    std::vector<StoredBlk> sb;
    std::vector<std::pair<uint64_t, uint64_t>> regs(n);
    for (size_t i = 0; i < n; i++) {
        szl_stream &s = streams[i];
        std::vector<uint64_t> chunks;
        for (uint64_t o = 0; o < s.in_len; o += (1ull << 30)) chunks.push_back(std::min<uint64_t>(1ull << 30, s.in_len - o));
        if (chunks.empty()) chunks.push_back(0);
        L0State st0; std::vector<L0Blk> blks;
        if (flags & SZL_F_SYNC_FLUSH_BEFORE_FINISH) { l0_replay(st0, chunks, chunks.size(), true, false, blks); l0_replay(st0, {}, 0, false, true, blks); }
        else l0_replay(st0, chunks, chunks.size(), false, true, blks);
        uint64_t o = s.out_off + (gzip ? 10 : (nowrap ? 0 : 2));
        for (auto &b : blks) { sb.push_back(StoredBlk{s.in_off + b.abs_off, o, b.len, b.last}); o += 5 + (uint64_t)b.len; }
        s.out_len = o - s.out_off + (gzip ? 8 : (nowrap ? 0 : 4));
        s.status = s.out_len <= s.out_cap ? 0 : SZL_E_OUTPUT_TOO_SMALL;
        if (s.status) { set_error("stream %zu: out_cap too small for level 0", i); return SZL_E_OUTPUT_TOO_SMALL; }
        regs[i] = {s.in_off, s.in_len};
    }
    int rc = e->e.deflate_stored(d_in, d_out, sb, 0, 0, 0, 0, 1, nullptr, nullptr, st);
    if (rc) return rc;
    unsigned want = (((flags & SZL_F_CRC32) || gzip) ? 1u : 0u) | (((flags & SZL_F_ADLER32) || !nowrap) ? 2u : 0u);
    std::vector<std::pair<uint32_t, uint32_t>> cks(n, {0u, 1u});
    if (want && (rc = szl::region_checksums(d_in, regs, want, cks, nullptr, st))) return rc;
    for (size_t i = 0; i < n; i++) {
        streams[i].crc32 = cks[i].first; streams[i].adler32 = cks[i].second;
        if (gzip) {    // ID1 ID2 CM FLG MTIME XFL OS ... CRC32 ISIZE (little endian), as k_finish writes them for the coded levels
            const uint32_t t = streams[i].reserved, c = cks[i].first, isz = (uint32_t)(streams[i].in_len & 0xffffffffu);
            uint8_t hb[10] = {0x1F, 0x8B, 8, 0, (uint8_t)t, (uint8_t)(t >> 8), (uint8_t)(t >> 16), (uint8_t)(t >> 24), 0, 255};
            uint8_t tb[8] = {(uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24), (uint8_t)isz, (uint8_t)(isz >> 8), (uint8_t)(isz >> 16), (uint8_t)(isz >> 24)};
            if (hipMemcpyAsync(d_out + streams[i].out_off, hb, 10, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(d_out + streams[i].out_off + streams[i].out_len - 8, tb, 8, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) return SZL_E_DEVICE;
        } else if (!nowrap) { // zlib header (level_flags = 3 for level 0: (0-1)>>1 < 0 -> 3, C/Deflater.cs:440-444) and Adler trailer
            int hdr = zlib_header(0);
            uint8_t hb[2] = {(uint8_t)(hdr >> 8), (uint8_t)hdr};
            uint32_t a = cks[i].second;
            uint8_t tb[4] = {(uint8_t)(a >> 24), (uint8_t)(a >> 16), (uint8_t)(a >> 8), (uint8_t)a};
            if (hipMemcpyAsync(d_out + streams[i].out_off, hb, 2, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(d_out + streams[i].out_off + streams[i].out_len - 4, tb, 4, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) return SZL_E_DEVICE;
        }
    }
    return 0;
}

int szl_deflate_batch_device(szl_engine *e, const void *d_in, void *d_out, szl_stream *streams, size_t n_streams, int level,
                             int strategy, unsigned flags, void *hip_stream) {
    if (!e || (!streams && n_streams)) return SZL_E_ARG;
    if (level == -1) level = 6;
    if (level == 0) return deflate_batch_stored(e, (const uint8_t *)d_in, (uint8_t *)d_out, streams, n_streams, flags, (hipStream_t)hip_stream);
    LevelParams P;
    int rc = level_params(level, strategy, &P);
    if (rc) { set_error("level %d: %s", level, szl_strerror(rc)); return rc; }
    if (strategy < 0 || strategy > 2) return SZL_E_ARG;
    if (((uintptr_t)d_out) & 3) { set_error("d_out must be 4-byte aligned"); return SZL_E_ARG; }
    std::vector<SegDev> segs; std::vector<uint64_t> bnds; uint64_t in_total, out_total;
    if ((rc = build_batch(streams, n_streams, flags, level, segs, bnds, &in_total, &out_total))) return rc;
    std::vector<SegOut> res;
    unsigned want = ((flags & (SZL_F_CRC32 | SZL_F_GZIP)) ? 1u : 0u) | (((flags & SZL_F_ADLER32) || !(flags & (SZL_F_NOWRAP | SZL_F_GZIP))) ? 2u : 0u);
    hipStream_t st = (hipStream_t)hip_stream;
    rc = e->e.deflate((const uint8_t *)d_in, in_total, (uint8_t *)d_out, out_total, segs, bnds, P, want, res, st);
    if (rc) return rc;
    for (size_t i = 0; i < n_streams; i++) {
        streams[i].out_len = res[i].out_bytes;
        streams[i].crc32 = res[i].crc32;
        streams[i].adler32 = res[i].adler32;
        streams[i].status = res[i].out_bytes <= streams[i].out_cap ? 0 : SZL_E_OUTPUT_TOO_SMALL;
    }
    return 0;
}

int szl_deflate_batch_host(szl_engine *e, const void *h_in, void *h_out, szl_stream *streams, size_t n_streams, int level,
                           int strategy, unsigned flags) {
    if (!e || (!streams && n_streams)) return SZL_E_ARG;
    uint64_t in_total = 0, out_total = 0;
    for (size_t i = 0; i < n_streams; i++) {
        in_total = std::max(in_total, streams[i].in_off + streams[i].in_len);
        out_total = std::max(out_total, streams[i].out_off + streams[i].out_cap);
    }
    int rc;
    if ((rc = e->e.stage_in.ensure(in_total + 64))) return rc;
    if ((rc = e->e.stage_out.ensure(out_total + 64))) return rc;
    // One long stream that will go through the window pipeline: the input is copied by a second host thread, 32 MiB at a time
    // on its own stream, while the engine already works on the windows that have arrived (Engine::in_ready).
    const int lv = level == -1 ? 6 : level;
    if (Engine::uses_window_pipeline(n_streams, lv >= 5, n_streams == 1 ? streams[0].in_len : 0, nullptr) && SZL_LABKNOB("SZL_H2D_OVERLAP", 1)) {
        volatile uint64_t ready = 0;
        int copy_rc = 0, dev = 0;
        (void)hipGetDevice(&dev);
        std::thread copier([&]() {
            if (hipSetDevice(dev) != hipSuccess) { copy_rc = SZL_E_DEVICE; ready = in_total; return; }
            hipStream_t cs = nullptr;
            if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { copy_rc = SZL_E_DEVICE; ready = in_total; return; }
            const uint64_t piece = 32ull << 20;
            for (uint64_t o = 0; o < in_total; o += piece) {
                const uint64_t k = std::min<uint64_t>(piece, in_total - o);
                if (hipMemcpyAsync((uint8_t *)e->e.stage_in.p + o, (const uint8_t *)h_in + o, k, hipMemcpyHostToDevice, cs) != hipSuccess ||
                    hipStreamSynchronize(cs) != hipSuccess) { copy_rc = SZL_E_DEVICE; break; }
                __atomic_store_n((uint64_t *)&ready, o + k, __ATOMIC_RELEASE);
            }
            __atomic_store_n((uint64_t *)&ready, in_total, __ATOMIC_RELEASE); // (also on failure: never leave the engine waiting)
            (void)hipStreamDestroy(cs);
        });
        e->e.in_ready = &ready;
        rc = szl_deflate_batch_device(e, e->e.stage_in.p, e->e.stage_out.p, streams, n_streams, level, strategy, flags, nullptr);
        e->e.in_ready = nullptr;
        copier.join();
        if (copy_rc) { set_error("H2D failed"); return copy_rc; }
        if (rc) return rc;
    } else {
    if (in_total && hipMemcpy(e->e.stage_in.p, h_in, in_total, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D failed"); return SZL_E_DEVICE; }
    rc = szl_deflate_batch_device(e, e->e.stage_in.p, e->e.stage_out.p, streams, n_streams, level, strategy, flags, nullptr);
    if (rc) return rc;
    }
    for (size_t i = 0; i < n_streams; i++) {
        if (streams[i].status) continue;
        if (streams[i].out_len && hipMemcpy((uint8_t *)h_out + streams[i].out_off, (uint8_t *)e->e.stage_out.p + streams[i].out_off,
                                            streams[i].out_len, hipMemcpyDeviceToHost) != hipSuccess) { set_error("D2H failed"); return SZL_E_DEVICE; }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Several devices behind the C ABI (configs 3 and 4 shard by independent streams — zip entries, gzip members).
// The streams are cut into n_dev contiguous groups of about equal input bytes; group g runs on devices[g] in its own host
// thread with its own engine (kept for the next call), staging only the bytes of its group.  No data-path collective: the
// groups never exchange anything.  One stream is never split.
struct MultiSlot { szl_engine *eng = nullptr; int device = -1; };
// The engines of the multi-device entry points are kept from call to call in this process-wide pool (slot g = group g).  The
// entry points take no handle, so they SERIALISE: g_multi_mu is held for the whole call (include/szl.h says so), and the pool is
// a deque — growing it never moves an engine another thread holds a reference to.
static std::mutex g_multi_mu;
static std::deque<MultiSlot> g_multi_slots;

static int multi_run(bool inflate, const int *devices, int n_dev, const void *h_in, void *h_out, szl_stream *streams, size_t n,
                     int level, int strategy, unsigned flags) {
    if ((!streams && n) || !devices || n_dev <= 0) return SZL_E_ARG;
    if (n == 0) return 0;
    int ndev_avail = 0;
    if (hipGetDeviceCount(&ndev_avail) != hipSuccess || ndev_avail <= 0) { set_error("no gfx950 device available"); return SZL_E_DEVICE; }
    for (int g = 0; g < n_dev; g++) if (devices[g] < 0 || devices[g] >= ndev_avail) { set_error("device ordinal %d out of range", devices[g]); return SZL_E_ARG; }
    // contiguous groups balanced by input bytes
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) total += streams[i].in_len + 1;
    std::vector<size_t> cut(n_dev + 1, n);
    cut[0] = 0;
    { uint64_t acc = 0; int g = 1;
      for (size_t i = 0; i < n && g < n_dev; i++) { acc += streams[i].in_len + 1; while (g < n_dev && acc * (uint64_t)n_dev >= total * (uint64_t)g) cut[g++] = i + 1; } }
    for (int g = 1; g <= n_dev; g++) if (cut[g] < cut[g - 1]) cut[g] = cut[g - 1];
    cut[n_dev] = n;
    std::lock_guard<std::mutex> multi_lock(g_multi_mu);          // one multi-device call at a time
    if ((int)g_multi_slots.size() < n_dev) g_multi_slots.resize(n_dev);
    std::vector<int> rcs(n_dev, 0);
    std::vector<std::string> errs(n_dev);
    auto work = [&](int g) {
        const size_t a = cut[g], b = cut[g + 1];
        if (a >= b) return;
        if (hipSetDevice(devices[g]) != hipSuccess) { rcs[g] = SZL_E_DEVICE; errs[g] = "hipSetDevice failed"; return; }
        MultiSlot &slot = g_multi_slots[g];                       // slot g is only ever used by group g's thread of one call at a time
        if (slot.eng && slot.device != devices[g]) { szl_engine_destroy(slot.eng); slot.eng = nullptr; }
        if (!slot.eng) { slot.eng = szl_engine_create(); slot.device = devices[g]; }
        if (!slot.eng) { rcs[g] = SZL_E_DEVICE; errs[g] = last_error(); return; }
        // rebase the group's byte spans so that only its own bytes cross PCIe
        uint64_t in_lo = ~0ull, in_hi = 0, out_lo = ~0ull, out_hi = 0;
        for (size_t i = a; i < b; i++) {
            in_lo = std::min(in_lo, streams[i].in_off); in_hi = std::max(in_hi, streams[i].in_off + streams[i].in_len);
            out_lo = std::min(out_lo, streams[i].out_off); out_hi = std::max(out_hi, streams[i].out_off + streams[i].out_cap);
        }
        out_lo &= ~3ull;                                          // output regions stay 4-byte aligned after the shift
        std::vector<szl_stream> local(streams + a, streams + b);
        for (auto &s : local) { s.in_off -= in_lo; s.out_off -= out_lo; }
        int rc = inflate ? szl_inflate_batch_host(slot.eng, (const uint8_t *)h_in + in_lo, (uint8_t *)h_out + out_lo, local.data(), local.size(), flags)
                         : szl_deflate_batch_host(slot.eng, (const uint8_t *)h_in + in_lo, (uint8_t *)h_out + out_lo, local.data(), local.size(), level, strategy, flags);
        if (rc) { rcs[g] = rc; errs[g] = last_error(); return; }
        for (size_t i = a; i < b; i++) {
            const szl_stream &r = local[i - a];
            streams[i].out_len = r.out_len; streams[i].crc32 = r.crc32; streams[i].adler32 = r.adler32; streams[i].status = r.status;
            streams[i].in_consumed = r.in_consumed;
        }
    };
    std::vector<std::thread> th;
    for (int g = 1; g < n_dev; g++) th.emplace_back(work, g);
    work(0);
    for (auto &t : th) t.join();
    (void)hipSetDevice(g_device);
    for (int g = 0; g < n_dev; g++) if (rcs[g]) { set_error("device %d (group %d): %s", devices[g], g, errs[g].c_str()); return rcs[g]; }
    return 0;
}
// ---------------------------------------------------------------------------------------------
// ONE stream over several engines / devices (levels 5-9): exact position-range partition, dynamically balanced.
//   * The stream is cut into UNITS at multiples of the stage-B tile — SZL_PART_UNITS (4) per device — and every device takes the
//     next unit when it is free: a device that gets cheap bytes (or is faster) simply takes more units, so the parts balance by
//     measured cost without an estimate, and without a collective (the devices never exchange input; all engines live in this
//     process).  A unit runs stages A, B and C with the window pipeline (Engine::PartRun): it needs nothing but its bytes, 64 KiB
//     of history and a few KiB of lookahead.
//   * What a unit cannot know is the iteration on which the true parse enters it — so it parses a warm-up stretch in front of
//     itself from an assumed clean state and drops those tokens.  Two parses that are clean at the same position are identical
//     from there on: the entry the warm-up arrives at is the true one iff the previous unit's parse leaves on it.  The host checks
//     exit(u-1) == entry(u) in unit order and re-runs a unit with the right entry where that fails (data whose parses never
//     re-synchronise: long runs of one byte).  No other coupling between the units.
//   * The tokens are gathered on devices[0] — which also holds the whole input — WHILE later units still run: as soon as units
//     0..u are done their offsets are known, and unit u's tokens travel on a stream of their own (peer copy where the devices can
//     reach each other, through a host buffer otherwise).  Stage D — block positions from the tokens' own lengths, Huffman trees,
//     bit packing, checksums, framing — then runs there once.
// Same bytes as one engine produces (tests/test_gpu_multi.py: several engines on one or more devices against it and the oracle).
// d_res != nullptr: the stream is RESIDENT — d_res[g] is the input arena in the memory of devices[g] (the caller uploaded it once),
// d_out_res the output arena on devices[0]; nothing travels between host and devices but the result fields.
static int stream_multi_run(const int *devices, int n_dev, const void *h_in, void *h_out, szl_stream *stream, int level, int strategy, unsigned flags,
                            const void *const *d_res = nullptr, void *d_out_res = nullptr) {
    LevelParams P;
    int rc = level_params(level, strategy, &P);
    if (rc) return rc;
    std::vector<SegDev> segs; std::vector<uint64_t> bnds_unused; uint64_t in_total0, out_total0;
    if ((rc = build_batch(stream, 1, flags, level, segs, bnds_unused, &in_total0, &out_total0))) return rc;
    const SegDev whole = segs[0];                       // buf_off = in_off, seg [0, N)
    const int64_t N = (int64_t)stream->in_len;
    const uint8_t *src = d_res ? nullptr : (const uint8_t *)h_in + stream->in_off;
    auto resident = [&](int g) -> const uint8_t * { return (const uint8_t *)d_res[g] + stream->in_off; };
    const uint64_t window = std::max<uint64_t>((uint64_t)szl::knob("SZL_WINDOW_KIB", 256 * 1024) * 1024 / B_TILE * B_TILE, B_TILE);
    const int64_t WARM = (int64_t)std::max(64, szl::knob("SZL_PART_WARM_KIB", 256)) * 1024;
    const int64_t LOOK = C_WIN_HALO + 1024 + MAX_MATCH + 64;      // bytes a unit sees beyond its end (halo of the last window + hand-over slack)
    int n_units = n_dev * std::max(1, szl::knob("SZL_PART_UNITS", 4));
    {   // a unit pays a warm-up stretch: not shorter than 16 times that (and never fewer units than devices)
        const int64_t min_unit = std::max<int64_t>(16 * WARM, (int64_t)B_TILE);
        while (n_units > n_dev && N / n_units < min_unit) n_units--;
    }
    std::vector<int64_t> cut(n_units + 1);
    for (int u = 0; u <= n_units; u++) cut[u] = u == n_units ? N : (int64_t)((uint64_t)N * (uint64_t)u / (uint64_t)n_units / B_TILE * B_TILE);
    std::lock_guard<std::mutex> multi_lock(g_multi_mu);          // one multi-device call at a time
    if ((int)g_multi_slots.size() < n_dev) g_multi_slots.resize(n_dev);
    struct Unit { int rc = 0; std::string err; int64_t entry = 0, exit = 0; uint64_t ntok = 0; int slot = -1; DevBuf toks; bool done = false; };
    std::vector<Unit> un(n_units);
    std::mutex mu; std::condition_variable cv;
    std::atomic<int> next_unit{0};
    std::vector<int> slot_rc(n_dev, 0);
    std::vector<std::string> slot_err(n_dev);
    auto free_units = [&]() { for (auto &x : un) if (x.toks.p) { if (x.slot >= 0) (void)hipSetDevice(devices[x.slot]); x.toks.release(); } (void)hipSetDevice(g_device); };

    // one unit on slot g's engine (the calling thread has made devices[g] current and the engine exists)
    auto run_unit = [&](int g, int u, int64_t force_entry) {
        Unit &o = un[u];
        o.rc = 0; o.slot = g;
        Engine &E = g_multi_slots[g].eng->e;
        const int64_t first = cut[u], pend = cut[u + 1];
        const int64_t warm_from = u == 0 ? -1 : std::max<int64_t>(0, first - WARM) / B_TILE * B_TILE;
        // bytes of this unit: 64 KiB of history in front of the warm-up, a little lookahead behind the unit; engine 0 holds the
        // whole stream (stage D and the checksums need it) and runs its units on that copy
        const int64_t b0 = g == 0 ? 0 : (u == 0 ? 0 : std::max<int64_t>(0, (force_entry >= 0 ? std::min(force_entry, first) : warm_from) - 65536));
        const int64_t b1 = g == 0 ? N : std::min<int64_t>(N, pend + LOOK);
        const uint64_t nb = (uint64_t)(b1 - b0);
        int r;
        if (g != 0 && !d_res) {
            if ((r = E.stage_in.ensure(nb + 64))) { o.rc = r; o.err = last_error(); return; }
            if (nb && hipMemcpy(E.stage_in.p, src + b0, nb, hipMemcpyHostToDevice) != hipSuccess) { o.rc = SZL_E_DEVICE; o.err = "H2D failed"; return; }
        }
        const uint8_t *unit_in = d_res ? resident(g) + b0 : (const uint8_t *)E.stage_in.p;   // (engine 0: b0 == 0, the whole stream)
        SegDev sg{};
        sg.buf_off = 0; sg.abs0 = (uint64_t)b0;               // window bases follow the absolute position (C/DeflaterEngine.cs:371,:771)
        sg.seg_start = (u == 0 ? 0 : (warm_from >= 0 ? warm_from : first)) - b0; sg.seg_end = b1 - b0;
        sg.bnd_off = 0; sg.bnd_cnt = 1;
        // the only boundary that matters is the true end of the stream (InsertString needs three bytes, :780): a unit that does not
        // see it has none within reach
        std::vector<uint64_t> bnds{(uint64_t)(b1 == N ? N - b0 : (b1 - b0) + 64)};
        sg.finish = whole.finish; sg.flags = 0; sg.out_off = 0; sg.out_cap = 0; sg.start_bit = 0; sg.adler_init = 1; sg.crc_init = 0;
        E.part = Engine::PartRun{};
        E.part.active = true; E.part.first = first - b0; E.part.parse_end = pend - b0;
        E.part.warm_from = (u == 0 || force_entry >= 0) ? -1 : warm_from - b0;
        E.part.force_entry = force_entry >= 0 ? force_entry - b0 : -1;
        if (E.part.force_entry >= 0) sg.seg_start = std::min<int64_t>(sg.seg_start, E.part.force_entry);
        std::vector<SegOut> res;
        r = E.deflate_windowed(unit_in, nb, nullptr, 0, sg, bnds, P, 0, res, nullptr, window);
        const Engine::PartRun pr = E.part;
        E.part = Engine::PartRun{};
        if (r == SZL_E_STATE && force_entry < 0 && u != 0) { o.entry = -1; o.exit = -1; o.ntok = 0; return; }   // the warm-up found no clean hand-over: decided in the chain below
        if (r) { o.rc = r; o.err = last_error(); return; }
        o.entry = pr.entry + b0; o.exit = pr.exit + b0; o.ntok = pr.tok_count;
        // the unit's tokens leave the engine's buffer (its next unit overwrites it)
        if ((r = o.toks.ensure((o.ntok + 16) * 4))) { o.rc = r; o.err = last_error(); return; }
        if (o.ntok && (hipMemcpy(o.toks.p, E.tokens.p, o.ntok * 4, hipMemcpyDeviceToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) {   // (the gather reads them on another stream)
            o.rc = SZL_E_DEVICE; o.err = "token copy failed"; return;
        }
    };
    auto worker = [&](int g) {
        auto fail = [&](int r, const std::string &e) { std::lock_guard<std::mutex> lk(mu); slot_err[g] = e; slot_rc[g] = r; cv.notify_all(); };   // (published under the mutex the waiter reads them with)
        if (hipSetDevice(devices[g]) != hipSuccess) { fail(SZL_E_DEVICE, "hipSetDevice failed"); return; }
        MultiSlot &slot = g_multi_slots[g];
        if (slot.eng && slot.device != devices[g]) { szl_engine_destroy(slot.eng); slot.eng = nullptr; }
        if (!slot.eng) { slot.eng = szl_engine_create(); slot.device = devices[g]; }
        if (!slot.eng) { fail(SZL_E_DEVICE, last_error()); return; }
        if (g == 0 && !d_res) {   // engine 0 holds the whole stream
            Engine &E = slot.eng->e;
            int r = E.stage_in.ensure((uint64_t)N + 64);
            if (r) { fail(r, last_error()); return; }
            if (N && hipMemcpy(E.stage_in.p, src, (size_t)N, hipMemcpyHostToDevice) != hipSuccess) { fail(SZL_E_DEVICE, "H2D failed"); return; }
        }
        for (;;) {
            const int u = next_unit.fetch_add(1);
            if (u >= n_units) break;
            run_unit(g, u, -1);
            { std::lock_guard<std::mutex> lk(mu); un[u].done = true; }
            cv.notify_all();
            if (un[u].rc) break;
        }
    };
    std::vector<std::thread> th;
    for (int g = 0; g < n_dev; g++) th.emplace_back(worker, g);

    // ---- the main thread: hand-over chain and token gather, in unit order, while the workers run
    if (hipSetDevice(devices[0]) != hipSuccess) { for (auto &t : th) t.join(); return SZL_E_DEVICE; }
    DevBuf nt;                                           // the stream's tokens on device 0 (a token covers at least one byte)
    hipStream_t gst = nullptr;
    std::vector<uint8_t> bounce;                         // devices that cannot reach device 0: through the host
    int fatal = 0; std::string fatal_msg;
    auto set_fatal = [&](int r, const std::string &m) { if (!fatal) { fatal = r; fatal_msg = m; } };
    // sized for text (a token covers three bytes on average there) and grown when a unit's tokens do not fit — a token covers at
    // least one byte, so 4 bytes per input byte is the ceiling, but reserving that up front held 4 GiB per GiB of input
    if ((rc = nt.ensure(std::min<uint64_t>(((uint64_t)N + 16) * 4, ((uint64_t)N / 2 + 65536) * 4)))) set_fatal(rc, last_error());
    if (!fatal && hipStreamCreateWithFlags(&gst, hipStreamNonBlocking) != hipSuccess) set_fatal(SZL_E_DEVICE, "stream for the token gather");
    uint64_t at = 0;
    int reruns = 0;
    bool workers_joined = false;
    auto join_workers = [&]() { if (!workers_joined) { for (auto &t : th) t.join(); workers_joined = true; (void)hipSetDevice(devices[0]); } };
    auto grow_nt = [&](uint64_t need_bytes) -> bool {       // (devices[0] is current; copies already queued on gst are waited for)
        if (need_bytes <= nt.cap) return true;
        if (gst && hipStreamSynchronize(gst) != hipSuccess) return false;
        uint64_t want = std::max<uint64_t>(need_bytes, 2 * (uint64_t)nt.cap);
        want = std::max<uint64_t>(need_bytes, std::min<uint64_t>(want, ((uint64_t)N + 16) * 4));
        DevBuf nb;
        if (nb.ensure(want)) return false;
        if (at && hipMemcpy(nb.p, nt.p, at * 4, hipMemcpyDeviceToDevice) != hipSuccess) { nb.release(); return false; }
        nt.release();
        nt = nb;
        return true;
    };
    auto gather = [&](int u) -> bool {
        Unit &x = un[u];
        if (!x.ntok) return true;
        if (!grow_nt((at + x.ntok + 16) * 4)) return false;
        const int sd = devices[x.slot];
        int can = 1;
        if (sd != devices[0] && hipDeviceCanAccessPeer(&can, devices[0], sd) != hipSuccess) can = 0;
        if (knob("SZL_PART_HOST_GATHER", 0) != 0) can = 0;     // (tools/gpu_two_device_check.sh: the host-staged form on a box whose devices do reach each other)
        if ((sd == devices[0] && knob("SZL_PART_HOST_GATHER", 0) == 0) || can) {
            if (hipMemcpyPeerAsync((uint32_t *)nt.p + at, devices[0], x.toks.p, sd, x.ntok * 4, gst) != hipSuccess) return false;
        } else {
            bounce.resize(x.ntok * 4);
            if (hipSetDevice(sd) != hipSuccess || hipMemcpy(bounce.data(), x.toks.p, x.ntok * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
            if (hipSetDevice(devices[0]) != hipSuccess || hipMemcpy((uint32_t *)nt.p + at, bounce.data(), x.ntok * 4, hipMemcpyHostToDevice) != hipSuccess) return false;
        }
        at += x.ntok;
        return true;
    };
    for (int u = 0; u < n_units && !fatal; u++) {
        if (!workers_joined) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { if (un[u].done) return true; for (int g = 0; g < n_dev; g++) if (slot_rc[g]) return true; return false; });
            if (!un[u].done) { lk.unlock(); join_workers(); }
        }
        for (int g = 0; g < n_dev && !fatal; g++) if (slot_rc[g]) set_fatal(slot_rc[g], "device " + std::to_string(devices[g]) + ": " + slot_err[g]);
        if (fatal) break;
        if (!un[u].done) { set_fatal(SZL_E_STATE, "a unit was never run"); break; }
        if (un[u].rc) { set_fatal(un[u].rc, "unit " + std::to_string(u) + " on device " + std::to_string(devices[un[u].slot]) + ": " + un[u].err); break; }
        if (u > 0 && un[u].entry != un[u - 1].exit) {
            // the warm-up did not arrive on the iteration the true parse enters this unit on: run it again from the right one, on
            // engine 0 (it holds the whole stream) once the workers are through — rare, and sequential by nature (its exit decides
            // about the next unit)
            join_workers();
            if (un[u].toks.p) { (void)hipSetDevice(devices[un[u].slot]); un[u].toks.release(); (void)hipSetDevice(devices[0]); }
            run_unit(0, u, un[u - 1].exit);
            reruns++;
            if (un[u].rc) { set_fatal(un[u].rc, "unit " + std::to_string(u) + " (re-run): " + un[u].err); break; }
            if (un[u].entry != un[u - 1].exit) { set_fatal(SZL_E_STATE, "unit " + std::to_string(u) + ": forced entry not honoured"); break; }
        }
        if (!gather(u)) set_fatal(SZL_E_DEVICE, "token gather from device " + std::to_string(devices[un[u].slot]) + " failed");
    }
    join_workers();
    if (!fatal && gst && hipStreamSynchronize(gst) != hipSuccess) set_fatal(SZL_E_DEVICE, "token gather failed");
    if (gst) (void)hipStreamDestroy(gst);
    if (szl::knob("SZL_DEBUG", 0) && !fatal) {
        fprintf(stderr, "[szl] one stream on %d engines, %d units: %d re-run;", n_dev, n_units, reruns);
        for (int u = 0; u < n_units; u++) fprintf(stderr, " [%lld,%lld)@%d", (long long)un[u].entry, (long long)un[u].exit, un[u].slot);
        fprintf(stderr, "\n");
    }
    free_units();
    if (fatal) { (void)hipSetDevice(devices[0]); nt.release(); (void)hipSetDevice(g_device); set_error("%s", fatal_msg.c_str()); return fatal; }
    // ---- finish on engine 0
    if (hipSetDevice(devices[0]) != hipSuccess) return SZL_E_DEVICE;
    Engine &E0 = g_multi_slots[0].eng->e;
    const uint64_t ntok = at;
    E0.tokens.release();
    E0.tokens = nt;                                      // (DevBuf is a plain pointer + capacity; E0 owns it from here)
    SegDev fin = whole;
    fin.buf_off = 0;                                     // engine 0's copy of the stream starts at its buffer's first byte
    if (!d_out_res && (rc = E0.stage_out.ensure(stream->out_cap + 64))) return rc;
    fin.out_off = 0;
    uint8_t *fin_out = d_out_res ? (uint8_t *)d_out_res + stream->out_off : (uint8_t *)E0.stage_out.p;
    std::vector<SegOut> res;
    const unsigned want = ((flags & (SZL_F_CRC32 | SZL_F_GZIP)) ? 1u : 0u) | (((flags & SZL_F_ADLER32) || !(flags & (SZL_F_NOWRAP | SZL_F_GZIP))) ? 2u : 0u);
    rc = E0.finish_tokens(d_res ? resident(0) : (const uint8_t *)E0.stage_in.p, (uint64_t)N, fin_out, fin, ntok, want, res, nullptr);
    if (rc) { (void)hipSetDevice(g_device); return rc; }
    stream->out_len = res[0].out_bytes; stream->crc32 = res[0].crc32; stream->adler32 = res[0].adler32;
    stream->status = res[0].out_bytes <= stream->out_cap ? 0 : SZL_E_OUTPUT_TOO_SMALL;
    if (!d_out_res && stream->status == 0 && stream->out_len &&
        hipMemcpy((uint8_t *)h_out + stream->out_off, E0.stage_out.p, stream->out_len, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipSetDevice(g_device); set_error("D2H failed"); return SZL_E_DEVICE; }
    (void)hipSetDevice(g_device);
    return 0;
}

int szl_trim(void) {            // everything the library holds for objects that do not exist any more
    (void)szl_multi_release();
    (void)szl::pin_pool_trim(0);
    return 0;
}
int szl_multi_release(void) {   // the engines (and their device memory) the multi-device entry points and the streaming objects' pool keep between calls
    engine_pool_release();
    std::lock_guard<std::mutex> multi_lock(g_multi_mu);
    for (auto &slot : g_multi_slots) if (slot.eng) { if (slot.device >= 0) (void)hipSetDevice(slot.device); szl_engine_destroy(slot.eng); slot.eng = nullptr; slot.device = -1; }
    (void)hipSetDevice(g_device);
    return 0;
}

int szl_deflate_stream_multi_device(const int *devices, int n_dev, const void *const *d_in, void *d_out0, szl_stream *stream, int level, int strategy, unsigned flags) {
    if (!devices || n_dev < 1 || !d_in || !d_out0 || !stream) return SZL_E_ARG;
    const int lv = level == -1 ? 6 : level;
    if (lv < 5 || lv > 9 || strategy < 0 || strategy > 2) { set_error("one stream over several devices: levels 5-9 (DeflateSlow) only"); return SZL_E_UNSUPPORTED; }
    int ndev_avail = 0;
    if (hipGetDeviceCount(&ndev_avail) != hipSuccess || ndev_avail <= 0) { set_error("no gfx950 device available"); return SZL_E_DEVICE; }
    for (int g = 0; g < n_dev; g++) if (devices[g] < 0 || devices[g] >= ndev_avail || !d_in[g]) { set_error("device ordinal %d out of range / no input for it", devices[g]); return SZL_E_ARG; }
    if (stream->in_len < (uint64_t)B_TILE * 4 * (uint64_t)n_dev) { set_error("the stream is too short to be cut into units for %d devices", n_dev); return SZL_E_ARG; }
    return stream_multi_run(devices, n_dev, nullptr, nullptr, stream, lv, strategy, flags, d_in, d_out0);
}

int szl_deflate_batch_multi_host(const int *devices, int n_dev, const void *h_in, void *h_out, szl_stream *streams, size_t n_streams,
                                 int level, int strategy, unsigned flags) {
    // one long stream at a DeflateSlow level: exact position-range partition over the devices (stream_multi_run); otherwise by stream
    const int lv = level == -1 ? 6 : level;
    if (n_streams == 1 && streams && n_dev > 1 && devices && lv >= 5 && lv <= 9 && strategy >= 0 && strategy <= 2 &&
        streams[0].in_len >= (uint64_t)std::max(1, szl::knob("SZL_PART_MIN_KIB", 64 * 1024)) * 1024 * (uint64_t)n_dev) {
        int ndev_avail = 0;
        if (hipGetDeviceCount(&ndev_avail) != hipSuccess || ndev_avail <= 0) { set_error("no gfx950 device available"); return SZL_E_DEVICE; }
        for (int g = 0; g < n_dev; g++) if (devices[g] < 0 || devices[g] >= ndev_avail) { set_error("device ordinal %d out of range", devices[g]); return SZL_E_ARG; }
        return stream_multi_run(devices, n_dev, h_in, h_out, &streams[0], lv, strategy, flags);
    }
    return multi_run(false, devices, n_dev, h_in, h_out, streams, n_streams, level, strategy, flags);
}
int szl_inflate_batch_multi_host(const int *devices, int n_dev, const void *h_in, void *h_out, szl_stream *streams, size_t n_streams,
                                 unsigned flags) {
    return multi_run(true, devices, n_dev, h_in, h_out, streams, n_streams, 0, 0, flags);
}

int szl_engine_debug_fetch(szl_engine *e, uint16_t *link, uint32_t *m2, uint32_t *mq, size_t n_positions, uint32_t *tokens,
                           size_t tok_cap, size_t *n_tokens) {
    if (!e) return SZL_E_ARG;
    Engine &E = e->e;
    size_t n = std::min<size_t>(n_positions, E.last_in_total);
    if (link && n && hipMemcpy(link, E.link.p, n * 2, hipMemcpyDeviceToHost) != hipSuccess) return SZL_E_DEVICE;
    if ((m2 || mq) && n) {
        if (!E.mtab.p || E.last_mt_stride < n) { // DeflateFast calls build no match tables
            if (m2) memset(m2, 0, n * 4);
            if (mq) memset(mq, 0, n * 4);
            m2 = mq = nullptr;
        }
        // the device keeps packed entries (szl_internal.h): the tap hands out what they mean — M2 and Mq as len | dist<<16
        const uint32_t *dm = (const uint32_t *)E.mtab.p;
        if (m2 || mq) {
            std::vector<uint32_t> pk(n), side(mq ? n : 0);
            if (hipMemcpy(pk.data(), dm, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return SZL_E_DEVICE;
            if (mq && hipMemcpy(side.data(), dm + E.last_mt_stride, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return SZL_E_DEVICE;
            for (size_t i = 0; i < n; i++) {
                const uint32_t e = pk[i];
                if (m2) m2[i] = e == M_UNSET ? e : mt_m2(e);
                if (mq) mq[i] = e == M_UNSET ? e : (mt_code(e) == 0u ? mt_m2(e) : (mt_code(e) == 1u ? 0u : side[i]));
            }
        }
    }
    size_t nt = std::min<size_t>(tok_cap, (size_t)E.timing.tokens);
    if (tokens && nt && hipMemcpy(tokens, E.tokens.p, nt * 4, hipMemcpyDeviceToHost) != hipSuccess) return SZL_E_DEVICE;
    if (n_tokens) *n_tokens = (size_t)E.timing.tokens;
    return 0;
}

// Parity tap: force the stage-B form of this engine (0 full search, 1 on demand, 2 pilot, -1 default); returns the form the
// last call used (1 = on demand) so that tests can tell what they exercised.
int szl_engine_debug_match_mode(szl_engine *e, int mode) {
    if (!e) return SZL_E_ARG;
    if (mode >= -1 && mode <= 2) e->e.match_mode_override = mode;
    return e->e.last_lazy ? 1 : 0;
}

// Experiment / parity knob (see knob() in szl_engine.hip): overrides the environment variable of the same name for this process.
int szl_debug_set(const char *name, int value) { return name ? szl::knob_set(name, value) : SZL_E_ARG; }

// Parity tap: device bytes held by the per-position side arrays (links, match tables, tokens, ...) at the peak of the last call.
uint64_t szl_engine_debug_workspace(const szl_engine *e) { return e ? e->e.last_workspace_bytes : 0; }
uint32_t szl_engine_debug_par_jobs(const szl_engine *e) { return e ? e->e.last_par_jobs : 0; }
int szl_engine_debug_text_form(const szl_engine *e) { return e ? e->e.last_text_form : SZL_E_ARG; }

// Parity tap: block table of the last call. rows of 8 x uint64: type,last,ntok,bit_start,opt_len,static_len,in_len,hdr_bits
int szl_engine_debug_blocks(szl_engine *e, uint64_t *rows, size_t cap_rows, size_t *n_rows) {
    if (!e) return SZL_E_ARG;
    Engine &E = e->e;
    size_t nslots = (size_t)(E.last_blk_slots);
    std::vector<BlockDesc> tmp(nslots);
    if (nslots && hipMemcpy(tmp.data(), E.descs.p, nslots * sizeof(BlockDesc), hipMemcpyDeviceToHost) != hipSuccess) return SZL_E_DEVICE;
    size_t k = 0;
    for (size_t i = 0; i < nslots; i++) {
        if (tmp[i].type == 0xFFu) continue;
        if (rows && k < cap_rows) {
            uint64_t *r = rows + 8 * k;
            r[0] = tmp[i].type; r[1] = tmp[i].last; r[2] = tmp[i].ntok; r[3] = tmp[i].bit_start; r[4] = tmp[i].opt_len;
            r[5] = tmp[i].static_len; r[6] = tmp[i].in_len; r[7] = tmp[i].hdr_bits;
        }
        k++;
    }
    if (n_rows) *n_rows = k;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Deflater (streaming object).  State constants of C/Deflater.cs:126-140.
enum { IS_SETDICT = 0x01, IS_FLUSHING = 0x04, IS_FINISHING = 0x08, INIT_STATE = 0x00, SETDICT_STATE = 0x01, BUSY_STATE = 0x10,
       FLUSHING_STATE = 0x14, FINISHING_STATE = 0x1c, FINISHED_STATE = 0x1e, CLOSED_STATE = 0x7f };

struct szl_deflater {
    int level = 6, strategy = 0, nowrap = 0, state = 0;
    int64_t total_in = 0, total_out = 0;
    std::vector<uint8_t> hist;      // tail (<= 65536 B) of the bytes already compressed
    std::vector<uint32_t> hist_flags; // levels 1-4: "inserted into the hash chains" bit per hist byte (DeflateFast skips long matches)
    uint64_t hist_abs = 0;          // absolute stream position of hist[0]
    std::vector<uint64_t> bounds;   // absolute positions of earlier segment ends that still lie inside hist
    // bytes given by SetInput since the last Flush(): pinned, and — at the coded levels — uploaded as they arrive (`up_done` of them lie
    // at d_in[up_H ..), `up_H` = the length of the history the layout assumed), so that Flush() / Finish() find the input on the device
    // instead of starting with a pageable copy of all of it (1 GiB: ~100 ms before the first kernel)
    PinVec pend;
    size_t up_done = 0, up_H = 0;
    hipStream_t up_stream = nullptr;
    // CRC-32 of the input, on the device beside the Adler-32 (szl_deflater_enable_crc32: what GZipOutputStream / ZipOutputStream keep on
    // the CPU over every Write, S/GZip/GzipOutputStream.cs:210, S/Zip/ZipOutputStream.cs:700)
    bool want_crc = false; uint32_t crc = 0;
    bool caller_drains = false;     // szl_deflater_caller_drains: the caller takes all Deflate() offers before it changes a parameter
    std::vector<uint64_t> chunks;   // SetInput sizes since the last Flush() (level 0 block cuts depend on them)
    uint64_t chunk_base = 0;        // offset of chunks[0] inside `pend` (not 0 after a function switch: the lookahead the old function left)
    L0State l0;
    size_t chunks_drained = 0;      // chunks after which Deflate() ran while no Flush/Finish was pending
    uint64_t l0_dict = 0;           // bytes of preset dictionary in front of the stream (window positions, not TotalIn)
    PinVec outq;                    // compressed bytes not yet handed out (pinned: filled by DMA)
    size_t outpos = 0;
    // The bytes of a Finish() come back in pieces (run_segment): outq has its final size at once, out_pieces names the pieces still on
    // their way, in order (end offset in outq, the event behind the piece's copy), out_confirmed the prefix known to have arrived.
    // Deflate() / DeflateView() hand out what is there and wait for the next piece only when nothing is (out_ready): the caller's copy —
    // or his write to the base stream — runs beside the DMA, 13 ms per GiB of text that used to be spent before the first byte was offered.
    struct OutPiece { size_t end; hipEvent_t ev; };
    std::deque<OutPiece> out_pieces;
    size_t out_confirmed = 0;
    std::vector<hipEvent_t> out_events;   // spare events
    uint32_t carry_bits = 0; uint8_t carry_byte = 0;
    // PendingBuffer.Reset() clears bitCount but not `bits` (C/PendingBuffer.cs:43): what an unfinished stream left in the bit buffer
    // is OR'ed into the first byte the next stream writes bit by bit (WriteBits :168-189, AlignToByte :143-155).  Survives
    // deflater_clear(); consumed by the first segment / stored block that produces such a byte.
    uint8_t stale = 0;
    uint32_t adler = 1;             // running Adler32.Value of everything compressed so far
    uint32_t dict_adler = 0;        // Adler-32 of the preset dictionary (SETDICT state)
    // SetLevel / SetStrategy while input is pending (same compression function): the parameters the first pending byte is
    // parsed with, and the changes after it as (absolute input position, level, strategy).  `engine_seen` = TotalIn the last
    // time the reference's engine would have run (a Deflate() call that drained the input, or a flush): its DeflateSlow /
    // DeflateFast loop stops at the first iteration start within MIN_LOOKAHEAD - 1 = 261 bytes of that point
    // (C/DeflaterEngine.cs:681,:759), and the new parameters apply to every iteration from there on (:304-361).
    int base_level = 6, base_strategy = 0;
    struct Sw { uint64_t abs_pos; int level, strategy; };
    std::vector<Sw> switches;
    int64_t engine_seen = 0;
    bool hist_has_gaps = false;     // the history holds positions a DeflateFast level did not insert (beyond the segment-end rule)
    szl_engine *eng = nullptr;
    DevBuf d_in, d_out;
    PinVec h_out;
    // ---- stages A-C while the caller still writes (round 6).  Levels 5-9 are chunk-independent: the tokens of
    // [0, x) do not depend on what follows x + lookahead.  Once a stream's first segment has `PIPE_PART` bytes uploaded, a worker thread
    // runs the engine's window pipeline over them part by part — the same part runs one stream over several devices is cut into
    // (Engine::PartRun: entered exactly where the last one left, no stage D) — and collects the tokens; Flush() / Finish() then parse
    // what is left and run stage D over all tokens (Engine::finish_tokens).  Whatever goes wrong, or any call that changes what the
    // engine would do (SetLevel, SetStrategy, Reset), drops the parts: the segment is then compressed in one piece as before.
    struct Pipe {
        std::thread th;
        std::mutex mu, buf_mu;          // mu: avail / stop;  buf_mu: held by the worker during a part, by the caller while d_in moves
        std::condition_variable cv;
        bool started = false, stop = false, failed = false;
        uint64_t avail = 0;             // pending bytes whose upload has been queued on up_stream
        struct Mark { uint64_t upto; hipEvent_t ev; };   // an upload that ends at pending byte `upto` and the event behind it: a part waits for the first
        std::deque<Mark> marks;                           // mark that covers what it reads, not for everything the caller has queued since (mu)
        std::vector<hipEvent_t> spare_events;             // (mu)
        int64_t exit = 0;               // the next part enters the parse here (buffer position)
        uint64_t ntok = 0; uint32_t parts = 0;
        LevelParams P{};
        int device = 0;
        hipStream_t st = nullptr;       // the parts' kernels: a stream of their own (a wait on the null stream stalls the caller's uploads: 20 -> 200 ms of Write() per GiB)
        uint32_t last_parts = 0;        // parts the last segment was parsed in (0: in one piece) — szl_deflater_debug_pipe_parts
    } pipe;
};

static void pipe_stop(szl_deflater *d, bool keep);   // (the pipelined first segment, below)
static void pipe_destroy(szl_deflater *d);
static void deflater_clear(szl_deflater *d) {
    d->state = d->nowrap ? BUSY_STATE : INIT_STATE;
    d->total_in = d->total_out = 0;
    d->hist.clear(); d->hist_flags.clear(); d->hist_abs = 0; d->bounds.clear(); d->pend.clear(); d->up_done = 0; d->outq.clear(); d->outpos = 0;
    d->chunks.clear(); d->chunks_drained = 0; d->chunk_base = 0; d->l0 = L0State{}; d->dict_adler = 0; d->l0_dict = 0;
    d->carry_bits = 0; d->carry_byte = 0; d->adler = 1; d->crc = 0; d->up_done = 0; d->up_H = 0;
    d->switches.clear(); d->engine_seen = 0; d->base_level = d->level; d->base_strategy = d->strategy; d->hist_has_gaps = false;
}

szl_deflater *szl_deflater_create(int level, int nowrap) {
    if (level == -1) level = 6;
    else if (level < 0 || level > 9) { set_error("level out of range"); return nullptr; } // C/Deflater.cs:184-187
    szl_deflater *d = new (std::nothrow) szl_deflater();
    if (!d) return nullptr;
    d->eng = engine_take();
    if (!d->eng) { delete d; return nullptr; }
    std::swap(d->d_in, d->eng->io_a); std::swap(d->d_out, d->eng->io_b);   // (the last owner's device buffers come with a pooled engine)
    d->up_stream = d->eng->st_a; d->eng->st_a = nullptr; d->pipe.st = d->eng->st_b; d->eng->st_b = nullptr;   // (... and its streams)
    d->level = level; d->nowrap = nowrap ? 1 : 0;
    deflater_clear(d);
    object_born();
    return d;
}
static void out_drain(szl_deflater *d);
void szl_deflater_destroy(szl_deflater *d) {
    if (!d) return;
    pipe_destroy(d);
    out_drain(d);
    for (hipEvent_t ev : d->out_events) (void)hipEventDestroy(ev);
    d->out_events.clear(); d->outq.busy = nullptr;
    if (d->up_stream) {   // (the buffers' destructors must not wait on a stream that is gone; the stream itself stays with the engine)
        (void)hipStreamSynchronize(d->up_stream); d->pend.busy = nullptr; d->outq.busy = nullptr;
        if (!d->eng->st_a) d->eng->st_a = d->up_stream; else (void)hipStreamDestroy(d->up_stream);
        d->up_stream = nullptr;
    }
    if (d->pipe.st) { if (!d->eng->st_b) d->eng->st_b = d->pipe.st; else (void)hipStreamDestroy(d->pipe.st); d->pipe.st = nullptr; }   // (synchronised by pipe_destroy's caller below)
    // (nothing of this object is in flight when its engine changes hands: every engine call ends with its stream synchronised, and the
    // uploads' stream was synchronised above)
    std::swap(d->d_in, d->eng->io_a); std::swap(d->d_out, d->eng->io_b);
    engine_give(d->eng);
    delete d;
    object_gone();
}
static int function_switch(szl_deflater *d, int level);
static int lvl_kind(int lv) { return lv == 0 ? 0 : (lv < 5 ? 1 : 2); }   // DEFLATE_STORED / DEFLATE_FAST / DEFLATE_SLOW (C/DeflaterConstants.cs:146)
static int cut_coded(szl_deflater *d, uint64_t seen, uint64_t *x_rel_out, uint64_t *end_bit_out = nullptr);
// What PendingBuffer.bits holds when Reset() arrives (C/Deflater.cs:204-210 -> C/PendingBuffer.cs:43 leaves it): the partial byte
// behind everything the reference has written by then.  That is the byte carried behind the last Flush() / function switch —
// plus, if the caller went on feeding input and draining Deflate() without a flush, the FULL blocks the engine has flushed since
// (16384 tokens each, C/DeflaterHuffman.cs:863; DeflateStored: its blocks end on a byte and AlignToByte clears `bits`).  Those
// blocks are produced here, once, to learn the bits that follow the last of them.  (A caller that resets a finished stream — every
// ZipOutputStream entry — has nothing pending and pays nothing.)
static int reset_stale_bits(szl_deflater *d, uint8_t *out) {
    uint8_t stale = d->carry_bits ? (uint8_t)(d->carry_byte & ((1u << d->carry_bits) - 1u)) : d->stale;
    *out = stale;
    if (d->state == FINISHED_STATE || d->pend.empty()) return 0;            // (Finish: AlignToByte has cleared `bits`; carry is 0 then)
    const int64_t pend_abs = d->total_in - (int64_t)d->pend.size();
    uint64_t seen = d->engine_seen > pend_abs ? (uint64_t)(d->engine_seen - pend_abs) : 0;
    if (seen > d->pend.size()) seen = d->pend.size();
    if (seen == 0) return 0;
    if (lvl_kind(d->level) == 0) {
        L0State l0 = d->l0;
        std::vector<L0Blk> blks;
        const size_t ndr = std::min(d->chunks_drained, d->chunks.size());
        for (size_t i = 0; i < ndr; i++) { uint64_t avail = d->chunks[i]; while (l0_engine_deflate(l0, avail, false, false, blks)) { } }
        if (!blks.empty()) *out = 0;
        return 0;
    }
    if (seen < 16384) return 0;                                               // no full block without 16384 tokens
    uint64_t x_rel = 0, end_bit = ~0ull;
    int rc = cut_coded(d, seen, &x_rel, &end_bit);                            // (mutates the object: the caller clears it next)
    if (rc) return rc;
    if (end_bit == ~0ull) return 0;                                           // the engine stands where the pending bytes begin
    size_t nrows = 0;
    if ((rc = szl_engine_debug_blocks(d->eng, nullptr, 0, &nrows))) return rc;
    std::vector<uint64_t> rows(8 * (nrows ? nrows : 1));
    if ((rc = szl_engine_debug_blocks(d->eng, rows.data(), nrows, &nrows))) return rc;
    int64_t last_full = -1;
    for (size_t i = 0; i < nrows; i++) if (rows[8 * i + 2] >= 16384) last_full = (int64_t)i;
    if (last_full < 0) return 0;
    const uint64_t fend = (size_t)last_full + 1 < nrows ? rows[8 * (last_full + 1) + 3] : end_bit;
    const uint32_t c = (uint32_t)(fend & 7);
    *out = c && (fend >> 3) < d->h_out.size() ? (uint8_t)(d->h_out[fend >> 3] & ((1u << c) - 1u)) : 0;
    return 0;
}
static int deflater_reset(szl_deflater *d) {
    if (!d) return SZL_E_ARG;
    pipe_stop(d, false);
    out_drain(d);
    // DeflaterEngine.Reset() does not touch inputBuf / inputOff / inputEnd (C/DeflaterEngine.cs:234-253): input the engine has not taken
    // yet — a SetInput that no Deflate() call has followed — is still there, IsNeedingInput stays false, and the first Deflate() of the
    // next stream compresses those bytes as its beginning.  (Round 4 dropped them: silently different bytes, and a second SetInput
    // succeeded where the reference throws "Old input was not completely processed".)
    std::vector<uint8_t> unseen;
    if (d->chunks_drained < d->chunks.size()) {
        const uint64_t k = d->chunks.back();
        if (k <= d->pend.size()) { unseen.assign(d->pend.end() - (ptrdiff_t)k, d->pend.end()); d->pend.resize(d->pend.size() - (size_t)k); d->chunks.pop_back(); d->total_in -= (int64_t)k; }
    }
    uint8_t stale = 0;
    const int rc = reset_stale_bits(d, &stale);
    deflater_clear(d);
    d->stale = rc ? 0 : stale;
    if (!unseen.empty()) {
        d->pend.append(unseen.data(), unseen.size());
        d->chunks.push_back((uint64_t)d->pend.size());
        d->total_in = (int64_t)d->pend.size();
    }
    return rc;
}
// a parameter change while bytes are pending: it takes effect where the reference's engine stands
// Where that is depends on how much output the caller has taken once the pending input could have filled a block (16384 tokens need at
// least as many bytes, C/DeflaterHuffman.cs:863): the reference's engine pauses while `pending` holds bytes (C/DeflaterEngine.cs:126-139).
// This backend compresses at Flush() / Finish(); before that Deflate() hands out nothing, so it cannot SEE whether its caller would have
// taken everything the reference offers.  It places the change where the engine stands for a caller who does — the reference's own
// stream classes (CS/DeflaterOutputStream.cs:242-272) — and only for a caller who has SAID so (szl_deflater_caller_drains: the stream
// classes of sharpziplib_amd/dotnet and streams.py do); anybody else's change with 16 KiB or more pending is refused, never answered
// with bytes that may differ from the reference's (round 6; SZL_STRICT=0 restores the silent assumption of rounds 3-5).
static int strict_refuses(const szl_deflater *d) {
    if (d->caller_drains || knob("SZL_STRICT", 1) == 0 || d->pend.size() < 16384) return 0;
    set_error("SetLevel / SetStrategy with %zu bytes pending: where the reference's engine stands depends on how much of Deflate()'s output the caller took "
              "(C/DeflaterEngine.cs:126-139); declare szl_deflater_caller_drains(d, 1) if every Deflate() loop runs until IsNeedingInput, or Flush() first", d->pend.size());
    return SZL_E_UNSUPPORTED;
}
static int pend_switch(szl_deflater *d, int level, int strategy) {
    const int64_t at = d->engine_seen > (MIN_LOOKAHEAD - 1) ? d->engine_seen - (MIN_LOOKAHEAD - 1) : 0;
    d->switches.push_back(szl_deflater::Sw{(uint64_t)at, level, strategy});
    return 0;
}
static int deflater_set_level(szl_deflater *d, int level) {
    if (!d) return SZL_E_ARG;
    if (level == -1) level = 6;
    else if (level < 0 || level > 9) return SZL_E_ARG;
    if (level == d->level) return 0;                       // C/Deflater.cs:357
    if (strict_refuses(d)) return SZL_E_UNSUPPORTED;
    pipe_stop(d, false);                                   // (parts parsed under the old parameters are of no use behind the switch)
    // Another compression function (DeflateStored / DeflateFast / DeflateSlow, C/DeflaterConstants.cs:146): the reference flushes a block
    // with the OLD function where its engine stands and continues with the new one (C/DeflaterEngine.cs:319-359) — with bytes pending,
    // to or from level 0, any number of times.  The three functions leave different hash chains behind (level 0 inserts nothing,
    // 1-4 skip the inside of long matches, :697): the per-byte "inserted" flags of the history carry that over.
    if (lvl_kind(level) != lvl_kind(d->level) && d->total_in != 0) {
        int rc = function_switch(d, level);
        if (rc) return rc;
        d->level = level; d->base_level = level; d->base_strategy = d->strategy;
        return 0;
    }
    if (!d->pend.empty() && level != 0) { int rc = pend_switch(d, level, d->strategy); if (rc) return rc; }
    else if (d->pend.empty()) d->base_level = level;
    d->level = level;
    return 0;
}
int szl_deflater_get_level(const szl_deflater *d) { return d ? d->level : SZL_E_ARG; }
static int deflater_set_strategy(szl_deflater *d, int s) {
    if (!d || s < 0 || s > 2) return SZL_E_ARG;
    if (s != d->strategy && strict_refuses(d)) return SZL_E_UNSUPPORTED;
    if (s != d->strategy) pipe_stop(d, false);
    if (s != d->strategy && !d->pend.empty() && d->level != 0) { int rc = pend_switch(d, d->level, s); if (rc) return rc; }
    else if (d->pend.empty()) d->base_strategy = s;
    d->strategy = s;
    return 0;
}
int szl_deflater_set_dictionary(szl_deflater *d, const uint8_t *p, int n) { // C/Deflater.cs:559 + C/DeflaterEngine.cs:198-229
    if (!d || n < 0 || (!p && n)) return SZL_E_ARG;
    if (d->state != INIT_STATE) { set_error("SetDictionary is only legal before the first Deflate of a zlib stream"); return SZL_E_STATE; } // :561-564
    uint32_t a = 1;
    int rc = szl_adler32(1, p, (size_t)n, &a);   // adler?.Update(dictionary) :204
    if (rc) return rc;
    d->dict_adler = a;
    d->state = SETDICT_STATE;
    if (n < MIN_MATCH) return 0;                   // :205-208: too short to be inserted, the window is not touched
    int off = 0, len = n;
    if (len > MAX_DIST) { off = len - MAX_DIST; len = MAX_DIST; } // :210-214
    // The dictionary is the history of the stream: window indices 1..len, inserted like any other position except its last
    // two bytes (the insert loop :218-225 stops at length-2), exactly what a segment boundary at `len` expresses.
    d->hist.assign(p + off, p + off + len);
    d->hist_flags.assign(((size_t)len + 31) / 32, 0u);
    for (int q = 0; q + 2 < len; q++) d->hist_flags[(size_t)q >> 5] |= 1u << (q & 31); // every dictionary position but the last two is inserted
    d->hist_abs = 0;
    d->bounds.assign(1, (uint64_t)len);
    d->l0.strstart = d->l0.blockStart = 1 + len;
    d->l0_dict = (uint64_t)len;
    return 0;
}
// ---- the pipelined first segment (szl_deflater::Pipe) ----------------------------------------------------------------------------------
// every piece of an asynchronous download has arrived (before anything else touches outq)
static void out_drain(szl_deflater *d) {
    for (auto &pc : d->out_pieces) { (void)hipEventSynchronize(pc.ev); d->out_events.push_back(pc.ev); }
    d->out_pieces.clear();
}
// bytes of outq that may be handed out; `block`: wait for the next piece if nothing beyond outpos is there yet
static size_t out_ready(szl_deflater *d, bool block) {
    while (!d->out_pieces.empty()) {
        szl_deflater::OutPiece &pc = d->out_pieces.front();
        hipError_t q = hipEventQuery(pc.ev);
        if (q != hipSuccess && block && d->out_confirmed <= d->outpos) q = hipEventSynchronize(pc.ev);
        if (q != hipSuccess) { if (q != hipErrorNotReady) { (void)hipGetLastError(); (void)hipEventSynchronize(pc.ev); } else break; }
        d->out_confirmed = pc.end;
        d->out_events.push_back(pc.ev);
        d->out_pieces.pop_front();
    }
    return d->out_pieces.empty() ? d->outq.size() : d->out_confirmed;
}
static double dbg_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static uint64_t pipe_part_bytes() { const int k = knob("SZL_PIPE_PART_KIB", 65536); return k <= 0 ? 0 : (uint64_t)k * 1024 / B_TILE * B_TILE; }
// the stream's first parts are short — a quarter of a part, then half — so that the device starts on the caller's bytes when 20 MiB of them
// are there, not 80: the device is what a caller who writes at memcpy speed waits for, and it was idle for the first 8 of his 27 ms per GiB
static uint64_t pipe_part_bytes(uint32_t k) {
    const uint64_t part = pipe_part_bytes();
    if (k >= 2 || knob("SZL_PIPE_RAMP", 1) == 0) return part;
    return std::max<uint64_t>((part >> (2 - k)) / B_TILE * B_TILE, std::min<uint64_t>(part, (uint64_t)B_TILE));
}
enum : int64_t { PIPE_LOOK = C_WIN_HALO + 1024 + MAX_MATCH + 64 };    // bytes a part sees beyond its end (stream_multi_run's LOOK)
// one part [exit, exit + part) — or, `to_end`, everything up to the segment's end n — on the object's engine; its tokens are appended
static int pipe_run_part(szl_deflater *d, uint64_t part, uint64_t visible, bool to_end) {
    szl_deflater::Pipe &pp = d->pipe;
    Engine &E = d->eng->e;
    const int64_t first = pp.exit, pend = to_end ? (int64_t)visible : first + (int64_t)part;
    SegDev sg{};
    sg.buf_off = 0; sg.abs0 = d->hist_abs;
    sg.seg_start = pp.parts == 0 ? 0 : first; sg.seg_end = (int64_t)visible;
    sg.bnd_off = 0; sg.bnd_cnt = 1;
    std::vector<uint64_t> bnds{to_end ? visible : visible + 64};      // the only boundary that matters is the stream's end (InsertString needs three bytes, :780)
    sg.finish = 0; sg.flags = 0; sg.out_off = 0; sg.out_cap = 0; sg.start_bit = 0; sg.adler_init = 1; sg.crc_init = 0;
    E.part = Engine::PartRun{};
    E.part.active = true; E.part.first = first; E.part.parse_end = pend; E.part.warm_from = -1;
    E.part.force_entry = pp.parts == 0 ? -1 : first;
    E.part.tok_start = pp.ntok;            // the engine's token buffer gathers the parts' tokens (a copy per part into a buffer of the object's own
                                           // meant a hipMalloc / hipFree on the worker's path whenever that buffer grew: 15 ms in which no part ran)
    std::vector<SegOut> res;
    const auto t_part = std::chrono::steady_clock::now();
    // (what is left at Flush() / Finish() is ONE window, as long as the window pipeline's own windows are: a part costs ~20 % more per byte
    // than a long launch — tiles that do not fill the last round of CUs, three host round trips — which pays while the caller writes, not after)
    uint64_t window = std::max<uint64_t>(part, B_TILE);
    // (... and no longer: the side arrays follow the window, 19 bytes per byte of it, and how much is left at Finish() depends on how far the
    // worker got — a remainder longer than any before it cost a Finish() 400 ms of hipMalloc / hipFree for 5 GB of tables)
    if (to_end) window = std::max<uint64_t>(window, std::min<uint64_t>((visible - (uint64_t)first + B_TILE - 1) / B_TILE * B_TILE, (uint64_t)std::max(1, knob("SZL_WINDOW_KIB", 256 * 1024)) * 1024 / B_TILE * B_TILE));
    int rc = E.deflate_windowed((const uint8_t *)d->d_in.p, visible, nullptr, 0, sg, bnds, pp.P, 0, res, pp.st, window);
    if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] t=%.2f part %u [%lld, %lld)%s: %.2f ms wall; device: links %.2f match %.2f parse %.2f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), pp.parts, (long long)first, (long long)pend, to_end ? " to the end" : "",
                                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_part).count(), E.timing.links_ms, E.timing.match_ms, E.timing.parse_ms);
    const Engine::PartRun pr = E.part;
    E.part = Engine::PartRun{};
    if (rc) return rc;
    if (pr.entry != first && pp.parts != 0) { set_error("pipelined segment: forced entry not honoured"); return SZL_E_STATE; }
    pp.ntok += pr.tok_count; pp.exit = pr.exit; pp.parts++;
    return 0;
}
static void pipe_worker(szl_deflater *d) {
    szl_deflater::Pipe &pp = d->pipe;
    if (hipSetDevice(pp.device) != hipSuccess) { std::lock_guard<std::mutex> lk(pp.mu); pp.failed = true; return; }
    for (;;) {
        uint64_t avail;
        hipEvent_t wait_ev = nullptr;
        uint64_t part = pipe_part_bytes(pp.parts);
        {
            std::unique_lock<std::mutex> lk(pp.mu);
            // a caller who is far ahead (he copies at memcpy speed, the device compresses at half of that) gets longer parts: two, then four
            // times the length — a 64 MiB part is twelve rounds of tiles, the last a fifth full, and three host round trips; 256 MiB is one
            // window of the pipeline as it runs for a resident stream
            if (pp.parts >= 2 && knob("SZL_PIPE_GROW", 1) != 0)
                for (int g = 0; g < 2 && pp.avail >= (uint64_t)pp.exit + 2 * part + part / 2 + (uint64_t)PIPE_LOOK; g++) part *= 2;
            // a part runs when its bytes and the lookahead behind them are there, and a quarter part more (the window pipeline lets no sliver
            // stand: with less behind it the part would be taken as the segment's last)
            const uint64_t need = (uint64_t)pp.exit + part + part / 4 + (uint64_t)PIPE_LOOK;
            pp.cv.wait(lk, [&]() { return pp.stop || pp.avail >= need; });
            if (pp.stop) return;
            avail = pp.avail;
            // the bytes the part sees end with the first upload that covers what it needs (the caller may be many uploads ahead of the copy
            // engine: waiting for all of them idled the device for up to 9 ms per part)
            while (!pp.marks.empty() && pp.marks.front().upto < need) { pp.spare_events.push_back(pp.marks.front().ev); pp.marks.pop_front(); }
            if (!pp.marks.empty()) { avail = pp.marks.front().upto; wait_ev = pp.marks.front().ev; pp.marks.pop_front(); }
        }
        const auto tw0 = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> bl(pp.buf_mu);
        const auto tw1 = std::chrono::steady_clock::now();
        const hipError_t upe = wait_ev ? hipEventSynchronize(wait_ev) : hipStreamSynchronize(d->up_stream);
        if (wait_ev) { std::lock_guard<std::mutex> lk(pp.mu); pp.spare_events.push_back(wait_ev); }
        if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] t=%.2f worker woke with avail %llu; buf_mu %.2f ms, upload sync %.2f ms\n", std::chrono::duration<double, std::milli>(tw0.time_since_epoch()).count(), (unsigned long long)avail,
                                      std::chrono::duration<double, std::milli>(tw1 - tw0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw1).count());
        if (upe != hipSuccess || pipe_run_part(d, part, avail, false) != 0) {
            std::lock_guard<std::mutex> lk(pp.mu);
            pp.failed = true;
            return;
        }
    }
}
// stop the worker; `keep`: its tokens stay for pipe_finish, otherwise everything is dropped
static void pipe_stop(szl_deflater *d, bool keep) {
    szl_deflater::Pipe &pp = d->pipe;
    if (pp.started) {
        { std::lock_guard<std::mutex> lk(pp.mu); pp.stop = true; }
        pp.cv.notify_all();
        if (pp.th.joinable()) pp.th.join();
    }
    if (!keep || pp.failed) { pp.ntok = 0; pp.exit = 0; pp.parts = 0; pp.failed = pp.failed && keep; }
    if (!keep) { pp.started = false; pp.stop = false; pp.failed = false; pp.avail = 0; for (auto &m : pp.marks) pp.spare_events.push_back(m.ev); pp.marks.clear(); }
}
static void pipe_destroy(szl_deflater *d) { pipe_stop(d, false); if (d->pipe.st) (void)hipStreamSynchronize(d->pipe.st); for (hipEvent_t ev : d->pipe.spare_events) (void)hipEventDestroy(ev); d->pipe.spare_events.clear(); }   // (the stream goes back to the engine: szl_deflater_destroy)
static void pipe_feed(szl_deflater *d) {      // after eager_upload: tell the worker, or start it
    szl_deflater::Pipe &pp = d->pipe;
    const uint64_t part = pipe_part_bytes();
    if (!part || d->level < 5 || !d->up_stream || d->up_done == 0) return;
    if (!pp.started) {
        if (!d->hist.empty() || !d->bounds.empty() || d->hist_has_gaps || d->l0_dict || !d->switches.empty() || d->hist_abs != 0) return;   // the stream's first segment only
        if (d->up_done < pipe_part_bytes(0u) + pipe_part_bytes(0u) / 4 + (uint64_t)PIPE_LOOK) return;   // (what the first part needs: see pipe_worker)
        if (level_params(d->level, d->strategy, &pp.P) != 0) return;
        (void)hipGetDevice(&pp.device);
        if (!pp.st && hipStreamCreateWithFlags(&pp.st, hipStreamNonBlocking) != hipSuccess) { pp.st = nullptr; (void)hipGetLastError(); return; }
        pp.stop = false; pp.failed = false; pp.exit = 0; pp.ntok = 0; pp.parts = 0; pp.avail = d->up_done;
        if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] t=%.2f worker starts (up_done %zu)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), (size_t)d->up_done);
        try { pp.th = std::thread(pipe_worker, d); } catch (...) { return; }
        pp.started = true;
        return;
    }
    {
        std::lock_guard<std::mutex> lk(pp.mu);
        if (d->up_done > pp.avail) {                       // a new upload has been queued: its mark
            hipEvent_t ev = nullptr;
            if (!pp.spare_events.empty()) { ev = pp.spare_events.back(); pp.spare_events.pop_back(); }
            else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ev = nullptr; (void)hipGetLastError(); }
            if (ev && hipEventRecord(ev, d->up_stream) == hipSuccess) pp.marks.push_back(szl_deflater::Pipe::Mark{(uint64_t)d->up_done, ev});
            else { if (ev) pp.spare_events.push_back(ev); for (auto &m : pp.marks) pp.spare_events.push_back(m.ev); pp.marks.clear(); }   // (no marks: the worker waits for the stream)
        }
        pp.avail = d->up_done;
    }
    pp.cv.notify_all();
}

// Pending bytes travel to the device while the caller is still writing (coded levels; level 0 lays its blocks out on the host and
// uploads at the flush).  Best effort: whatever fails here is simply uploaded at the flush.
static void eager_upload(szl_deflater *d) {
    const size_t UP_SLAB = (size_t)std::max(1, knob("SZL_UP_SLAB_KIB", 4096)) << 10;   // (tests: small slabs)
    if (d->level == 0 || d->pend.size() < d->up_done + UP_SLAB) return;
    const size_t H = d->hist.size();
    if (d->up_done && d->up_H != H) d->up_done = 0;                       // (the layout is [history | pending bytes])
    if (!d->up_stream && hipStreamCreateWithFlags(&d->up_stream, hipStreamNonBlocking) != hipSuccess) { d->up_stream = nullptr; (void)hipGetLastError(); return; }
    if (knob("SZL_DEBUG", 0) && H + d->pend.size() + 64 > d->d_in.cap) fprintf(stderr, "[szl] t=%.2f d_in grows: cap %zu, need %zu\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), d->d_in.cap, H + d->pend.size() + 64);
    if (H + d->pend.size() + 64 > d->d_in.cap && d->pipe.started) {                       // (the worker's part in flight reads d_in: wait for it, then move)
        bool moved;
        { std::lock_guard<std::mutex> bl(d->pipe.buf_mu); moved = d->d_in.ensure_keep(H + d->pend.size() + 64, d->up_done ? H + d->up_done : 0, d->up_stream) == 0; }
        if (!moved) { d->up_done = 0; pipe_stop(d, false); return; }
    } else if (d->d_in.ensure_keep(H + d->pend.size() + 64, d->up_done ? H + d->up_done : 0, d->up_stream)) { d->up_done = 0; return; }
    d->up_H = H;
    d->pend.busy = d->up_stream;
    if (hipMemcpyAsync((uint8_t *)d->d_in.p + H + d->up_done, d->pend.data() + d->up_done, d->pend.size() - d->up_done, hipMemcpyHostToDevice, d->up_stream) != hipSuccess) { (void)hipGetLastError(); d->up_done = 0; return; }
    d->up_done = d->pend.size();
}
// the rest of the pending bytes (and the history in front of them) at a flush; afterwards d_in = [hist | pend]
static int finish_upload(szl_deflater *d, uint64_t H, uint64_t n) {
    int rc;
    // (the object's own stream: its uploads, and since round 5 its kernels too — streaming Deflaters driven by several host threads
    // overlap on the device instead of queueing on the default stream)
    if (!d->up_stream && hipStreamCreateWithFlags(&d->up_stream, hipStreamNonBlocking) != hipSuccess) { d->up_stream = nullptr; (void)hipGetLastError(); }
    if (d->up_done > n || d->up_H != H) d->up_done = 0;
    if ((rc = d->d_in.ensure_keep(H + n + 64, d->up_done ? H + d->up_done : 0, d->up_stream))) return rc;
    if (H && hipMemcpyAsync(d->d_in.p, d->hist.data(), H, hipMemcpyHostToDevice, d->up_stream) != hipSuccess) { set_error("H2D failed"); return SZL_E_DEVICE; }
    if (n > d->up_done && hipMemcpyAsync((uint8_t *)d->d_in.p + H + d->up_done, d->pend.data() + d->up_done, n - d->up_done, hipMemcpyHostToDevice, d->up_stream) != hipSuccess) { set_error("H2D failed"); return SZL_E_DEVICE; }
    if (hipStreamSynchronize(d->up_stream) != hipSuccess) { set_error("H2D failed"); return SZL_E_DEVICE; }
    d->up_done = (size_t)n; d->up_H = (size_t)H;
    return 0;
}
static int deflater_set_input(szl_deflater *d, const uint8_t *p, int n) {
    if (!d) return SZL_E_ARG;
    if ((d->state & IS_FINISHING) != 0) { set_error("Finish() already called"); return SZL_E_STATE; } // :333-336
    if (n < 0 || (!p && n)) return SZL_E_ARG;
    if (d->chunks_drained != d->chunks.size()) { set_error("Old input was not completely processed"); return SZL_E_STATE; } // C/DeflaterEngine.cs:163-166
    d->pend.append(p, (size_t)n);
    eager_upload(d);
    pipe_feed(d);
    d->chunks.push_back((uint64_t)n);
    d->total_in += n;
    return 0;
}
int szl_deflater_flush(szl_deflater *d) { if (!d) return SZL_E_ARG; d->state |= IS_FLUSHING; return 0; }
int szl_deflater_finish(szl_deflater *d) { if (!d) return SZL_E_ARG; d->state |= (IS_FLUSHING | IS_FINISHING); return 0; }
int szl_deflater_needs_input(const szl_deflater *d) { return d && d->chunks_drained == d->chunks.size(); } // true again after the next Deflate() call
int szl_deflater_is_finished(const szl_deflater *d) { return d && d->state == FINISHED_STATE && d->outpos == d->outq.size(); }
int64_t szl_deflater_total_in(const szl_deflater *d) { return d ? d->total_in : 0; }
int64_t szl_deflater_total_out(const szl_deflater *d) { return d ? d->total_out : 0; }
uint32_t szl_deflater_adler(const szl_deflater *d) {
    if (!d || d->nowrap) return 0; // engine has no Adler32 when noZlibHeaderOrFooter (C/DeflaterEngine.cs:84-85)
    if (d->pend.empty()) return d->adler;
    uint32_t v = d->adler;
    if (szl_adler32(d->adler, d->pend.data(), d->pend.size(), &v) != 0) return d->adler;
    return v;
}

// Compress the pending bytes as one segment on the device and append the produced bytes to outq.
// Level 0: render the stored blocks `blks` (absolute positions; all inside the pending bytes) and queue their bytes.
// fed_now: bytes of the pending data FillWindow took in (what the running Adler-32 covers).
static int stored_emit(szl_deflater *d, const std::vector<L0Blk> &blks, uint64_t fed_now, bool finish) {
    // absolute positions count the preset dictionary (if any) as a prefix of the stream
    const uint64_t n = d->pend.size(), pend_abs = (uint64_t)d->total_in - n + d->l0_dict;
    std::vector<StoredBlk> sb(blks.size());
    uint64_t out_total = 0;
    for (size_t i = 0; i < blks.size(); i++) {
        if (blks[i].abs_off < pend_abs || blks[i].abs_off + blks[i].len > pend_abs + n) { set_error("level-0 block outside the pending data"); return SZL_E_STATE; }
        sb[i] = StoredBlk{blks[i].abs_off - pend_abs, out_total, blks[i].len, blks[i].last};
        out_total += 5 + (uint64_t)blks[i].len;
    }
    int rc;
    if (d->up_stream) (void)hipStreamSynchronize(d->up_stream);          // (uploads of an earlier, coded level into d_in)
    d->up_done = 0;
    if ((rc = d->d_in.ensure(n + 64)) || (rc = d->d_out.ensure(out_total + 64))) return rc;
    if (n && hipMemcpy(d->d_in.p, d->pend.data(), n, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D failed"); return SZL_E_DEVICE; }
    uint32_t adler = d->adler, crc = d->crc;
    rc = d->eng->e.deflate_stored((const uint8_t *)d->d_in.p, (uint8_t *)d->d_out.p, sb, (d->nowrap ? 0u : 2u) | (d->want_crc ? 1u : 0u), 0, fed_now < n ? fed_now : n, d->crc, d->adler, &crc, &adler, nullptr);
    if (rc) return rc;
    if (!d->nowrap) d->adler = adler;
    if (d->want_crc) d->crc = crc;
    // A stored block after a compressed era starts inside a byte: its three header bits follow the carried bits, then the stream is
    // byte aligned (FlushStoredBlock: WriteBits(3) + AlignToByte, C/DeflaterHuffman.cs:766-779, C/PendingBuffer.cs:143-155)
    size_t old = d->outq.size();
    if (d->carry_bits && !sb.empty()) {
        const uint32_t c = d->carry_bits;
        d->outq.push_back((uint8_t)(d->carry_byte | ((sb[0].last ? 1u : 0u) << c)));
        if (c + 3 > 8) d->outq.push_back(0);
        d->carry_bits = 0; d->carry_byte = 0;
        old = d->outq.size() - 1;                         // the block's own first byte (device: the header at bit 0) is replaced by the above
        d->outq.resize(old + out_total + (finish && !d->nowrap ? 4 : 0));
        const uint8_t keep = d->outq[old];
        if (out_total && hipMemcpy(d->outq.data() + old, d->d_out.p, out_total, hipMemcpyDeviceToHost) != hipSuccess) { set_error("D2H failed"); return SZL_E_DEVICE; }
        d->outq[old] = keep;
    } else {
    d->outq.resize(old + out_total + (finish && !d->nowrap ? 4 : 0));
    if (out_total && hipMemcpy(d->outq.data() + old, d->d_out.p, out_total, hipMemcpyDeviceToHost) != hipSuccess) { set_error("D2H failed"); return SZL_E_DEVICE; }
    }
    if (d->stale && !sb.empty()) { d->outq[old] |= d->stale; d->stale = 0; }   // the header byte AlignToByte writes is the whole `bits` (:143-155)
    if (finish && !d->nowrap) { // C/Deflater.cs:510-515
        uint8_t *t = d->outq.data() + old + out_total;
        t[0] = (uint8_t)(d->adler >> 24); t[1] = (uint8_t)(d->adler >> 16); t[2] = (uint8_t)(d->adler >> 8); t[3] = (uint8_t)d->adler;
    }
    return 0;
}
enum { TAIL0 = WSIZE + MAX_DIST /* 65274: an engine that stands at this window index or further slides first (:371) */ };
static int64_t base_of_host(int64_t wp) { int64_t idx = wp + 1; if (idx < TAIL0) return 0; return ((idx - (TAIL0 - 1) + 32767) >> 15) << 15; }   // DESIGN App. A.2

// the bytes of `pend` in front of position X_rel become history (flags: which of them are in the hash chains), the rest stays pending
static void advance_history(szl_deflater *d, uint64_t X_rel, const std::vector<uint32_t> *new_flags /* bit q = pend position q; null = none inserted */, bool all_inserted) {
    const uint64_t H = d->hist.size();
    const uint64_t keep = std::min<uint64_t>(H + X_rel, 65536);
    const uint64_t first = H + X_rel - keep;                       // buffer position (hist + pend) of the first byte kept
    std::vector<uint8_t> nh; nh.reserve(keep);
    std::vector<uint32_t> nf((keep + 31) / 32, 0u);
    bool gaps = false;
    for (uint64_t q = 0; q < keep; q++) {
        const uint64_t bp = first + q;
        bool ins;
        if (bp < H) { nh.push_back(d->hist[bp]); ins = d->hist_flags.empty() ? true : ((size_t)(bp >> 5) < d->hist_flags.size() && ((d->hist_flags[(size_t)(bp >> 5)] >> (bp & 31)) & 1u)); }
        else { const uint64_t pq = bp - H; nh.push_back(d->pend[pq]); ins = all_inserted ? true : (new_flags && (size_t)(pq >> 5) < new_flags->size() && (((*new_flags)[(size_t)(pq >> 5)] >> (pq & 31)) & 1u)); }
        if (ins) nf[q >> 5] |= 1u << (q & 31); else gaps = true;
    }
    d->hist_abs = d->hist_abs + first;
    d->hist.swap(nh); d->hist_flags.swap(nf); d->hist_has_gaps = gaps;
    d->pend.erase_front((size_t)X_rel); d->up_done = 0;
    d->bounds.erase(std::remove_if(d->bounds.begin(), d->bounds.end(), [&](uint64_t b) { return b <= d->hist_abs; }), d->bounds.end());
}
static int run_segment_stored(szl_deflater *d, bool finish) {
    std::vector<L0Blk> blks;
    const int64_t wp0 = d->total_in - (int64_t)d->pend.size() + (int64_t)d->l0_dict;
    l0_replay(d->l0, d->chunks, d->chunks_drained, !finish, finish, blks);
    // Bytes the engine really took in.  Normally all of them; the reference stops early when Finish() precedes the first
    // Deflate() on > 64 KiB of level-0 input (DeflateStored marks the block final while input remains, :630-631) — the Adler-32
    // trailer then only covers what FillWindow copied (:389), and so does ours.
    const int64_t fed_end = (int64_t)(d->l0.strstart + d->l0.lookahead) - 1 + d->l0.base;
    const uint64_t fed_now = fed_end > wp0 ? (uint64_t)(fed_end - wp0) : 0;
    d->chunks.clear(); d->chunks_drained = 0; d->chunk_base = 0;
    int rc = stored_emit(d, blks, fed_now, finish);
    if (rc) return rc;
    if (finish) {   // the stream is over: no history to keep (only Reset() makes the object usable again) — as run_segment does for the coded levels
        d->hist.clear(); d->hist_flags.clear(); d->hist_has_gaps = false; d->bounds.clear(); d->pend.clear(); d->up_done = 0;
        d->hist_abs = (uint64_t)d->total_in + d->l0_dict;
    } else advance_history(d, d->pend.size(), nullptr, false);   // stored bytes are in the window, but in no hash chain
    d->engine_seen = d->total_in;
    return 0;
}

// The parameter changes inside the pending bytes (SetLevel / SetStrategy within one compression function) as buffer positions for
// the engine — and, for DeflateFast, the SetInput boundaries: the reference's engine stops at the first iteration start within
// MIN_LOOKAHEAD - 1 bytes of the input it has (C/DeflaterEngine.cs:681), and the Deflate() call that brings the next chunk starts
// with FillWindow(), which slides at window index >= 65274 where DeflateFast's own test is > 65274 (:371 vs :680).  An iteration
// that starts exactly at index 65274 behind such a boundary therefore runs on the slid window (one candidate at distance 32506
// becomes index 0 = "no entry").  `nbounds` boundaries: chunk_base + chunks[0] + ... (pend-relative ends of the first nbounds chunks).
static int segment_switches(szl_deflater *d, const LevelParams &P, uint64_t H, size_t nbounds, std::vector<int64_t> &sw_pos, std::vector<LevelParams> &sw_P) {
    struct Ev { int64_t pos; int kind; LevelParams P; };
    std::vector<Ev> ev;
    const int64_t pend_abs = d->total_in - (int64_t)d->pend.size();
    int rc;
    for (const auto &w : d->switches) {
        LevelParams Pk;
        if ((rc = level_params(w.level, w.strategy, &Pk))) return rc;
        if (Pk.fast != P.fast) { set_error("internal: compression function changed inside a segment"); return SZL_E_STATE; }
        int64_t rel = (int64_t)w.abs_pos - pend_abs;
        if (rel < 0) rel = 0;
        ev.push_back(Ev{(int64_t)H + rel, 0, Pk});
    }
    if (P.fast) {
        uint64_t b = d->chunk_base;
        for (size_t i = 0; i < nbounds && i < d->chunks.size(); i++) {
            b += d->chunks[i];
            ev.push_back(Ev{(int64_t)H + (int64_t)b - (int64_t)(MIN_LOOKAHEAD - 1), 1, P});
        }
    }
    std::stable_sort(ev.begin(), ev.end(), [](const Ev &a, const Ev &b) { return a.pos != b.pos ? a.pos < b.pos : a.kind < b.kind; });
    LevelParams cur = P;
    sw_pos.clear(); sw_P.clear();
    for (auto &e : ev) {
        if (e.kind == 0) cur = e.P;
        LevelParams q = cur;
        if (e.kind == 1) q.fast |= 2;
        sw_pos.push_back(e.pos); sw_P.push_back(q);
    }
    return 0;
}

static int run_segment(szl_deflater *d, bool finish) {
    if (d->level == 0) return run_segment_stored(d, finish);
    // SetInput boundaries behind which a Deflate() call ran FillWindow() with the engine out of lookahead: all but the last chunk's
    // end — and that one too if the caller drained Deflate() once more before Flush() / Finish()
    const size_t fill_bounds = d->chunks.empty() ? 0 : (d->chunks_drained >= d->chunks.size() ? d->chunks.size() : d->chunks.size() - 1);
    std::vector<int64_t> sw_pos; std::vector<LevelParams> sw_P;
    {
        LevelParams P0;
        int rc0 = level_params(d->switches.empty() ? d->level : d->base_level, d->switches.empty() ? d->strategy : d->base_strategy, &P0);
        if (rc0) return rc0;
        if ((rc0 = segment_switches(d, P0, d->hist.size(), fill_bounds, sw_pos, sw_P))) return rc0;
    }
    d->chunks.clear(); d->chunks_drained = 0; d->chunk_base = 0;
    LevelParams P;
    int rc = level_params(d->switches.empty() ? d->level : d->base_level, d->switches.empty() ? d->strategy : d->base_strategy, &P);
    if (rc) return rc;
    const uint64_t H = d->hist.size(), n = d->pend.size();
    const uint64_t in_total = H + n;
    const uint64_t cap = (szl_deflate_bound(n) + 16 + 3) & ~3ull;
    if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] t=%.2f run_segment: %llu bytes\n", dbg_now_ms(), (unsigned long long)n);
    if ((rc = finish_upload(d, H, n))) return rc;                        // (most of the pending bytes are there already: eager_upload)
    if ((rc = d->d_out.ensure(cap + 64))) return rc;
    if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] t=%.2f input on the device, output buffer there\n", dbg_now_ms());
    std::vector<SegDev> segs(1);
    std::vector<uint64_t> bnds;
    for (uint64_t b : d->bounds) if (b > d->hist_abs) bnds.push_back(b - d->hist_abs);
    bnds.push_back(H + n);
    SegDev &s = segs[0];
    s = SegDev{};
    s.buf_off = 0; s.abs0 = d->hist_abs; s.seg_start = (int64_t)H; s.seg_end = (int64_t)(H + n);
    s.bnd_off = 0; s.bnd_cnt = (uint32_t)bnds.size();
    s.finish = finish ? 1 : 0;
    s.flags = finish ? ((d->nowrap ? 0u : (uint32_t)SEG_ZLIB_TRAILER)) : (uint32_t)SEG_SYNC_PAD;
    s.out_off = 0; s.out_cap = cap; s.start_bit = d->carry_bits; s.adler_init = d->adler; s.crc_init = d->crc;
    std::vector<SegOut> res;
    Engine &E = d->eng->e;
    E.sw_pos_in = sw_pos; E.sw_P_in = sw_P;
    E.fast_hist_in.clear(); E.fast_want_tail = false;
    if (P.fast) { E.fast_hist_in = d->hist_flags; E.fast_hist_in.resize((H + 31) / 32, 0u); E.fast_want_tail = !finish; /* (the inserted bits of the tail are history for a next segment only) */ }
    else if (d->hist_has_gaps && H) { E.fast_hist_in = d->hist_flags; E.fast_hist_in.resize((H + 31) / 32, 0u); }   // stage A must skip what DeflateFast skipped
    const unsigned want_ck = (d->nowrap ? 0u : 2u) | (d->want_crc ? 1u : 0u);
    bool piped = false;
    d->pipe.last_parts = 0;
    if (d->pipe.started) {   // parts of this segment were parsed while the caller wrote (szl_deflater::Pipe): parse the rest, then stage D over all tokens
        szl_deflater::Pipe &pp = d->pipe;
        if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] t=%.2f flush: worker stop\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count());
        pipe_stop(d, true);
        const bool usable = !pp.failed && pp.parts > 0 && H == 0 && sw_pos.empty() && !P.fast && E.fast_hist_in.empty() &&
                            pp.P.good == P.good && pp.P.nice == P.nice && pp.P.max_chain == P.max_chain && pp.P.strategy == P.strategy && (uint64_t)pp.exit < n;
        if (usable && pipe_run_part(d, pipe_part_bytes(), n, true) == 0) {
            const uint64_t ntok = pp.ntok;                 // (all in the engine's token buffer, part behind part)
            E.sw_pos_in.clear(); E.sw_P_in.clear();
            rc = E.finish_tokens((const uint8_t *)d->d_in.p, in_total, (uint8_t *)d->d_out.p, s, ntok, want_ck, res, d->up_stream);
            if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] t=%.2f stage D done\n", dbg_now_ms());
            piped = rc == 0;
            pp.last_parts = piped ? pp.parts : 0;
            if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] pipelined segment: %u parts, %llu tokens, rc %d\n", pp.parts, (unsigned long long)ntok, rc);
        }
        pipe_stop(d, false);                               // (also forgets a failed attempt: the segment is then compressed in one piece, below)
    }
    if (!piped) rc = E.deflate((const uint8_t *)d->d_in.p, in_total, (uint8_t *)d->d_out.p, cap, segs, bnds, P, want_ck, res, d->up_stream);
    E.fast_hist_in.clear(); E.fast_want_tail = false; E.sw_pos_in.clear(); E.sw_P_in.clear();
    if (rc) return rc;
    const uint64_t end_bit = res[0].end_bit;
    const uint64_t bytes = (end_bit + 7) >> 3;
    // the segment's bytes come straight into the output queue (pinned: one DMA, no second host copy)
    const size_t q0 = d->outq.size();
    if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] t=%.2f segment compressed: %llu bytes\n", dbg_now_ms(), (unsigned long long)bytes);
    d->outq.resize(q0 + (size_t)bytes + 1);
    if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] t=%.2f output queue has room\n", dbg_now_ms());
    uint8_t *ho = d->outq.data() + q0;
    // a Finish() of many megabytes: the first piece now, the rest on its way while the caller takes what is there (szl_deflater::out_pieces)
    const uint64_t PIECE = (uint64_t)std::max(1, knob("SZL_OUT_PIECE_KIB", 16384)) << 10;
    uint64_t sync_bytes = bytes;
    if (finish && d->up_stream && bytes >= 2 * PIECE) sync_bytes = PIECE;
    if (sync_bytes && hipMemcpy(ho, d->d_out.p, sync_bytes, hipMemcpyDeviceToHost) != hipSuccess) { set_error("D2H failed"); return SZL_E_DEVICE; }
    if (sync_bytes < bytes) {
        d->out_confirmed = q0 + (size_t)sync_bytes;
        d->outq.busy = d->up_stream;
        for (uint64_t at = sync_bytes; at < bytes; at += PIECE) {
            const uint64_t k = std::min<uint64_t>(PIECE, bytes - at);
            hipEvent_t ev = nullptr;
            if (!d->out_events.empty()) { ev = d->out_events.back(); d->out_events.pop_back(); }
            else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
            if (!ev || hipMemcpyAsync(ho + at, (const uint8_t *)d->d_out.p + at, k, hipMemcpyDeviceToHost, d->up_stream) != hipSuccess || hipEventRecord(ev, d->up_stream) != hipSuccess) {
                // (no event / no queue slot: the rest in one synchronous copy)
                (void)hipGetLastError();
                if (ev) d->out_events.push_back(ev);
                out_drain(d);
                if (hipMemcpy(ho + at, (const uint8_t *)d->d_out.p + at, bytes - at, hipMemcpyDeviceToHost) != hipSuccess) { set_error("D2H failed"); return SZL_E_DEVICE; }
                break;
            }
            d->out_pieces.push_back(szl_deflater::OutPiece{q0 + (size_t)(at + k), ev});
        }
    }
    if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] t=%.2f bytes on the host (%zu pieces on their way)\n", dbg_now_ms(), d->out_pieces.size());
    if (d->carry_bits && bytes) ho[0] |= d->carry_byte;
    if (d->stale && bytes) { ho[0] |= d->stale; d->stale = 0; }
    if (!d->nowrap) d->adler = res[0].adler32;
    if (d->want_crc) d->crc = res[0].crc32;
    uint64_t whole = finish ? bytes : (end_bit >> 3);
    if (finish) { d->carry_bits = 0; d->carry_byte = 0; }
    else {
        d->carry_bits = (uint32_t)(end_bit & 7);
        d->carry_byte = d->carry_bits ? ho[whole] : 0;
    }
    d->outq.resize(q0 + (size_t)whole);
    if (finish) {   // the stream is over (only Reset() makes the object usable again, and it clears all of this): no history to keep —
                    // the per-entry pattern of ZipOutputStream (Reset + SetInput + Finish) paid 0.1-0.2 ms of host time for it
        d->hist.clear(); d->hist_flags.clear(); d->hist_has_gaps = false; d->bounds.clear(); d->pend.clear(); d->up_done = 0;
        d->hist_abs = (uint64_t)d->total_in + d->l0_dict;
        d->switches.clear(); d->base_level = d->level; d->base_strategy = d->strategy; d->engine_seen = d->total_in;
        return 0;
    }
    // slide the history: keep the last 65536 bytes (candidates reach 32506 back, their links another 32767)
    d->bounds.push_back(d->hist_abs + H + n);
    std::vector<uint8_t> nh;
    const uint64_t keep = std::min<uint64_t>(H + n, 65536);
    nh.reserve(keep);
    if (keep > n) nh.insert(nh.end(), d->hist.end() - (keep - n), d->hist.end());
    nh.insert(nh.end(), d->pend.end() - std::min<uint64_t>(keep, n), d->pend.end());
    if (P.fast) { // inserted bits of the bytes that stay as history (the engine returns those of the last 32 Ki positions)
        std::vector<uint32_t> nf((keep + 31) / 32, 0u);
        const int64_t first = (int64_t)(H + n - keep);
        for (uint64_t q = 0; q < keep; q++) {
            const int64_t p = first + (int64_t)q - E.fast_tail_start;
            if (p >= 0 && (size_t)(p >> 5) < E.fast_tail_bits.size() && ((E.fast_tail_bits[(size_t)(p >> 5)] >> (p & 31)) & 1u)) nf[q >> 5] |= 1u << (q & 31);
        }
        d->hist_flags.swap(nf);
        d->hist_has_gaps = true;
    } else { // DeflateSlow inserts every position that has three bytes of lookahead (:780): all but the last two of the segment
        std::vector<uint32_t> nf((keep + 31) / 32, 0u);
        const uint64_t first = H + n - keep;
        bool gaps = false;
        for (uint64_t q = 0; q < keep; q++) {
            const uint64_t bp = first + q;                 // buffer position before the slide
            bool ins;
            if (bp < H) ins = !d->hist_flags.empty() && (size_t)(bp >> 5) < d->hist_flags.size() ? ((d->hist_flags[(size_t)(bp >> 5)] >> (bp & 31)) & 1u) != 0 : true;
            else ins = bp + 3 <= H + n;
            if (ins) nf[q >> 5] |= 1u << (q & 31);
            else if (bp < H) gaps = true;
        }
        if (d->hist_has_gaps) d->hist_has_gaps = gaps;   // fast-era bytes have left the history
        d->hist_flags.swap(nf);
    }
    d->hist_abs = d->hist_abs + H + n - keep;
    d->hist.swap(nh);
    d->pend.clear(); d->up_done = 0;
    d->switches.clear(); d->base_level = d->level; d->base_strategy = d->strategy; d->engine_seen = d->total_in;
    d->bounds.erase(std::remove_if(d->bounds.begin(), d->bounds.end(), [&](uint64_t b) { return b <= d->hist_abs; }), d->bounds.end());
    return 0;
}

// ---- SetLevel to another compression function in mid-stream (C/DeflaterEngine.cs:304-361) ----------------------------------------
// Where does the reference's engine stand when the call arrives?  It has SEEN `engine_seen` bytes of input (everything up to the
// last Deflate() call that drained it; later SetInput bytes still sit in the Deflater's input buffer).  DeflateStored consumes all
// it sees; DeflateFast / DeflateSlow stop at the first iteration start with less than MIN_LOOKAHEAD bytes in front of it
// (`while (lookahead >= MIN_LOOKAHEAD || flush)`, :681,:759).  The pending bytes in front of that point are compressed now — with
// the old function, as a block that is flushed without the sync padding of a Flush() — and the rest stays pending for the new one.
// DeflateFast / DeflateSlow -> another function: run the bytes the engine has seen through the device with SEG_SWITCH_CUT
static int cut_coded(szl_deflater *d, uint64_t seen, uint64_t *x_rel_out, uint64_t *end_bit_out) {
    *x_rel_out = 0;
    if (end_bit_out) *end_bit_out = ~0ull;
    const int64_t pend_abs = d->total_in - (int64_t)d->pend.size();
    const int64_t T_abs = d->engine_seen - (int64_t)(MIN_LOOKAHEAD - 1);   // first position whose iteration did not run
    if (seen == 0 || T_abs <= pend_abs) return 0;                          // the engine stands where the pending bytes begin: no block to flush
    LevelParams P;
    int rc = level_params(d->switches.empty() ? d->level : d->base_level, d->switches.empty() ? d->strategy : d->base_strategy, &P);
    if (rc) return rc;
    const uint64_t H = d->hist.size(), n = seen;
    const uint64_t in_total = H + n;
    const uint64_t cap = (szl_deflate_bound(n) + 16 + 3) & ~3ull;
    if ((rc = finish_upload(d, H, n)) || (rc = d->d_out.ensure(cap + 64))) return rc;
    std::vector<SegDev> segs(1);
    std::vector<uint64_t> bnds;
    for (uint64_t b : d->bounds) if (b > d->hist_abs) bnds.push_back(b - d->hist_abs);
    bnds.push_back(H + n);
    SegDev &s = segs[0];
    s = SegDev{};
    s.buf_off = 0; s.abs0 = d->hist_abs; s.seg_start = (int64_t)H; s.seg_end = (int64_t)(H + n);
    s.bnd_off = 0; s.bnd_cnt = (uint32_t)bnds.size();
    s.finish = 0; s.flags = (uint32_t)SEG_SWITCH_CUT; s.cut_pos = (int64_t)H + (T_abs - pend_abs);
    s.out_off = 0; s.out_cap = cap; s.start_bit = d->carry_bits; s.adler_init = 1; s.crc_init = 0;
    Engine &E = d->eng->e;
    {   // parameter changes of the old function inside these bytes, and (DeflateFast) the SetInput boundaries in front of the last one seen
        const size_t ndr = std::min(d->chunks_drained, d->chunks.size());
        if ((rc = segment_switches(d, P, H, ndr ? ndr - 1 : 0, E.sw_pos_in, E.sw_P_in))) return rc;
    }
    E.fast_hist_in.clear(); E.fast_want_tail = false;
    if (P.fast) { E.fast_hist_in = d->hist_flags; E.fast_hist_in.resize((H + 31) / 32, 0u); E.fast_want_tail = true; }
    else if (d->hist_has_gaps && H) { E.fast_hist_in = d->hist_flags; E.fast_hist_in.resize((H + 31) / 32, 0u); }
    std::vector<SegOut> res;
    rc = E.deflate((const uint8_t *)d->d_in.p, in_total, (uint8_t *)d->d_out.p, cap, segs, bnds, P, 0u, res, nullptr);
    E.fast_hist_in.clear(); E.fast_want_tail = false; E.sw_pos_in.clear(); E.sw_P_in.clear();
    if (rc) return rc;
    const int64_t X = res[0].cut_x;
    if (X < (int64_t)H || X > (int64_t)(H + n)) { set_error("internal: the switch cut landed outside the pending bytes"); return SZL_E_STATE; }
    const uint64_t X_rel = (uint64_t)(X - (int64_t)H);
    // the block(s) in front of the cut: whole bytes go out, the partial byte is carried (no padding: FlushBlock(.., false))
    const uint64_t end_bit = res[0].end_bit;
    const uint64_t bytes = (end_bit + 7) >> 3;
    d->h_out.resize(bytes + 1);
    if (bytes && hipMemcpy(d->h_out.data(), d->d_out.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) { set_error("D2H failed"); return SZL_E_DEVICE; }
    if (d->carry_bits && bytes) d->h_out[0] |= d->carry_byte;
    if (d->stale && bytes) { d->h_out[0] |= d->stale; d->stale = 0; }
    if (end_bit_out) *end_bit_out = end_bit;
    const uint64_t whole = end_bit >> 3;
    d->outq.append(d->h_out.data(), (size_t)whole);
    d->carry_bits = (uint32_t)(end_bit & 7);
    d->carry_byte = d->carry_bits ? d->h_out[whole] : 0;
    if (!d->nowrap && X_rel) { uint32_t a = d->adler; if ((rc = szl_adler32(d->adler, d->pend.data(), (size_t)X_rel, &a))) return rc; d->adler = a; }
    if (d->want_crc && X_rel) { uint32_t c = d->crc; if ((rc = szl_crc32(d->crc, d->pend.data(), (size_t)X_rel, &c))) return rc; d->crc = c; }
    // history: DeflateSlow inserted every position in front of the cut (each had MIN_LOOKAHEAD bytes in front of it); DeflateFast the
    // ones its flags say
    std::vector<uint32_t> ff;
    if (P.fast) {
        ff.assign((X_rel + 31) / 32, 0u);
        for (uint64_t q = 0; q < X_rel; q++) {
            const int64_t p = (int64_t)(H + q) - E.fast_tail_start;
            if (p >= 0 && (size_t)(p >> 5) < E.fast_tail_bits.size() && ((E.fast_tail_bits[(size_t)(p >> 5)] >> (p & 31)) & 1u)) ff[q >> 5] |= 1u << (q & 31);
        }
    }
    advance_history(d, X_rel, P.fast ? &ff : nullptr, !P.fast);
    *x_rel_out = X_rel;
    return 0;
}

static int function_switch(szl_deflater *d, int level) {
    const int old_kind = lvl_kind(d->level), new_kind = lvl_kind(level);
    const int64_t pend_abs = d->total_in - (int64_t)d->pend.size();
    uint64_t seen = d->engine_seen > pend_abs ? (uint64_t)(d->engine_seen - pend_abs) : 0;
    if (seen > d->pend.size()) seen = d->pend.size();
    int rc;
    const size_t ndr = std::min(d->chunks_drained, d->chunks.size());
    std::vector<uint64_t> unseen_chunks(d->chunks.begin() + (ptrdiff_t)ndr, d->chunks.end());
    uint64_t look = 0;                                 // bytes the engine keeps in front of it (its lookahead)
    uint64_t seen_left = 0;                            // seen bytes that stay pending: the unseen chunks begin behind them
    if (old_kind == 0) {
        // DeflateStored has consumed what it was given (everything, normally): FlushStoredBlock(blockStart .. strstart, false)
        // if that is not empty, then UpdateHash() (:327-333)
        std::vector<L0Blk> blks;
        const int64_t wp0 = pend_abs + (int64_t)d->l0_dict;
        L0State l0 = d->l0;                           // (replayed on a copy: the object is untouched if the call fails below)
        for (size_t i = 0; i < ndr; i++) {
            uint64_t avail = d->chunks[i];
            while (l0_engine_deflate(l0, avail, false, false, blks)) { }
        }
        // ins_h = window[strstart] << 5 ^ window[strstart + 1] (:409).  The value itself never matters — FillWindow() calls UpdateHash()
        // again as soon as three bytes of lookahead are there (:396-399) and InsertString() needs as many — but the READ does: with
        // DeflateStored at one of the last two indices of a full window it is past the array, and the reference throws.
        if ((int64_t)l0.strstart + 1 >= 2 * WSIZE) { set_error("SetLevel: the reference's UpdateHash() reads past its window array here (IndexOutOfRangeException, C/DeflaterEngine.cs:409)"); return SZL_E_INDEX; }
        d->l0 = l0;
        if (d->l0.strstart > d->l0.blockStart) {
            blks.push_back(L0Blk{(uint64_t)(d->l0.blockStart - 1 + d->l0.base), (uint32_t)(d->l0.strstart - d->l0.blockStart), 0u});
            d->l0.blockStart = d->l0.strstart;
        }
        const int64_t X_w = (int64_t)d->l0.strstart - 1 + d->l0.base;
        const uint64_t X_rel = X_w > wp0 ? (uint64_t)(X_w - wp0) : 0;
        if (X_rel > d->pend.size()) { set_error("internal: level-0 replay ran past the pending bytes"); return SZL_E_STATE; }
        if (!blks.empty()) { if ((rc = stored_emit(d, blks, X_rel, false))) return rc; }
        look = (uint64_t)d->l0.lookahead;
        advance_history(d, X_rel, nullptr, false);      // stored bytes are in the window but in no hash chain
        seen_left = seen > X_rel ? seen - X_rel : 0;
    } else {
        uint64_t X_rel = 0;
        if ((rc = cut_coded(d, seen, &X_rel))) return rc;
        look = seen - X_rel;
        seen_left = look;
    }
    d->switches.clear();
    d->chunks = unseen_chunks; d->chunks_drained = 0; d->chunk_base = seen_left;
    if (new_kind == 0) {
        // DeflateStored continues in the engine's window: strstart at the cut, `look` bytes of lookahead already there
        const int64_t X_w = (d->total_in - (int64_t)d->pend.size()) + (int64_t)d->l0_dict;
        d->l0 = L0State{};
        d->l0.base = base_of_host(X_w);
        d->l0.strstart = d->l0.blockStart = (int)(X_w + 1 - d->l0.base);
        d->l0.lookahead = (int)look;
        d->l0.fed = (uint64_t)d->engine_seen;
    }
    return 0;
}

static int deflater_deflate(szl_deflater *d, uint8_t *out, int length) { // C/Deflater.cs:427
    if (!d || length < 0 || (!out && length)) return SZL_E_ARG;
    if (d->state == CLOSED_STATE) return SZL_E_STATE;
    const int orig = length;
    if (d->state < BUSY_STATE) { // zlib header :436-464
        int hdr = zlib_header(d->level, (d->state & IS_SETDICT) != 0);
        d->outq.push_back((uint8_t)(hdr >> 8)); d->outq.push_back((uint8_t)hdr);
        if (d->state & IS_SETDICT) { // the dictionary's Adler-32, then the running Adler restarts :458-463
            const uint32_t a = d->dict_adler;
            d->outq.push_back((uint8_t)(a >> 24)); d->outq.push_back((uint8_t)(a >> 16)); d->outq.push_back((uint8_t)(a >> 8)); d->outq.push_back((uint8_t)a);
        }
        d->state = BUSY_STATE | (d->state & (IS_FLUSHING | IS_FINISHING));
    }
    for (;;) {
        size_t avail = out_ready(d, length > 0) - d->outpos;
        size_t k = std::min<size_t>(avail, (size_t)length);
        if (k) { memcpy(out, d->outq.data() + d->outpos, k); d->outpos += k; out += k; length -= (int)k; d->total_out += (int64_t)k; }
        if (d->outpos == d->outq.size()) { d->outq.clear(); d->outpos = 0; }
        if (length == 0 || d->state == FINISHED_STATE) break;
        if (d->state == BUSY_STATE) {                    // "We need more input now" :482-484 (the engine has seen every chunk)
            d->chunks_drained = d->chunks.size(); d->engine_seen = d->total_in;
            if (d->level == 0 && d->l0.lookahead > 0) {  // DeflateStored takes the lookahead a coded function left in the window (:621-623)
                uint64_t none = 0; std::vector<L0Blk> nb;
                while (l0_engine_deflate(d->l0, none, false, false, nb)) { }
                if (!nb.empty()) { set_error("internal: DeflateStored emitted a block from the lookahead alone"); return SZL_E_STATE; }   // (never: at most 261 bytes wait there)
            }
            break;
        }
        int rc;
        if (d->state == FLUSHING_STATE) { if ((rc = run_segment(d, false))) return rc; d->state = BUSY_STATE; }
        else if (d->state == FINISHING_STATE) { if ((rc = run_segment(d, true))) return rc; d->state = FINISHED_STATE; }
        else return SZL_E_STATE;
    }
    return orig - length;
}

// Deflate() without the copy (include/szl.h: szl_deflater_deflate_view): the same state machine, but the bytes stay where they are — in
// the object's pinned queue, which the device filled by DMA — and the caller gets their address.  They count as handed out (TotalOut)
// and stay readable until the next call on the object.
static int deflater_deflate_view(szl_deflater *d, const uint8_t **p, int64_t *n) {
    if (!d || !p || !n) return SZL_E_ARG;
    *p = nullptr; *n = 0;
    if (d->state == CLOSED_STATE) return SZL_E_STATE;
    if (d->state < BUSY_STATE) { uint8_t none; const int rc = deflater_deflate(d, &none, 0); if (rc < 0) return rc; }   // (queues the zlib header, :436-464)
    for (;;) {
        if (d->outpos == d->outq.size()) { d->outq.clear(); d->outpos = 0; }      // (the previous view has been read)
        const size_t avail = out_ready(d, true) - d->outpos;
        if (avail) { *p = d->outq.data() + d->outpos; *n = (int64_t)avail; d->outpos += avail; d->total_out += (int64_t)avail; return 0; }
        if (d->state == FINISHED_STATE) return 0;
        if (d->state == BUSY_STATE) { uint8_t none; const int rc = deflater_deflate(d, &none, 1); return rc < 0 ? rc : (rc == 0 ? 0 : SZL_E_STATE); }   // "We need more input now": Deflate()'s own bookkeeping; it has nothing to hand out
        int rc;
        if (d->state == FLUSHING_STATE) { if ((rc = run_segment(d, false))) return rc; d->state = BUSY_STATE; }
        else if (d->state == FINISHING_STATE) { if ((rc = run_segment(d, true))) return rc; d->state = FINISHED_STATE; }
        else return SZL_E_STATE;
    }
}

// The entry points proper: no C++ exception leaves the library (an allocation that fails is SZL_E_NOMEM — round-4 ADVICE)
#define SZL_GUARDED(call) do { try { return (call); } catch (const std::bad_alloc &) { set_error("out of host memory"); return SZL_E_NOMEM; } catch (...) { set_error("internal error"); return SZL_E_STATE; } } while (0)
int szl_deflater_set_input(szl_deflater *d, const uint8_t *p, int n) { SZL_GUARDED(deflater_set_input(d, p, n)); }
int szl_deflater_deflate(szl_deflater *d, uint8_t *out, int length) { SZL_GUARDED(deflater_deflate(d, out, length)); }
int szl_deflater_deflate_view(szl_deflater *d, const uint8_t **p, int64_t *n) { SZL_GUARDED(deflater_deflate_view(d, p, n)); }
int szl_deflater_reset(szl_deflater *d) { SZL_GUARDED(deflater_reset(d)); }
int szl_deflater_set_level(szl_deflater *d, int level) { SZL_GUARDED(deflater_set_level(d, level)); }
int szl_deflater_set_strategy(szl_deflater *d, int s) { SZL_GUARDED(deflater_set_strategy(d, s)); }
int szl_deflater_debug_pipe_parts(const szl_deflater *d) { return d ? (int)d->pipe.last_parts : SZL_E_ARG; }
int szl_deflater_caller_drains(szl_deflater *d, int on) {
    if (!d) return SZL_E_ARG;
    d->caller_drains = on != 0;
    return 0;
}
int szl_deflater_enable_crc32(szl_deflater *d, int on) {
    if (!d) return SZL_E_ARG;
    if (d->total_in != 0 && (on != 0) != d->want_crc) { set_error("the CRC-32 is switched before the first SetInput (or after Reset)"); return SZL_E_STATE; }
    d->want_crc = on != 0;
    return 0;
}
uint32_t szl_deflater_crc32(const szl_deflater *d) {   // of all the input given so far (compressed or still pending), like TotalIn counts it
    if (!d || !d->want_crc) return 0;
    uint32_t v = d->crc;
    if (!d->pend.empty() && szl_crc32(d->crc, d->pend.data(), d->pend.size(), &v) != 0) return d->crc;
    return v;
}

// ---------------------------------------------------------------------------------------------
// Host-memory checksum helpers (device kernels; used by the shim's Adler getter and by tests)
int szl_crc32(uint32_t value, const void *host_data, size_t n, uint32_t *out);
int szl_adler32(uint32_t value, const void *host_data, size_t n, uint32_t *out);

} // extern "C"

namespace szl {
void launch_checksums(const uint8_t *in, const SegDev *segs, uint32_t nseg, const uint64_t *chunk_off, uint64_t nchunks, void *parts,
                      SegOut *so, unsigned want, hipStream_t st);
size_t checksum_partial_bytes();
}

// Workspaces of the host-buffer checksum helpers, kept per thread and device (Deflater.Adler with pending data and the stream
// mirrors call these once per Write: five hipMalloc/hipFree pairs per call were most of their cost).
struct CkCache { int dev = -1; DevBuf din, dseg, doff, dparts, dso; };
static int checksum_host(unsigned want, uint32_t value, const void *data, size_t n, uint32_t *out) {
    if (!out || (!data && n)) return SZL_E_ARG;
    if (szl_device_count() <= 0) { set_error("no gfx950 device available"); return SZL_E_DEVICE; }
    static thread_local CkCache C;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (C.dev != dev) { C.din.release(); C.dseg.release(); C.doff.release(); C.dparts.release(); C.dso.release(); C.dev = dev; }
    int rc = 0;
    SegDev s{}; s.seg_start = 0; s.seg_end = (int64_t)n; s.adler_init = value; s.crc_init = value;
    uint64_t off[2] = {0, (n + 4095) / 4096};
    SegOut so{};
    if ((rc = C.din.ensure(n + 64)) || (rc = C.dseg.ensure(sizeof s)) || (rc = C.doff.ensure(sizeof off)) ||
        (rc = C.dparts.ensure((off[1] + 1) * checksum_partial_bytes())) || (rc = C.dso.ensure(sizeof so))) return rc;
    if ((n && hipMemcpy(C.din.p, data, n, hipMemcpyHostToDevice) != hipSuccess) || hipMemcpy(C.dseg.p, &s, sizeof s, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(C.doff.p, off, sizeof off, hipMemcpyHostToDevice) != hipSuccess) return SZL_E_DEVICE;
    launch_checksums((const uint8_t *)C.din.p, (const SegDev *)C.dseg.p, 1, (const uint64_t *)C.doff.p, off[1], C.dparts.p, (SegOut *)C.dso.p, want, nullptr);
    if (hipMemcpy(&so, C.dso.p, sizeof so, hipMemcpyDeviceToHost) != hipSuccess) return SZL_E_DEVICE;
    *out = (want & 1) ? so.crc32 : so.adler32;
    if (C.din.cap > (64u << 20)) C.din.release();        // (do not sit on a large input copy)
    return 0;
}
namespace szl { void launch_tree_probe(const int *freqs, int n, int numSymbols, int minCodes, int maxLength, unsigned char *len_out, int *ncodes_out, hipStream_t st); }
extern "C" int szl_debug_tree_lengths(const int32_t *freqs, int n, int num_symbols, int min_codes, int max_length, uint8_t *lengths_out, int32_t *num_codes_out) {
    if (!freqs || !lengths_out || !num_codes_out || n < 0 || num_symbols < 1 || num_symbols > LIT_NUM || max_length < 1 || max_length > 15) return SZL_E_ARG;
    if (szl_device_count() <= 0) { set_error("no gfx950 device available"); return SZL_E_DEVICE; }
    if (!n) return 0;
    DevBuf df, dl, dn;
    int rc = 0;
    const size_t cells = (size_t)n * (size_t)num_symbols;
    if ((rc = df.ensure(cells * 4)) || (rc = dl.ensure(cells)) || (rc = dn.ensure((size_t)n * 4))) { df.release(); dl.release(); dn.release(); return rc; }
    if (hipMemcpy(df.p, freqs, cells * 4, hipMemcpyHostToDevice) != hipSuccess) rc = SZL_E_DEVICE;
    if (!rc) {
        launch_tree_probe((const int *)df.p, n, num_symbols, min_codes, max_length, (unsigned char *)dl.p, (int *)dn.p, nullptr);
        if (hipMemcpy(lengths_out, dl.p, cells, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(num_codes_out, dn.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = SZL_E_DEVICE;
    }
    df.release(); dl.release(); dn.release();
    return rc;
}

// ---- pinned I/O buffers of the device-aware stream classes (include/szl.h) ---------------------------------------------------------
namespace {
struct PinnedRange { const uint8_t *p; size_t n; bool ours; };
std::mutex g_pin_mu;
std::vector<PinnedRange> g_pinned;
}
namespace szl {
bool host_is_pinned(const void *p, size_t n) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (const PinnedRange &r : g_pinned) if ((const uint8_t *)p >= r.p && (const uint8_t *)p + n <= r.p + r.n) return true;
    return false;
}
}
extern "C" void *szl_host_alloc(size_t n) {
    if (szl_device_count() <= 0) { set_error("no gfx950 device available"); return nullptr; }
    size_t cap = 0;
    uint8_t *p = pin_alloc(n ? n : 1, &cap);              // (the process-wide pool of pinned blocks: a stream's buffer is the next stream's)
    if (!p) { set_error("pinned host memory (%zu bytes)", n); return nullptr; }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned.push_back(PinnedRange{p, cap, true});
    return p;
}
static bool pinned_forget(void *p, bool ours, size_t *cap = nullptr) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (size_t i = 0; i < g_pinned.size(); i++)
        if (g_pinned[i].p == (const uint8_t *)p && g_pinned[i].ours == ours) { if (cap) *cap = g_pinned[i].n; g_pinned.erase(g_pinned.begin() + (ptrdiff_t)i); return true; }
    return false;
}
extern "C" void szl_host_free(void *p) { size_t cap = 0; if (p && pinned_forget(p, true, &cap)) pin_free((uint8_t *)p, cap); }
extern "C" int szl_host_register(void *p, size_t n) {
    if (!p || !n) return SZL_E_ARG;
    if (szl_device_count() <= 0) { set_error("no gfx950 device available"); return SZL_E_DEVICE; }
    if (hipHostRegister(p, n, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); set_error("hipHostRegister(%zu bytes) failed", n); return SZL_E_NOMEM; }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned.push_back(PinnedRange{(const uint8_t *)p, n, false});
    return 0;
}
extern "C" int szl_host_unregister(void *p) {
    if (!p || !pinned_forget(p, false)) return SZL_E_ARG;
    return hipHostUnregister(p) == hipSuccess ? 0 : SZL_E_DEVICE;
}

extern "C" int szl_crc32(uint32_t value, const void *p, size_t n, uint32_t *out) { return checksum_host(1, value, p, n, out); }
extern "C" int szl_adler32(uint32_t value, const void *p, size_t n, uint32_t *out) { return checksum_host(2, value, p, n, out); }
