// szl_engine.hip — host side of the device pipeline: workspace, work tables, launches, timing.
// Product code: never includes or links anything from oracle/.
#include <hip/hip_runtime.h>
#include <chrono>
#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <mutex>
#include <thread>
#include <cstring>
#include <vector>
#include "szl_engine.h"

namespace szl {

// ---- launch wrappers implemented in the kernel translation units
void launch_links(const uint8_t *in, uint64_t in_total, const SegDev *segs, const uint64_t *bnds, const SpanDev *spans, int nspans,
                  uint16_t *link, const uint32_t *hflags, unsigned long long *guard_flag, uint64_t span_bytes, hipStream_t st,
                  hipStream_t guard_st, hipEvent_t guard_ev);
void links_distrust_ticket_form();
uint32_t links_guard_trips();
enum : int { CNT_WORDS = 64, CNT_LINKS_GUARD = 32 };   // counters: 64 x u64; [32] = links the per-call guard found wrong (szl_kernels_match.hip)
enum : int { SZL_I_RETRY_LINKS = -1000 };             // internal: run the call again (the ticket form of stage A is distrusted from now on)
static int links_guard_tripped() {
    links_distrust_ticket_form();
    fprintf(stderr, "[szl] stage A: the per-call guard found a link k_links3 got wrong (LDS exchange order under load); this process uses the "
                    "bucketed form (k_links2) from now on and the call is run again\n");
    return SZL_I_RETRY_LINKS;
}
hipError_t launch_match(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int ntiles, const uint16_t *link, MTab mtab,
                        LevelParams P, unsigned long long *dbg, hipStream_t st);
#if SZL_LAB   // laboratory forms of the full search (libszl_amd_lab.so only; Makefile)
void launch_links4(const uint8_t *in, const uint16_t *link, int64_t lo, int64_t hi, int64_t n_end, uint16_t *link4, uint8_t *skip4, uint16_t *e3d,
                   uint8_t *e3h, hipStream_t st);
int match3_tile();
#else
static void launch_links4(const uint8_t *, const uint16_t *, int64_t, int64_t, int64_t, uint16_t *, uint8_t *, uint16_t *, uint8_t *, hipStream_t) {}
static int match3_tile() { return B_TILE; }
#endif
// SZL_MATCH_KERNEL=3: the full search of stage B walks four-byte sub-chains (szl_kernels_match3.hip); hop counts are bytes there
#if SZL_LAB
hipError_t launch_match_ring(const uint8_t *in, const SegDev *segs, const TileDev *stripes, int nstripes, const uint16_t *link, MTab mtab, LevelParams P,
                             unsigned long long *dbg, hipStream_t st);
#else
static hipError_t launch_match_ring(const uint8_t *, const SegDev *, const TileDev *, int, const uint16_t *, MTab, LevelParams, unsigned long long *, hipStream_t) { return hipErrorNotSupported; }
#endif
int match2_tile();
// SZL_MATCH_KERNEL=5: the full search in bucket order (szl_kernels_match5.hip): tiles of up to match5_tile() positions, a scratch slot per resident workgroup
#if SZL_LAB
hipError_t launch_match5(const uint8_t *in, uint64_t in_total, const SegDev *segs, const uint64_t *bnds, const TileDev *tiles, int ntiles,
                         MTab mtab, LevelParams P, uint8_t *scratch, int nslots, unsigned long long *dbg, hipStream_t st);
size_t match5_scratch_bytes(int nslots);
int match5_tile();
#else
static hipError_t launch_match5(const uint8_t *, uint64_t, const SegDev *, const uint64_t *, const TileDev *, int, MTab, LevelParams, uint8_t *, int, unsigned long long *, hipStream_t) { return hipErrorNotSupported; }
static size_t match5_scratch_bytes(int) { return 0; }
static int match5_tile() { return B_TILE; }
#endif
static int match5_slots() {   // one workgroup per CU is resident (its LDS): one scratch slot per CU
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
}
// which form of the full search a call uses: SZL_MATCH_KERNEL, with 5 (bucket order) only where it applies — DeflateSlow with a chain budget
// of 4 or more, and no history whose insertions a DeflateFast level decided (`hist_flags`: only the link pass knows those)
static int match_form(const LevelParams &P, bool hist_flags) {
#if !SZL_LAB
    return 2;   // the product library holds one form of the full search: k_match4
#endif
    const int which = SZL_LABKNOB("SZL_MATCH_KERNEL", SZL_MATCH_KERNEL_DEFAULT);
    if (which == 5 && (P.fast || P.strategy == 2 || P.max_chain < 4 || hist_flags)) return 2;
    return which;
}
// The full search of stage B has a work list of its own: k_match4 (SZL_MATCH_KERNEL=2, the default) takes tiles as long as its LDS window
// allows — longer than the B_TILE the other forms of stage B are built for; the ring form (=4, lab) takes stripes of any length.
// `emit` = positions searched, `tile_len` = what the call's size made of B_TILE.  0: use the common tile list.
static int64_t full_search_len(uint64_t emit, int64_t tile_len, int which) {
    if (which == 5) {
        int64_t len = match5_tile();
        if (SZL_LABKNOB("SZL_TILE_LEN", 0) >= 1024) len = std::min<int64_t>(len, SZL_LABKNOB("SZL_TILE_LEN", 0) / 64 * 64);   // (lab)
        while (len > 8192 && emit / (uint64_t)len < 128) len >>= 1;     // a small call gets shorter tiles: more workgroups (a tile's sort has a fixed cost: not below 8 Ki)
        return len;
    }
    if (which == 2) {
        int64_t len = match2_tile();
        if (SZL_LABKNOB("SZL_TILE_LEN", 0) >= 1024) len = std::min<int64_t>(len, SZL_LABKNOB("SZL_TILE_LEN", 0) / 64 * 64);   // (lab)
        const int64_t floor_len = std::max(2048, SZL_LABKNOB("SZL_TILE_FLOOR", 2048));   // (lab / tools/gfxsim: full-length tiles on a small input — the steady state of a long stream)
        while (len > floor_len && emit / (uint64_t)len < 128) len >>= 1;     // a small call gets shorter tiles (see tile_len)
        return len;
    }
    if (which != 4) return 0;
    const uint64_t min_stripes = (uint64_t)std::max(1, SZL_LABKNOB("SZL_STRIPE_MIN", 512));   // (1: tests — long stripes on small inputs)
    if (tile_len < B_TILE && min_stripes > 1) return 0;
    int64_t len = (int64_t)std::max(16, SZL_LABKNOB("SZL_STRIPE_KIB", 256)) << 10;
    while (len > B_TILE && emit / (uint64_t)len < min_stripes) len >>= 1;
    return len;
}
static bool use_match3(const LevelParams &P) { return SZL_LAB && SZL_LABKNOB("SZL_MATCH_KERNEL", SZL_MATCH_KERNEL_DEFAULT) == 3 && P.max_chain >= 4 && P.max_chain <= 128 && P.strategy != 2; }
void launch_block_positions(const uint32_t *tokens, uint64_t ntok, int64_t seg_start, uint64_t *sums, uint32_t *lastlen, int64_t *bsp, int64_t *blp,
                            hipStream_t st);
hipError_t launch_match_lazy(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int nblocks, int tile_first, int tile_step,
                             const uint16_t *link, MTab mtab, LevelParams P, unsigned long long *dbg, hipStream_t st);
void launch_spec(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, uint64_t nranges,
                 LevelParams P, RangeDev *ranges, uint32_t *visited, unsigned long long *counters, uint32_t *spec_tok, hipStream_t st);
bool emit_copy_enabled();
void launch_switch_cut(const uint8_t *in, const SegDev *segs, uint32_t *tokens, SegOut *so, const uint64_t *blk_off, const int64_t *bsp, int64_t *blp, hipStream_t st);
void launch_emit_copy(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, uint64_t nranges,
                      LevelParams P, const RangeDev *ranges, const uint32_t *visited, const uint32_t *spec_tok,
                      const uint64_t *range_tok, const SegOut *so, uint32_t *tokens, const uint64_t *blk_off, int64_t *blk_start_pos,
                      int64_t *blk_lasttok_pos, hipStream_t st);
void launch_fix(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, uint64_t nranges,
                LevelParams P, RangeDev *ranges, const uint32_t *visited, unsigned long long *counters, uint32_t *bad_slot,
                uint64_t *bad_range, hipStream_t st);
void launch_resolve(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, LevelParams P,
                    RangeDev *ranges, const uint32_t *visited, unsigned long long *counters, hipStream_t st);
int exitmap_width();
void launch_exitmaps(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, LevelParams P,
                     RangeDev *ranges, const uint32_t *visited, const uint32_t *bad_slot, const uint64_t *bad_range, uint64_t nbad,
                     uint16_t *exmap, uint16_t *cnmap, unsigned long long *counters, hipStream_t st, void *chain_scratch, uint32_t range_cnt0);
size_t exitchain_scratch_bytes(uint64_t range_cnt);
void launch_range_counts(const RangeDev *ranges, uint64_t nranges, uint32_t *counts, hipStream_t st);
bool counts_to_blocks_fits(uint64_t nranges, uint32_t nseg);
void launch_counts_to_blocks(const RangeDev *ranges, uint64_t nranges, const SegDev *segs, uint32_t nseg, uint32_t *counts, uint64_t *range_tok, SegOut *so,
                             uint32_t *blk_counts, uint64_t *blk_off, hipStream_t st);
void launch_seg_tokens(const SegDev *segs, uint32_t nseg, const uint64_t *range_tok, SegOut *so, uint32_t *blk_counts, hipStream_t st);
void launch_emit(const uint8_t *in, const uint16_t *link, MTab mtab, const SegDev *segs, uint32_t nseg, uint64_t nranges,
                 LevelParams P, const RangeDev *ranges, const uint64_t *range_tok, const SegOut *so, uint32_t *tokens,
                 const uint64_t *blk_off, int64_t *blk_start_pos, int64_t *blk_lasttok_pos, unsigned long long *counters, hipStream_t st);
void launch_seg_blocks(const SegDev *segs, uint32_t nseg, const uint32_t *tokens, const uint64_t *blk_off, SegOut *so, int fast, hipStream_t st);
hipError_t launch_fast(const uint8_t *in, const uint16_t *link, const SegDev *segs, uint32_t nseg, LevelParams P, uint32_t *fbits,
                       SegOut *so, uint32_t *tokens, const uint64_t *blk_off, int64_t *bsp, int64_t *blp, hipStream_t st);
void launch_block_build(const SegDev *segs, uint32_t nseg, const SegOut *so, const uint64_t *blk_off, const uint32_t *tokens,
                        const int64_t *bsp, const int64_t *blp, BlockDesc *descs, uint32_t nslots, int fast, hipStream_t st);
void launch_block_scan(const SegDev *segs, uint32_t nseg, SegOut *so, BlockDesc *descs, hipStream_t st);
void launch_block_encode(const uint8_t *in, uint8_t *out, const SegDev *segs, const BlockDesc *descs, const uint32_t *tokens,
                         uint32_t nslots, hipStream_t st);
void launch_seg_finish(const SegDev *segs, uint32_t nseg, SegOut *so, uint8_t *out, hipStream_t st);
void launch_checksums(const uint8_t *in, const SegDev *segs, uint32_t nseg, const uint64_t *chunk_off, uint64_t nchunks, void *parts,
                      SegOut *so, unsigned want, hipStream_t st);
size_t checksum_partial_bytes();
void launch_stored(const uint8_t *in, uint8_t *out, const StoredBlk *blks, uint32_t n, hipStream_t st);
void launch_zero_regions(const SegDev *segs, uint32_t nseg, const uint64_t *zoff, uint64_t npieces, uint8_t *out, hipStream_t st);
void launch_zero_many(void *const *ptrs, const size_t *bytes, int n, hipStream_t st);
void launch_copy_small(void *dst, const void *src, size_t bytes, hipStream_t st);
void launch_hop_stat(const uint16_t *link, int64_t lo, int64_t n, uint32_t *out_pinned, hipStream_t st);
size_t exscan_tmp_bytes(uint64_t n);
hipError_t launch_exscan(const uint32_t *in, uint64_t *out, uint64_t n, void *tmp, hipStream_t st);
int zero_piece_bytes();

// Tuning knobs for experiments and parity taps: a value set through szl_debug_set() wins, else the environment variable of
// the same name, else the default.  Read at every launch (a handful of string compares).
struct Knob { char name[32]; int value; };
static Knob g_knobs[32];
static int g_nknobs = 0;
int knob_set(const char *name, int value) {   // value INT_MIN: forget the name (environment / default apply again)
    for (int i = 0; i < g_nknobs; i++) if (!strcmp(g_knobs[i].name, name)) {
        if (value == INT32_MIN) { g_knobs[i] = g_knobs[g_nknobs - 1]; g_nknobs--; } else g_knobs[i].value = value;
        return 0;
    }
    if (value == INT32_MIN) return 0;
    if (g_nknobs >= 32 || strlen(name) >= sizeof g_knobs[0].name) return SZL_E_ARG;
    strcpy(g_knobs[g_nknobs].name, name); g_knobs[g_nknobs].value = value; g_nknobs++;
    return 0;
}
int knob(const char *name, int dflt) {
    for (int i = 0; i < g_nknobs; i++) if (!strcmp(g_knobs[i].name, name)) return g_knobs[i].value;
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
const char *last_error() { return g_err; }

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); return SZL_E_DEVICE; } } while (0)

int level_params(int level, int strategy, LevelParams *P) { // C/DeflaterConstants.cs:124-144
    static const int GOOD[10] = {0, 4, 4, 4, 4, 8, 8, 8, 32, 32};
    static const int NICE[10] = {0, 8, 16, 32, 16, 32, 128, 128, 258, 258};
    static const int CHAIN[10] = {0, 4, 8, 32, 16, 32, 128, 256, 1024, 4096};
    if (level == -1) level = 6;
    if (level < 0 || level > 9) return SZL_E_ARG;
    static const int LAZY[10] = {0, 4, 5, 6, 4, 16, 16, 32, 128, 258};
    if (level < 1) return SZL_E_UNSUPPORTED; // DeflateStored has its own entry point (Engine::deflate_stored)
    P->good = GOOD[level]; P->nice = NICE[level]; P->max_chain = CHAIN[level]; P->strategy = strategy;
    P->max_lazy = LAZY[level]; P->fast = level < 5; // COMPR_FUNC :144
    return 0;
}

int DevBuf::ensure(size_t n) {
    if (n <= cap) return 0;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = n + (n >> 3) + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; return SZL_E_NOMEM; }
    cap = want;
    return 0;
}
int DevBuf::ensure_keep(size_t n, size_t keep, hipStream_t st) {
    if (n <= cap) return 0;
    const size_t want = std::max(n + (n >> 3) + 256, 2 * cap);   // (a buffer that follows a stream being written: few moves)
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return SZL_E_NOMEM; }
    if (p && keep) {
        if (hipMemcpyAsync(q, p, std::min(keep, cap), hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipFree(q); set_error("device copy failed"); return SZL_E_DEVICE; }
    } else if (p) (void)hipStreamSynchronize(st);
    if (p) (void)hipFree(p);
    p = q; cap = want;
    return 0;
}
void DevBuf::release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }

namespace {
struct PinBlock { uint8_t *p; size_t cap; };
std::mutex g_pool_mu;
std::vector<PinBlock> g_pool;
size_t g_pool_bytes = 0;
}
// `growing`: the block is for a vector that has outgrown one already and is long (a stream somebody is still writing): the LARGEST pooled
// block that fits, so that it does not move again — a gigabyte written through GZipOutputStream used to move six times, a copy of all it
// held and a wait for the uploads in flight each time.  Otherwise best fit.
uint8_t *pin_alloc(size_t want, size_t *cap_out, bool growing) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = g_pool.size();
        for (size_t i = 0; i < g_pool.size(); i++) {
            if (g_pool[i].cap < want) continue;
            if (growing) { if (best == g_pool.size() || g_pool[i].cap > g_pool[best].cap) best = i; }
            else if (g_pool[i].cap <= 4 * want + (1u << 20) && (best == g_pool.size() || g_pool[i].cap < g_pool[best].cap)) best = i;
        }
        if (best != g_pool.size()) {
            const PinBlock b = g_pool[best];
            g_pool.erase(g_pool.begin() + (ptrdiff_t)best);
            g_pool_bytes -= b.cap;
            *cap_out = b.cap;
            return b.p;
        }
    }
    const size_t ncap = (want + (1u << 16) - 1) & ~(size_t)((1u << 16) - 1);
    uint8_t *q = nullptr;
    if (hipHostMalloc((void **)&q, ncap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *cap_out = ncap;
    return q;
}
void pin_free(uint8_t *p, size_t cap) {
    if (!p) return;
    const size_t limit = (size_t)std::max(0, knob("SZL_PIN_POOL_MIB", 4096)) << 20;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (cap >= (1u << 20) && g_pool_bytes + cap <= limit && g_pool.size() < 64) { g_pool.push_back(PinBlock{p, cap}); g_pool_bytes += cap; return; }
    }
    (void)hipHostFree(p);
}

size_t pin_pool_trim(size_t keep) {     // frees pooled blocks, largest first, until the pool holds at most `keep` bytes; returns what it still holds
    std::vector<PinBlock> drop;
    size_t left;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        std::sort(g_pool.begin(), g_pool.end(), [](const PinBlock &a, const PinBlock &b) { return a.cap > b.cap; });
        size_t k = 0;
        while (k < g_pool.size() && g_pool_bytes > keep) { g_pool_bytes -= g_pool[k].cap; drop.push_back(g_pool[k]); k++; }
        g_pool.erase(g_pool.begin(), g_pool.begin() + (ptrdiff_t)k);
        left = g_pool_bytes;
    }
    for (auto &b : drop) (void)hipHostFree(b.p);
    return left;
}

void PinVec::reserve(size_t want) {
    if (want <= cap) return;
    size_t ncap = 0;
    uint8_t *q = pin_alloc(std::max(want, 2 * cap), &ncap, cap != 0 && want >= (8u << 20));
    if (!q) throw std::bad_alloc();
    if (busy) (void)hipStreamSynchronize(busy);       // copies out of the old memory
    host_copy(q, p, n);
    pin_free(p, cap);
    p = q; cap = ncap;
}
// A long host copy on several cores: one core moves ~13 GB/s, which made the copy of SetInput's bytes into pinned memory the longest
// part of GZipOutputStream over a gigabyte (82 of 190 ms, profiles/r05/write_path.log).  SZL_COPY_THREADS (4; 1 = plain memcpy), pieces of
// 4 MiB or more only — a thread costs tens of microseconds to start.
void host_copy(void *dst, const void *src, size_t k) {
    const size_t MIN_PER_THREAD = 2u << 20;
    int nt = knob("SZL_COPY_THREADS", 4);
    if ((size_t)nt > k / MIN_PER_THREAD) nt = (int)(k / MIN_PER_THREAD);
    if (nt < 2) { if (k) memcpy(dst, src, k); return; }
    const size_t per = ((k / (size_t)nt) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    size_t done = per;                                  // (the calling thread takes the first piece)
    try {
        for (int t = 1; t < nt && done < k; t++, done += per) {
            const size_t off = done, len = std::min(per, k - done);
            th.emplace_back([=] { memcpy((uint8_t *)dst + off, (const uint8_t *)src + off, len); });
        }
    } catch (...) { /* no more threads: this one does the rest */ }
    memcpy(dst, src, std::min(per, k));
    const size_t started = std::min(k, per * (th.size() + 1));
    if (started < k) memcpy((uint8_t *)dst + started, (const uint8_t *)src + started, k - started);
    for (auto &t : th) t.join();
}
void PinVec::append(const uint8_t *src, size_t k) { reserve(n + k); host_copy(p + n, src, k); n += k; }
void PinVec::erase_front(size_t k) {
    if (busy) (void)hipStreamSynchronize(busy);
    if (k >= n) { n = 0; return; }
    memmove(p, p + k, n - k); n -= k;
}
void PinVec::release() { if (busy) (void)hipStreamSynchronize(busy); pin_free(p, cap); p = nullptr; n = cap = 0; }

Engine::Engine() {
    for (auto &e : ev) e = nullptr;
}
std::vector<DevBuf *> Engine::all_bufs() {
    return {&tabs, &link, &mtab, &tokens, &visited, &ranges, &counts, &range_tok, &descs, &d_segs, &d_bnds, &d_spans, &d_tiles,
                      &d_so, &blk_counts, &blk_off, &bsp, &blp, &counters, &ckparts, &ckoff, &cubtmp, &stage_in, &stage_out, &bad_slot, &bad_range, &exmap, &cnmap, &chain_buf, &d_stored, &spec_tok, &d_zoff, &inf_sym, &inf_wins, &inf_jobs, &inf_states, &inf_misc, &inf_groups, &hist_flags_dev, &m5_scratch, &d_sw_pos, &d_sw_P, &d_stripes, &link4, &skip4, &e3dist, &e3hops};
}
size_t Engine::device_bytes() { size_t t = 0; for (DevBuf *b : all_bufs()) t += b->cap; return t; }
void Engine::trim(size_t keep) {
    std::vector<DevBuf *> v = all_bufs();
    std::sort(v.begin(), v.end(), [](DevBuf *a, DevBuf *b) { return a->cap > b->cap; });
    size_t total = device_bytes();
    for (DevBuf *b : v) { if (total <= keep) break; total -= b->cap; b->release(); }
    if (tab_pin && total + tab_pin_cap > keep) { (void)hipHostFree(tab_pin); tab_pin = nullptr; tab_pin_cap = 0; }   // (the tables' pinned staging comes back with the first call)
    if (stat_pin && keep == 0) { (void)hipHostFree(stat_pin); stat_pin = nullptr; }
    if (ring && keep == 0) { if (ring_st) (void)hipStreamSynchronize(ring_st); (void)hipHostFree(ring); ring = nullptr; ring_at = 0; ring_st = nullptr; }
}
Engine::~Engine() {
    for (DevBuf *b : all_bufs())
        b->release();
    for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_guard) (void)hipEventDestroy(ev_guard);
    if (ev_gjoin) (void)hipEventDestroy(ev_gjoin);
    if (ev_zfork) (void)hipEventDestroy(ev_zfork);
    if (ev_zjoin) (void)hipEventDestroy(ev_zjoin);
    if (side) (void)hipStreamDestroy(side);
    if (pin) (void)hipHostFree(pin);
    if (tab_pin) (void)hipHostFree(tab_pin);
    if (ring) (void)hipHostFree(ring);
    if (stat_pin) (void)hipHostFree(stat_pin);
}

// Small tables and structs go to the device — and a few counters come back — through k_copy_small, out of / into a ring of mapped pinned
// memory: no copy engine, whose queue they would share with the long uploads of a caller who is still writing (szl_kernels_block.hip).
// Anything longer, misaligned, or with SZL_SMALL_BY_KERNEL=0: hipMemcpyAsync as before.
int Engine::h2d_small(void *dst, const void *src, size_t n, hipStream_t st) {
    if (!n) return 0;
    const size_t RING = 1u << 20;
    if (n > RING / 4 || (n & 3) || ((uintptr_t)dst & 3) || knob("SZL_SMALL_BY_KERNEL", 1) == 0) {
        if (hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st) != hipSuccess) { set_error("H2D table copy failed"); return SZL_E_DEVICE; }
        return 0;
    }
    if (!ring) { if (hipHostMalloc((void **)&ring, RING, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ring = nullptr; set_error("pinned memory for the tables' ring"); return SZL_E_NOMEM; } ring_at = 0; ring_st = st; }
    const size_t n16 = (n + 15) & ~(size_t)15;
    if (ring_st != st) { if (ring_st) (void)hipStreamSynchronize(ring_st); ring_st = st; ring_at = 0; }     // (another stream's kernels may still read the ring)
    if (ring_at + n16 > RING) { if (hipStreamSynchronize(st) != hipSuccess) { set_error("stream"); return SZL_E_DEVICE; } ring_at = 0; }
    memcpy(ring + ring_at, src, n);
    launch_copy_small(dst, ring + ring_at, n, st);
    ring_at += n16;
    return 0;
}
// Which form of k_match9's instruction text a launch runs (szl_match9_asm.h SZL9_V; szl_kernels_match9.hip).  Form 1 — the first filter
// byte follows the walk's last failed compare — is 9 % faster where chains are dense (logs) and 1 % slower on text, and a kernel that
// holds both forms and lets every tile choose costs text that 1 % too (33.72 -> 34.01 ms per GiB, A/B of two builds in one process,
// profiles/r06/stage_b_forms_per_launch.log).  So the LAUNCH chooses: a sample of the prev[] hops stage A has just written (131072 links
// at evenly spaced places, ~5 us and one synchronisation of the stream: calls of 8 MiB or more only, shorter ones run form 0) — the share of
// hops below 256 is 0.23 on text, 0.92 on logs: below 0.35 form 0, above 0.65 form 1, between them the kernel whose tiles choose.
int Engine::pick_text_form(const uint16_t *lk, int64_t lo, int64_t n, hipStream_t st) {
    last_text_form = 0;
    const int force = knob("SZL_TEXT_FORM", -1);                     // (0 / 1 / 2: no sample)
    if (force >= 0 && force <= 2) return last_text_form = force;
    if (n < (8 << 20)) return 0;
    if (!stat_pin && hipHostMalloc((void **)&stat_pin, 512, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); stat_pin = nullptr; return 0; }
    memset(stat_pin, 0, 512);
    launch_hop_stat(lk, lo, n, stat_pin, st);
    if (hipStreamSynchronize(st) != hipSuccess) return 0;
    uint64_t seen = 0, sh = 0;
    for (int b = 0; b < 64; b++) { seen += ((volatile uint32_t *)stat_pin)[b]; sh += ((volatile uint32_t *)stat_pin)[64 + b]; }
    if (!seen) return 0;
    const double share = (double)sh / (double)seen;
    last_text_form = share < 0.35 ? 0 : (share > 0.65 ? 1 : 2);
    if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] stage B: %.3f of %llu sampled prev[] hops are below 256: form %d of the text\n", share, (unsigned long long)seen, last_text_form);
    return last_text_form;
}
// device -> the engine's pinned page `pin` (read after the stream is synchronised)
int Engine::d2h_small(void *pin_dst, const void *src, size_t n, hipStream_t st) {
    if (!n) return 0;
    if ((n & 3) || ((uintptr_t)src & 3) || ((uintptr_t)pin_dst & 3) || knob("SZL_SMALL_BY_KERNEL", 1) == 0) {
        if (hipMemcpyAsync(pin_dst, src, n, hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("D2H copy failed"); return SZL_E_DEVICE; }
        return 0;
    }
    launch_copy_small(pin_dst, src, n, st);
    return 0;
}
template <typename T> static int upload(Engine &E, DevBuf &b, const std::vector<T> &v, hipStream_t st) {
    int rc = b.ensure(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (rc) return rc;
    return v.empty() ? 0 : E.h2d_small(b.p, v.data(), v.size() * sizeof(T), st);
}

// THE decision "this call goes through the window pipeline" — Engine::deflate takes it, and szl_deflate_batch_host asks the same
// function before it starts the overlapped host->device copy (only deflate_windowed waits for Engine::in_ready: two copies of
// the predicate that drift apart would let the one-piece form read bytes that have not arrived).
// (from SZL_WINDOW_FROM_KIB of input on — 2 GiB by default: below that the 19 B per byte of the one-pass form fit easily and it is
// ~10 % faster, profiles/r02/bench_windowed_vs_monolithic.txt; 0 = as soon as the stream is longer than a window)
bool Engine::uses_window_pipeline(size_t n_segments, bool deflate_slow, uint64_t stream_len, uint64_t *window_out) {
    const uint64_t window = (uint64_t)knob("SZL_WINDOW_KIB", 256 * 1024) * 1024;
    const uint64_t from = (uint64_t)knob("SZL_WINDOW_FROM_KIB", 2048 * 1024) * 1024;
    if (window_out) *window_out = window / B_TILE * B_TILE;
    return n_segments == 1 && deflate_slow && window >= (uint64_t)B_TILE && stream_len > window + window / 4 && stream_len >= from;
}

int Engine::deflate(const uint8_t *d_in, uint64_t in_total, uint8_t *d_out, uint64_t out_total, std::vector<SegDev> &segs,
                    const std::vector<uint64_t> &bnds, LevelParams P, unsigned want_ck, std::vector<SegOut> &results, hipStream_t st) {
    const std::vector<SegDev> segs0 = segs;
    int rc = deflate_impl(d_in, in_total, d_out, out_total, segs, bnds, P, want_ck, results, st);
    if (rc == SZL_I_RETRY_LINKS) { segs = segs0; rc = deflate_impl(d_in, in_total, d_out, out_total, segs, bnds, P, want_ck, results, st); }
    return rc == SZL_I_RETRY_LINKS ? SZL_E_STATE : rc;
}
int Engine::deflate_windowed(const uint8_t *d_in, uint64_t in_total, uint8_t *d_out, uint64_t out_total, SegDev seg,
                             const std::vector<uint64_t> &bnds, LevelParams P, unsigned want_ck, std::vector<SegOut> &results, hipStream_t st,
                             uint64_t window) {
    const PartRun part0 = part;
    int rc = deflate_windowed_impl(d_in, in_total, d_out, out_total, seg, bnds, P, want_ck, results, st, window);
    if (rc == SZL_I_RETRY_LINKS) { part = part0; rc = deflate_windowed_impl(d_in, in_total, d_out, out_total, seg, bnds, P, want_ck, results, st, window); }
    return rc == SZL_I_RETRY_LINKS ? SZL_E_STATE : rc;
}

int Engine::deflate_impl(const uint8_t *d_in, uint64_t in_total, uint8_t *d_out, uint64_t out_total, std::vector<SegDev> &segs,
                         const std::vector<uint64_t> &bnds, LevelParams P, unsigned want_ck, std::vector<SegOut> &results, hipStream_t st) {
    const uint32_t nseg = (uint32_t)segs.size();
    results.assign(nseg, SegOut{});
    if (nseg == 0) return 0;
    for (auto &s : segs) if (s.look_end < s.seg_end) s.look_end = s.seg_end;
    {   // a single long stream at a DeflateSlow level goes through the window pipeline (workspace of one window, not of the stream)
        uint64_t window = 0;
        if (!(segs[0].flags & SEG_SWITCH_CUT) && sw_pos_in.empty() && uses_window_pipeline(nseg, !P.fast, (uint64_t)(segs[0].seg_end - segs[0].seg_start), &window))
            return deflate_windowed(d_in, in_total, d_out, out_total, segs[0], bnds, P, want_ck, results, st, window);
    }
    memset(&timing, 0, sizeof timing);
    for (auto &e : ev) if (!e) HIPCHK(hipEventCreate(&e));
    if (!pin) HIPCHK(hipHostMalloc((void **)&pin, 256, hipHostMallocDefault));

    // ---------------- work tables
    uint64_t total_emit = 0, nranges = 0, vis_words = 0, nchunks = 0, ntiles = 0, blk_slots = 0, seg_bytes = 0;
    for (auto &s : segs) total_emit += (uint64_t)(s.seg_end - std::max<int64_t>(0, s.seg_start - WSIZE));
    uint64_t span_len = (total_emit + 511) / 512;
    span_len = std::min<uint64_t>(std::max<uint64_t>(span_len, 1u << 17), 1u << 22);
    span_len = (span_len + 63) & ~63ull;
    // Stage C walks each range serially (one lane per range): a small call gets shorter ranges, i.e. more lanes and shorter
    // walks — the latency of ONE 64 KiB entry through the streaming object is dominated by that walk otherwise (2.1 of 3.2 ms)
    // (256 Ki ranges fill the device: 256 CUs x 64 lanes x 16 wavefronts.  Round 5: the rule said 64 Ki, i.e. four wavefronts per CU, and
    // a launch is as long as ONE range's walk — 3.3 ms for 4096 positions — however few of them there are: a 256 MiB window of the window
    // pipeline spent 3.3 ms in k_spec_win where 1 GiB in one piece spends 4.3)
    constexpr uint64_t FILL_RANGES = 262144;
    uint32_t range_len = C_RANGE;
    if (SZL_LABKNOB("SZL_RANGE_LEN", 0) >= 32) range_len = (uint32_t)SZL_LABKNOB("SZL_RANGE_LEN", 0) / 32 * 32;   // (lab)
    else {
        // (round 6: one entry of the unchanged ZipOutputStream path — a call of a few hundred KiB at most — spent 205 of its 890 us of device
        // time walking ranges of 256 positions on four wavefronts; a call that small gets ranges of 64)
        // (by data class, tools/lab/small_call_ranges.py, profiles/r06/small_call_ranges.log: text gains from 64 up to 256 KiB — 0.156 against 0.194 /
        // 0.297 ms with 128 / 256 — and from 128 at 1 MiB, 0.207 against 0.267 / 0.304; logs, whose matches are as long as such a range, never
        // merge in ranges of 64 — 4693 unmerged ranges in 1 MiB, 0.407 ms against 0.293 / 0.179 — and from 2 MiB on 128 costs them 2-8 x)
        const uint32_t min_range = total_emit <= (1u << 18) ? 64u : (total_emit <= (1u << 20) ? 128u : 256u);
        while (range_len > min_range && total_emit / range_len < FILL_RANGES) range_len >>= 1;
    }
    // Stage B: a small call gets shorter tiles (a tile is one workgroup; its lanes walk 16 positions each, one after the other)
    bool m3 = !P.fast && use_match3(P);
    if (!sw_pos_in.empty()) {   // SetLevel / SetStrategy inside the (single) segment: any number of changes, carried in device arrays
        if (nseg != 1 || sw_pos_in.size() != sw_P_in.size()) { set_error("parameter changes inside a segment are a single-stream feature"); return SZL_E_UNSUPPORTED; }
        segs[0].sw_cnt = (uint32_t)sw_pos_in.size();
        m3 = false;
    } else for (auto &s : segs) { s.sw_cnt = 0; s.sw_pos = nullptr; s.sw_P = nullptr; }
    int64_t tile_len = m3 ? match3_tile() : B_TILE;
    if (SZL_LABKNOB("SZL_TILE_LEN", 0) >= 1024) tile_len = std::min<int64_t>(tile_len, SZL_LABKNOB("SZL_TILE_LEN", 0) / 64 * 64);   // (lab)
    while (tile_len > 2048 && total_emit / (uint64_t)tile_len < 128) tile_len >>= 1;
    const bool hist_flags = !P.fast && !fast_hist_in.empty() && nseg == 1 && segs[0].seg_start > 0;
    const int form = match_form(P, hist_flags);
    const int64_t stripe_len = (!P.fast && !m3) ? full_search_len(total_emit, tile_len, form) : 0;
    std::vector<SpanDev> spans;
    std::vector<TileDev> tiles, stripes;
    std::vector<uint64_t> chunk_off(nseg + 1), zero_off(nseg + 1);
    uint64_t nzero = 0;
    for (uint32_t i = 0; i < nseg; i++) { zero_off[i] = nzero; nzero += (segs[i].out_cap + (uint64_t)zero_piece_bytes() - 1) / (uint64_t)zero_piece_bytes(); }
    zero_off[nseg] = nzero;
    if (nzero > 0x7FFFFFFFull) { set_error("batch too large"); return SZL_E_ARG; }
    const bool fast = P.fast != 0;             // DeflateFast: one wavefront per segment instead of stages B and C
    bool lazy = false;                         // stage B ran in its on-demand form
    bool has_switch = false;                   // some segment changes LevelParams inside (SegDev.sw_*)
    std::vector<uint64_t> fast_blk_off;        // (its block slots are laid out here, not by a device scan)
    if (fast) fast_blk_off.resize(nseg + 1);
    for (uint32_t i = 0; i < nseg; i++) {
        SegDev &s = segs[i];
        const uint64_t n = (uint64_t)(s.seg_end - s.seg_start);
        if (fast) {
            int64_t e0f = std::max<int64_t>(0, s.seg_start - WSIZE);
            if (n > 0)
                for (int64_t a = e0f; a < s.seg_end; a += (int64_t)span_len)
                    spans.push_back(SpanDev{i, 0, a, std::min<int64_t>(a + (int64_t)span_len, s.seg_end)});
            s.range_off = seg_bytes; // token base
            s.range_cnt = 0;
            s.vis_word_off = vis_words;
            vis_words += ((uint64_t)s.seg_end + 31) / 32 + 1;
            fast_blk_off[i] = blk_slots;
            blk_slots += n / BLOCK_TOKENS + 2;
            seg_bytes += n;
            chunk_off[i] = nchunks;
            nchunks += (n + 4095) / 4096;
            continue;
        }
        seg_bytes += n;
        int64_t e0 = std::max<int64_t>(0, s.seg_start - WSIZE);
        if (n > 0)
            for (int64_t a = e0; a < s.seg_end; a += (int64_t)span_len)
                spans.push_back(SpanDev{i, 0, a, std::min<int64_t>(a + (int64_t)span_len, s.seg_end)});
        if (s.sw_cnt == 0) {
            for (int64_t a = s.seg_start; a < s.seg_end; a += tile_len)
                tiles.push_back(TileDev{i, 0, a, (int32_t)std::min<int64_t>(tile_len, s.seg_end - a), 0});
            // (tiles: equal parts, so that a 64 KiB entry is 4 x 16384 and not 3 x 21504 + 1024 — a sliver still costs a tile's fixed
            // time: 50000 such entries 212.6 -> 174.9 ms in stage B, profiles/r02/config3_tile_balance.log)
            int64_t part = stripe_len;
            if (part && (form == 2 || form == 5) && s.seg_end > s.seg_start) {
                const int64_t nt = (s.seg_end - s.seg_start + part - 1) / part;
                part = std::min<int64_t>(part, ((s.seg_end - s.seg_start + nt - 1) / nt + 63) & ~(int64_t)63);
            }
            for (int64_t a = s.seg_start; part && a < s.seg_end; a += part)
                stripes.push_back(TileDev{i, 0, a, (int32_t)std::min<int64_t>(part, s.seg_end - a), 0});
        } else { // SetLevel / SetStrategy inside the segment: tiles end at the switch positions, each searched with its own parameters
            has_switch = true;
            int64_t lo = s.seg_start;
            for (uint32_t k = 0; k <= s.sw_cnt; k++) {
                int64_t hi = k < s.sw_cnt ? std::min<int64_t>(std::max<int64_t>(sw_pos_in[k], lo), s.seg_end) : s.seg_end;
                for (int64_t a = lo; a < hi; a += B_TILE)
                    tiles.push_back(TileDev{i, 0, a, (int32_t)std::min<int64_t>(B_TILE, hi - a), (int32_t)k});
                lo = hi;
            }
        }
        s.range_off = nranges;
        s.range_len = range_len;
        s.range_cnt = (uint32_t)((n + range_len - 1) / range_len);
        nranges += s.range_cnt;
        s.vis_word_off = vis_words;
        vis_words += (n + 31) / 32 + 1;
        chunk_off[i] = nchunks;
        nchunks += (n + 4095) / 4096;
        blk_slots += n / BLOCK_TOKENS + 1;
    }
    chunk_off[nseg] = nchunks;
    if (fast) fast_blk_off[nseg] = blk_slots;
    ntiles = tiles.size();
    if (has_switch) {
        if (nseg != 1) { set_error("parameter changes inside a segment are a single-stream feature"); return SZL_E_UNSUPPORTED; }
        std::stable_sort(tiles.begin(), tiles.end(), [](const TileDev &a, const TileDev &b) { return a.pad2 < b.pad2; });
    }
    if (blk_slots > 0xFFFFFFF0ull || spans.size() > 0x7FFFFFFFull || ntiles > 0x7FFFFFFFull) { set_error("batch too large"); return SZL_E_ARG; }

    // ---------------- workspace
    int rc;
    if ((rc = link.ensure(in_total * 2 + 64))) return rc;
    const size_t mt_stride = (in_total + 63) & ~(size_t)63; // M2 array, then Mq array
    if (!fast && (rc = mtab.ensure(mt_stride * 8 + 64))) return rc;
    MTab mt = {(uint32_t *)mtab.p, (uint32_t *)mtab.p + mt_stride};
    last_mt_stride = fast ? 0 : mt_stride;
    if ((rc = tokens.ensure((seg_bytes + 16) * 4))) return rc;
    if ((rc = visited.ensure((vis_words + 4) * 4))) return rc;
    if ((rc = ranges.ensure((nranges + 1) * sizeof(RangeDev)))) return rc;
    if ((rc = counts.ensure((nranges + 2) * 4))) return rc;
    if ((rc = bad_slot.ensure((nranges + 2) * 4))) return rc;
    if ((rc = bad_range.ensure((nranges + 2) * 8))) return rc;
    if ((rc = range_tok.ensure((nranges + 2) * 8))) return rc;
    if ((rc = descs.ensure((blk_slots + 1) * sizeof(BlockDesc)))) return rc;
    if ((rc = d_so.ensure(nseg * sizeof(SegOut)))) return rc;
    if ((rc = blk_counts.ensure((nseg + 2) * 4))) return rc;
    if ((rc = blk_off.ensure((nseg + 2) * 8))) return rc;
    if ((rc = bsp.ensure((blk_slots + 1) * 8))) return rc;
    if ((rc = blp.ensure((blk_slots + 1) * 8))) return rc;
    if ((rc = counters.ensure(CNT_WORDS * 8))) return rc;
    if (want_ck && (rc = ckparts.ensure((nchunks + 1) * checksum_partial_bytes()))) return rc;
    if (!sw_pos_in.empty()) {
        if ((rc = upload(*this, d_sw_pos, sw_pos_in, st)) || (rc = upload(*this, d_sw_P, sw_P_in, st))) return rc;
        segs[0].sw_pos = (const int64_t *)d_sw_pos.p; segs[0].sw_P = (const LevelParams *)d_sw_P.p;
    }
    const void *d_segs_view = nullptr;
    const bool own_list = stripe_len > 0 && !has_switch && !stripes.empty();   // the full search's own work list
    // the call's tables travel in ONE copy out of pinned memory (seven copies out of pageable vectors cost a small call ~60 us of
    // device time in gaps alone, and the host a staging copy each)
    const void *p_bnds, *p_spans, *p_tiles, *p_stripes, *p_ckoff, *p_zoff;
    {
        struct Part { const void *src; size_t n; size_t off; };
        size_t total = 0;
        auto add = [&](const void *src, size_t n) { Part q{src, n, total}; total += (std::max<size_t>(n, 1) + 255) & ~(size_t)255; return q; };
        const Part a[7] = {add(segs.data(), segs.size() * sizeof(SegDev)), add(bnds.data(), bnds.size() * 8), add(spans.data(), spans.size() * sizeof(SpanDev)),
                           add(tiles.data(), tiles.size() * sizeof(TileDev)), add(own_list ? stripes.data() : nullptr, own_list ? stripes.size() * sizeof(TileDev) : 0),
                           add(chunk_off.data(), chunk_off.size() * 8), add(zero_off.data(), zero_off.size() * 8)};
        if ((rc = tabs.ensure(total))) return rc;
        if (tab_pin_cap < total) {
            if (tab_pin) { (void)hipHostFree(tab_pin); tab_pin = nullptr; tab_pin_cap = 0; }
            const size_t want = std::max<size_t>(total + total / 2, 1u << 16);
            if (hipHostMalloc((void **)&tab_pin, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); tab_pin = nullptr; set_error("pinned memory for the call's tables"); return SZL_E_NOMEM; }
            tab_pin_cap = want;
        }
        for (const Part &q : a) if (q.n) memcpy(tab_pin + q.off, q.src, q.n);
        if (total <= (1u << 18) && knob("SZL_SMALL_BY_KERNEL", 1) != 0) launch_copy_small(tabs.p, tab_pin, total, st);   // (no copy engine: h2d_small)
        else HIPCHK(hipMemcpyAsync(tabs.p, tab_pin, total, hipMemcpyHostToDevice, st));
        uint8_t *b = (uint8_t *)tabs.p;
        d_segs_view = b + a[0].off; p_bnds = b + a[1].off; p_spans = b + a[2].off; p_tiles = b + a[3].off; p_stripes = b + a[4].off; p_ckoff = b + a[5].off; p_zoff = b + a[6].off;
    }
    if ((rc = cubtmp.ensure(std::max(exscan_tmp_bytes(nranges + 1), exscan_tmp_bytes(nseg + 1)) + 256))) return rc;

    const SegDev *dsegs = (const SegDev *)d_segs_view;
    SegOut *dso = (SegOut *)d_so.p;
    unsigned long long *dcnt = (unsigned long long *)counters.p;

    HIPCHK(hipEventRecord(ev[0], st));
    // the output regions are zeroed (k_block_encode ORs bits into them) beside stages A-C, on the side stream: nothing reads or writes
    // them before stage D3 (0.2 ms per GiB off the critical path)
    if (!side) HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    if (!ev_zfork) HIPCHK(hipEventCreateWithFlags(&ev_zfork, hipEventDisableTiming));
    if (!ev_zjoin) HIPCHK(hipEventCreateWithFlags(&ev_zjoin, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev_zfork, st));
    HIPCHK(hipStreamWaitEvent(side, ev_zfork, 0));
    launch_zero_regions(dsegs, nseg, (const uint64_t *)p_zoff, nzero, d_out, side); // only the streams' own regions (szl.h)
    HIPCHK(hipEventRecord(ev_zjoin, side));
    // every exit between here and the join in front of stage D3 leaves that kernel in flight on `side`, reading d_zoff / dsegs and writing
    // the caller's d_out (round-4 ADVICE): an early return waits for it, so that nothing of this call touches memory the caller may free
    // or the next call re-uploads
    struct SideJoin { hipStream_t s; bool armed; ~SideJoin() { if (armed && s) (void)hipStreamSynchronize(s); } } side_join{side, true};
    {
        void *const zp[5] = {visited.p, counters.p, d_so.p, counts.p, blk_counts.p};
        const size_t zb[5] = {(size_t)(vis_words + 4) * 4, (size_t)CNT_WORDS * 8, (size_t)nseg * sizeof(SegOut), (size_t)(nranges + 2) * 4, (size_t)(nseg + 2) * 4};
        launch_zero_many(zp, zb, 5, st);
        HIPCHK(hipGetLastError());
    }
    // checksums (also seeds so[].adler32 / crc32 with the running values when not requested)
#if SZL_LAB
    static const bool ck_overlap = !(getenv("SZL_CK_OVERLAP") && atoi(getenv("SZL_CK_OVERLAP")) == 0);
#else
    const bool ck_overlap = true;
#endif
    const bool forked = want_ck && ck_overlap;
    if (forked) {
        if (!side) HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        if (!ev_fork) HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        if (!ev_join) HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev_fork, st));
        HIPCHK(hipStreamWaitEvent(side, ev_fork, 0));
        launch_checksums(d_in, dsegs, nseg, (const uint64_t *)p_ckoff, nchunks, ckparts.p, dso, want_ck, side);
        HIPCHK(hipEventRecord(ev_join, side));
    } else {
        launch_checksums(d_in, dsegs, nseg, (const uint64_t *)p_ckoff, want_ck ? nchunks : 0, ckparts.p, dso, want_ck, st);
    }
    HIPCHK(hipEventRecord(ev[1], st));
    // A: hash links
    const uint32_t *d_hflags = nullptr;
    if (!fast && !fast_hist_in.empty() && nseg == 1 && segs[0].seg_start > 0) {
        // the history holds bytes a DeflateFast level compressed (SetLevel 1-4 -> 5-9 at a flush): only the positions it
        // inserted are in the hash chains (C/DeflaterEngine.cs:697-712); bit q = buffer position q
        std::vector<uint32_t> hf(fast_hist_in);
        hf.resize(((size_t)segs[0].seg_start + 31) / 32 + 1, 0u);
        if ((rc = upload(*this, hist_flags_dev, hf, st))) return rc;
        d_hflags = (const uint32_t *)hist_flags_dev.p;
    }
    if (!side) HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    if (!ev_guard) HIPCHK(hipEventCreateWithFlags(&ev_guard, hipEventDisableTiming));
    if (!ev_gjoin) HIPCHK(hipEventCreateWithFlags(&ev_gjoin, hipEventDisableTiming));
    launch_links(d_in, in_total, dsegs, (const uint64_t *)p_bnds, (const SpanDev *)p_spans, (int)spans.size(), (uint16_t *)link.p, d_hflags, dcnt + CNT_LINKS_GUARD, total_emit, st,
                 side, ev_guard);
    HIPCHK(hipEventRecord(ev_gjoin, side));      // (joined in front of the read-back of the counters)
    HIPCHK(hipEventRecord(ev[2], st));
    if (fast) {
        // B+C for DeflateFast: sequential greedy parse, one wavefront per segment (szl_kernels_fast.hip)
        if ((rc = upload(*this, blk_off, fast_blk_off, st))) return rc;
        if (!fast_hist_in.empty() && nseg == 1) // inserted bits of the history (streaming Deflater / preset dictionary)
            HIPCHK(hipMemcpyAsync((uint32_t *)visited.p + segs[0].vis_word_off, fast_hist_in.data(),
                                  std::min<size_t>(fast_hist_in.size(), ((size_t)segs[0].seg_start + 31) / 32) * 4, hipMemcpyHostToDevice, st));
        HIPCHK(launch_fast(d_in, (const uint16_t *)link.p, dsegs, nseg, P, (uint32_t *)visited.p, dso, (uint32_t *)tokens.p,
                           (const uint64_t *)blk_off.p, (int64_t *)bsp.p, (int64_t *)blp.p, st));
        HIPCHK(hipEventRecord(ev[3], st));
        if (fast_want_tail && nseg == 1) {
            const int64_t t0 = std::max<int64_t>(0, segs[0].seg_end - WSIZE);
            fast_tail_start = t0 & ~(int64_t)31;
            const size_t w0 = (size_t)(fast_tail_start >> 5), w1 = ((size_t)segs[0].seg_end + 31) >> 5;
            fast_tail_bits.assign(w1 - w0, 0u);
            if (w1 > w0)
                HIPCHK(hipMemcpyAsync(fast_tail_bits.data(), (const uint32_t *)visited.p + segs[0].vis_word_off + w0, (w1 - w0) * 4,
                                      hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipEventRecord(ev[4], st));
    } else {
    // B: match tables
    // Two forms of stage B (szl_kernels_match.hip): search every position, or only the positions a parse can reach.
    // The second evaluates far fewer positions on repetitive data but each evaluation costs ~3x more, so a pilot on a
    // sample of tiles measures the evaluated fraction first (results are identical either way).
    // Round 3: measured again on ten data classes at levels 6 and 9 (tools/gpu_forms_classes.py, profiles/r03/forms_by_class.log), the
    // full search wins everywhere, by 1.3x (logs) to 5x (four-symbol text) — k_match4 is 1.5x the search the 0.25 threshold was set
    // against — so the default is the full search, without the pilot (0.75 ms of the 1 GiB pass; config 5, 1 GiB of logs at level 9,
    // 261 -> 190 ms).  The on-demand form and the pilot stay behind SZL_MATCH_MODE = 1 / 2 and the tests that force them.
    // Round 5: the product library holds ONE form of stage B; the on-demand form (k_match_lazy) and its pilot live in the laboratory library.
#if SZL_LAB
    static const int match_mode_env = getenv("SZL_MATCH_MODE") ? atoi(getenv("SZL_MATCH_MODE")) : 0; // 0 full, 1 on demand, 2 pilot
    const int match_mode = has_switch ? 0 : (match_mode_override >= 0 ? match_mode_override : match_mode_env);
#else
    const int match_mode = 0;
#endif
#if SZL_LAB
    static const double lazy_max_frac = getenv("SZL_LAZY_FRAC") ? atof(getenv("SZL_LAZY_FRAC")) : 0.25; // break-even was 0.36-0.40 against k_match; the full search is 1.4-1.55x faster now
#else
    const double lazy_max_frac = 0.25;
#endif
    last_pilot_frac = -1.0;
    bool b_event = false; // ev[7]: start of the stage-B search proper (after the pilot)
    // (the pilot is one on-demand walk of a tile sample, ~0.5 ms whatever the input: more than the whole search of a few MiB — calls of
    // 256 KiB-1 MiB spent 0.46-0.56 ms in it and 0.1-0.15 ms in the search it was to speed up)
    const bool pilot_worth = total_emit >= ((uint64_t)std::max(1, SZL_LABKNOB("SZL_PILOT_MIN_MIB", 8)) << 20) && ntiles >= 64;
    if (match_mode == 1 || (match_mode == 2 && pilot_worth)) {
        if (match_mode == 2) { // the pilot's own entries are overwritten by whichever form runs afterwards
            const uint64_t step = ntiles >= 16384 ? 256 : (ntiles >= 4096 ? 128 : (ntiles >= 1024 ? 64 : 8)); // >= 64 sampled tiles
            const int nb = (int)((ntiles + step - 1) / step);
            uint64_t sampled = 0;
            for (uint64_t t = 0; t < ntiles; t += step) sampled += (uint64_t)tiles[t].len;
            HIPCHK(launch_match_lazy(d_in, dsegs, (const TileDev *)p_tiles, nb, 0, (int)step, (const uint16_t *)link.p, mt, P, dcnt, st));
            if ((rc = d2h_small(pin + 192, (unsigned long long *)counters.p + 6, 8, st))) return rc;
            HIPCHK(hipStreamSynchronize(st));
            const unsigned long long ne = *(volatile unsigned long long *)(pin + 192);
            last_pilot_frac = sampled ? (double)ne / (double)sampled : 1.0;
            lazy = last_pilot_frac < lazy_max_frac;
        } else lazy = true;
        HIPCHK(hipEventRecord(ev[7], st));
        b_event = true;
        if (lazy) {
            HIPCHK(hipMemsetAsync(mt.m2, 0xFF, mt_stride * 4, st)); // M_UNSET
            HIPCHK(launch_match_lazy(d_in, dsegs, (const TileDev *)p_tiles, (int)ntiles, 0, 1, (const uint16_t *)link.p, mt, P, dcnt, st));
        }
    }
    if (!lazy) mt.form = pick_text_form((const uint16_t *)link.p, 0, (int64_t)in_total, st);   // (one synchronisation, calls of 8 MiB or more)
    if (!b_event) HIPCHK(hipEventRecord(ev[7], st));
    if (!lazy && m3) {
        if ((rc = link4.ensure(in_total * 2 + 64)) || (rc = skip4.ensure(in_total + 64)) || (rc = e3dist.ensure(in_total * 2 + 64)) ||
            (rc = e3hops.ensure(in_total + 64))) return rc;
        launch_links4(d_in, (const uint16_t *)link.p, 0, (int64_t)in_total, (int64_t)in_total, (uint16_t *)link4.p, (uint8_t *)skip4.p,
                      (uint16_t *)e3dist.p, (uint8_t *)e3hops.p, st);
        MTab mt3 = mt;
        mt3.link4 = (const uint16_t *)link4.p; mt3.skip4 = (const uint8_t *)skip4.p;
        mt3.e3d = (const uint16_t *)e3dist.p; mt3.e3h = (const uint8_t *)e3hops.p;
        HIPCHK(launch_match(d_in, dsegs, (const TileDev *)p_tiles, (int)ntiles, (const uint16_t *)link.p, mt3, P, dcnt, st));
    } else if (!lazy && own_list) {
        if (form == 4) HIPCHK(launch_match_ring(d_in, dsegs, (const TileDev *)p_stripes, (int)stripes.size(), (const uint16_t *)link.p, mt, P, dcnt, st));
        else if (form == 5) {
            const int nslots = match5_slots();
            if ((rc = m5_scratch.ensure(match5_scratch_bytes(nslots)))) return rc;
            HIPCHK(launch_match5(d_in, in_total, dsegs, (const uint64_t *)p_bnds, (const TileDev *)p_stripes, (int)stripes.size(), mt, P, (uint8_t *)m5_scratch.p, nslots, dcnt, st));
        }
        else HIPCHK(launch_match(d_in, dsegs, (const TileDev *)p_stripes, (int)stripes.size(), (const uint16_t *)link.p, mt, P, dcnt, st));
    }
    else if (!lazy && !has_switch) HIPCHK(launch_match(d_in, dsegs, (const TileDev *)p_tiles, (int)ntiles, (const uint16_t *)link.p, mt, P, dcnt, st));
    if (has_switch) { // tiles are grouped by parameter set: group 0 = the call's P, group k = sw_P[k-1] of the (single) switching segment
        size_t a = 0;
        while (a < tiles.size()) {
            size_t b = a;
            while (b < tiles.size() && tiles[b].pad2 == tiles[a].pad2) b++;
            LevelParams Pk = P;
            if (tiles[a].pad2 > 0) Pk = sw_P_in[(size_t)tiles[a].pad2 - 1];
            HIPCHK(launch_match(d_in, dsegs, (const TileDev *)p_tiles + a, (int)(b - a), (const uint16_t *)link.p, mt, Pk, dcnt, st));
            a = b;
        }
    }
    HIPCHK(hipEventRecord(ev[3], st));
    // C: parse
    const bool emit_copy = emit_copy_enabled(); // the speculative walk keeps its tokens; emission copies them (szl_kernels_parse.hip)
    if (emit_copy && (rc = spec_tok.ensure(in_total * 4 + 1024))) return rc;
    launch_spec(d_in, (const uint16_t *)link.p, mt, dsegs, nseg, nranges, P, (RangeDev *)ranges.p, (uint32_t *)visited.p, dcnt,
                emit_copy ? (uint32_t *)spec_tok.p : nullptr, st);
    launch_fix(d_in, (const uint16_t *)link.p, mt, dsegs, nseg, nranges, P, (RangeDev *)ranges.p, (const uint32_t *)visited.p, dcnt,
               (uint32_t *)bad_slot.p, (uint64_t *)bad_range.p, st);
    {   // how many ranges never merged?  (one 8-byte read-back; the common answer is 0)
        if ((rc = d2h_small(pin, counters.p, 8, st))) return rc;   // (by kernel into mapped pinned memory: no copy engine between the device and the host's wait)
        HIPCHK(hipStreamSynchronize(st));
        const unsigned long long nbad = *(volatile unsigned long long *)pin;
        if (nbad > 0 && lazy) // ranges that never re-synchronise are chained position by position: evaluate everything first
            HIPCHK(launch_match(d_in, dsegs, (const TileDev *)p_tiles, (int)ntiles, (const uint16_t *)link.p, mt, P, dcnt, st));
        if (nbad > 0 && nbad <= 48) {
            launch_resolve(d_in, (const uint16_t *)link.p, mt, dsegs, nseg, P, (RangeDev *)ranges.p, (const uint32_t *)visited.p, dcnt, st);
        } else if (nbad > 48) {
            const size_t W = (size_t)exitmap_width();
            if ((rc = exmap.ensure(nbad * W * 2 + 64))) return rc;
            if ((rc = cnmap.ensure(nbad * W * 2 + 64))) return rc;
            if (nseg == 1 && (rc = chain_buf.ensure(exitchain_scratch_bytes(segs[0].range_cnt)))) return rc;
            launch_exitmaps(d_in, (const uint16_t *)link.p, mt, dsegs, nseg, P, (RangeDev *)ranges.p, (const uint32_t *)visited.p,
                            (const uint32_t *)bad_slot.p, (const uint64_t *)bad_range.p, nbad, (uint16_t *)exmap.p, (uint16_t *)cnmap.p, dcnt, st,
                            nseg == 1 ? chain_buf.p : nullptr, segs[0].range_cnt);
        }
    }
    if (nranges > 0 && counts_to_blocks_fits(nranges, nseg) && knob("SZL_SMALL_TAIL", 1) != 0)   // (a small call: the four steps in one launch)
        launch_counts_to_blocks((const RangeDev *)ranges.p, nranges, dsegs, nseg, (uint32_t *)counts.p, (uint64_t *)range_tok.p, dso, (uint32_t *)blk_counts.p, (uint64_t *)blk_off.p, st);
    else {
        launch_range_counts((const RangeDev *)ranges.p, nranges, (uint32_t *)counts.p, st);
        HIPCHK(launch_exscan((const uint32_t *)counts.p, (uint64_t *)range_tok.p, nranges + 1, cubtmp.p, st));
        launch_seg_tokens(dsegs, nseg, (const uint64_t *)range_tok.p, dso, (uint32_t *)blk_counts.p, st);
        HIPCHK(launch_exscan((const uint32_t *)blk_counts.p, (uint64_t *)blk_off.p, (uint64_t)nseg + 1, cubtmp.p, st));
    }
    if (emit_copy)
        launch_emit_copy(d_in, (const uint16_t *)link.p, mt, dsegs, nseg, nranges, P, (const RangeDev *)ranges.p, (const uint32_t *)visited.p,
                         (const uint32_t *)spec_tok.p, (const uint64_t *)range_tok.p, dso, (uint32_t *)tokens.p, (const uint64_t *)blk_off.p,
                         (int64_t *)bsp.p, (int64_t *)blp.p, st);
    else
        launch_emit(d_in, (const uint16_t *)link.p, mt, dsegs, nseg, nranges, P, (const RangeDev *)ranges.p,
                (const uint64_t *)range_tok.p, dso, (uint32_t *)tokens.p, (const uint64_t *)blk_off.p, (int64_t *)bsp.p, (int64_t *)blp.p, dcnt, st);
    HIPCHK(hipEventRecord(ev[4], st));
    }
    // a segment that ends where the reference's engine stands at a SetLevel to another function (SEG_SWITCH_CUT): DeflateFast cut itself
    // (k_fast); DeflateSlow's token stream is cut here
    if (!fast && nseg == 1 && (segs[0].flags & SEG_SWITCH_CUT))
        launch_switch_cut(d_in, dsegs, (uint32_t *)tokens.p, dso, (const uint64_t *)blk_off.p, (const int64_t *)bsp.p, (int64_t *)blp.p, st);
    // D: blocks
    launch_seg_blocks(dsegs, nseg, (const uint32_t *)tokens.p, (const uint64_t *)blk_off.p, dso, fast ? 1 : 0, st);
    launch_block_build(dsegs, nseg, dso, (const uint64_t *)blk_off.p, (const uint32_t *)tokens.p, (const int64_t *)bsp.p, (const int64_t *)blp.p,
                       (BlockDesc *)descs.p, (uint32_t)blk_slots, fast ? 1 : 0, st);
    launch_block_scan(dsegs, nseg, dso, (BlockDesc *)descs.p, st);
    HIPCHK(hipEventRecord(ev[5], st));
    HIPCHK(hipStreamWaitEvent(st, ev_zjoin, 0));
    side_join.armed = false;
    launch_block_encode(d_in, d_out, dsegs, (const BlockDesc *)descs.p, (const uint32_t *)tokens.p, (uint32_t)blk_slots, st);
    if (forked) HIPCHK(hipStreamWaitEvent(st, ev_join, 0));
    HIPCHK(hipStreamWaitEvent(st, ev_gjoin, 0));
    launch_seg_finish(dsegs, nseg, dso, d_out, st);
    HIPCHK(hipEventRecord(ev[6], st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(results.data(), d_so.p, nseg * sizeof(SegOut), hipMemcpyDeviceToHost, st));
    unsigned long long hc[48] = {0};
    HIPCHK(hipMemcpyAsync(hc, counters.p, sizeof hc, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (hc[CNT_LINKS_GUARD]) return links_guard_tripped();

    float ms[6] = {0};
    for (int i = 0; i < 6; i++) (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    timing.checksum_ms = ms[0]; timing.links_ms = ms[1]; timing.match_ms = ms[2];
    timing.parse_ms = ms[3];
    if (!fast) { // ev[7] separates the pilot (tile sample + read-back) from the search proper
        float pm = 0, mm = 0;
        (void)hipEventElapsedTime(&pm, ev[2], ev[7]);
        (void)hipEventElapsedTime(&mm, ev[7], ev[3]);
        timing.pilot_ms = pm; timing.match_ms = mm;
    }
    timing.blocks_ms = ms[4]; timing.encode_ms = ms[5];
    (void)hipEventElapsedTime(&timing.total_ms, ev[0], ev[6]);
    timing.in_bytes = seg_bytes;
    timing.ranges_unmerged = hc[0];
    timing.fallback_walks = hc[1];
    last_evaluated = hc[6]; last_eval_fallbacks = hc[7]; last_lazy = lazy;
    if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] match: quick wave-steps %llu (avg lanes %.1f), verify wave-steps %llu (avg lanes %.1f), positions %llu\n", hc[2], hc[2] ? (double)hc[3] / hc[2] : 0.0, hc[5], hc[5] ? (double)hc[4] / hc[5] : 0.0, (unsigned long long)seg_bytes);
    if (knob("SZL_DEBUG", 0) && hc[22])   // k_match8 (ring): per-wave loop statistics
        fprintf(stderr, "[szl] match8: engine calls %llu (busy contexts at entry %.1f) | starving %llu | chunks staged %llu, staging refused %llu, lock busy %llu | idle sleeps %llu\n",
                hc[22], (double)hc[28] / hc[22], hc[23], hc[27], hc[24], hc[25], hc[26]);
    if (knob("SZL_DEBUG", 0) && hc[31])   // k_match5 (bucket order): lockstep wave-steps and the lanes that took part
        fprintf(stderr, "[szl] k_match5: %.2f candidates per position, %.1f %% of the lane-steps do work, %.3f byte comparisons and %.3f best_len updates per position\n",
                (double)hc[30] / (double)seg_bytes, 100.0 * (double)hc[30] / (64.0 * (double)hc[31]), (double)hc[29] / (double)seg_bytes, (double)hc[28] / (double)seg_bytes);
    if (knob("SZL_DEBUG", 0) && hc[31]) { // (wall_clock64 runs at 100 MHz)
        const double tot = (double)(hc[9] + hc[10] + hc[11] + hc[12] + hc[13]);
        const double u = 1e-5 / (double)match5_slots();   // ticks of 10 ns summed over the workgroups -> ms per workgroup
        fprintf(stderr, "[szl] k_match5 phases (ms per workgroup): clear+histogram %.2f, scan %.2f, scatter %.2f, window bytes %.2f, search %.2f (sum %.2f)\n",
                u * hc[9], u * hc[10], u * hc[11], u * hc[12], u * hc[13], u * tot);
    }
    if (knob("SZL_DEBUG", 0) && hc[31]) {
        const double tot = (double)(hc[17] + hc[18] + hc[19] + hc[20] + hc[21]);
        const double u = 1e-5 / (16.0 * (double)match5_slots());
        fprintf(stderr, "[szl] k_match5 search (ms per wave): slice setup + staging %.2f, candidate counts %.2f, pass 1 %.2f, pass 2 %.2f, settle + store + restage %.2f (sum %.2f)\n",
                u * hc[17], u * hc[18], u * hc[19], u * hc[20], u * hc[21], u * tot);
    }
    if (knob("SZL_DEBUG", 0) && hc[16] && !hc[31]) { // k_match4 (two-context engine): loop iterations of each phase and the contexts (of 128) that took part
        const double np = (double)seg_bytes;
        fprintf(stderr, "[szl] match4: engine calls %llu | fetch visits %llu lanes/visit %.1f | QUICK iterations %llu contexts/iter %.1f (x2 steps) | VERIFY iterations %llu contexts/iter %.1f | "
                        "COMPLETE passes %llu contexts/pass %.1f | per position: quick ctx-steps %.2f verify ctx-steps %.2f completes %.2f\n",
                hc[8], hc[14], hc[14] ? (double)hc[15] / hc[14] : 0.0, hc[16], (double)hc[17] / hc[16], hc[18], hc[18] ? (double)hc[19] / hc[18] : 0.0,
                hc[20], hc[20] ? (double)hc[21] / hc[20] : 0.0, 2.0 * hc[17] / np, hc[19] / np, hc[21] / np);
    }
    if (knob("SZL_DEBUG", 0) && hc[43]) {   // k_match9: the tiles' timeline (wall_clock64: 10 ns ticks), averaged over the tiles
        const double u = 0.01 / (double)hc[43];
        fprintf(stderr, "[szl] stage B tiles: %llu, per tile: staging %.1f us, until the first wavefront runs out of positions %.1f us, from there to the last wavefront's end %.1f us\n",
                hc[43], u * hc[40], u * hc[41], u * hc[42]);
        fprintf(stderr, "[szl] stage B tails: last wavefront out of positions %.1f us after the first; a wavefront's own tail: mean %.1f us, longest of the tile %.1f us\n",
                u * hc[45], u * hc[44] / 16.0, u * hc[46]);
    }
    if (knob("SZL_DEBUG", 0)) fprintf(stderr, "[szl] stage B %s (pilot fraction %.3f): %llu of %llu positions evaluated by walkers, %llu by the parse (eval_global), slow walks %llu, unmerged %llu\n", lazy ? "on demand" : "full", last_pilot_frac, hc[6], (unsigned long long)seg_bytes, hc[7], hc[1], hc[0]);
    for (auto &r : results) { timing.out_bytes += r.out_bytes; timing.tokens += r.tok_count; timing.blocks += r.blk_count; }
    last_nranges = nranges; last_in_total = in_total; last_blk_slots = blk_slots;
    last_workspace_bytes = 0;
    for (DevBuf *b : {&link, &mtab, &tokens, &visited, &ranges, &counts, &range_tok, &descs, &bsp, &blp, &spec_tok, &bad_slot, &bad_range, &ckparts})
        last_workspace_bytes += b->cap;
    return 0;
}

// ---- window pipeline -----------------------------------------------------------------------------------------------------
__global__ void k_add_base(uint64_t *v, uint64_t n, uint64_t base) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] += base;
}

static int grow_preserve(DevBuf &b, size_t need, size_t used, hipStream_t st) { // like ensure(), but keeps the first `used` bytes
    if (need <= b.cap) return 0;
    void *np = nullptr;
    const size_t want = need + (need >> 1) + 256;
    if (hipMalloc(&np, want) != hipSuccess) { set_error("hipMalloc(%zu) failed", want); return SZL_E_NOMEM; }
    if (b.p && used) {
        if (hipMemcpyAsync(np, b.p, used, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipFree(np); return SZL_E_DEVICE; }
    }
    if (b.p) (void)hipFree(b.p);
    b.p = np; b.cap = want;
    return 0;
}

int Engine::deflate_windowed_impl(const uint8_t *d_in, uint64_t in_total, uint8_t *d_out, uint64_t out_total, SegDev seg,
                             const std::vector<uint64_t> &bnds, LevelParams P, unsigned want_ck, std::vector<SegOut> &results, hipStream_t st,
                             uint64_t window) {
    (void)out_total;
    memset(&timing, 0, sizeof timing);
    const bool dbg_lap = part.active && knob("SZL_DEBUG", 0) > 1;
    auto lap_t0 = std::chrono::steady_clock::now();
    auto LAP = [&](const char *w) { if (dbg_lap) { const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[szl]    lap %-28s %.2f ms\n", w, std::chrono::duration<double, std::milli>(t - lap_t0).count()); lap_t0 = t; } };
    for (auto &e : ev) if (!e) HIPCHK(hipEventCreate(&e));
    if (!pin) HIPCHK(hipHostMalloc((void **)&pin, 256, hipHostMallocDefault));
    const int64_t S0 = seg.seg_start, N = seg.seg_end;           // the real segment [S0, N) inside its stream buffer
    const uint64_t n = (uint64_t)(N - S0);
    uint32_t range_len = C_RANGE;                                 // (as in deflate_impl: enough ranges per window to fill the device)
    while (range_len > 256 && window / range_len < 262144) range_len >>= 1;
    seg.look_end = N; seg.range_off = 0; seg.vis_word_off = 0; seg.range_len = range_len;
    // part of a stream (PartRun): ranges end at NP, not at the end of the bytes the engine sees; no stage D here
    const bool is_part = part.active;
    const int64_t NP = is_part ? part.parse_end : N;
    const bool final_part = NP >= N;
    int rc;
    // ---- whole-stream tables and buffers
    const uint64_t blk_slots = n / BLOCK_TOKENS + 1;
    const uint64_t nchunks = (n + 4095) / 4096;
    std::vector<uint64_t> chunk_off{0, nchunks}, zero_off{0, (seg.out_cap + (uint64_t)zero_piece_bytes() - 1) / (uint64_t)zero_piece_bytes()};
    std::vector<uint64_t> fixed_blk_off{0, blk_slots};
    if ((rc = descs.ensure((blk_slots + 1) * sizeof(BlockDesc))) || (rc = d_so.ensure(2 * sizeof(SegOut))) || (rc = blk_counts.ensure(16)) ||
        (rc = bsp.ensure((blk_slots + 1) * 8)) || (rc = blp.ensure((blk_slots + 1) * 8)) || (rc = counters.ensure(CNT_WORDS * 8)) ||
        (want_ck && (rc = ckparts.ensure((nchunks + 1) * checksum_partial_bytes()))) ||
        (rc = upload(*this, d_bnds, bnds, st)) || (rc = upload(*this, ckoff, chunk_off, st)) || (rc = upload(*this, d_zoff, zero_off, st)) || (rc = upload(*this, blk_off, fixed_blk_off, st)))
        return rc;
    LAP("whole-stream tables");
    // d_segs: [0] = the real segment (checksums, zeroing, stage D), [1] = the current window's pseudo-segment (stages A-C)
    if ((rc = d_segs.ensure(2 * sizeof(SegDev)))) return rc;
    if ((rc = h2d_small(d_segs.p, &seg, sizeof seg, st))) return rc;
    const SegDev *dseg_real = (const SegDev *)d_segs.p, *dseg_win = dseg_real + 1;
    SegOut *dso = (SegOut *)d_so.p;                               // [0] real, [1] window view (token indices are global: tok_first 0)
    unsigned long long *dcnt = (unsigned long long *)counters.p;
    HIPCHK(hipEventRecord(ev[0], st));
    auto wait_input = [&](uint64_t upto) { // overlapped H2D of the arena (szl_deflate_batch_host): bytes [0, upto) must have landed
        if (!in_ready) return;
        const uint64_t need = seg.buf_off + upto;
        while (*in_ready < (need < in_total ? need : in_total)) std::this_thread::yield();
    };
    if (!is_part) launch_zero_regions(dseg_real, 1, (const uint64_t *)d_zoff.p, zero_off[1], d_out, st);
    HIPCHK(hipMemsetAsync(d_so.p, 0, 2 * sizeof(SegOut), st));
    HIPCHK(hipMemsetAsync(counters.p, 0, CNT_WORDS * 8, st));
    const bool forked = want_ck && !in_ready && !is_part;          // (with an overlapped copy the checksums run at the end, when all bytes are there)
    if (is_part) {
    } else if (forked) {
        if (!side) HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        if (!ev_fork) HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        if (!ev_join) HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev_fork, st));
        HIPCHK(hipStreamWaitEvent(side, ev_fork, 0));
        launch_checksums(d_in, dseg_real, 1, (const uint64_t *)ckoff.p, nchunks, ckparts.p, dso, want_ck, side);
        HIPCHK(hipEventRecord(ev_join, side));
    } else if (!want_ck) {
        launch_checksums(d_in, dseg_real, 1, (const uint64_t *)ckoff.p, 0, ckparts.p, dso, 0, st); // seeds the running values
    }
    const uint64_t max_ranges = (window + window / 4) / range_len + 2;
    if ((rc = counts.ensure((max_ranges + 2) * 4)) || (rc = range_tok.ensure((max_ranges + 2) * 8)) || (rc = ranges.ensure((max_ranges + 1) * sizeof(RangeDev))) ||
        (rc = bad_slot.ensure((max_ranges + 2) * 4)) || (rc = bad_range.ensure((max_ranges + 2) * 8))) return rc;
    if ((rc = cubtmp.ensure(exscan_tmp_bytes(max_ranges + 1) + 256))) return rc;

#if SZL_LAB
    const int match_mode = match_mode_override >= 0 ? match_mode_override : SZL_LABKNOB("SZL_MATCH_MODE", 0);
#else
    const int match_mode = 0;      // (the product library holds one form of stage B)
#endif
#if SZL_LAB
    static const double lazy_max_frac = getenv("SZL_LAZY_FRAC") ? atof(getenv("SZL_LAZY_FRAC")) : 0.25;
#else
    const double lazy_max_frac = 0.25;
#endif
    const bool emit_copy = emit_copy_enabled();
    bool lazy = match_mode == 1;
    last_pilot_frac = -1.0;
    uint64_t tok_base = 0, total_unmerged = 0, peak = 0;
    float ms_links = 0, ms_match = 0, ms_parse = 0, ms_pilot = 0;
    int64_t e = S0;                                                // clean iteration the next window starts on
    bool warming = false;
    if (is_part) {
        if (part.force_entry >= 0) e = part.force_entry;
        else if (part.warm_from >= 0) { e = part.warm_from; warming = true; }
        else e = part.first;
        part.entry = e;
    }
    for (uint32_t wi = 0; e < NP || (wi == 0 && n == 0); wi++) {
        int64_t wend = e + (int64_t)window;
        if (NP - wend < (int64_t)(window / 4)) wend = NP;          // no sliver at the end
        if (warming) wend = part.first;                            // the warm-up stretch is a window of its own
        const bool part_end = wend >= NP;                          // the part's last window (not the stream's, unless final_part)
        const bool last = part_end && final_part;
        if (part_end) wend = NP;
        const int64_t hi = last ? N : std::min<int64_t>(N, wend + C_WIN_HALO);   // links / table entries exist for [.., hi)
        const int64_t lo = std::max<int64_t>(0, e - WSIZE);                      // links from here (stage B stages 32512 of history)
        wait_input((uint64_t)std::min<int64_t>(N, hi + MAX_MATCH + 8));
        SegDev w = seg;
        w.seg_start = e; w.seg_end = wend; w.look_end = N;
        const uint64_t wn = (uint64_t)(wend - e);
        const uint64_t nranges = (wn + range_len - 1) / range_len;
        w.range_cnt = (uint32_t)nranges;
        if (nranges > max_ranges) { set_error("window bookkeeping"); return SZL_E_STATE; }
        if ((rc = h2d_small((void *)dseg_win, &w, sizeof w, st))) return rc;
        std::vector<SpanDev> spans;
        std::vector<TileDev> tiles;
        uint64_t span_len = ((uint64_t)(hi - lo) + 511) / 512;
        span_len = std::min<uint64_t>(std::max<uint64_t>(span_len, 1u << 17), 1u << 22);
        span_len = (span_len + 63) & ~63ull;
        for (int64_t a = lo; a < hi; a += (int64_t)span_len) spans.push_back(SpanDev{1, 0, a, std::min<int64_t>(a + (int64_t)span_len, hi)});
        const bool m3 = use_match3(P);
        int64_t tile_len = m3 ? match3_tile() : B_TILE;
        if (SZL_LABKNOB("SZL_TILE_LEN", 0) >= 1024) tile_len = std::min<int64_t>(tile_len, SZL_LABKNOB("SZL_TILE_LEN", 0) / 64 * 64);   // (lab)
        for (int64_t a = e; a < wend; a += tile_len) tiles.push_back(TileDev{1, 0, a, (int32_t)std::min<int64_t>(tile_len, wend - a), 0});
        const uint64_t ntiles = tiles.size();
        std::vector<TileDev> stripes;
        const int form = match_form(P, false);
        const int64_t stripe_len = m3 ? 0 : full_search_len((uint64_t)(wend - e), tile_len, form);
        for (int64_t a = e; stripe_len && a < wend; a += stripe_len) stripes.push_back(TileDev{1, 0, a, (int32_t)std::min<int64_t>(stripe_len, wend - a), 0});
        if (!stripes.empty() && (rc = upload(*this, d_stripes, stripes, st))) return rc;
        // side arrays of this window, addressed with the stream's own indices (pointer minus the window's first index)
        const uint64_t nlink = (uint64_t)(hi - lo), ntab = (uint64_t)(hi - e);
        const size_t mt_stride = (ntab + 63) & ~(size_t)63;
        if ((rc = link.ensure(nlink * 2 + 64)) || (rc = mtab.ensure(mt_stride * 8 + 64)) || (rc = visited.ensure((wn / 32 + 8) * 4)) ||
            (emit_copy && (rc = spec_tok.ensure(ntab * 4 + 1024))) || (rc = upload(*this, d_spans, spans, st)) || (rc = upload(*this, d_tiles, tiles, st))) return rc;
        uint16_t *lk = (uint16_t *)link.p - (seg.buf_off + (uint64_t)lo);
        MTab mt = {(uint32_t *)mtab.p - (seg.buf_off + (uint64_t)e), (uint32_t *)mtab.p + mt_stride - (seg.buf_off + (uint64_t)e)};
        uint32_t *stok = emit_copy ? (uint32_t *)spec_tok.p - (seg.buf_off + (uint64_t)e) : nullptr;
        LAP("window: side arrays");
        const uint64_t tok0 = is_part ? part.tok_start : 0;        // a part's tokens follow those of the parts before it (same buffer; indices are the part's own)
        if ((rc = grow_preserve(tokens, (tok0 + tok_base + wn + 16) * 4, (tok0 + tok_base) * 4, st))) return rc;
        HIPCHK(hipMemsetAsync(visited.p, 0, (wn / 32 + 8) * 4, st));
        HIPCHK(hipMemsetAsync(counts.p, 0, (nranges + 2) * 4, st));
        HIPCHK(hipMemsetAsync(counters.p, 0, 8, st));             // counter 0: ranges of THIS window that never merged
        HIPCHK(hipEventRecord(ev[1], st));
        // the launch wrappers index segs[span.seg] / segs[tile.seg]: entry 1 of d_segs is the window
        launch_links(d_in, in_total, dseg_real, (const uint64_t *)d_bnds.p, (const SpanDev *)d_spans.p, (int)spans.size(), lk, nullptr, dcnt + CNT_LINKS_GUARD, (uint64_t)(hi - lo), st, nullptr, nullptr);
        HIPCHK(hipEventRecord(ev[2], st));
        if (wi == 0 && match_mode == 2 && ntiles >= 64 && !warming) { // the pilot (see deflate()): once, on the first window
            const uint64_t step = ntiles >= 16384 ? 256 : (ntiles >= 4096 ? 128 : (ntiles >= 1024 ? 64 : 8));
            uint64_t sampled = 0;
            for (uint64_t t = 0; t < ntiles; t += step) sampled += (uint64_t)tiles[t].len;
            HIPCHK(launch_match_lazy(d_in, dseg_real, (const TileDev *)d_tiles.p, (int)((ntiles + step - 1) / step), 0, (int)step, lk, mt, P, dcnt, st));
            if ((rc = d2h_small(pin + 192, (unsigned long long *)counters.p + 6, 8, st))) return rc;
            HIPCHK(hipStreamSynchronize(st));
            const unsigned long long ne = *(volatile unsigned long long *)(pin + 192);
            last_pilot_frac = sampled ? (double)ne / (double)sampled : 1.0;
            lazy = last_pilot_frac < lazy_max_frac;
        }
        if (!lazy) mt.form = pick_text_form(lk + seg.buf_off, e, wend - e, st);   // (the window's own links: every window chooses)
        HIPCHK(hipEventRecord(ev[7], st));
        if (lazy) {
            HIPCHK(hipMemsetAsync(mtab.p, 0xFF, mt_stride * 4, st)); // M_UNSET
            HIPCHK(launch_match_lazy(d_in, dseg_real, (const TileDev *)d_tiles.p, (int)ntiles, 0, 1, lk, mt, P, dcnt, st));
        } else {
            MTab mtw = mt;
            if (m3) {   // four-byte links of the window: [lo, wend) is all a tile's LDS window holds
                if ((rc = link4.ensure(nlink * 2 + 64)) || (rc = skip4.ensure(nlink + 64)) || (rc = e3dist.ensure(nlink * 2 + 64)) ||
                    (rc = e3hops.ensure(nlink + 64))) return rc;
                const uint64_t bias = seg.buf_off + (uint64_t)lo;
                uint16_t *l4 = (uint16_t *)link4.p - bias, *ed = (uint16_t *)e3dist.p - bias;
                uint8_t *s4 = (uint8_t *)skip4.p - bias, *eh = (uint8_t *)e3hops.p - bias;
                launch_links4(d_in, lk, (int64_t)seg.buf_off + lo, (int64_t)seg.buf_off + wend, (int64_t)seg.buf_off + N, l4, s4, ed, eh, st);
                mtw.link4 = l4; mtw.skip4 = s4; mtw.e3d = ed; mtw.e3h = eh;
            }
            if (!stripes.empty() && form == 4) HIPCHK(launch_match_ring(d_in, dseg_real, (const TileDev *)d_stripes.p, (int)stripes.size(), lk, mtw, P, dcnt, st));
            else if (!stripes.empty() && form == 5) {
                const int nslots = match5_slots();
                if ((rc = m5_scratch.ensure(match5_scratch_bytes(nslots)))) return rc;
                HIPCHK(launch_match5(d_in, in_total, dseg_real, (const uint64_t *)d_bnds.p, (const TileDev *)d_stripes.p, (int)stripes.size(), mtw, P, (uint8_t *)m5_scratch.p, nslots, dcnt, st));
            }
            else if (!stripes.empty()) HIPCHK(launch_match(d_in, dseg_real, (const TileDev *)d_stripes.p, (int)stripes.size(), lk, mtw, P, dcnt, st));
            else HIPCHK(launch_match(d_in, dseg_real, (const TileDev *)d_tiles.p, (int)ntiles, lk, mtw, P, dcnt, st));
            if (!last) HIPCHK(hipMemsetAsync((uint32_t *)mtab.p + wn, 0xFF, (size_t)(hi - wend) * 4, st)); // the tail past the parse end: evaluated on demand
        }
        HIPCHK(hipEventRecord(ev[3], st));
        // ---- stage C on the window's ranges (as in deflate(); the window is "a segment with history" that starts on a clean iteration)
        launch_spec(d_in, lk, mt, dseg_win, 1, nranges, P, (RangeDev *)ranges.p, (uint32_t *)visited.p, dcnt, stok, st);
        launch_fix(d_in, lk, mt, dseg_win, 1, nranges, P, (RangeDev *)ranges.p, (const uint32_t *)visited.p, dcnt, (uint32_t *)bad_slot.p, (uint64_t *)bad_range.p, st);
        if ((rc = d2h_small(pin, counters.p, 8, st))) return rc;
        if ((rc = d2h_small(pin + 8, (unsigned long long *)counters.p + CNT_LINKS_GUARD, 8, st))) return rc;
        HIPCHK(hipStreamSynchronize(st));
        if (*(volatile unsigned long long *)(pin + 8)) return links_guard_tripped();      // (the window's links: see launch_links)
        const unsigned long long nbad = *(volatile unsigned long long *)pin;
        total_unmerged += nbad;
        if (nbad > 0 && lazy) HIPCHK(launch_match(d_in, dseg_real, (const TileDev *)d_tiles.p, (int)ntiles, lk, mt, P, dcnt, st));
        if (nbad > 0 && nbad <= 48) launch_resolve(d_in, lk, mt, dseg_win, 1, P, (RangeDev *)ranges.p, (const uint32_t *)visited.p, dcnt, st);
        else if (nbad > 48) {
            const size_t W = (size_t)exitmap_width();
            if ((rc = exmap.ensure(nbad * W * 2 + 64)) || (rc = cnmap.ensure(nbad * W * 2 + 64)) || (rc = chain_buf.ensure(exitchain_scratch_bytes(nranges)))) return rc;
            launch_exitmaps(d_in, lk, mt, dseg_win, 1, P, (RangeDev *)ranges.p, (const uint32_t *)visited.p, (const uint32_t *)bad_slot.p, (const uint64_t *)bad_range.p,
                            nbad, (uint16_t *)exmap.p, (uint16_t *)cnmap.p, dcnt, st, chain_buf.p, (uint32_t)nranges);
        }
        launch_range_counts((const RangeDev *)ranges.p, nranges, (uint32_t *)counts.p, st);
        HIPCHK(launch_exscan((const uint32_t *)counts.p, (uint64_t *)range_tok.p, nranges + 1, cubtmp.p, st));
        hipLaunchKernelGGL(k_add_base, dim3((unsigned)((nranges + 1 + 255) / 256)), dim3(256), 0, st, (uint64_t *)range_tok.p, nranges + 1, tok_base);
        // window's end: the clean iteration the true parse leaves it on, and its token count
        if ((rc = d2h_small(pin + 64, (RangeDev *)ranges.p + (nranges - 1), sizeof(RangeDev), st))) return rc;
        if ((rc = d2h_small(pin + 128, (uint64_t *)range_tok.p + nranges, 8, st))) return rc;
        HIPCHK(hipStreamSynchronize(st));
        RangeDev lastr;
        memcpy(&lastr, pin + 64, sizeof lastr);
        const uint64_t tok_after = *(volatile uint64_t *)(pin + 128);
        LAP("window: A-C, counts back");
        // emission view of the window: global token indices; only the stream's very last token closes a block early
        SegOut view{};
        view.tok_first = 0; view.tok_count = last ? tok_after : ~0ull >> 2;
        if ((rc = h2d_small(dso + 1, &view, sizeof view, st))) return rc;
        if (emit_copy)
            launch_emit_copy(d_in, lk, mt, dseg_win, 1, nranges, P, (const RangeDev *)ranges.p, (const uint32_t *)visited.p, stok, (const uint64_t *)range_tok.p,
                             dso + 1, (uint32_t *)tokens.p + tok0, (const uint64_t *)blk_off.p, (int64_t *)bsp.p, (int64_t *)blp.p, st);
        else
            launch_emit(d_in, lk, mt, dseg_win, 1, nranges, P, (const RangeDev *)ranges.p, (const uint64_t *)range_tok.p, dso + 1, (uint32_t *)tokens.p + tok0,
                        (const uint64_t *)blk_off.p, (int64_t *)bsp.p, (int64_t *)blp.p, dcnt, st);
        HIPCHK(hipEventRecord(ev[4], st));
        HIPCHK(hipStreamSynchronize(st));
        LAP("window: emit");
        float a = 0, b = 0, c = 0, pm = 0;
        (void)hipEventElapsedTime(&a, ev[1], ev[2]); (void)hipEventElapsedTime(&pm, ev[2], ev[7]); (void)hipEventElapsedTime(&b, ev[7], ev[3]); (void)hipEventElapsedTime(&c, ev[3], ev[4]);
        ms_links += a; ms_pilot += pm; ms_match += b; ms_parse += c;
        uint64_t ws = 0;
        for (DevBuf *bb : {&link, &mtab, &tokens, &visited, &ranges, &counts, &range_tok, &descs, &bsp, &blp, &spec_tok, &bad_slot, &bad_range, &ckparts}) ws += bb->cap;
        peak = std::max(peak, ws);
        tok_base = tok_after;
        if (last) { if (is_part) part.exit = N; break; }
        e = lastr.exit_true;                                       // >= wend: the next window starts here, on a clean iteration
        if (e < wend || e > wend + 1024) { set_error("window hand-over out of range"); return SZL_E_STATE; }
        if (warming) { warming = false; tok_base = 0; part.entry = e; continue; }   // the warm-up's tokens are dropped
        if (part_end) { part.exit = e; break; }                    // the next part's parse is entered here
        if (e >= N) { set_error("window hand-over reached the end of the stream"); return SZL_E_STATE; } // (the last window absorbs slivers)
    }
    if (is_part) {
        part.tok_count = tok_base;
        timing.links_ms = ms_links; timing.match_ms = ms_match; timing.parse_ms = ms_parse; timing.pilot_ms = ms_pilot;
        timing.in_bytes = (uint64_t)(part.exit - part.entry); timing.ranges_unmerged = total_unmerged; timing.tokens = tok_base;
        last_workspace_bytes = peak;
        return 0;
    }
    // ---- stage D over the whole token stream (as in deflate())
    SegOut whole{};
    whole.tok_first = 0; whole.tok_count = tok_base;
    if (!want_ck || forked) { // keep the checksum fields the kernels wrote into dso[0]
        HIPCHK(hipMemcpyAsync((char *)dso, &whole, offsetof(SegOut, blk_first), hipMemcpyHostToDevice, st));
    }
    if (want_ck && !forked) { // overlapped copy: every byte is there now
        wait_input((uint64_t)N);
        HIPCHK(hipMemcpyAsync((char *)dso, &whole, offsetof(SegOut, blk_first), hipMemcpyHostToDevice, st));
        launch_checksums(d_in, dseg_real, 1, (const uint64_t *)ckoff.p, nchunks, ckparts.p, dso, want_ck, st);
    }
    HIPCHK(hipEventRecord(ev[4], st));
    launch_seg_blocks(dseg_real, 1, (const uint32_t *)tokens.p, (const uint64_t *)blk_off.p, dso, 0, st);
    launch_block_build(dseg_real, 1, dso, (const uint64_t *)blk_off.p, (const uint32_t *)tokens.p, (const int64_t *)bsp.p, (const int64_t *)blp.p,
                       (BlockDesc *)descs.p, (uint32_t)blk_slots, 0, st);
    launch_block_scan(dseg_real, 1, dso, (BlockDesc *)descs.p, st);
    HIPCHK(hipEventRecord(ev[5], st));
    launch_block_encode(d_in, d_out, dseg_real, (const BlockDesc *)descs.p, (const uint32_t *)tokens.p, (uint32_t)blk_slots, st);
    if (forked) HIPCHK(hipStreamWaitEvent(st, ev_join, 0));
    launch_seg_finish(dseg_real, 1, dso, d_out, st);
    HIPCHK(hipEventRecord(ev[6], st));
    HIPCHK(hipGetLastError());
    results.assign(1, SegOut{});
    HIPCHK(hipMemcpyAsync(results.data(), d_so.p, sizeof(SegOut), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    float d1 = 0, d2 = 0;
    (void)hipEventElapsedTime(&d1, ev[4], ev[5]); (void)hipEventElapsedTime(&d2, ev[5], ev[6]); (void)hipEventElapsedTime(&timing.total_ms, ev[0], ev[6]);
    timing.links_ms = ms_links; timing.match_ms = ms_match; timing.parse_ms = ms_parse; timing.pilot_ms = ms_pilot; timing.blocks_ms = d1; timing.encode_ms = d2;
    timing.in_bytes = n; timing.ranges_unmerged = total_unmerged;
    timing.out_bytes = results[0].out_bytes; timing.tokens = results[0].tok_count; timing.blocks = results[0].blk_count;
    last_lazy = lazy; last_in_total = in_total; last_blk_slots = blk_slots; last_nranges = 0; last_mt_stride = 0;
    last_workspace_bytes = peak;
    return 0;
}

// Stage D of a stream whose tokens were produced elsewhere (PartRun): everything deflate_windowed does outside its window loop.
int Engine::finish_tokens(const uint8_t *d_in, uint64_t in_total, uint8_t *d_out, SegDev seg, uint64_t tok_total, unsigned want_ck,
                          std::vector<SegOut> &results, hipStream_t st) {
    for (auto &e : ev) if (!e) HIPCHK(hipEventCreate(&e));
    const int64_t S0 = seg.seg_start, N = seg.seg_end;
    const uint64_t n = (uint64_t)(N - S0);
    seg.look_end = N; seg.range_off = 0; seg.vis_word_off = 0; seg.range_len = C_RANGE;
    int rc;
    const uint64_t blk_slots = n / BLOCK_TOKENS + 1;
    const uint64_t nchunks = (n + 4095) / 4096;
    std::vector<uint64_t> chunk_off{0, nchunks}, zero_off{0, (seg.out_cap + (uint64_t)zero_piece_bytes() - 1) / (uint64_t)zero_piece_bytes()};
    std::vector<uint64_t> fixed_blk_off{0, blk_slots};
    if ((rc = descs.ensure((blk_slots + 1) * sizeof(BlockDesc))) || (rc = d_so.ensure(2 * sizeof(SegOut))) || (rc = blk_counts.ensure(16)) ||
        (rc = bsp.ensure((blk_slots + 1) * 8)) || (rc = blp.ensure((blk_slots + 1) * 8)) || (rc = counters.ensure(CNT_WORDS * 8)) ||
        (rc = ckparts.ensure((nchunks + 1) * checksum_partial_bytes())) || (rc = cubtmp.ensure((blk_slots + 1) * 12 + 256)) ||
        (rc = upload(*this, ckoff, chunk_off, st)) || (rc = upload(*this, d_zoff, zero_off, st)) || (rc = upload(*this, blk_off, fixed_blk_off, st)) ||
        (rc = d_segs.ensure(2 * sizeof(SegDev))))
        return rc;
    HIPCHK(hipMemcpyAsync(d_segs.p, &seg, sizeof seg, hipMemcpyHostToDevice, st));
    const SegDev *dseg_real = (const SegDev *)d_segs.p;
    SegOut *dso = (SegOut *)d_so.p;
    HIPCHK(hipEventRecord(ev[0], st));
    launch_zero_regions(dseg_real, 1, (const uint64_t *)d_zoff.p, zero_off[1], d_out, st);
    HIPCHK(hipMemsetAsync(d_so.p, 0, 2 * sizeof(SegOut), st));
    SegOut whole{};
    whole.tok_first = 0; whole.tok_count = tok_total;
    HIPCHK(hipMemcpyAsync((char *)dso, &whole, offsetof(SegOut, blk_first), hipMemcpyHostToDevice, st));
    launch_checksums(d_in, dseg_real, 1, (const uint64_t *)ckoff.p, want_ck ? nchunks : 0, ckparts.p, dso, want_ck, st);
    launch_block_positions((const uint32_t *)tokens.p, tok_total, S0, (uint64_t *)cubtmp.p, (uint32_t *)((uint64_t *)cubtmp.p + blk_slots + 1),
                           (int64_t *)bsp.p, (int64_t *)blp.p, st);
    HIPCHK(hipEventRecord(ev[4], st));
    launch_seg_blocks(dseg_real, 1, (const uint32_t *)tokens.p, (const uint64_t *)blk_off.p, dso, 0, st);
    launch_block_build(dseg_real, 1, dso, (const uint64_t *)blk_off.p, (const uint32_t *)tokens.p, (const int64_t *)bsp.p, (const int64_t *)blp.p,
                       (BlockDesc *)descs.p, (uint32_t)blk_slots, 0, st);
    launch_block_scan(dseg_real, 1, dso, (BlockDesc *)descs.p, st);
    HIPCHK(hipEventRecord(ev[5], st));
    launch_block_encode(d_in, d_out, dseg_real, (const BlockDesc *)descs.p, (const uint32_t *)tokens.p, (uint32_t)blk_slots, st);
    launch_seg_finish(dseg_real, 1, dso, d_out, st);
    HIPCHK(hipEventRecord(ev[6], st));
    HIPCHK(hipGetLastError());
    results.assign(1, SegOut{});
    HIPCHK(hipMemcpyAsync(results.data(), d_so.p, sizeof(SegOut), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    float d1 = 0, d2 = 0;
    (void)hipEventElapsedTime(&d1, ev[4], ev[5]); (void)hipEventElapsedTime(&d2, ev[5], ev[6]);
    timing.blocks_ms = d1; timing.encode_ms = d2;
    timing.in_bytes = n; timing.out_bytes = results[0].out_bytes; timing.tokens = results[0].tok_count; timing.blocks = results[0].blk_count;
    last_in_total = in_total; last_blk_slots = blk_slots;
    return 0;
}

int Engine::deflate_stored(const uint8_t *d_in, uint8_t *d_out, const std::vector<StoredBlk> &blks, unsigned want_ck, uint64_t ck_off,
                           uint64_t ck_len, uint32_t crc_init, uint32_t adler_init, uint32_t *crc_out, uint32_t *adler_out, hipStream_t st) {
    int rc;
    if ((rc = upload(*this, d_stored, blks, st))) return rc;
    launch_stored(d_in, d_out, (const StoredBlk *)d_stored.p, (uint32_t)blks.size(), st);
    if (want_ck) {
        std::vector<SegDev> segs(1);
        segs[0] = SegDev{};
        segs[0].buf_off = ck_off; segs[0].seg_start = 0; segs[0].seg_end = (int64_t)ck_len; segs[0].crc_init = crc_init; segs[0].adler_init = adler_init;
        std::vector<uint64_t> coff{0, (ck_len + 4095) / 4096};
        if ((rc = upload(*this, d_segs, segs, st)) || (rc = upload(*this, ckoff, coff, st)) || (rc = ckparts.ensure((coff[1] + 1) * checksum_partial_bytes())) ||
            (rc = d_so.ensure(sizeof(SegOut)))) return rc;
        launch_checksums(d_in, (const SegDev *)d_segs.p, 1, (const uint64_t *)ckoff.p, coff[1], ckparts.p, (SegOut *)d_so.p, want_ck, st);
        SegOut so{};
        HIPCHK(hipMemcpyAsync(&so, d_so.p, sizeof so, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (crc_out) *crc_out = so.crc32;
        if (adler_out) *adler_out = so.adler32;
    } else HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    return 0;
}

} // namespace szl
