// szl_api_inflate.hip — Inflater entry points of include/szl.h over the device decoder (k_inflate).
// Mirrors C/Inflater.cs: SetInput :629, Inflate :715, IsNeedingInput :783, IsFinished :806, RemainingInput :878,
// TotalIn/TotalOut :862/:848, Adler :823, Reset :188.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>
#include "szl_engine.h"
#include "szl_inflate.h"

using namespace szl;

namespace szl {
void launch_inflate(const uint8_t *in, uint8_t *out, InfJob *jobs, InfState *states, uint32_t njobs, bool one_shot, hipStream_t st);
void launch_checksums(const uint8_t *in, const SegDev *segs, uint32_t nseg, const uint64_t *chunk_off, uint64_t nchunks, void *parts,
                      SegOut *so, unsigned want, hipStream_t st);
size_t checksum_partial_bytes();
}
struct szl_engine { Engine e; };

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); return SZL_E_DEVICE; } } while (0)

// checksums of device regions described by (off,len) pairs; results[i] = {crc, adler}
namespace szl {
int region_checksums(const uint8_t *base, const std::vector<std::pair<uint64_t, uint64_t>> &regs, unsigned want,
                            std::vector<std::pair<uint32_t, uint32_t>> &out, const std::vector<std::pair<uint32_t, uint32_t>> *init, hipStream_t st) {
    const uint32_t n = (uint32_t)regs.size();
    out.assign(n, {0u, 1u});
    if (!n) return 0;
    std::vector<SegDev> segs(n);
    std::vector<uint64_t> coff(n + 1);
    uint64_t nch = 0;
    for (uint32_t i = 0; i < n; i++) {
        SegDev s{};
        s.buf_off = regs[i].first; s.seg_start = 0; s.seg_end = (int64_t)regs[i].second;
        s.crc_init = init ? (*init)[i].first : 0u; s.adler_init = init ? (*init)[i].second : 1u;
        segs[i] = s;
        coff[i] = nch; nch += (regs[i].second + 4095) / 4096;
    }
    coff[n] = nch;
    DevBuf dseg, doff, dparts, dso;
    int rc = 0;
    std::vector<SegOut> so(n);
    if ((rc = dseg.ensure(n * sizeof(SegDev))) || (rc = doff.ensure((n + 1) * 8)) || (rc = dparts.ensure((nch + 1) * checksum_partial_bytes())) ||
        (rc = dso.ensure(n * sizeof(SegOut)))) goto done;
    if (hipMemcpyAsync(dseg.p, segs.data(), n * sizeof(SegDev), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(doff.p, coff.data(), (n + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess) { rc = SZL_E_DEVICE; goto done; }
    launch_checksums(base, (const SegDev *)dseg.p, n, (const uint64_t *)doff.p, nch, dparts.p, (SegOut *)dso.p, want, st);
    if (hipMemcpyAsync(so.data(), dso.p, n * sizeof(SegOut), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = SZL_E_DEVICE; goto done; }
    for (uint32_t i = 0; i < n; i++) out[i] = {so[i].crc32, so[i].adler32};
done:
    dseg.release(); doff.release(); dparts.release(); dso.release();
    return rc;
}
} // namespace szl

extern "C" {

int szl_inflate_batch_device(szl_engine *e, const void *d_in, void *d_out, szl_stream *streams, size_t n, unsigned flags, void *hip_stream) {
    if (!e || (!streams && n)) return SZL_E_ARG;
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFull) return SZL_E_ARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const bool nowrap = flags & SZL_F_NOWRAP;
    std::vector<InfJob> jobs(n);
    std::vector<InfState> states(n);
    for (size_t i = 0; i < n; i++) {
        InfJob j{};
        j.in_off = streams[i].in_off; j.in_len = streams[i].in_len; j.out_off = streams[i].out_off; j.out_cap = streams[i].out_cap;
        j.window = nullptr; j.zlib = nowrap ? 0 : 1; j.keep_window = 0;
        jobs[i] = j;
        InfState s{};
        s.mode = nowrap ? INF_M_HEADER : INF_M_ZHEADER;
        states[i] = s;
    }
    DevBuf djobs, dstates;
    int rc;
    if ((rc = djobs.ensure(n * sizeof(InfJob))) || (rc = dstates.ensure(n * sizeof(InfState)))) { djobs.release(); dstates.release(); return rc; }
    auto cleanup = [&]() { djobs.release(); dstates.release(); };
    if (hipMemcpyAsync(djobs.p, jobs.data(), n * sizeof(InfJob), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dstates.p, states.data(), n * sizeof(InfState), hipMemcpyHostToDevice, st) != hipSuccess) { cleanup(); set_error("H2D failed"); return SZL_E_DEVICE; }
    for (int i = 0; i < 2; i++) if (!e->e.ev[i]) (void)hipEventCreate(&e->e.ev[i]);
    (void)hipEventRecord(e->e.ev[0], st);
    launch_inflate((const uint8_t *)d_in, (uint8_t *)d_out, (InfJob *)djobs.p, (InfState *)dstates.p, (uint32_t)n, true, st);
    (void)hipEventRecord(e->e.ev[1], st);
    if (hipMemcpyAsync(jobs.data(), djobs.p, n * sizeof(InfJob), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(states.data(), dstates.p, n * sizeof(InfState), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { cleanup(); set_error("inflate kernel/D2H failed: %s", hipGetErrorString(hipGetLastError())); return SZL_E_DEVICE; }
    cleanup();
    (void)hipEventElapsedTime(&e->e.timing.inflate_ms, e->e.ev[0], e->e.ev[1]);
    unsigned want = ((flags & SZL_F_CRC32) ? 1u : 0u) | (((flags & SZL_F_ADLER32) || !nowrap) ? 2u : 0u);
    std::vector<std::pair<uint32_t, uint32_t>> cks;
    if (want) {
        std::vector<std::pair<uint64_t, uint64_t>> regs(n);
        for (size_t i = 0; i < n; i++) regs[i] = {jobs[i].out_off, jobs[i].out_written};
        if ((rc = region_checksums((const uint8_t *)d_out, regs, want, cks, nullptr, st))) return rc;
    }
    for (size_t i = 0; i < n; i++) {
        szl_stream &s = streams[i];
        s.out_len = jobs[i].out_written;
        s.in_consumed = jobs[i].consumed;
        s.reserved = 0;
        s.crc32 = want ? cks[i].first : 0; s.adler32 = want ? cks[i].second : 1;
        int stt = jobs[i].status;
        if (stt == INF_FINISHED) {
            s.status = 0;
            if (!nowrap && states[i].adler_read != cks[i].second) s.status = SZL_E_ADLER_MISMATCH; // C/Inflater.cs:411-414
        } else if (stt == INF_NEED_INPUT) s.status = SZL_E_UNEXPECTED_EOF;
        else if (stt == INF_OUTPUT_FULL) s.status = SZL_E_OUTPUT_TOO_SMALL;
        else if (stt == INF_NEED_DICT) s.status = SZL_E_UNSUPPORTED; // batch call cannot supply a preset dictionary
        else s.status = stt < 0 ? stt : SZL_E_STATE;
    }
    return 0;
}

int szl_inflate_batch_host(szl_engine *e, const void *h_in, void *h_out, szl_stream *streams, size_t n, unsigned flags) {
    if (!e || (!streams && n)) return SZL_E_ARG;
    uint64_t in_total = 0, out_total = 0;
    for (size_t i = 0; i < n; i++) {
        in_total = std::max(in_total, streams[i].in_off + streams[i].in_len);
        out_total = std::max(out_total, streams[i].out_off + streams[i].out_cap);
    }
    int rc;
    if ((rc = e->e.stage_in.ensure(in_total + 64))) return rc;
    if ((rc = e->e.stage_out.ensure(out_total + 64))) return rc;
    if (in_total && hipMemcpy(e->e.stage_in.p, h_in, in_total, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D failed"); return SZL_E_DEVICE; }
    rc = szl_inflate_batch_device(e, e->e.stage_in.p, e->e.stage_out.p, streams, n, flags, nullptr);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++)
        if (streams[i].out_len && hipMemcpy((uint8_t *)h_out + streams[i].out_off, (uint8_t *)e->e.stage_out.p + streams[i].out_off,
                                            streams[i].out_len, hipMemcpyDeviceToHost) != hipSuccess) { set_error("D2H failed"); return SZL_E_DEVICE; }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Inflater (streaming object).  The decoder runs ahead of the caller: each kernel call decodes as
// much of the input given so far as fits a 256 KiB device buffer; Inflate() hands the bytes out.
struct szl_inflater {
    int no_header = 0;
    std::vector<uint8_t> hin;      // compressed bytes not yet consumed (hin[0] is stream byte `in_base`)
    uint64_t given = 0;            // total bytes ever passed to SetInput
    uint64_t in_base = 0;          // stream offset of hin[0]
    std::vector<uint8_t> pend;     // decoded bytes not yet handed out
    size_t pend_pos = 0;
    int64_t total_out = 0;
    InfState st{};                 // host mirror of the device state (bitpos relative to in_base)
    int dec_status = INF_NEED_INPUT;  // last status of the decoder
    int err = 0;                   // sticky error
    bool fresh_input = false;      // bytes were added since the decoder last reported NEED_INPUT
    bool have_dict = false;        // a preset dictionary sits in the device window
    uint32_t adler = 1;            // Adler-32 of the bytes handed out so far (zlib mode), excluding `unsummed`
    std::vector<uint8_t> unsummed; // handed-out bytes not yet folded into `adler` (folded on the device, lazily)
    uint32_t adler_dec = 1;        // Adler-32 of everything decoded so far
    DevBuf d_in, d_out, d_win, d_job, d_state;
    static constexpr size_t OUT_CHUNK = 256 * 1024;
};

static void inflater_clear(szl_inflater *s) {
    s->hin.clear(); s->given = 0; s->in_base = 0; s->pend.clear(); s->pend_pos = 0; s->total_out = 0;
    s->st = InfState{};
    s->st.mode = s->no_header ? INF_M_HEADER : INF_M_ZHEADER;
    s->dec_status = INF_NEED_INPUT; s->err = 0; s->fresh_input = false; s->have_dict = false; s->adler = 1; s->adler_dec = 1; s->unsummed.clear();
}

szl_inflater *szl_inflater_create(int no_header) {
    if (szl_device_count() <= 0) { set_error("no gfx950 device available"); return nullptr; }
    szl_inflater *s = new (std::nothrow) szl_inflater();
    if (!s) return nullptr;
    s->no_header = no_header ? 1 : 0;
    inflater_clear(s);
    return s;
}
void szl_inflater_destroy(szl_inflater *s) {
    if (!s) return;
    s->d_in.release(); s->d_out.release(); s->d_win.release(); s->d_job.release(); s->d_state.release();
    delete s;
}
int szl_inflater_reset(szl_inflater *s) { if (!s) return SZL_E_ARG; inflater_clear(s); return 0; }

static uint64_t consumed_bytes(const szl_inflater *s) { // stream bytes the decoder has consumed (partial byte counts, :131-137)
    return s->in_base + ((s->st.bitpos + 7) >> 3);
}
int szl_inflater_remaining_input(const szl_inflater *s) { // C/Inflater.cs:878
    if (!s) return 0;
    uint64_t c = consumed_bytes(s);
    return (int)(s->given > c ? s->given - c : 0);
}
int szl_inflater_needs_input(const szl_inflater *s) { // :783 — all given input was taken by the decoder
    if (!s) return 0;
    if (s->dec_status == INF_FINISHED) return szl_inflater_remaining_input(s) == 0;
    if (s->dec_status == INF_NEED_DICT) return s->hin.size() * 8 <= s->st.bitpos;
    return s->dec_status == INF_NEED_INPUT && !s->fresh_input;
}
int szl_inflater_needs_dictionary(const szl_inflater *s) { return s && s->dec_status == INF_NEED_DICT; } // :794
int szl_inflater_is_finished(const szl_inflater *s) { return s && s->dec_status == INF_FINISHED && s->pend_pos == s->pend.size(); } // :806
int64_t szl_inflater_total_in(const szl_inflater *s) { return s ? (int64_t)s->given - szl_inflater_remaining_input(s) : 0; } // :862
int64_t szl_inflater_total_out(const szl_inflater *s) { return s ? s->total_out : 0; }
static void fold_adler(szl_inflater *s) {
    if (s->unsummed.empty()) return;
    uint32_t v = s->adler;
    if (szl_adler32(s->adler, s->unsummed.data(), s->unsummed.size(), &v) == 0) s->adler = v;
    s->unsummed.clear();
}
uint32_t szl_inflater_adler(const szl_inflater *cs) { // :823
    szl_inflater *s = const_cast<szl_inflater *>(cs);
    if (!s || s->no_header) return 0;
    if (s->dec_status == INF_NEED_DICT) return s->st.adler_read; // IsNeedingDictionary => readAdler (:827-830)
    fold_adler(s);
    return s->adler;
}

int szl_inflater_set_input(szl_inflater *s, const uint8_t *p, int n) { // :629
    if (!s || n < 0 || (!p && n)) return SZL_E_ARG;
    if (!szl_inflater_needs_input(s) && s->given != 0) { set_error("Old input was not completely processed"); return SZL_E_STATE; }
    s->hin.insert(s->hin.end(), p, p + n);
    s->given += (uint64_t)n;
    if (n) s->fresh_input = true;
    return 0;
}
int szl_inflater_set_dictionary(szl_inflater *s, const uint8_t *p, int n) { // :563
    if (!s || n < 0 || (!p && n)) return SZL_E_ARG;
    if (s->dec_status != INF_NEED_DICT) { set_error("Dictionary is not needed"); return SZL_E_STATE; } // :573-576
    uint32_t a = 1;
    int rc = szl_adler32(1, p, (size_t)n, &a);
    if (rc) return rc;
    if (a != s->st.adler_read) { set_error("Wrong adler checksum"); s->err = SZL_E_ADLER_MISMATCH; return SZL_E_ADLER_MISMATCH; } // :583-586
    // OutputWindow.CopyDict (CS/OutputWindow.cs:130): the last 32 KiB of the dictionary become the history of the stream.
    // The device window is circular with index = position & 32767, so a byte at distance k before output position 0 lives at 32768-k.
    const int len = n > 32768 ? 32768 : n;
    if ((rc = s->d_win.ensure(32768))) return rc;
    if (hipMemset(s->d_win.p, 0, 32768) != hipSuccess) return SZL_E_DEVICE;
    if (len && hipMemcpy((uint8_t *)s->d_win.p + (32768 - len), p + (n - len), (size_t)len, hipMemcpyHostToDevice) != hipSuccess) return SZL_E_DEVICE;
    s->have_dict = true;
    s->dec_status = INF_NEED_INPUT;
    s->fresh_input = !s->hin.empty(); // the bytes after the DICTID are still waiting
    return 0;
}

// Run the decoder once over the input given so far.
static int inflater_step(szl_inflater *s) {
    int rc;
    const size_t nin = s->hin.size();
    if ((rc = s->d_in.ensure(nin + 64)) || (rc = s->d_out.ensure(szl_inflater::OUT_CHUNK + 64)) || (rc = s->d_win.ensure(32768)) ||
        (rc = s->d_job.ensure(sizeof(InfJob))) || (rc = s->d_state.ensure(sizeof(InfState)))) return rc;
    // One step produces at most OUT_CHUNK bytes, so it cannot need more than about that much input (stored data is 1:1):
    // upload a bounded prefix instead of the whole unconsumed input every step (a large SetInput would cost O(n^2) H2D).
    const size_t nup = std::min<size_t>(nin, szl_inflater::OUT_CHUNK + (64u << 10));
    InfJob j{};
    j.in_off = 0; j.in_len = nup; j.out_off = 0; j.out_cap = szl_inflater::OUT_CHUNK;
    j.window = (uint8_t *)s->d_win.p; j.zlib = s->no_header ? 0 : 1; j.keep_window = 1; j.load_window = s->have_dict ? 1 : 0;
    if (nup) HIPCHK(hipMemcpy(s->d_in.p, s->hin.data(), nup, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_job.p, &j, sizeof j, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_state.p, &s->st, sizeof s->st, hipMemcpyHostToDevice));
    launch_inflate((const uint8_t *)s->d_in.p, (uint8_t *)s->d_out.p, (InfJob *)s->d_job.p, (InfState *)s->d_state.p, 1, false, nullptr);
    HIPCHK(hipMemcpy(&j, s->d_job.p, sizeof j, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&s->st, s->d_state.p, sizeof s->st, hipMemcpyDeviceToHost));
    s->fresh_input = false;
    // A corrupt token stops the decoder, but everything it decoded before that point is still delivered (the reference hands
    // those bytes out over earlier Inflate() calls and throws only when it reaches the bad token): record the error, keep the
    // bytes; szl_inflater_inflate returns the error once they are drained.
    if (j.status < 0) s->err = j.status;
    else s->dec_status = j.status;
    if (j.status == INF_NEED_INPUT && nup < nin) s->fresh_input = true; // only the uploaded prefix ran dry
    if (j.out_written) {
        size_t old = s->pend.size();
        if (s->pend_pos == old) { s->pend.clear(); s->pend_pos = 0; old = 0; }
        s->pend.resize(old + j.out_written);
        HIPCHK(hipMemcpy(s->pend.data() + old, s->d_out.p, j.out_written, hipMemcpyDeviceToHost));
        if (!s->no_header) { // running Adler-32 of the decoded bytes, on the device (K/Adler32.cs)
            std::vector<std::pair<uint64_t, uint64_t>> regs{{0, j.out_written}};
            std::vector<std::pair<uint32_t, uint32_t>> init{{0u, s->adler_dec}}, out;
            if ((rc = region_checksums((const uint8_t *)s->d_out.p, regs, 2u, out, &init, nullptr))) return rc;
            s->adler_dec = out[0].second;
        }
    }
    if (s->err) return 0;
    if (s->dec_status == INF_FINISHED && !s->no_header && s->st.adler_read != s->adler_dec) { s->err = SZL_E_ADLER_MISMATCH; return 0; }
    // drop the consumed whole dwords of input; keep bitpos relative to the new base
    uint64_t drop = (s->st.bitpos >> 3) & ~3ull;
    if (s->st.mode == INF_M_ZHEADER) drop = 0;
    if (drop > s->hin.size()) drop = s->hin.size() & ~3ull;
    if (drop) {
        s->hin.erase(s->hin.begin(), s->hin.begin() + (ptrdiff_t)drop);
        s->in_base += drop;
        s->st.bitpos -= 8 * drop;
    }
    return 0;
}

int szl_inflater_inflate(szl_inflater *s, uint8_t *out, int count) { // :715
    if (!s || count < 0 || (!out && count)) return SZL_E_ARG;
    if (s->err && s->pend_pos == s->pend.size()) return s->err;
    int copied = 0;
    for (;;) {
        size_t avail = s->pend.size() - s->pend_pos;
        if (s->err && avail == 0) return copied ? copied : s->err; // bytes decoded before the error went out first
        if (avail && count) {
            size_t k = std::min<size_t>(avail, (size_t)count);
            memcpy(out, s->pend.data() + s->pend_pos, k);
            if (!s->no_header) { // Adler of what has been handed out == adler.Update in Inflate (:752-756); folded lazily on the device
                s->unsummed.insert(s->unsummed.end(), out, out + k);
                if (s->unsummed.size() > (4u << 20)) fold_adler(s);
            }
            s->pend_pos += k; out += k; count -= (int)k; copied += (int)k; s->total_out += (int64_t)k;
            if (count == 0) return copied;
        }
        if (s->dec_status == INF_FINISHED) return copied;
        if (s->dec_status == INF_NEED_DICT) return copied;                      // IsNeedingDictionary: the caller must SetDictionary
        if (s->dec_status == INF_NEED_INPUT && !s->fresh_input) return copied; // IsNeedingInput
        if (s->err) return copied; // (count == 0 with bytes still pending: nothing to hand out, nothing more to decode)
        int rc = inflater_step(s);
        if (rc) return rc;
        if (count == 0) return copied; // Inflate(…, 0): "count may be zero" still advances the decoder (:738-745)
    }
}

} // extern "C"
