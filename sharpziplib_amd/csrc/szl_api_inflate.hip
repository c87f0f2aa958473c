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
void launch_inflate_chunks(const uint8_t *in, InfJob *jobs, InfState *states, uint32_t njobs, int pass, hipStream_t st);
void launch_find_blocks(const uint8_t *in, uint64_t in_len, uint64_t chunk_bytes, uint32_t nchunks, uint64_t *start_bit, hipStream_t st);
void launch_resolve_wins(const uint16_t *sym, const uint64_t *out_off, uint32_t njobs, uint8_t *wins, hipStream_t st);
void launch_convert(const uint16_t *sym, const uint64_t *out_off, uint32_t njobs, const uint8_t *wins, uint8_t *out, uint64_t total, hipStream_t st);
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

// ---------------------------------------------------------------------------------------------
// One member decoded by many wavefronts (SURVEY §7.5 stage 1; kernels in szl_kernels_inflate_par.hip).
//   1. k_find_blocks: a dynamic block header in every chunk of compressed bytes -> candidate start bits
//   2. count pass (k_inflate<.,1>): each chunk decodes from its start to the next chunk's start, producing nothing but its
//      output length; chunk k must END exactly on chunk k+1's start bit — the proof that every start is a real block boundary
//      (chunk 0 starts at the member's first block, so the chain is anchored).  Starts that are not on the chain are dropped,
//      a chain end nobody starts at becomes a new start, and the pass is repeated for the changed chunks.
//   3. symbol pass (k_inflate<.,2>): the same decode into 16-bit symbols at each chunk's final output offset; a back-reference
//      that reaches in front of the chunk becomes "byte i of the preceding 32 KiB"
//   4. k_resolve_wins (front to back, 32 KiB per chunk) and k_convert (everything else, in parallel).
// Returns 1 = decoded (*res filled like a finished sequential job), 0 = not taken (anything unusual: stream errors, truncated
// input, output too small, no chain) — the caller then runs the ordinary decoder, whose status/partial-output behaviour is
// the one checked against the reference — or a negative szl_status for device failures.
struct ParResult { uint64_t out_written, consumed; uint32_t adler_read; };

static int inflate_member_parallel(Engine &E, const uint8_t *d_in, uint8_t *d_out, const szl_stream &s, bool zlib, hipStream_t st, ParResult *res) {
    const uint64_t chunk_bytes = (uint64_t)std::max(16, knob("SZL_INF_CHUNK_KIB", 128)) * 1024;
    if (s.in_len < 4 * chunk_bytes || s.in_len >= (1ull << 60)) return 0;
    const uint8_t *in = d_in + s.in_off;
    uint64_t first_bit = 0;
    if (zlib) { // C/Inflater.cs:211-249; a preset dictionary or a bad header is the sequential decoder's business
        uint8_t h[2];
        HIPCHK(hipMemcpyAsync(E.pin, in, 2, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
        memcpy(h, E.pin, 2);
        const uint32_t hv = ((uint32_t)h[0] << 8) | h[1];
        if (hv % 31 != 0 || (hv & 0x0f00) != (8u << 8) || (hv & 0x0020)) return 0;
        first_bit = 16;
    }
    const uint32_t nchunks = (uint32_t)std::min<uint64_t>((s.in_len + chunk_bytes - 1) / chunk_bytes, 1u << 20);
    int rc;
    if ((rc = E.inf_misc.ensure((uint64_t)nchunks * 8 + 64))) return rc;
    uint64_t *d_start = (uint64_t *)E.inf_misc.p;
    launch_find_blocks(in, s.in_len, chunk_bytes, nchunks, d_start, st);
    std::vector<uint64_t> starts(nchunks);
    HIPCHK(hipMemcpyAsync(starts.data(), d_start, (uint64_t)nchunks * 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
    std::vector<uint64_t> sb;            // start bits of the jobs, ascending
    sb.push_back(first_bit);
    for (uint32_t c = 1; c < nchunks; c++) if (starts[c] != ~0ull && starts[c] > sb.back()) sb.push_back(starts[c]);
    if (sb.size() < 4) return 0;

    struct Cnt { uint64_t end_bit, out; int status; };
    std::vector<InfJob> jobs;
    std::vector<Cnt> cnt;                // count-pass result per job, keyed like sb
    std::vector<char> have;
    auto run_pass = [&](int pass, const std::vector<uint32_t> &which, uint16_t *sym, const std::vector<uint64_t> *ooff) -> int {
        const uint32_t n = (uint32_t)which.size();
        jobs.assign(n, InfJob{});
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t j = which[k];
            InfJob &jb = jobs[k];
            jb.in_off = 0; jb.in_len = s.in_len; jb.out_cap = ~0ull >> 2;
            jb.start_bit = sb[j]; jb.stop_bit = j + 1 < sb.size() ? sb[j + 1] : ~0ull;
            jb.sym_out = sym ? sym + (*ooff)[j] : nullptr;
        }
        int r;
        if ((r = E.inf_jobs.ensure((uint64_t)n * sizeof(InfJob))) || (r = E.inf_states.ensure((uint64_t)n * sizeof(InfState)))) return r;
        HIPCHK(hipMemcpyAsync(E.inf_jobs.p, jobs.data(), (uint64_t)n * sizeof(InfJob), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(E.inf_states.p, 0, (uint64_t)n * sizeof(InfState), st));
        launch_inflate_chunks(in, (InfJob *)E.inf_jobs.p, (InfState *)E.inf_states.p, n, pass, st);
        HIPCHK(hipMemcpyAsync(jobs.data(), E.inf_jobs.p, (uint64_t)n * sizeof(InfJob), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    };

    // ---- count pass + chain repair
    cnt.assign(sb.size(), Cnt{});
    have.assign(sb.size(), 0);
    bool ok = false;
    for (int iter = 0; iter < 6 && !ok; iter++) {
        std::vector<uint32_t> which;
        for (uint32_t j = 0; j < sb.size(); j++) if (!have[j]) which.push_back(j);
        if (!which.empty()) {
            if ((rc = run_pass(1, which, nullptr, nullptr))) return rc;
            for (uint32_t k = 0; k < which.size(); k++) { cnt[which[k]] = Cnt{jobs[k].end_bit, jobs[k].out_written, jobs[k].status}; have[which[k]] = 1; }
        }
        // walk the chain from job 0; keep the starts it visits
        std::vector<uint64_t> nsb; std::vector<Cnt> ncnt; std::vector<char> nhave;
        uint32_t j = 0;
        ok = true;
        for (;;) {
            nsb.push_back(sb[j]); ncnt.push_back(cnt[j]); nhave.push_back(1);
            const Cnt &c = cnt[j];
            if (c.status == INF_FINISHED) break;                   // the member's last block ended inside this job
            if (c.status != INF_CHUNK_END) return 0;               // an error on the chain is a real error of the stream
            uint32_t m = j + 1;
            while (m < sb.size() && sb[m] < c.end_bit) m++;        // starts the real decode ran over: false candidates
            if (m < sb.size() && sb[m] == c.end_bit) { j = m; continue; }
            // nobody starts where this job ended: a new job starts there (its stop is the next candidate), counted next round.
            // The job that ended there ran with a different stop before, but a decode that stops at the first block boundary
            // >= stop also stops there for any stop in (previous boundary, end_bit] — its count stays valid.
            ok = false;
            nsb.push_back(c.end_bit); ncnt.push_back(Cnt{}); nhave.push_back(0);
            for (; m < sb.size(); m++) { nsb.push_back(sb[m]); ncnt.push_back(cnt[m]); nhave.push_back(have[m]); }
            break;
        }
        if (!ok) {
            // jobs in front of a changed stop must be recounted unless they ended on the chain already (those are kept above)
            sb.swap(nsb); cnt.swap(ncnt); have.swap(nhave);
            continue;
        }
        sb.swap(nsb); cnt.swap(ncnt); have.swap(nhave);
    }
    if (!ok) return 0;
    const uint32_t nj = (uint32_t)sb.size();
    if (cnt[nj - 1].status != INF_FINISHED) return 0;              // truncated member: NEED_INPUT semantics belong to the sequential path
    std::vector<uint64_t> ooff(nj + 1);
    uint64_t total = 0;
    for (uint32_t j = 0; j < nj; j++) { ooff[j] = total; total += cnt[j].out; }
    ooff[nj] = total;
    if (total > s.out_cap) return 0;                               // SZL_E_OUTPUT_TOO_SMALL with the bytes that fit: sequential path
    uint64_t end_byte = (cnt[nj - 1].end_bit + 7) >> 3;
    uint32_t adler_read = 0;
    if (zlib) {
        if (end_byte + 4 > s.in_len) return 0;
        HIPCHK(hipMemcpyAsync(E.pin, in + end_byte, 4, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
        adler_read = ((uint32_t)E.pin[0] << 24) | ((uint32_t)E.pin[1] << 16) | ((uint32_t)E.pin[2] << 8) | E.pin[3];
        end_byte += 4;
    }
    // ---- symbol pass, windows, bytes
    if ((rc = E.inf_sym.ensure(total * 2 + 64)) || (rc = E.inf_wins.ensure((uint64_t)(nj + 1) * 32768)) ||
        (rc = E.inf_misc.ensure((uint64_t)(nj + 1) * 8))) return rc;
    std::vector<uint32_t> all(nj);
    for (uint32_t j = 0; j < nj; j++) all[j] = j;
    if ((rc = run_pass(2, all, (uint16_t *)E.inf_sym.p, &ooff))) return rc;
    for (uint32_t j = 0; j < nj; j++)
        if (jobs[j].end_bit != cnt[j].end_bit || jobs[j].out_written != cnt[j].out) { set_error("parallel inflate: pass 2 disagrees with pass 1 at job %u", j); return SZL_E_STATE; }
    HIPCHK(hipMemcpyAsync(E.inf_misc.p, ooff.data(), (uint64_t)(nj + 1) * 8, hipMemcpyHostToDevice, st));
    launch_resolve_wins((const uint16_t *)E.inf_sym.p, (const uint64_t *)E.inf_misc.p, nj, (uint8_t *)E.inf_wins.p, st);
    launch_convert((const uint16_t *)E.inf_sym.p, (const uint64_t *)E.inf_misc.p, nj, (const uint8_t *)E.inf_wins.p, d_out + s.out_off, total, st);
    HIPCHK(hipStreamSynchronize(st));
    res->out_written = total; res->consumed = end_byte; res->adler_read = adler_read;
    E.last_par_jobs = nj;
    return 1;
}

extern "C" {

int szl_inflate_batch_device(szl_engine *e, const void *d_in, void *d_out, szl_stream *streams, size_t n_all, unsigned flags, void *hip_stream) {
    if (!e || (!streams && n_all)) return SZL_E_ARG;
    if (n_all == 0) return 0;
    if (n_all > 0x7FFFFFFFull) return SZL_E_ARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const bool nowrap = flags & SZL_F_NOWRAP;
    const unsigned want = ((flags & SZL_F_CRC32) ? 1u : 0u) | (((flags & SZL_F_ADLER32) || !nowrap) ? 2u : 0u);
    int rc;
    for (int i = 0; i < 2; i++) if (!e->e.ev[i]) (void)hipEventCreate(&e->e.ev[i]);
    if (!e->e.pin && hipHostMalloc((void **)&e->e.pin, 256) != hipSuccess) { set_error("hipHostMalloc failed"); return SZL_E_NOMEM; }
    e->e.timing.inflate_ms = 0;
    e->e.last_par_jobs = 0;

    // long members first: each is decoded by many wavefronts (inflate_member_parallel); whatever it does not take joins the batch
    const uint64_t par_min = (uint64_t)std::max(64, knob("SZL_INF_PAR_MIN_KIB", 2048)) * 1024;
    std::vector<size_t> idx;             // streams for the one-wavefront-per-stream decoder
    std::vector<char> par_done(n_all, 0);
    std::vector<ParResult> par_res(n_all);
    // (each long member costs a handful of host round trips: with many of them in one call the batch already fills the device
    // with one wavefront per stream, and that form has no per-stream host work at all)
    size_t n_long = 0;
    for (size_t i = 0; i < n_all; i++) n_long += streams[i].in_len >= par_min ? 1 : 0;
    const bool use_par = n_long > 0 && n_long <= (size_t)std::max(1, knob("SZL_INF_PAR_MAX_STREAMS", 32));
    for (size_t i = 0; i < n_all; i++) {
        if (use_par && streams[i].in_len >= par_min) {
            (void)hipEventRecord(e->e.ev[0], st);
            rc = inflate_member_parallel(e->e, (const uint8_t *)d_in, (uint8_t *)d_out, streams[i], !nowrap, st, &par_res[i]);
            if (rc < 0) return rc;
            if (rc == 1) {
                (void)hipEventRecord(e->e.ev[1], st); (void)hipEventSynchronize(e->e.ev[1]);
                float ms = 0; (void)hipEventElapsedTime(&ms, e->e.ev[0], e->e.ev[1]); e->e.timing.inflate_ms += ms;
                par_done[i] = 1; continue;
            }
        }
        idx.push_back(i);
    }
    const size_t n = idx.size();
    std::vector<InfJob> jobs(n);
    std::vector<InfState> states(n);
    for (size_t k = 0; k < n; k++) {
        const szl_stream &s = streams[idx[k]];
        InfJob j{};
        j.in_off = s.in_off; j.in_len = s.in_len; j.out_off = s.out_off; j.out_cap = s.out_cap;
        j.window = nullptr; j.zlib = nowrap ? 0 : 1; j.keep_window = 0;
        jobs[k] = j;
        InfState is{};
        is.mode = nowrap ? INF_M_HEADER : INF_M_ZHEADER;
        states[k] = is;
    }
    if (n) {
        DevBuf djobs, dstates;
        if ((rc = djobs.ensure(n * sizeof(InfJob))) || (rc = dstates.ensure(n * sizeof(InfState)))) { djobs.release(); dstates.release(); return rc; }
        auto cleanup = [&]() { djobs.release(); dstates.release(); };
        if (hipMemcpyAsync(djobs.p, jobs.data(), n * sizeof(InfJob), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(dstates.p, states.data(), n * sizeof(InfState), hipMemcpyHostToDevice, st) != hipSuccess) { cleanup(); set_error("H2D failed"); return SZL_E_DEVICE; }
        (void)hipEventRecord(e->e.ev[0], st);
        launch_inflate((const uint8_t *)d_in, (uint8_t *)d_out, (InfJob *)djobs.p, (InfState *)dstates.p, (uint32_t)n, true, st);
        (void)hipEventRecord(e->e.ev[1], st);
        if (hipMemcpyAsync(jobs.data(), djobs.p, n * sizeof(InfJob), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(states.data(), dstates.p, n * sizeof(InfState), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { cleanup(); set_error("inflate kernel/D2H failed: %s", hipGetErrorString(hipGetLastError())); return SZL_E_DEVICE; }
        cleanup();
        float ms = 0; (void)hipEventElapsedTime(&ms, e->e.ev[0], e->e.ev[1]); e->e.timing.inflate_ms += ms;
    }
    std::vector<std::pair<uint32_t, uint32_t>> cks;
    if (want) {
        std::vector<std::pair<uint64_t, uint64_t>> regs(n_all);
        size_t k = 0;
        for (size_t i = 0; i < n_all; i++) {
            if (par_done[i]) regs[i] = {streams[i].out_off, par_res[i].out_written};
            else { regs[i] = {jobs[k].out_off, jobs[k].out_written}; k++; }
        }
        if ((rc = region_checksums((const uint8_t *)d_out, regs, want, cks, nullptr, st))) return rc;
    }
    size_t k = 0;
    for (size_t i = 0; i < n_all; i++) {
        szl_stream &s = streams[i];
        s.reserved = 0;
        s.crc32 = want ? cks[i].first : 0; s.adler32 = want ? cks[i].second : 1;
        if (par_done[i]) {
            s.out_len = par_res[i].out_written; s.in_consumed = par_res[i].consumed;
            s.status = (!nowrap && par_res[i].adler_read != cks[i].second) ? SZL_E_ADLER_MISMATCH : 0;
            continue;
        }
        s.out_len = jobs[k].out_written;
        s.in_consumed = jobs[k].consumed;
        int stt = jobs[k].status;
        if (stt == INF_FINISHED) {
            s.status = 0;
            if (!nowrap && states[k].adler_read != cks[i].second) s.status = SZL_E_ADLER_MISMATCH; // C/Inflater.cs:411-414
        } else if (stt == INF_NEED_INPUT) s.status = SZL_E_UNEXPECTED_EOF;
        else if (stt == INF_OUTPUT_FULL) s.status = SZL_E_OUTPUT_TOO_SMALL;
        else if (stt == INF_NEED_DICT) s.status = SZL_E_UNSUPPORTED; // batch call cannot supply a preset dictionary
        else s.status = stt < 0 ? stt : SZL_E_STATE;
        k++;
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
