// szl_inflate_sizing.h — chunk sizes of the chunk-parallel Inflater (inflate_members_parallel): pure arithmetic, shared by the
// product and tests/sizing_harness.cpp.
//
// The symbol pass is a fixed number of wavefront SLOTS (jobs a CU holds x CUs) times the one-wavefront decode of a chunk, so its time
// is ceil(jobs / slots) rounds of one chunk's decode: what counts is that the jobs fill whole rounds (round 3: a 1 GiB text member in
// 128 KiB chunks = 3034 jobs = 1.5 rounds of 2048 slots, 56 ms; 185 KiB = 2044 jobs = one round, 46 ms).  inflate_chunk_max does that for
// the call as a whole; with many members the per-member round-up of the chunk count can still leave a few stragglers for a last round.
// `trim_tail` (SZL_INF_TRIM_TAIL=1, not the default: unmeasured) removes such a tail: when the jobs exceed a whole number of rounds by
// less than an eighth of a round, every member's chunks grow by the few percent that bring the count down to it (never beyond 256 KiB,
// never below 8 chunks per member).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace szl {

struct ChunkPlan { uint64_t chunk_bytes = 0; uint32_t nchunks = 0; };   // chunk_bytes 0: the member is not decoded in chunks

inline uint64_t inflate_chunk_max(uint64_t total_in, uint64_t slots) {   // r whole rounds of the slots with chunks of at most ~192 KiB, at least 32 KiB
    const uint64_t rounds = std::max<uint64_t>(1, (total_in + slots * (192ull << 10) - 1) / (slots * (192ull << 10)));
    return std::min<uint64_t>(std::max<uint64_t>((total_in / (slots * rounds) + 1023) & ~1023ull, 32ull << 10), 256ull << 10);
}

inline ChunkPlan inflate_chunk_plan_one(uint64_t in_len, uint64_t cb) {
    ChunkPlan p;
    if (in_len < 8 * cb || in_len >= (1ull << 60)) return p;
    p.chunk_bytes = cb;
    p.nchunks = (uint32_t)std::min<uint64_t>((in_len + cb - 1) / cb, 1u << 20);
    return p;
}

// in_len of every candidate member -> its plan.  chunk_max: the cap on a chunk (from inflate_chunk_max, or the SZL_INF_CHUNK_KIB knob).
inline std::vector<ChunkPlan> inflate_chunk_plans(const std::vector<uint64_t> &in_len, uint64_t chunk_max, uint64_t slots, bool trim_tail) {
    std::vector<ChunkPlan> plan(in_len.size());
    uint64_t jobs = 0;
    for (size_t i = 0; i < in_len.size(); i++) {
        uint64_t cb = in_len[i] / 32;                             // short members get smaller chunks: at least ~32 of them
        cb = std::min<uint64_t>(std::max<uint64_t>(cb & ~1023ull, 16384), chunk_max);
        plan[i] = inflate_chunk_plan_one(in_len[i], cb);
        jobs += plan[i].nchunks;
    }
    if (!trim_tail || slots == 0 || jobs <= slots) return plan;
    const uint64_t whole = jobs / slots * slots, tail = jobs - whole;
    if (tail == 0 || tail * 8 > slots) return plan;
    for (int pct = 1; pct <= 20; pct++) {                          // the smallest growth (in percent steps above the exact ratio) that fits
        const double scale = (double)jobs / (double)whole * (1.0 + 0.01 * (pct - 1));
        std::vector<ChunkPlan> q(plan);
        uint64_t j2 = 0;
        for (size_t i = 0; i < in_len.size(); i++) {
            if (!plan[i].chunk_bytes) continue;
            uint64_t cb = ((uint64_t)((double)plan[i].chunk_bytes * scale) + 1023) & ~1023ull;
            cb = std::min<uint64_t>(cb, 256ull << 10);
            if (in_len[i] < 8 * cb) cb = plan[i].chunk_bytes;     // (a member keeps its 8 chunks)
            q[i] = inflate_chunk_plan_one(in_len[i], cb);
            j2 += q[i].nchunks;
        }
        if (j2 <= whole) return q;
    }
    return plan;
}

}  // namespace szl
