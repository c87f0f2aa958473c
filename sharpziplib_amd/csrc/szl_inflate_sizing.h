// szl_inflate_sizing.h — chunk sizes of the chunk-parallel Inflater (inflate_members_parallel): pure arithmetic, shared by the
// product and tests/sizing_harness.cpp.
//
// The symbol pass is a fixed number of wavefront SLOTS (jobs a CU holds x CUs) times the one-wavefront decode of a chunk, so its time
// is ceil(jobs / slots) rounds of one chunk's decode: what counts is that the jobs fill whole rounds (round 3: a 1 GiB text member in
// 128 KiB chunks = 3034 jobs = 1.5 rounds of 2048 slots, 56 ms; 185 KiB = 2044 jobs = one round, 46 ms).  inflate_chunk_max does that for
// the call as a whole.  (Round 4 also built a `trim_tail` that grew every member's chunks by a few percent when the jobs exceeded whole
// rounds by a few stragglers; measured in round 5 — 64 x 1 MiB members 18.19 vs 18.12 ms, one member 28.40 vs 28.41 — it moved nothing and
// is gone, profiles/r05/r5_ab.log.)
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
inline std::vector<ChunkPlan> inflate_chunk_plans(const std::vector<uint64_t> &in_len, uint64_t chunk_max) {
    std::vector<ChunkPlan> plan(in_len.size());
    for (size_t i = 0; i < in_len.size(); i++) {
        uint64_t cb = in_len[i] / 32;                             // short members get smaller chunks: at least ~32 of them
        cb = std::min<uint64_t>(std::max<uint64_t>(cb & ~1023ull, 16384), chunk_max);
        plan[i] = inflate_chunk_plan_one(in_len[i], cb);
    }
    return plan;
}

}  // namespace szl
