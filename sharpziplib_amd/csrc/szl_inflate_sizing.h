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

inline uint64_t inflate_chunk_max(uint64_t total_in, uint64_t slots, uint64_t members = 1) {   // r whole rounds of the slots with chunks of at most ~192 KiB, at least 16 KiB
    const uint64_t rounds = std::max<uint64_t>(1, (total_in + slots * (192ull << 10) - 1) / (slots * (192ull << 10)));
    uint64_t jobs = slots * rounds;
    if (members > 1 && members * 2 < jobs) jobs -= members;      // (every member's last chunk is a short one: a job more than its bytes ask for)
    // (the floor was 32 KiB through round 6's second third.  A block of a level-6 stream is ~20 KiB of compressed bytes, a job is at least a
    // block, and a call that leaves slots empty is as fast as its LONGEST job: ONE 32 MiB text member 8.4 -> 4.9 ms with chunks of 16 KiB,
    // 64 MiB 12.1 -> 8.4, 4 MiB 7.8 -> 5.7; logs 64 MiB 9.6 -> 5.4 — profiles/r06/inflate_one_chunk_floor.log)
    return std::min<uint64_t>(std::max<uint64_t>((total_in / jobs + 1023) & ~1023ull, 16ull << 10), 256ull << 10);
}

// How many chunks a short member is cut into at least.  32 while the call leaves slots empty (a few short members: more, smaller jobs);
// a call that fills the slots at chunk_max — chunk_max above 32 KiB — keeps chunk_max for every member that holds eight of
// them: round 6 measured 256 x 4 MiB members 40.2 -> 34.9 ms, 1024 of them 140 -> 129, 512 x 1 MiB 26.0 -> 23.4 that way (32 chunks
// a member there are three partial rounds of jobs a quarter the size; profiles/r06/inflate_min_chunks.log).
inline uint64_t inflate_min_chunks(uint64_t chunk_max) { return chunk_max > (32ull << 10) ? 8 : 32; }

inline ChunkPlan inflate_chunk_plan_one(uint64_t in_len, uint64_t cb) {
    ChunkPlan p;
    if (in_len < 8 * cb || in_len >= (1ull << 60)) return p;
    p.chunk_bytes = cb;
    p.nchunks = (uint32_t)std::min<uint64_t>((in_len + cb - 1) / cb, 1u << 20);
    return p;
}

// in_len of every candidate member -> its plan.  chunk_max: the cap on a chunk (from inflate_chunk_max, or the SZL_INF_CHUNK_KIB knob).
inline std::vector<ChunkPlan> inflate_chunk_plans(const std::vector<uint64_t> &in_len, uint64_t chunk_max, uint64_t min_chunks = 32) {
    std::vector<ChunkPlan> plan(in_len.size());
    for (size_t i = 0; i < in_len.size(); i++) {
        uint64_t cb = in_len[i] / std::max<uint64_t>(min_chunks, 8);   // short members get smaller chunks: at least ~32 of them
        cb = std::min<uint64_t>(std::max<uint64_t>(cb & ~1023ull, 16384), chunk_max);
        plan[i] = inflate_chunk_plan_one(in_len[i], cb);
    }
    return plan;
}

}  // namespace szl
