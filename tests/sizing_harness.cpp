// CPU harness for csrc/szl_inflate_sizing.h (tests/test_inflate_sizing.py): the product's own chunk-sizing arithmetic, compiled with g++.
#include "../sharpziplib_amd/csrc/szl_inflate_sizing.h"

extern "C" {
unsigned long long sz_chunk_max(unsigned long long total_in, unsigned long long slots) { return szl::inflate_chunk_max(total_in, slots); }
unsigned long long sz_chunk_max_members(unsigned long long total_in, unsigned long long slots, unsigned long long members) { return szl::inflate_chunk_max(total_in, slots, members); }
unsigned long long sz_min_chunks(unsigned long long chunk_max) { return szl::inflate_min_chunks(chunk_max); }
// the call as the library plans it since round 6 (szl_api_inflate.hip: chunk_max from the call's bytes AND members, the least number of chunks from chunk_max)
unsigned long long sz_plans_auto(const unsigned long long *in_len, int n, unsigned long long slots, unsigned long long *out_cb, unsigned *out_n, unsigned long long *out_cm) {
    std::vector<uint64_t> v(in_len, in_len + n);
    uint64_t total = 0;
    for (uint64_t x : v) total += x;
    const uint64_t cm = szl::inflate_chunk_max(total, slots, v.size());
    const std::vector<szl::ChunkPlan> p = szl::inflate_chunk_plans(v, cm, szl::inflate_min_chunks(cm));
    unsigned long long jobs = 0;
    for (int i = 0; i < n; i++) { out_cb[i] = p[i].chunk_bytes; out_n[i] = p[i].nchunks; jobs += p[i].nchunks; }
    *out_cm = cm;
    return jobs;
}
// plans for n members; out_cb / out_n receive chunk bytes and chunk counts; returns the total number of jobs
unsigned long long sz_plans(const unsigned long long *in_len, int n, unsigned long long chunk_max, unsigned long long *out_cb, unsigned *out_n) {
    std::vector<uint64_t> v(in_len, in_len + n);
    const std::vector<szl::ChunkPlan> p = szl::inflate_chunk_plans(v, chunk_max);
    unsigned long long jobs = 0;
    for (int i = 0; i < n; i++) { out_cb[i] = p[i].chunk_bytes; out_n[i] = p[i].nchunks; jobs += p[i].nchunks; }
    return jobs;
}
}
