// CPU harness for csrc/szl_inflate_sizing.h (tests/test_inflate_sizing.py): the product's own chunk-sizing arithmetic, compiled with g++.
#include "../sharpziplib_amd/csrc/szl_inflate_sizing.h"

extern "C" {
unsigned long long sz_chunk_max(unsigned long long total_in, unsigned long long slots) { return szl::inflate_chunk_max(total_in, slots); }
// plans for n members; out_cb / out_n receive chunk bytes and chunk counts; returns the total number of jobs
unsigned long long sz_plans(const unsigned long long *in_len, int n, unsigned long long chunk_max, unsigned long long *out_cb, unsigned *out_n) {
    std::vector<uint64_t> v(in_len, in_len + n);
    const std::vector<szl::ChunkPlan> p = szl::inflate_chunk_plans(v, chunk_max);
    unsigned long long jobs = 0;
    for (int i = 0; i < n; i++) { out_cb[i] = p[i].chunk_bytes; out_n[i] = p[i].nchunks; jobs += p[i].nchunks; }
    return jobs;
}
}
