// szl_kernels_checksum.hip — CRC-32 and Adler-32 of each segment's bytes on the device.
// Restates K/Crc32.cs (poly 0xEDB88320 reflected, init/xor 0xFFFFFFFF, :50,:138) and K/Adler32.cs:134-161.
// Both checksums are sums over GF(2)/Z_65521 of per-byte terms, so chunks are computed independently
// and folded: CRC with the x^(8·len) mod P operator, Adler with its closed form.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "szl_internal.h"

namespace szl {

enum : uint32_t { CRC_POLY = 0xEDB88320u };
enum : int { CK_CHUNK = 4096, CK_THREADS = 256 };

__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b) { // a(x)*b(x) mod P, reflected bit order
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
__device__ uint32_t x2nmodp(uint64_t n, unsigned k) { // x^(n * 2^k) mod P
    uint32_t p = 1u << 31;
    uint32_t sq = 1u << 30; // x^1
    for (unsigned i = 0; i < k; i++) sq = multmodp(sq, sq);
    while (n) {
        if (n & 1) p = multmodp(sq, p);
        sq = multmodp(sq, sq);
        n >>= 1;
    }
    return p;
}

struct CkPartial { uint32_t crc; uint32_t a; uint32_t b; uint32_t len; }; // crc of the chunk (standard), adler raw sums

// grid: one thread per 4 KiB chunk of a segment's [seg_start, seg_end); chunk table built arithmetically
__global__ __launch_bounds__(CK_THREADS) void k_ck_partial(const uint8_t *__restrict__ in, const SegDev *__restrict__ segs,
                                                           uint32_t nseg, const uint64_t *__restrict__ chunk_off,
                                                           uint64_t nchunks, CkPartial *parts, unsigned want) {
    __shared__ uint32_t tab[4][256];
    for (int i = threadIdx.x; i < 256; i += CK_THREADS) {
        uint32_t r = (uint32_t)i;
        for (int k = 0; k < 8; k++) r = (r & 1) ? CRC_POLY ^ (r >> 1) : r >> 1;
        tab[0][i] = r;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += CK_THREADS) {
        uint32_t r = tab[0][i];
        for (int j = 1; j < 4; j++) { r = tab[0][r & 0xFF] ^ (r >> 8); tab[j][i] = r; }
    }
    __syncthreads();
    uint64_t c = (uint64_t)blockIdx.x * CK_THREADS + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t lo = 0, hi = nseg - 1;
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (chunk_off[mid] <= c) lo = mid; else hi = mid - 1; }
    const SegDev s = segs[lo];
    const int64_t start = s.seg_start + (int64_t)(c - chunk_off[lo]) * CK_CHUNK;
    const int64_t end = start + CK_CHUNK < s.seg_end ? start + CK_CHUNK : s.seg_end;
    const uint8_t *p = in + s.buf_off + start;
    const int len = (int)(end - start);
    uint32_t crc = 0xFFFFFFFFu;
    uint32_t a = 0, b = 0; // a = sum d_i ; b = sum (len - i) d_i   (both < 2^32: 4096*255*4096 < 2^32)
    int i = 0;
    // 16 bytes per load when the chunk start is 16-byte aligned (a lane strides 4 KiB: fewer, wider requests per cache line)
    if ((((uintptr_t)p) & 15) == 0) {
        for (; i + 16 <= len; i += 16) {
            const uint4 v = *(const uint4 *)(p + i);
            const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t w = ws[q];
                const int ii = i + 4 * q;
                if (want & 1) {
                    crc ^= w;
                    crc = tab[3][crc & 0xFF] ^ tab[2][(crc >> 8) & 0xFF] ^ tab[1][(crc >> 16) & 0xFF] ^ tab[0][crc >> 24];
                }
                if (want & 2) {
                    uint32_t d0 = w & 0xFF, d1 = (w >> 8) & 0xFF, d2 = (w >> 16) & 0xFF, d3 = w >> 24;
                    a += d0 + d1 + d2 + d3;
                    b += (uint32_t)(len - ii) * d0 + (uint32_t)(len - ii - 1) * d1 + (uint32_t)(len - ii - 2) * d2 + (uint32_t)(len - ii - 3) * d3;
                }
            }
        }
    }
    for (; i + 4 <= len; i += 4) {
        uint32_t w;
        __builtin_memcpy(&w, p + i, 4);
        if (want & 1) {
            crc ^= w;
            crc = tab[3][crc & 0xFF] ^ tab[2][(crc >> 8) & 0xFF] ^ tab[1][(crc >> 16) & 0xFF] ^ tab[0][crc >> 24];
        }
        if (want & 2) {
            uint32_t d0 = w & 0xFF, d1 = (w >> 8) & 0xFF, d2 = (w >> 16) & 0xFF, d3 = w >> 24;
            a += d0 + d1 + d2 + d3;
            b += (uint32_t)(len - i) * d0 + (uint32_t)(len - i - 1) * d1 + (uint32_t)(len - i - 2) * d2 + (uint32_t)(len - i - 3) * d3;
        }
    }
    for (; i < len; i++) {
        uint32_t d = p[i];
        if (want & 1) crc = tab[0][(crc ^ d) & 0xFF] ^ (crc >> 8);
        if (want & 2) { a += d; b += (uint32_t)(len - i) * d; }
    }
    CkPartial r;
    r.crc = ~crc; r.a = a % 65521u; r.b = b % 65521u; r.len = (uint32_t)len;
    parts[c] = r;
}

// One workgroup per segment folds its chunk partials.  CRC: crc(A||B) = multmodp(x^(8|B|), crc(A)) ^ crc(B)
// (the standard combine identity on finalised CRCs); all full chunks share one operator, so a log-tree over
// power-of-two groups uses operators obtained by repeated squaring.
__global__ __launch_bounds__(CK_THREADS) void k_ck_fold(const SegDev *__restrict__ segs, uint32_t nseg,
                                                        const uint64_t *__restrict__ chunk_off, const CkPartial *parts,
                                                        SegOut *so, unsigned want) {
    __shared__ uint32_t s_crc[CK_THREADS];
    __shared__ uint64_t s_len[CK_THREADS];
    __shared__ unsigned long long s_s1, s_s2;
    __shared__ uint32_t s_mt[4][256];
    uint32_t si = blockIdx.x;
    if (si >= nseg) return;
    const SegDev s = segs[si];
    const uint64_t c0 = chunk_off[si], nc = chunk_off[si + 1] - c0;
    const uint64_t n = (uint64_t)(s.seg_end - s.seg_start);
    const int tid = threadIdx.x;
    if (tid == 0) { s_s1 = 0; s_s2 = 0; }
    __syncthreads();
    // ---- Adler: s1 = s1_0 + sum a_j ; s2 = s2_0 + n*s1_0 + sum ((n - o_j - l_j) a_j + b_j)   (mod 65521)
    if (want & 2) {
        unsigned long long t1 = 0, t2 = 0;
        for (uint64_t j = tid; j < nc; j += CK_THREADS) {
            CkPartial p = parts[c0 + j];
            uint64_t after = n - j * CK_CHUNK - p.len;
            t1 += p.a;
            t2 += (after % 65521ull) * p.a % 65521ull + p.b;
        }
        t1 %= 65521ull; t2 %= 65521ull;
        atomicAdd(&s_s1, t1);
        atomicAdd(&s_s2, t2);
    }
    // ---- CRC: each thread folds a contiguous run of chunks sequentially, then threads are folded in order
    uint32_t mycrc = 0; uint64_t mylen = 0;
    if (want & 1) {
        uint64_t per = (nc + CK_THREADS - 1) / CK_THREADS;
        uint64_t j0 = (uint64_t)tid * per, j1 = j0 + per < nc ? j0 + per : nc;
        // every full chunk is folded with the same operator, x^(8 * CK_CHUNK): the product with a CONSTANT is linear in the other factor,
        // so it is four table look-ups (the operator times every byte value at every byte position) instead of multmodp's 32 rounds — the
        // fold of a 1 GiB segment, 1024 chunks to a thread, took 3.5 ms with the whole device idle but for this one workgroup
        const uint32_t opfull = x2nmodp(CK_CHUNK, 3);
        for (int i = tid; i < 1024; i += CK_THREADS) s_mt[i >> 8][i & 255] = multmodp(opfull, (uint32_t)(i & 255) << (8 * (i >> 8)));
        __syncthreads();
        for (uint64_t j = j0; j < j1; j++) {
            CkPartial p = parts[c0 + j];
            if (!mylen) mycrc = p.crc;
            else if (p.len == CK_CHUNK) mycrc = s_mt[0][mycrc & 0xFF] ^ s_mt[1][(mycrc >> 8) & 0xFF] ^ s_mt[2][(mycrc >> 16) & 0xFF] ^ s_mt[3][mycrc >> 24] ^ p.crc;
            else mycrc = multmodp(x2nmodp(p.len, 3), mycrc) ^ p.crc;
            mylen += p.len;
        }
    }
    // (crc, x^(8*len)) pairs combine associatively: crc = op_B * crc_A ^ crc_B ; op = op_A * op_B.  Log-tree over threads.
    __shared__ uint32_t s_op[CK_THREADS];
    s_crc[tid] = mycrc; s_len[tid] = mylen;
    s_op[tid] = (want & 1) ? x2nmodp(mylen, 3) : (1u << 31);
    __syncthreads();
    if (want & 1) {
        for (int stride = 1; stride < CK_THREADS; stride <<= 1) {
            if ((tid & (2 * stride - 1)) == 0) {
                const int r = tid + stride;
                const uint32_t opB = s_op[r];
                s_crc[tid] = multmodp(opB, s_crc[tid]) ^ s_crc[r];
                s_op[tid] = multmodp(s_op[tid], opB);
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        uint32_t adler = s.adler_init;
        if (want & 2) {
            unsigned long long s1_0 = adler & 0xFFFF, s2_0 = adler >> 16;
            unsigned long long s1 = (s1_0 + s_s1) % 65521ull;
            unsigned long long s2 = (s2_0 + (n % 65521ull) * s1_0 + s_s2) % 65521ull;
            adler = (uint32_t)((s2 << 16) | s1);
        }
        uint32_t crc = s.crc_init;
        if (want & 1) crc = (crc ? multmodp(s_op[0], crc) : 0u) ^ s_crc[0];
        so[si].crc32 = crc;
        so[si].adler32 = adler;
    }
}

void launch_checksums(const uint8_t *in, const SegDev *segs, uint32_t nseg, const uint64_t *chunk_off, uint64_t nchunks,
                      void *parts, SegOut *so, unsigned want, hipStream_t st) {
    if (nchunks)
        hipLaunchKernelGGL(k_ck_partial, dim3((unsigned)((nchunks + CK_THREADS - 1) / CK_THREADS)), dim3(CK_THREADS), 0, st, in, segs,
                           nseg, chunk_off, nchunks, (CkPartial *)parts, want);
    hipLaunchKernelGGL(k_ck_fold, dim3(nseg), dim3(CK_THREADS), 0, st, segs, nseg, chunk_off, (const CkPartial *)parts, so, want);
}
size_t checksum_partial_bytes() { return sizeof(CkPartial); }

} // namespace szl
