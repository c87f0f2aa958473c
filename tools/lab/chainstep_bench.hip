// chainstep_bench — what ONE chain step of FindLongestMatch costs on gfx950 under the three data layouts round 4's verdict named for stage B
// (C/DeflaterEngine.cs:474-612: a step = prev[] hop + the two filter bytes scan_end1 / scan_end of the candidate):
//   lds3    today's k_match9: u16 link and two u8 filter bytes, three LDS reads per step (3 B of LDS per window position)
//   packed  form (b): link + the two filter bytes of best_len = 2 in ONE aligned dword, one LDS read per step (4 B per position:
//           a tile of ~7.5 Ki positions behind the 32512-byte history instead of 21.5 Ki)
//   l2link  form (a): bytes in LDS, the link gathered from global memory (L2-resident: the window's links are 2 x 54 KiB per workgroup),
//           at 16 and at 32 wavefronts per CU (1 B of LDS per position leaves room for two workgroups)
// Every lane walks two independent chains (k_match9's two contexts) of random hops through a 54016-position window, as the engine's walkers do;
// hops and bytes are random, so bank conflicts and cache-line divergence are those of the real walk.  Output: ns per step per CU-resident
// wavefront and steps per second of the whole device.  A measurement tool (tools/README.md), not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o chainstep_bench tools/lab/chainstep_bench.hip && ./chainstep_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int W = 54016;            // window positions of a k_match9 tile (32512 history + 21504 tile)
constexpr int WP = 40944;           // form (b): what 160 KiB hold at 4 B per position (32512 history + 8432 tile)
constexpr int STEPS = 2048;

__device__ __forceinline__ uint32_t lds_u8(uint32_t a) { uint32_t v; asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ uint32_t lds_u16(uint32_t a) { uint32_t v; asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ uint32_t lds_b32(uint32_t a) { uint32_t v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }

// MODE 0 lds3, 1 packed, 2 l2link
template <int MODE>
__global__ __launch_bounds__(1024) void k_steps(const uint16_t *g_link, const uint8_t *g_byte, uint32_t *sink, int nwin) {
    extern __shared__ uint8_t lds[];
    const int tid = threadIdx.x;
    constexpr int W = MODE == 1 ? WP : ::W;
    const uint16_t *lk = g_link + (size_t)(blockIdx.x % nwin) * W;
    const uint8_t *by = g_byte + (size_t)(blockIdx.x % nwin) * W;
    // stage the window
    if (MODE == 0) {
        for (int i = tid; i < W; i += 1024) { lds[i] = by[i]; ((uint16_t *)(lds + 54336))[i] = lk[i]; }
    } else if (MODE == 1) {
        for (int i = tid; i < W; i += 1024) ((uint32_t *)lds)[i] = (uint32_t)lk[i] | ((uint32_t)by[i + 1 < W ? i + 1 : i] << 16) | ((uint32_t)by[i + 2 < W ? i + 2 : i] << 24);
    } else {
        for (int i = tid; i < W; i += 1024) lds[i] = by[i];
    }
    __syncthreads();
    uint32_t xa = W - 1 - (uint32_t)((tid * 37u) % 8000u), xb = W - 1 - (uint32_t)((tid * 53u + 11u) % 8000u), acc = 0;
    for (int s = 0; s < STEPS; s++) {
        uint32_t ha, hb, fa, fb;
        if (MODE == 0) {
            ha = lds_u16(54336 + 2 * xa); hb = lds_u16(54336 + 2 * xb);
            fa = lds_u8(xa + 1) | (lds_u8(xa + 2) << 16); fb = lds_u8(xb + 1) | (lds_u8(xb + 2) << 16);
        } else if (MODE == 1) {
            const uint32_t va = lds_b32(4 * xa), vb = lds_b32(4 * xb);
            ha = va & 0xFFFF; hb = vb & 0xFFFF; fa = va >> 16; fb = vb >> 16;
        } else {
            ha = __builtin_nontemporal_load(lk + xa); hb = __builtin_nontemporal_load(lk + xb);
            fa = lds_u8(xa + 1) | (lds_u8(xa + 2) << 16); fb = lds_u8(xb + 1) | (lds_u8(xb + 2) << 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
        acc += (fa == 0x00410042u) + (fb == 0x00410042u);
        xa = xa > ha + 64 ? xa - ha : W - 1 - ((xa * 2654435761u) >> 19);     // the chain ends: a new walk starts near the tile
        xb = xb > hb + 64 ? xb - hb : W - 1 - ((xb * 2246822519u) >> 19);
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <int MODE>
static void run(const char *name, const uint16_t *d_link, const uint8_t *d_byte, uint32_t *d_sink, int nwin, int cus, int wg_per_cu, size_t lds_bytes) {
    CK(hipFuncSetAttribute((const void *)k_steps<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const int grid = cus * wg_per_cu;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_steps<MODE>, dim3(grid), dim3(1024), lds_bytes, 0, d_link, d_byte, d_sink, nwin);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_steps<MODE>, dim3(grid), dim3(1024), lds_bytes, 0, d_link, d_byte, d_sink, nwin);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double steps = (double)grid * 1024.0 * 2.0 * STEPS;              // lane-steps
    printf("%-34s %d workgroup(s) per CU, %6zu B LDS each | %8.3f ms | %7.2f G lane-steps/s on the device | %6.2f ns per wave-step per CU\n",
           name, wg_per_cu, lds_bytes, best, steps / (best * 1e-3) / 1e9, best * 1e6 / ((double)wg_per_cu * 16.0 * 2.0 * STEPS));
}

int main() {
    int dev = 0, cus = 256;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int nwin = 64;
    std::vector<uint16_t> link((size_t)nwin * W);
    std::vector<uint8_t> byte((size_t)nwin * W);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (size_t i = 0; i < link.size(); i++) {
        const uint32_t r = rnd();
        // hop distances as the walks of text see them: mostly hundreds to thousands of positions, a tail to 32506
        link[i] = (uint16_t)((r & 7) ? 16 + (rnd() % 4000) : 16 + (rnd() % 32000));
        byte[i] = (uint8_t)(rnd() % 96 + 32);
    }
    uint16_t *d_link; uint8_t *d_byte; uint32_t *d_sink;
    CK(hipMalloc(&d_link, link.size() * 2)); CK(hipMalloc(&d_byte, byte.size())); CK(hipMalloc(&d_sink, 64));
    CK(hipMemcpy(d_link, link.data(), link.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_byte, byte.data(), byte.size(), hipMemcpyHostToDevice));
    printf("chain steps of two contexts per lane, %d dependent steps per context, %d CUs\n", STEPS, cus);
    run<0>("lds3   (k_match9 today)", d_link, d_byte, d_sink, nwin, cus, 1, 54336 + 2 * W + 64);
    run<1>("packed (form b: one dword)", d_link, d_byte, d_sink, nwin, cus, 1, 4 * (size_t)WP + 64);
    run<2>("l2link (form a: links from L2)", d_link, d_byte, d_sink, nwin, cus, 1, (size_t)W + 64);
    run<2>("l2link (form a: links from L2)", d_link, d_byte, d_sink, nwin, cus, 2, (size_t)W + 64);
    return 0;
}
