// fakehip — the subset of the HIP runtime API the product library (libszl_amd.so) uses, on the CPU, for tools/gfxsim.
//
// The product's object files are linked against THIS library instead of libamdhip64 (tools/gfxsim/build.py ->
// libszl_amd_sim.so), so the real host code of the engine runs unchanged in a container without a GPU:
//   * "device memory" is a bump allocator inside one big anonymous mapping (the arena); device pointers are real addresses,
//     hipMemcpy is memcpy;
//   * every stream is synchronous; a kernel launch calls back into Python (gfxsim.runtime interprets the kernel's gfx950
//     machine code on the arena) and returns when the kernel has finished;
//   * events carry the host clock.
// One device named gfx950 is reported.  Test infrastructure only — never loaded by the product.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace {
std::mutex g_mu;
unsigned char *g_arena = nullptr;
unsigned char *g_shadow = nullptr;                 // one byte per arena byte: 0 = never allocated, 1 = allocated, 2 = allocated and written (gfxsim memcheck)
size_t g_arena_size = 0, g_top = 4096;
std::map<const void *, std::string> g_kernels;     // host stub -> device (mangled) name
std::map<void *, size_t> g_allocs;
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local CallCfg t_cfg;
typedef int (*launch_cb_t)(const char *name, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, void **args, size_t shmem);
launch_cb_t g_cb = nullptr;
int g_ndev = 1;
int cus() { static const int n = getenv("FAKEHIP_CUS") ? atoi(getenv("FAKEHIP_CUS")) : 256; return n > 0 ? n : 256; }   // (a small device reproduces the library's sizing decisions at small sizes)
thread_local int t_dev = 0;
thread_local hipError_t t_last = hipSuccess;

void ensure_arena() {
    if (g_arena) return;
    const char *e = getenv("FAKEHIP_ARENA_GIB");
    size_t gib = e ? strtoull(e, nullptr, 10) : 64;
    g_arena_size = gib << 30;
    void *p = mmap(nullptr, g_arena_size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("fakehip: mmap"); abort(); }
    g_arena = (unsigned char *)p;
    void *q = mmap(nullptr, g_arena_size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (q == MAP_FAILED) { perror("fakehip: mmap (shadow)"); abort(); }
    g_shadow = (unsigned char *)q;
}
bool in_arena(const void *p) { return g_arena && (const unsigned char *)p >= g_arena && (const unsigned char *)p < g_arena + g_arena_size; }
void shadow_set(const void *p, unsigned char v, size_t n) { if (n && in_arena(p)) memset(g_shadow + ((const unsigned char *)p - g_arena), v, n); }
void shadow_copy(const void *d, const void *s, size_t n) {          // what a memcpy does to the shadow of its destination
    if (!n || !in_arena(d)) return;
    if (in_arena(s)) memmove(g_shadow + ((const unsigned char *)d - g_arena), g_shadow + ((const unsigned char *)s - g_arena), n);
    else shadow_set(d, 2, n);                                       // bytes from host memory are defined
}
void *arena_alloc(size_t n, size_t align = 256) {
    std::lock_guard<std::mutex> lk(g_mu);
    ensure_arena();
    size_t p = (g_top + align - 1) / align * align;
    if (p + n + 256 > g_arena_size) return nullptr;
    g_top = p + n + 256;                            // 256 bytes between allocations
    g_allocs[g_arena + p] = n;
    memset(g_shadow + p, 1, n);
    static const bool poison = getenv("GFXSIM_POISON") != nullptr;   // device memory is not zero after hipMalloc
    if (poison) memset(g_arena + p, 0xCD, n);
    return g_arena + p;
}
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct FakeEvent { double t; };
}  // namespace

extern "C" {
// ---- hooks for the Python side -----------------------------------------------------------------------------------
void fakehip_set_launch_callback(launch_cb_t cb) { g_cb = cb; }
void *fakehip_arena_base() { ensure_arena(); return g_arena; }
void *fakehip_shadow_base() { ensure_arena(); return g_shadow; }
size_t fakehip_arena_size() { ensure_arena(); return g_arena_size; }
size_t fakehip_arena_top() { return g_top; }
void *fakehip_alloc(size_t n) { return arena_alloc(n); }
void fakehip_set_device_count(int n) { g_ndev = n; }
size_t fakehip_alloc_size(void *p) { std::lock_guard<std::mutex> lk(g_mu); auto it = g_allocs.find(p); return it == g_allocs.end() ? 0 : it->second; }

// ---- registration (called by the module constructors hipcc emits) ------------------------------------------------
void **__hipRegisterFatBinary(const void *) { static void *handle[4]; return handle; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *hostFunction, char *, const char *deviceName, unsigned, void *, void *, void *, void *, int *) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_kernels[hostFunction] = deviceName;
}
void __hipRegisterVar(void **, void *, char *, char *, int, size_t, int, int) {}
void __hipRegisterManagedVar(void *, void **, void *, const char *, size_t, unsigned) {}
hipError_t __hipPushCallConfiguration(dim3 gridDim, dim3 blockDim, size_t sharedMem, hipStream_t stream) {
    t_cfg = {gridDim, blockDim, sharedMem, stream};
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *gridDim, dim3 *blockDim, size_t *sharedMem, hipStream_t *stream) {
    *gridDim = t_cfg.grid; *blockDim = t_cfg.block; *sharedMem = t_cfg.shmem; *stream = t_cfg.stream;
    return hipSuccess;
}

// ---- devices -----------------------------------------------------------------------------------------------------
hipError_t hipGetDeviceCount(int *n) { *n = g_ndev; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= g_ndev) return t_last = hipErrorInvalidDevice; t_dev = d; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int) {
    memset(p, 0, sizeof *p);
    strcpy(p->name, "gfxsim (CPU interpreter)");
    strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = cus(); p->warpSize = 64; p->totalGlobalMem = g_arena_size ? g_arena_size : (64ull << 30);
    p->sharedMemPerBlock = 65536; p->maxSharedMemoryPerMultiProcessor = 163840; p->maxThreadsPerBlock = 1024;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int) {
    switch (a) {
    case hipDeviceAttributeMultiprocessorCount: *v = cus(); break;
    case hipDeviceAttributeWarpSize: *v = 64; break;
    case hipDeviceAttributeMaxSharedMemoryPerBlock: *v = 65536; break;
    case hipDeviceAttributeMaxThreadsPerBlock: *v = 1024; break;
    default: *v = 0; break;
    }
    return hipSuccess;
}
hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { hipError_t e = t_last; t_last = hipSuccess; return e; }
hipError_t hipPeekAtLastError() { return t_last; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fakehip error"; }
const char *hipGetErrorName(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipError"; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

// ---- memory ------------------------------------------------------------------------------------------------------
// (live bytes by kind, for tests/test_memory_trim.py: what the library holds after its objects are gone)
static std::map<void *, size_t> g_live_dev, g_live_host;
static void live_add(std::map<void *, size_t> &m, void *p, size_t n) { if (p) { std::lock_guard<std::mutex> lk(g_mu); m[p] = n; } }
static void live_del(std::map<void *, size_t> &m, void *p) { std::lock_guard<std::mutex> lk(g_mu); m.erase(p); }
size_t fakehip_live_device_bytes() { std::lock_guard<std::mutex> lk(g_mu); size_t t = 0; for (auto &kv : g_live_dev) t += kv.second; return t; }
size_t fakehip_live_host_bytes() { std::lock_guard<std::mutex> lk(g_mu); size_t t = 0; for (auto &kv : g_live_host) t += kv.second; return t; }
hipError_t hipMalloc(void **p, size_t n) { *p = arena_alloc(n ? n : 1); live_add(g_live_dev, *p, n); return *p ? hipSuccess : (t_last = hipErrorOutOfMemory); }
hipError_t hipFree(void *p) { live_del(g_live_dev, p); return hipSuccess; }   // bump allocator: nothing is reused, so stale pointers stay visible to the checks
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = arena_alloc(n ? n : 1); shadow_set(*p, 2, n ? n : 1); live_add(g_live_host, *p, n); return *p ? hipSuccess : (t_last = hipErrorOutOfMemory); }   // (the host writes it without telling anyone: defined)
hipError_t hipHostFree(void *p) { live_del(g_live_host, p); return hipSuccess; }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { ensure_arena(); *tot = g_arena_size; *fr = g_arena_size - g_top; return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) { memmove(d, s, n); shadow_copy(d, s, n); } return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) { memmove(d, s, n); shadow_copy(d, s, n); } return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { if (n) { memmove(d, s, n); shadow_copy(d, s, n); } return hipSuccess; }
hipError_t hipMemcpyPeer(void *d, int, const void *s, int, size_t n) { if (n) { memmove(d, s, n); shadow_copy(d, s, n); } return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { if (n) { memset(d, v, n); shadow_set(d, 2, n); } return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) { memset(d, v, n); shadow_set(d, 2, n); } return hipSuccess; }

// ---- streams and events (everything is synchronous) --------------------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(16); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)malloc(16); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (hipStream_t)malloc(16); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { if (least) *least = 1; if (greatest) *greatest = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
int hipGetStreamDeviceId(hipStream_t) { return t_dev; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t) new FakeEvent{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete (FakeEvent *)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { ((FakeEvent *)e)->t = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(((FakeEvent *)b)->t - ((FakeEvent *)a)->t); return hipSuccess; }

// ---- launch ------------------------------------------------------------------------------------------------------
hipError_t hipLaunchKernel(const void *f, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t) {
    std::string name;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_kernels.find(f);
        if (it == g_kernels.end()) { fprintf(stderr, "fakehip: launch of an unregistered kernel %p\n", f); return t_last = hipErrorInvalidDeviceFunction; }
        name = it->second;
    }
    if (!g_cb) { fprintf(stderr, "fakehip: no launch callback installed (kernel %s)\n", name.c_str()); return t_last = hipErrorLaunchFailure; }
    int rc = g_cb(name.c_str(), grid.x, grid.y, grid.z, block.x, block.y, block.z, args, shmem);
    return rc == 0 ? hipSuccess : (t_last = hipErrorLaunchFailure);
}
}  // extern "C"
