// emu_rt.hpp -- TEST INFRASTRUCTURE ONLY.  A tiny host stand-in for the slice of the HIP
// runtime that fhe.rs_amd/csrc uses, so the *same kernel sources* can be compiled with g++
// and executed in a container without a GPU (tests/emu/build.sh -> tests/emu/_build/libfhe_emu.so).
//
// Workgroups run one at a time; every "thread" of a workgroup is a ucontext fiber and
// __syncthreads() yields to a round-robin scheduler, which gives real barrier semantics
// (all fibers reach barrier k before any passes it).  This checks indexing, barrier placement
// and arithmetic against the oracle.  It says nothing about performance, bank conflicts or
// memory-model hazards -- those are checked on the MI355X by `pytest -m gpu`.
// The shipped package never loads this library (see fhe.rs_amd/_lib.py).
#pragma once
#include <ucontext.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <mutex>
#include <vector>

#define __global__ static
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct State {
    dim3 threadIdx, blockIdx, blockDim, gridDim;
    unsigned char *smem = nullptr;
    ucontext_t main_ctx;
    std::vector<ucontext_t> fibers;
    std::vector<unsigned char *> stacks;
    std::vector<char> done;
    unsigned current = 0;
    const std::function<void()> *body = nullptr;
};
inline State &st() {
    static thread_local State s;   // one fiber scheduler per host thread (a host may drive one "device" per thread)
    return s;
}
inline void fiber_entry() {
    State &s = st();
    (*s.body)();
    s.done[s.current] = 1;
    swapcontext(&s.fibers[s.current], &s.main_ctx);
}
inline void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()> &body) {
    State &s = st();
    const unsigned nthreads = block.x * block.y * block.z;
    constexpr size_t STACK = 96 * 1024;
    while (s.stacks.size() < nthreads) s.stacks.push_back((unsigned char *)malloc(STACK));
    s.fibers.resize(nthreads);
    s.done.assign(nthreads, 0);
    std::vector<unsigned char> smem(smem_bytes + 64);
    s.smem = (unsigned char *)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
    s.blockDim = block;
    s.gridDim = grid;
    s.body = &body;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                s.blockIdx = dim3(bx, by, bz);
                for (unsigned t = 0; t < nthreads; t++) {
                    getcontext(&s.fibers[t]);
                    s.fibers[t].uc_stack.ss_sp = s.stacks[t];
                    s.fibers[t].uc_stack.ss_size = STACK;
                    s.fibers[t].uc_link = &s.main_ctx;
                    makecontext(&s.fibers[t], (void (*)())fiber_entry, 0);
                    s.done[t] = 0;
                }
                unsigned live = nthreads;
                while (live) {
                    live = 0;
                    for (unsigned t = 0; t < nthreads; t++) {
                        if (s.done[t]) continue;
                        s.current = t;
                        s.threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                        swapcontext(&s.main_ctx, &s.fibers[t]);
                        if (!s.done[t]) live++;
                    }
                }
            }
    s.body = nullptr;
}
inline void syncthreads() {
    State &s = st();
    swapcontext(&s.fibers[s.current], &s.main_ctx);
}
}  // namespace emu

#define threadIdx (emu::st().threadIdx)
#define blockIdx (emu::st().blockIdx)
#define blockDim (emu::st().blockDim)
#define gridDim (emu::st().gridDim)
inline void __syncthreads() { emu::syncthreads(); }
inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}
#define FHE_DYN_SMEM(type, name) type *name = reinterpret_cast<type *>(emu::st().smem)

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); })
// (hipExtLaunchKernelGGL: the launch that carries its own start / stop events; the two are stamped around the run)
#define hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, ev_start, ev_stop, flags, ...) \
    do {                                                                                        \
        (void)hipEventRecord((ev_start), (stream));                                             \
        emu::launch((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); });                   \
        (void)hipEventRecord((ev_stop), (stream));                                              \
    } while (0)

// ---- runtime API subset ---------------------------------------------------------------
typedef int hipError_t;
typedef void *hipStream_t;
struct emuEvent {
    std::chrono::steady_clock::time_point t;
};
typedef emuEvent *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorContextIsDestroyed = 709 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
inline const char *hipGetErrorString(hipError_t) { return "emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// (no code objects here: the profiler's symbol lookup falls back to the launch label)
inline const char *hipKernelNameRefByPtr(const void *, hipStream_t) { return nullptr; }
// FHE_EMU_DEVICES=<n> (read once): the emulator reports n "devices" (all of them this host), each thread has a
// current one -- enough to walk the engine's per-device bookkeeping (handles, scratch pools, the sharded multiply)
// in CI without a second GPU
inline int emu_device_count() {
    static const int n = [] {
        const char *e = getenv("FHE_EMU_DEVICES");
        const int v = e ? atoi(e) : 1;
        return v >= 1 && v <= 16 ? v : 1;
    }();
    return n;
}
inline int &emu_current_device() {
    static thread_local int d = 0;
    return d;
}
inline hipError_t hipGetDeviceCount(int *n) {
    *n = emu_device_count();
    return hipSuccess;
}
inline hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= emu_device_count()) return hipErrorInvalidValue;
    emu_current_device() = d;
    return hipSuccess;
}
inline hipError_t hipGetDevice(int *d) {
    *d = emu_current_device();
    return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t n) {
    *p = aligned_alloc(64, (n + 63) / 64 * 64 + 64);
    return *p ? hipSuccess : hipErrorInvalidValue;
}
inline hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) {
    memset(d, v, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
    memset(d, v, n);
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) {
    *e = new emuEvent();
    return hipSuccess;
}
inline hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
enum { hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
// Streams: distinct non-null handles (the emulator runs everything in order) with a liveness table, so that
// hipStreamQuery on a destroyed handle answers what the HIP runtime answers (an invalid-handle error) and handle
// values are recycled like real ones.
struct emuStreams {
    std::mutex mu;
    char pool[4096];
    bool live[4096] = {};
    int next = 0;
};
inline emuStreams &emu_streams() {
    static emuStreams s;
    return s;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    emuStreams &t = emu_streams();
    std::lock_guard<std::mutex> g(t.mu);
    for (int k = 0; k < 4096; k++) {
        const int i = (t.next + k) % 4096;
        if (!t.live[i]) {
            t.live[i] = true;
            t.next = i + 1;
            *s = (hipStream_t)&t.pool[i];
            return hipSuccess;
        }
    }
    return hipErrorInvalidValue;
}
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) {
    emuStreams &t = emu_streams();
    std::lock_guard<std::mutex> g(t.mu);
    const ptrdiff_t i = (char *)s - t.pool;
    if (i < 0 || i >= 4096 || !t.live[i]) return hipErrorContextIsDestroyed;
    t.live[i] = false;
    return hipSuccess;
}
inline hipError_t hipStreamQuery(hipStream_t s) {
    if (!s) return hipSuccess;
    emuStreams &t = emu_streams();
    std::lock_guard<std::mutex> g(t.mu);
    const ptrdiff_t i = (char *)s - t.pool;
    return (i >= 0 && i < 4096 && t.live[i]) ? hipSuccess : hipErrorContextIsDestroyed;
}
// (allocation counters: what the workspace tests read instead of hipMemGetInfo)
inline std::atomic<long long> &emu_live_bytes() {
    static std::atomic<long long> v{0};
    return v;
}
inline hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { return hipMalloc(p, n); }
inline hipError_t hipFreeAsync(void *p, hipStream_t) { return hipFree(p); }
typedef void *hipMemPool_t;
enum { hipMemPoolAttrReleaseThreshold = 4, hipMemPoolAttrReservedMemCurrent = 5, hipMemPoolAttrUsedMemCurrent = 7 };
enum hipMemAllocationType { hipMemAllocationTypePinned = 1 };
enum hipMemLocationType { hipMemLocationTypeDevice = 1 };
struct hipMemLocation {
    hipMemLocationType type;
    int id;
};
struct hipMemPoolProps {
    hipMemAllocationType allocType;
    int handleTypes;
    hipMemLocation location;
    void *win32SecurityAttributes;
    size_t maxSize;
    unsigned char reserved[56];
};
inline hipError_t hipMemPoolCreate(hipMemPool_t *pool, const hipMemPoolProps *) {
    static char pools[16];
    static int next = 0;
    *pool = &pools[next++ % 16];
    return hipSuccess;
}
inline hipError_t hipMallocFromPoolAsync(void **p, size_t n, hipMemPool_t, hipStream_t) { return hipMalloc(p, n); }
inline hipError_t hipDeviceGetDefaultMemPool(hipMemPool_t *pool, int) { *pool = nullptr; return hipSuccess; }
inline hipError_t hipMemPoolSetAttribute(hipMemPool_t, int, void *) { return hipSuccess; }
inline hipError_t hipMemPoolTrimTo(hipMemPool_t, size_t) { return hipSuccess; }
inline hipError_t hipMemPoolGetAttribute(hipMemPool_t, int, void *v) {   // (no pools here: every free is immediate)
    *(uint64_t *)v = 0;
    return hipSuccess;
}
inline hipError_t hipDeviceTotalMem(size_t *bytes, int) {
    *bytes = (size_t)1 << 34;
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
    *free_b = *total_b = (size_t)1 << 34;
    return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
template <class F>
inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) {
    return hipSuccess;
}

// ---- test hooks (exported by the emulation library only): a "foreign" stream, i.e. one the host creates and destroys
// without telling the engine, as torch or a Rust host with its own HIP bindings would ----
extern "C" {
inline __attribute__((visibility("default"), used)) void *fhe_emu_stream_create() {
    hipStream_t s = nullptr;
    (void)hipStreamCreateWithFlags(&s, 0);
    return s;
}
inline __attribute__((visibility("default"), used)) int fhe_emu_stream_destroy(void *s) { return hipStreamDestroy((hipStream_t)s); }
}
