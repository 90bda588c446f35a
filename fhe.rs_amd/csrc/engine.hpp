// engine.hpp -- host runtime above the kernels: handles (contexts, scalers, keys,
// multiplicators, parameter sets), stream-ordered workspace, launch geometry and the batched
// pipelines (Scaler::scale, key switch, relinearise, rotate, modulus switch, ct x ct).
// C++ because the reference's host is compiled code (Rust) and no Rust toolchain exists in
// this image; the C ABI in fhe_hip.cpp is a thin layer over these classes.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "hostmath.hpp"
#include "kernels.hpp"

namespace fhe {

// ------------------------------------------------------------------ error plumbing ----
enum : int {
    E_OK = 0, E_ARG = -1, E_HIP = -2, E_INVALID_MODULUS = -3, E_INVALID_DEGREE = -4, E_NTT_UNAVAILABLE = -5,
    E_CONTEXT_MISMATCH = -6, E_DEGREE_MISMATCH = -7, E_NO_MORE_CONTEXT = -8, E_CONTEXT_NOT_REACHABLE = -9,
    E_INVALID_SUBST = -10, E_PARAMETER_MISMATCH = -11, E_INVALID_LEVEL = -12, E_MUL_POLY_COUNT = -13,
    E_EMPTY_MODULI = -14, E_NON_COPRIME = -15, E_NOT_ENOUGH_PRIMES = -16, E_KEYSWITCH_UNSUPPORTED = -17,
    E_NO_DEVICE = -18, E_EMPTY_DOT = -19, E_EXPANSION_SIZE = -20, E_EXPANSION_UNSUPPORTED = -21
};

#define FHE_HIP_CHECK(expr)                                                                        \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            throw StatusError(E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

inline void require(bool cond, int code, const char *msg) {
    if (!cond) throw StatusError(code, msg);
}

// The release library reads NO environment variable: FHE_LAB_FLAG / FHE_LAB_INT are compile-time constants there
// (the variable names do not even reach the binary).  -DFHE_LAB builds (A/B tooling, never loaded by the package)
// read FHE_LAB_* switches that choose between exact kernel variants; FHE_LAB_SYNC synchronises after every launch.
#if defined(FHE_LAB)
inline bool lab_env_flag(const char *name) { return std::getenv(name) != nullptr; }
inline int lab_env_int(const char *name, int dflt) {
    const char *e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}
#define FHE_LAB_FLAG(name) ::fhe::lab_env_flag("FHE_LAB_" name)
#define FHE_LAB_INT(name, dflt) ::fhe::lab_env_int("FHE_LAB_" name, dflt)
#else
#define FHE_LAB_FLAG(name) false
#define FHE_LAB_INT(name, dflt) (dflt)
#endif
inline bool debug_sync() {
    static const bool on = FHE_LAB_FLAG("SYNC");
    return on;
}
// The F64 kernels for moduli below 2^50 (round 6) are the product default.  fhe_engine_set_f64(0) sends every launch to
// the integer kernels instead -- an execution option like fhe_ksk_set_mode (same results either way; tests run both, A/B
// timings flip it inside one process).  Read once per launch.
inline std::atomic<bool> &f64_enabled_flag() {
    static std::atomic<bool> on{true};
    return on;
}
inline bool f64_disabled() { return !f64_enabled_flag().load(std::memory_order_relaxed); }
// compute units of a device (persistent launches size their grids with it)
inline int device_cus(int device) {
#if defined(FHE_HOST_EMULATION)
    (void)device;
    return 3;   // (few, so that the emulated suite runs several items through one workgroup)
#else
    static std::mutex mu;
    static std::vector<int> cache;
    std::lock_guard<std::mutex> g(mu);
    if (device < 0) return 1;
    if ((size_t)device >= cache.size()) cache.resize((size_t)device + 1, 0);
    if (!cache[(size_t)device]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
        cache[(size_t)device] = n;
    }
    return cache[(size_t)device];
#endif
}

// ------------------------------------------------------------------------ profiling ----
// Optional per-kernel timing with HIP events recorded on the launching stream
// (bench.py reads these to compute the dominant kernel's achieved bytes/s live).
class Profiler {
public:
    static Profiler &get() {
        static Profiler p;
        return p;
    }
    bool enabled = false;
    // launches of THIS thread that must not be recorded (the microbenchmarks time themselves): a per-thread count, so a
    // thread profiling real work at the same time loses nothing (ubench used to flip `enabled` for everyone)
    static int &suppressed() {
        static thread_local int n = 0;
        return n;
    }
    struct Suppress {
        Suppress() { ++suppressed(); }
        ~Suppress() { --suppressed(); }
    };
    struct Pending {
        int id;
        hipEvent_t a, b;
    };
    // One entry per (label, kernel symbol): two instantiations of one template launched under one label -- the wide and
    // the narrow ntt_kernel<false, 13, ...> behind "ntt_fwd" -- stay apart, as rocprofv3 keeps them apart (VERDICT r05:
    // the label-keyed table named the wrong dominant kernel).  fhe_prof_get reports (label, launches, ms) per entry,
    // fhe_prof_get_symbol the kernel's demangled symbol; callers that want families sum the entries of a label.
    struct Entry {
        std::string name;
        const void *fn = nullptr;
        uint64_t launches = 0;
        double ms = 0;
    };
    int id_of(const char *name, const void *fn) {
        for (size_t i = 0; i < entries.size(); i++)
            if (entries[i].fn == fn && entries[i].name == name) return (int)i;
        entries.push_back(Entry{name, fn, 0, 0});
        return (int)entries.size() - 1;
    }
    hipEvent_t take_event() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        FHE_HIP_CHECK(hipEventCreate(&e));
        return e;
    }
    // The two events of a launch travel WITH it (hipExtLaunchKernelGGL stamps them from the dispatch itself) instead
    // of being recorded around it: the interval is the kernel's own duration, and the stream carries no extra
    // barrier packets -- round 3: the record-around form cost the timed region of bench.py 2.4-3.5 %.
    Pending &begin(const char *name, const void *fn) {
        cur = Pending{id_of(name, fn), take_event(), take_event()};
        return cur;
    }
    void end() { pending.push_back(cur); }
    void drain() {
        for (auto &p : pending) {
            FHE_HIP_CHECK(hipEventSynchronize(p.b));
            float ms = 0;
            FHE_HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
            entries[p.id].launches++;
            entries[p.id].ms += ms;
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
    void reset() {
        drain();
        entries.clear();
    }
    std::vector<Entry> entries;
    std::mutex mu;

private:
    Pending cur{};
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
};

#define FHE_LAUNCH(name, kernel, grid, block, smem, stream, ...)                         \
    do {                                                                                 \
        Profiler &_pf = Profiler::get();                                                 \
        if (_pf.enabled && !Profiler::suppressed()) {                                    \
            std::lock_guard<std::mutex> _lk(_pf.mu);                                     \
            auto &_ev = _pf.begin(name, reinterpret_cast<const void *>(kernel));         \
            hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, _ev.a, _ev.b, 0, __VA_ARGS__); \
            _pf.end();                                                                   \
        } else {                                                                         \
            hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);          \
        }                                                                                \
        FHE_HIP_CHECK(hipGetLastError());                                                \
        if (debug_sync()) FHE_HIP_CHECK(hipStreamSynchronize(stream));                   \
    } while (0)

// --------------------------------------------------------------- device allocations ----
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    void reset() {
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
    }
    void alloc(size_t n) {
        reset();
        if (n) FHE_HIP_CHECK(hipMalloc((void **)&p, n * sizeof(T)));
        count = n;
    }
    void upload(const std::vector<T> &h) {
        alloc(h.size());
        if (!h.empty()) FHE_HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    }
};

// Two PRIVATE stream-ordered memory pools per device (hipMemPoolCreate): SCRATCH serves the engine's workspace blocks,
// BUFFERS the ABI's fhe_buf_alloc_async.  (Round 3 raised the release threshold of the device's DEFAULT pool instead, which
// changed the behaviour of every other hipMallocAsync user in the process.)
// Round 5: the scratch pool's release threshold IS the workspace's total limit -- what the engine evicts
// goes back to the driver (hipMemPoolTrimTo right after an eviction, and the threshold at the next synchronisation for
// blocks whose stream-ordered free had not retired yet), so fhe_workspace_set_limit bounds the device memory the process
// holds for scratch, not only the bookkeeping.  The buffers pool keeps what is freed into it (a host that allocates and
// drops a result per call must not pay the driver for it each time) until fhe_workspace_trim.
class DevPools {
public:
    enum Kind : int { SCRATCH = 0, BUFFERS = 1 };
    static DevPools &get() {
        static DevPools p;
        return p;
    }
    hipMemPool_t pool(int device, Kind kind) {
        std::lock_guard<std::mutex> g(mu);
        return pool_locked(device, kind);
    }
    void *alloc(int device, size_t bytes, hipStream_t s, Kind kind) {
        void *p = nullptr;
        FHE_HIP_CHECK(hipMallocFromPoolAsync(&p, bytes ? bytes : 8, pool(device, kind), s));
        return p;
    }
    // bytes of reserved scratch the pools may keep once it is idle (0 = everything): Workspace's total limit
    void set_scratch_threshold(int device, size_t bytes) {
        std::lock_guard<std::mutex> g(mu);
        if ((size_t)device >= thresholds.size()) thresholds.resize((size_t)device + 1, ~0ull);
        const uint64_t keep = bytes ? (uint64_t)bytes : ~0ull;
        if (thresholds[(size_t)device] == keep && (size_t)(2 * device) < pools.size() && pools[(size_t)(2 * device)]) return;
        thresholds[(size_t)device] = keep;
        hipMemPool_t mp = pool_locked(device, SCRATCH);
        uint64_t v = keep;
        (void)hipMemPoolSetAttribute(mp, hipMemPoolAttrReleaseThreshold, &v);
    }
    // hands idle scratch beyond `keep` bytes back to the driver now -- when the pool holds more than `slack` bytes over
    // it (a working set sitting AT the limit must not free to the driver and re-map on every call; the
    // pool's release threshold takes care of the rest at the next synchronisation)
    void trim_scratch_to(int device, size_t keep, size_t slack = 0) {
        std::lock_guard<std::mutex> g(mu);
        const size_t i = (size_t)(2 * device + SCRATCH);
        if (i >= pools.size() || !pools[i]) return;
        if (slack) {
            uint64_t reserved = 0;
            if (hipMemPoolGetAttribute(pools[i], hipMemPoolAttrReservedMemCurrent, &reserved) == hipSuccess &&
                reserved <= (uint64_t)keep + slack)
                return;
        }
        (void)hipMemPoolTrimTo(pools[i], keep);
    }
    // hands the pools' idle memory back to the driver
    void trim() {
        std::lock_guard<std::mutex> g(mu);
        for (hipMemPool_t mp : pools)
            if (mp) (void)hipMemPoolTrimTo(mp, 0);
    }
    // what the driver says the pools hold: reserved (backed by device memory) and used (handed out) bytes
    void stats(int device, size_t out[4]) {
        std::lock_guard<std::mutex> g(mu);
        for (int k = 0; k < 2; k++) {
            uint64_t r = 0, u = 0;
            const size_t i = (size_t)(2 * device + k);
            if (device >= 0 && i < pools.size() && pools[i]) {
                (void)hipMemPoolGetAttribute(pools[i], hipMemPoolAttrReservedMemCurrent, &r);
                (void)hipMemPoolGetAttribute(pools[i], hipMemPoolAttrUsedMemCurrent, &u);
            }
            out[2 * k] = (size_t)r;
            out[2 * k + 1] = (size_t)u;
        }
    }

private:
    hipMemPool_t pool_locked(int device, Kind kind) {
        const size_t i = (size_t)(2 * device + (int)kind);
        if (i >= pools.size()) pools.resize(i + 1, nullptr);
        if (!pools[i]) {
            hipMemPoolProps props;
            std::memset(&props, 0, sizeof(props));
            props.allocType = hipMemAllocationTypePinned;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = device;
            hipMemPool_t mp = nullptr;
            FHE_HIP_CHECK(hipMemPoolCreate(&mp, &props));
            uint64_t keep = ~0ull;
            if (kind == SCRATCH && (size_t)device < thresholds.size()) keep = thresholds[(size_t)device];
            FHE_HIP_CHECK(hipMemPoolSetAttribute(mp, hipMemPoolAttrReleaseThreshold, &keep));
            pools[i] = mp;
        }
        return pools[i];
    }
    std::mutex mu;
    std::vector<hipMemPool_t> pools;        // [2 * device + kind]
    std::vector<uint64_t> thresholds;       // scratch release threshold per device
};

// Scratch blocks, reused in stream order: a block released by stream S is handed out again only to work enqueued on
// S, so there is no cross-stream hazard and no allocation per call.  Round 4: bounded.
//  * Blocks come from the device's private stream-ordered pool (DevPools).  A stream's own too-small blocks go back to
//    it in stream order (hipFreeAsync on the acquiring stream, which is alive by construction), so growing a stream's
//    block no longer synchronises the device.
//  * Limits (fhe_workspace_set_limit): `per_stream` and `total` bound the bytes the engine RETAINS -- idle blocks
//    beyond them are evicted least-recently-used first, on release and before growing.  Blocks in use are never
//    refused: a call that needs more than the limit still runs, its blocks just are not kept afterwards.  Blocks of
//    OTHER streams are evicted with hipFree (it waits for the device): the engine cannot know whether their stream still
//    exists -- a host that makes and destroys its own HIP streams never says so, and this runtime's hipStreamQuery
//    dereferences a destroyed handle (segmentation fault, tools/probe/hip_pool_probe.cpp) -- and hipFree is correct
//    either way.  That is also what bounds such a host: dead streams' blocks are simply the least recently used.
//    Default: no per-stream bound, total = a quarter of the device's memory (at least 8 GiB); 0 = unbounded.
//    `total` is PER DEVICE (round 5, ADVICE r04: it used to be one sum over all devices with a default taken from
//    whichever device was current first, so an in-process multi-GPU host evicted other devices' blocks -- hipSetDevice +
//    hipFree under the global mutex -- in steady state): each device's blocks are counted and evicted against its own
//    bound, and the default is a quarter of THAT device's memory.  What is evicted also leaves the scratch pool
//    (DevPools above): the bound holds for the device memory the process keeps, not just for this table.
//  * A recycled handle value that inherits an old block is harmless: the block was idle, and whatever ran on the old
//    stream is ordered before the new owner's work by the device itself (same queue slot) or long finished.
class Workspace {
public:
    static Workspace &get() {
        static Workspace w;
        return w;
    }
    void *acquire(size_t bytes, hipStream_t s);
    void release(void *p) {
        std::unique_lock<std::mutex> lk(mu);
        hipStream_t owner = nullptr;
        int dev = -1;
        for (auto &b : blocks)
            if (b.ptr == p) {
                b.in_use = false;
                b.last_use = ++tick;
                owner = b.stream;
                dev = b.device;
            }
        // (the releasing call ran on `owner`, which therefore exists: its blocks may go back in stream order)
        enforce_limits_locked(owner, dev, 0);
        flush_trims(lk);
    }
    // A stream is about to be destroyed: its idle blocks can never be handed out again.
    void drop_stream(hipStream_t s) {
        std::lock_guard<std::mutex> lk(mu);
        drop_stream_locked(s, true);
    }
    // Frees every idle block (all devices); returns the number of bytes released.
    size_t trim() {
        std::lock_guard<std::mutex> lk(mu);
        int cur = 0;
        (void)hipGetDevice(&cur);
        size_t freed = 0;
        for (auto &b : blocks)
            if (!b.in_use && b.ptr) {
                (void)hipSetDevice(b.device);
                (void)hipFree(b.ptr);   // (waits for the device: whatever still read the block is done)
                freed += b.bytes;
                b.ptr = nullptr;
            }
        (void)hipSetDevice(cur);
        compact_locked();
        return freed;
    }
    static constexpr size_t LIMIT_DEFAULT = ~(size_t)0;   // "a quarter of the device's memory, at least 8 GiB"
    void set_limits(size_t per_stream, size_t total) {
        std::unique_lock<std::mutex> lk(mu);
        limit_stream = per_stream;
        limit_total = total;
        thresholds_set.clear();   // (every device's scratch pool learns the new bound at its next use)
        for (auto &b : blocks) sync_threshold_locked(b.device);
        enforce_limits_locked(nullptr, -1, 0);
        for (auto &t : deferred_trims) t.slack = 0;   // (a new bound is applied to the letter, now)
        flush_trims(lk);
    }
    void get_limits(size_t *per_stream, size_t *total) {
        std::lock_guard<std::mutex> lk(mu);
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (per_stream) *per_stream = limit_stream;
        if (total) *total = total_limit_locked(dev);
    }
    // bytes held (idle + in use), bytes in use, number of blocks, number of distinct (device, stream) owners
    void stats(size_t *held, size_t *in_use, size_t *nblocks, size_t *nstreams) {
        std::lock_guard<std::mutex> lk(mu);
        size_t h = 0, u = 0;
        std::vector<std::pair<int, hipStream_t>> owners;
        for (auto &b : blocks) {
            h += b.bytes;
            if (b.in_use) u += b.bytes;
            if (std::find(owners.begin(), owners.end(), std::make_pair(b.device, b.stream)) == owners.end())
                owners.emplace_back(b.device, b.stream);
        }
        if (held) *held = h;
        if (in_use) *in_use = u;
        if (nblocks) *nblocks = blocks.size();
        if (nstreams) *nstreams = owners.size();
    }
    // (AuxStreams tells the pool which of its keys are not real streams, and which internal streams it destroyed)
    void drop_internal_stream(hipStream_t aux) {
        std::lock_guard<std::mutex> lk(mu);
        drop_stream_locked(aux, true);
    }

private:
    struct Block {
        void *ptr = nullptr;
        size_t bytes = 0;
        hipStream_t stream = nullptr;
        int device = 0;
        bool in_use = false;
        uint64_t last_use = 0;
    };
    std::vector<Block> blocks;
    std::mutex mu;
    uint64_t tick = 0;
    size_t limit_stream = 0, limit_total = LIMIT_DEFAULT;
    std::vector<size_t> default_total;   // per device, resolved on first use
    std::vector<char> thresholds_set;    // per device: the scratch pool has been told the current bound

    size_t total_limit_locked(int dev) {
        if (limit_total != LIMIT_DEFAULT) return limit_total;
        if (dev < 0) dev = 0;
        if ((size_t)dev >= default_total.size()) default_total.resize((size_t)dev + 1, 0);
        if (!default_total[(size_t)dev]) {
            size_t total_b = 0;
            if (hipDeviceTotalMem(&total_b, dev) != hipSuccess || !total_b) total_b = (size_t)32 << 30;
            default_total[(size_t)dev] = std::max<size_t>((size_t)8 << 30, total_b / 4);
        }
        return default_total[(size_t)dev];
    }
    void sync_threshold_locked(int dev) {
        if (dev < 0) return;
        if ((size_t)dev >= thresholds_set.size()) thresholds_set.resize((size_t)dev + 1, 0);
        if (thresholds_set[(size_t)dev]) return;
        DevPools::get().set_scratch_threshold(dev, total_limit_locked(dev));
        thresholds_set[(size_t)dev] = 1;
    }
    void compact_locked() {
        blocks.erase(std::remove_if(blocks.begin(), blocks.end(), [](const Block &b) { return !b.ptr; }), blocks.end());
    }
    // frees one idle block: stream-ordered on its own stream when that stream is known to be alive, else hipFree
    void free_block_locked(Block &b, bool owner_alive) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != b.device) (void)hipSetDevice(b.device);
        if (!owner_alive || hipFreeAsync(b.ptr, b.stream) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(b.ptr);
        }
        if (cur != b.device) (void)hipSetDevice(cur);
        b.ptr = nullptr;
        b.bytes = 0;
    }
    void drop_stream_locked(hipStream_t s, bool owner_alive) {
        for (auto &b : blocks)
            if (!b.in_use && b.ptr && b.stream == s) free_block_locked(b, owner_alive);
        compact_locked();
    }
    // Evicts idle blocks, least recently used first, until the retained bytes respect the limits; `extra` bytes on
    // (`dev`, `s`) are about to be added (acquire) and count against both.  Only blocks of (`dev`, `s`) -- the stream
    // the current call runs on -- are returned in stream order; every other owner's go through hipFree.
    void enforce_limits_locked(hipStream_t s, int dev, size_t extra) {
        if (!limit_stream && limit_total == 0) return;
        std::vector<int> trimmed;
        for (;;) {
            // per device: the device whose retained bytes exceed its bound gives up its least recently used idle block
            Block *victim = nullptr;
            std::vector<int> devs;
            for (auto &b : blocks)
                if (std::find(devs.begin(), devs.end(), b.device) == devs.end()) devs.push_back(b.device);
            if (dev >= 0 && std::find(devs.begin(), devs.end(), dev) == devs.end()) devs.push_back(dev);
            for (int d : devs) {
                const size_t lim_total = total_limit_locked(d);
                if (!lim_total) continue;
                size_t total = d == dev ? extra : 0;
                for (auto &b : blocks)
                    if (b.device == d) total += b.bytes;
                if (total <= lim_total) continue;
                for (auto &b : blocks)
                    if (b.device == d && !b.in_use && b.ptr && (!victim || b.last_use < victim->last_use)) victim = &b;
                if (victim) break;
            }
            if (!victim && limit_stream) {
                // any (device, stream) owner over its own limit gives up its oldest idle block
                for (auto &b : blocks) {
                    if (b.in_use || !b.ptr) continue;
                    size_t own = (dev >= 0 && b.device == dev && b.stream == s) ? extra : 0;
                    for (auto &o : blocks)
                        if (o.device == b.device && o.stream == b.stream) own += o.bytes;
                    if (own > limit_stream && (!victim || b.last_use < victim->last_use)) victim = &b;
                }
            }
            if (!victim) break;
            const int vdev = victim->device;
            free_block_locked(*victim, dev >= 0 && victim->device == dev && victim->stream == s);
            compact_locked();
            if (std::find(trimmed.begin(), trimmed.end(), vdev) == trimmed.end()) trimmed.push_back(vdev);
        }
        // what was evicted leaves the scratch pool too (blocks whose stream-ordered free has not retired yet follow at the
        // next synchronisation: the pool's release threshold is this bound).  The driver call happens in flush_trims, after
        // the table's mutex is dropped, and only when the pool holds more than an eighth of the bound over what is kept
        // (hipMemPoolTrimTo under the global mutex after every eviction serialised every thread and device
        // behind an unmap, and a working set at the limit paid unmap + map per call).
        for (int d : trimmed) {
            size_t keep = d == dev ? extra : 0;
            for (auto &b : blocks)
                if (b.device == d) keep += b.bytes;
            deferred_trims.push_back(Trim{d, keep, total_limit_locked(d) / 8});
        }
    }
    struct Trim {
        int device;
        size_t keep, slack;
    };
    std::vector<Trim> deferred_trims;
    void flush_trims(std::unique_lock<std::mutex> &lk) {
        if (deferred_trims.empty()) return;
        std::vector<Trim> todo;
        todo.swap(deferred_trims);
        lk.unlock();
        for (const Trim &t : todo) DevPools::get().trim_scratch_to(t.device, t.keep, t.slack);
    }
};
// `wipe`: the block held secret-dependent data (decryption intermediates, which the reference keeps in
// Zeroizing buffers, F/bfv/keys/secret_key.rs:198-226): it is cleared on its stream before it returns to the pool.
struct WsGuard {
    void *p;
    size_t bytes;
    hipStream_t stream;
    bool wipe;
    WsGuard(size_t bytes_, hipStream_t s, bool wipe_ = false)
        : p(Workspace::get().acquire(bytes_, s)), bytes(bytes_), stream(s), wipe(wipe_) {}
    ~WsGuard() {
        if (wipe && bytes) (void)hipMemsetAsync(p, 0, bytes, stream);
        Workspace::get().release(p);
    }
    u64 *u() const { return (u64 *)p; }
};

// ---------------------------------------------------------------------- rq::Context ----
struct Ctx {
    int device = -1;
    size_t n = 0, logn = 0, L = 0;
    std::vector<u64> moduli;
    const Ctx *root = nullptr;  // owner of the tables (children are moduli prefixes of it)
    // host tables (root only)
    std::vector<NttTables> tabs;
    std::vector<ModConsts> mods;
    // one 64-bit fingerprint per modulus over the four NTT tables and N^-1 (root only): two contexts agree on the
    // evaluation order of a shared modulus -- and may exchange Ntt-form rows -- exactly when these are equal
    std::vector<u64> tab_fp;
    // per-context
    std::vector<u64> inv_last, inv_last_shoup;  // M/rq/context.rs:66-73
    std::unique_ptr<Ctx> next;                  // next_context
    // device tables (owned by root; children alias them)
    DevBuf<DevMod> d_mods;
    DevBuf<k::u64x2> d_tw, d_itw, d_ninv, d_inv_last, d_pow2;
    // Round 6: the F64 twins of d_tw / d_itw / d_ninv -- {w, w / p} as doubles, same indexing -- for the rows whose
    // modulus is below 2^50 (zq_f64.hpp; rows of wider moduli are left zero and never read: a launch takes the F64
    // kernels only when every one of its rows qualifies, f64_class below).  Empty when no modulus qualifies.
    DevBuf<k::u64x2> d_tw_f, d_itw_f, d_ninv_f;

    const k::u64x2 *dtw_f() const { return root->d_tw_f.p; }
    const k::u64x2 *ditw_f() const { return root->d_itw_f.p; }
    const k::u64x2 *dninv_f() const { return root->d_ninv_f.p; }
    // The F64 class HR of the moduli [first, first + rows) of the ROOT's list: 5 when all are below 2^48, 4 below 2^49,
    // 3 below 2^50, 0 when one of them is wider (integer kernels).  The kernels are instantiated for these three.
    int f64_class(size_t first, size_t rows) const {
        if (!root->d_tw_f.p || rows == 0 || f64_disabled()) return 0;
        u64 mx = 0;
        for (size_t i = first; i < first + rows; i++) mx = std::max(mx, root->moduli[i]);
        if (mx >> 50) return 0;
        return (mx >> 49) ? 3 : (mx >> 48) ? 4 : 5;
    }
    const DevMod *dmods() const { return root->d_mods.p; }
    const k::u64x2 *dtw() const { return root->d_tw.p; }
    const k::u64x2 *ditw() const { return root->d_itw.p; }
    const k::u64x2 *dninv() const { return root->d_ninv.p; }
    const k::u64x2 *dpow2() const { return root->d_pow2.p; }
    const NttTables &tab(size_t i) const { return root->tabs[i]; }
    // Do the first `rows` moduli of this context and of `o` use the same NTT tables?  (Handles built from different
    // table sources -- the engine's own psi, fhe_ctx_create with host tables, fhe_params_create_with_tables -- can
    // share moduli and still disagree on the evaluation order; Ntt-form rows then must not cross between them.)
    bool same_tables(const Ctx &o, size_t rows) const {
        if (rows > L || rows > o.L) return false;
        for (size_t i = 0; i < rows; i++)
            if (moduli[i] != o.moduli[i] || root->tab_fp[i] != o.root->tab_fp[i]) return false;
        return true;
    }
    size_t poly_elems() const { return L * n; }
    void need_device() const { require(device >= 0, E_NO_DEVICE, "handle was created host-only (device = -1)"); }
    bool same_ring(const Ctx &o) const { return n == o.n && moduli == o.moduli; }
    const Ctx *at_level(size_t i) const {
        const Ctx *c = this;
        for (size_t k = 0; k < i; k++) {
            if (!c->next) return nullptr;
            c = c->next.get();
        }
        return c;
    }
    // Context::niterations_to (M/rq/context.rs:117-141); -1 if unreachable
    long niterations_to(const Ctx &to) const {
        long it = 0;
        for (const Ctx *c = this; c; c = c->next.get(), it++)
            if (c->same_ring(to)) return it;
        return -1;
    }
};

inline void ctx_fill_inv_last(Ctx &c) {
    c.inv_last.clear();
    c.inv_last_shoup.clear();
    const u64 q_last = c.moduli.back();
    for (size_t i = 0; i + 1 < c.L; i++) {
        const u64 qi = c.moduli[i];
        const u64 inv = powmod(q_last % qi, qi - 2, qi);
        c.inv_last.push_back(inv);
        c.inv_last_shoup.push_back(shoup(inv, qi));
    }
    if (c.device >= 0) {
        std::vector<k::u64x2> h(std::max<size_t>(c.L - 1, 1), k::u64x2{0, 0});
        for (size_t i = 0; i + 1 < c.L; i++) h[i] = k::u64x2{c.inv_last[i], c.inv_last_shoup[i]};
        c.d_inv_last.upload(h);
    }
}

inline std::unique_ptr<Ctx> ctx_create(int device, size_t degree, const std::vector<u64> &moduli,
                                       const u64 *omegas, const u64 *omegas_shoup, const u64 *zetas_inv,
                                       const u64 *zetas_inv_shoup, const u64 *size_inv,
                                       const u64 *size_inv_shoup) {
    require(degree >= 8 && (degree & (degree - 1)) == 0 && degree <= 65536, E_INVALID_DEGREE,
            "InvalidPolynomialDegree: degree must be a power of two in [8, 65536]");
    RnsContext rns_check(moduli);  // EmptyModuli / NonCoprimeModuli / InvalidModulus
    auto c = std::make_unique<Ctx>();
    c->device = device;
    c->n = degree;
    while (((size_t)1 << c->logn) < degree) c->logn++;
    c->L = moduli.size();
    c->moduli = moduli;
    c->root = c.get();
    const bool have_tables = omegas != nullptr;
    if (have_tables)
        require(omegas_shoup && zetas_inv && zetas_inv_shoup && size_inv && size_inv_shoup, E_ARG,
                "either all six NTT tables or none must be supplied");
    for (size_t i = 0; i < c->L; i++) {
        c->mods.push_back(make_mod_consts(moduli[i]));
        if (have_tables) {
            if (!supports_ntt(moduli[i], degree)) throw StatusError(E_NTT_UNAVAILABLE, "NttOperatorUnavailable");
            NttTables t;
            t.omegas.assign(omegas + i * degree, omegas + (i + 1) * degree);
            t.omegas_shoup.assign(omegas_shoup + i * degree, omegas_shoup + (i + 1) * degree);
            t.zetas_inv.assign(zetas_inv + i * degree, zetas_inv + (i + 1) * degree);
            t.zetas_inv_shoup.assign(zetas_inv_shoup + i * degree, zetas_inv_shoup + (i + 1) * degree);
            t.size_inv = size_inv[i];
            t.size_inv_shoup = size_inv_shoup[i];
            c->tabs.push_back(std::move(t));
        } else {
            c->tabs.push_back(make_ntt_tables(moduli[i], degree));
        }
    }
    for (size_t i = 0; i < c->L; i++) {
        const NttTables &t = c->tabs[i];
        u64 h = splitmix64(moduli[i] ^ degree);
        for (size_t j = 0; j < degree; j++) {
            h = splitmix64(h ^ t.omegas[j]) + t.omegas_shoup[j];
            h = splitmix64(h ^ t.zetas_inv[j]) + t.zetas_inv_shoup[j];
        }
        c->tab_fp.push_back(splitmix64(h ^ t.size_inv) + t.size_inv_shoup);
    }
    if (device >= 0) {
        int ndev = 0;
        FHE_HIP_CHECK(hipGetDeviceCount(&ndev));
        require(device < ndev, E_HIP, "no such HIP device");
        FHE_HIP_CHECK(hipSetDevice(device));
        std::vector<DevMod> hm(c->L);
        static_assert(sizeof(DevMod) == sizeof(ModConsts), "DevMod layout");
        std::memcpy(hm.data(), c->mods.data(), c->L * sizeof(DevMod));
        c->d_mods.upload(hm);
        std::vector<k::u64x2> tw(c->L * degree), itw(c->L * degree), ninv(2 * c->L);
        for (size_t i = 0; i < c->L; i++) {
            const NttTables &t = c->tabs[i];
            for (size_t j = 0; j < degree; j++) {
                tw[i * degree + j] = k::u64x2{t.omegas[j], t.omegas_shoup[j]};
                itw[i * degree + j] = k::u64x2{t.zetas_inv[j], t.zetas_inv_shoup[j]};
            }
            // {N^-1, shoup} and {z_last * N^-1, shoup}: the last inverse stage (one block, twiddle
            // zetas_inv[N-2]) absorbs the N^-1 scaling
            const u64 zn = mulmod(t.zetas_inv[degree - 2], t.size_inv, moduli[i]);
            ninv[2 * i] = k::u64x2{t.size_inv, t.size_inv_shoup};
            ninv[2 * i + 1] = k::u64x2{zn, shoup(zn, moduli[i])};
        }
        c->d_tw.upload(tw);
        c->d_itw.upload(itw);
        c->d_ninv.upload(ninv);
        // F64 twins for moduli below 2^50 (host doubles: conversion exact, division correctly rounded)
        bool any_f64 = false;
        for (size_t i = 0; i < c->L; i++) any_f64 = any_f64 || (moduli[i] >> 50) == 0;
        if (any_f64) {
            auto bits = [](double d) {
                u64 u;
                std::memcpy(&u, &d, sizeof(u));
                return u;
            };
            std::vector<k::u64x2> twf(c->L * degree, k::u64x2{0, 0}), itwf(c->L * degree, k::u64x2{0, 0}), ninvf(2 * c->L, k::u64x2{0, 0});
            for (size_t i = 0; i < c->L; i++) {
                if (moduli[i] >> 50) continue;
                const double pd = (double)moduli[i];
                auto pair = [&](u64 w) { return k::u64x2{bits((double)w), bits((double)w / pd)}; };
                for (size_t j = 0; j < degree; j++) {
                    twf[i * degree + j] = pair(tw[i * degree + j].x);
                    itwf[i * degree + j] = pair(itw[i * degree + j].x);
                }
                ninvf[2 * i] = pair(ninv[2 * i].x);
                ninvf[2 * i + 1] = pair(ninv[2 * i + 1].x);
            }
            c->d_tw_f.upload(twf);
            c->d_itw_f.upload(itwf);
            c->d_ninv_f.upload(ninvf);
        }
        std::vector<k::u64x2> pow2(c->L);
        for (size_t i = 0; i < c->L; i++) {
            const u64 two64 = (u64)((((u128)1) << 64) % moduli[i]);
            pow2[i] = k::u64x2{two64, mulmod(two64, two64, moduli[i])};
        }
        c->d_pow2.upload(pow2);
    }
    ctx_fill_inv_last(*c);
    // next_context chain (M/rq/context.rs:75-79): prefixes sharing the root's tables
    Ctx *parent = c.get();
    for (size_t l = c->L - 1; l >= 1; l--) {
        auto ch = std::make_unique<Ctx>();
        ch->device = device;
        ch->n = degree;
        ch->logn = c->logn;
        ch->L = l;
        ch->moduli.assign(moduli.begin(), moduli.begin() + l);
        ch->root = c.get();
        ctx_fill_inv_last(*ch);
        parent->next = std::move(ch);
        parent = parent->next.get();
    }
    return c;
}

// ------------------------------------------------------------------- launch helpers ----
inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }
inline unsigned blocks_for(u64 total, unsigned threads) { return (unsigned)((total + threads - 1) / threads); }
constexpr unsigned EW_THREADS = 256;

template <class K>
inline void allow_big_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        FHE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

#if defined(FHE_LAB)
// rejected kernel variants, selected by FHE_LAB_* environment switches in lab builds only (lab/lab_engine.hpp)
struct Ksk;
inline bool lab_try_ntt_fwd(const Ctx &c, unsigned rows_total, bool narrow, const u64 *in, u64 *out, const k::RowMap &map,
                            hipStream_t s);
template <int LOGN>
inline bool lab_try_ks_pair(const Ksk &k_, const u64 *p, u64 p_stride, u64 *o0, u64 *o1, u64 out_stride, const u64 *a0,
                            const u64 *a1, u64 a_stride, size_t npolys, hipStream_t s);
#endif

template <bool INV, bool NARROW = false, bool GATHER = false>
inline void launch_ntt_lds(const char *name, uint32_t logm, unsigned grid, hipStream_t s, const u64 *in, u64 *out,
                           const k::RowMap &map, const DevMod *mods, const k::u64x2 *tw, const k::u64x2 *ninv,
                           uint32_t logn) {
    const size_t lds = k::lds_words(1u << logm) * sizeof(u64);
#define FHE_NTT_CASE(LM)                                                                                        \
    case LM:                                                                                                    \
        allow_big_lds((k::ntt_kernel<INV, LM, NARROW, 1, GATHER>), lds);                                        \
        FHE_LAUNCH(name, (k::ntt_kernel<INV, LM, NARROW, 1, GATHER>), dim3(grid), dim3(k::ntt_threads_c(LM)), lds, s, in, \
                   out, map, mods, tw, ninv, logn);                                                             \
        break;
    if constexpr (GATHER) {   // (galois_apply folds the substitution from N = 4096 on: three tile sizes)
        switch (logm) {
            FHE_NTT_CASE(12) FHE_NTT_CASE(13) FHE_NTT_CASE(14)
            default: throw StatusError(E_ARG, "unsupported NTT tile size for the gathering loader");
        }
    } else {
        switch (logm) {
            FHE_NTT_CASE(3) FHE_NTT_CASE(4) FHE_NTT_CASE(5) FHE_NTT_CASE(6) FHE_NTT_CASE(7) FHE_NTT_CASE(8)
            FHE_NTT_CASE(9) FHE_NTT_CASE(10) FHE_NTT_CASE(11) FHE_NTT_CASE(12) FHE_NTT_CASE(13) FHE_NTT_CASE(14)
            default: throw StatusError(E_ARG, "unsupported NTT tile size");
        }
    }
#undef FHE_NTT_CASE
}

// The F64 instances of ntt_kernel (whole rows of 4096 / 8192 / 16384 points, every modulus of the launch below 2^50:
// Ctx::f64_class): false when the launch does not qualify and takes the integer kernels.
template <bool INV, bool GATHER>
inline bool launch_ntt_f64(const Ctx &c, const char *name, int hr, uint32_t logn, unsigned grid, hipStream_t s, const u64 *in,
                           u64 *out, const k::RowMap &map) {
    if (!hr || logn < 12 || logn > 14) return false;
    const size_t lds = k::lds_words(1u << logn) * sizeof(u64);
    const k::u64x2 *tw = INV ? c.ditw_f() : c.dtw_f();
#define FHE_NTT_F64(LM, HR)                                                                                            \
    do {                                                                                                               \
        allow_big_lds((k::ntt_kernel<INV, LM, false, 1, GATHER, HR>), lds);                                            \
        FHE_LAUNCH(name, (k::ntt_kernel<INV, LM, false, 1, GATHER, HR>), dim3(grid), dim3(k::ntt_threads_c(LM)), lds, s, in, \
                   out, map, c.dmods(), tw, c.dninv_f(), logn);                                                        \
    } while (0)
#define FHE_NTT_F64_HR(LM)                      \
    case LM:                                    \
        if (hr == 3) FHE_NTT_F64(LM, 3);        \
        else if (hr == 4) FHE_NTT_F64(LM, 4);   \
        else FHE_NTT_F64(LM, 5);                \
        break;
    switch (logn) { FHE_NTT_F64_HR(12) FHE_NTT_F64_HR(13) FHE_NTT_F64_HR(14) }
#undef FHE_NTT_F64_HR
#undef FHE_NTT_F64
    return true;
}

// Forward / inverse NTT of `npolys * map.rows` residue rows.  N <= 16384: one LDS-resident
// kernel.  N = 32768 / 65536: G0 global radix stages + LDS kernel on 8192-point sub-blocks.
inline void launch_ntt(const Ctx &c, bool inverse, const u64 *in, u64 *out, k::RowMap map, size_t npolys,
                       hipStream_t s) {
    if (npolys == 0 || map.rows == 0) return;
    const uint32_t logn = (uint32_t)c.logn;
    const unsigned rows_total = (unsigned)(npolys * map.rows);
    if (logn <= 14) {
        // every modulus of the launch below 2^50: the FP64-FMA instances (round 6)
        const int hr = c.f64_class((size_t)((int32_t)map.row_begin + map.mod_offset), map.rows);
        if (!inverse) {
            if (launch_ntt_f64<false, false>(c, "ntt_fwd_f64", hr, logn, rows_total, s, in, out, map)) return;
            // every modulus of the launch below 2^60: the transform without per-stage conditional subtractions
            bool narrow = !FHE_LAB_FLAG("NO_NARROW");
            for (uint32_t r = 0; r < map.rows; r++)
                narrow = narrow && (c.root->moduli[(size_t)((int32_t)(map.row_begin + r) + map.mod_offset)] >> 60) == 0;
#if defined(FHE_LAB)
            if (lab_try_ntt_fwd(c, rows_total, narrow, in, out, map, s)) return;   // lab/lab_engine.hpp
#endif
            if (narrow)
                launch_ntt_lds<false, true>("ntt_fwd", logn, rows_total, s, in, out, map, c.dmods(), c.dtw(), c.dninv(),
                                            logn);
            else
                launch_ntt_lds<false>("ntt_fwd", logn, rows_total, s, in, out, map, c.dmods(), c.dtw(), c.dninv(), logn);
        } else {
            bool narrow = !FHE_LAB_FLAG("NO_NARROW");
            for (uint32_t r = 0; r < map.rows; r++)
                narrow = narrow && (c.root->moduli[(size_t)((int32_t)(map.row_begin + r) + map.mod_offset)] >> 60) == 0;
            if (map.subst_exp ? launch_ntt_f64<true, true>(c, "ntt_inv_f64", hr, logn, rows_total, s, in, out, map)
                               : launch_ntt_f64<true, false>(c, "ntt_inv_f64", hr, logn, rows_total, s, in, out, map))
                return;
            if (map.subst_exp) {   // the source rows are read through the substitution x -> x^subst_exp (galois_apply)
                if (narrow)
                    launch_ntt_lds<true, true, true>("ntt_inv", logn, rows_total, s, in, out, map, c.dmods(), c.ditw(), c.dninv(), logn);
                else
                    launch_ntt_lds<true, false, true>("ntt_inv", logn, rows_total, s, in, out, map, c.dmods(), c.ditw(), c.dninv(), logn);
            } else if (narrow)
                launch_ntt_lds<true, true>("ntt_inv", logn, rows_total, s, in, out, map, c.dmods(), c.ditw(), c.dninv(),
                                           logn);
            else
                launch_ntt_lds<true>("ntt_inv", logn, rows_total, s, in, out, map, c.dmods(), c.ditw(), c.dninv(), logn);
        }
        return;
    }
    const uint32_t logm = 13, g0 = logn - logm, m = 1u << logm;
    const unsigned gth = 256, gblocks = rows_total * (m / gth);
    // every modulus of the launch below 2^60: the LDS halves take the bound-tracked narrow passes here too (round 4;
    // rounds 1-3 ran rows larger than LDS on the general passes whatever the moduli)
    bool narrow = !FHE_LAB_FLAG("NO_NARROW") && !FHE_LAB_FLAG("NO_NARROW_SUB");
    for (uint32_t r = 0; r < map.rows; r++)
        narrow = narrow && (c.root->moduli[(size_t)((int32_t)(map.row_begin + r) + map.mod_offset)] >> 60) == 0;
    k::RowMap inplace = map;
    inplace.src_poly_stride = map.dst_poly_stride;
    inplace.src_row_fixed = -1;
    inplace.in2 = nullptr;   // (the second half of the transform works in `out`: one array)
    inplace.subst_exp = 0;
    if (!inverse) {
        if (g0 == 2)
            FHE_LAUNCH("ntt_fwd_global", (k::ntt_global_kernel<false, 2>), dim3(gblocks), dim3(gth), 0, s, in, out,
                       map, c.dmods(), c.dtw(), c.dninv(), logn);
        else
            FHE_LAUNCH("ntt_fwd_global", (k::ntt_global_kernel<false, 3>), dim3(gblocks), dim3(gth), 0, s, in, out,
                       map, c.dmods(), c.dtw(), c.dninv(), logn);
        if (narrow) {   // (the global stages above leave values below 4p: FWD_B0 = 4)
            const size_t lds = k::lds_words(1u << 13) * sizeof(u64);
            allow_big_lds((k::ntt_kernel<false, 13, true, 4>), lds);
            FHE_LAUNCH("ntt_fwd", (k::ntt_kernel<false, 13, true, 4>), dim3(rows_total << g0), dim3(k::ntt_threads_c(13)), lds,
                       s, out, out, inplace, c.dmods(), c.dtw(), c.dninv(), logn);
        } else {
            launch_ntt_lds<false>("ntt_fwd", logm, rows_total << g0, s, out, out, inplace, c.dmods(), c.dtw(), c.dninv(),
                                  logn);
        }
    } else {
        if (map.subst_exp) {
            if (narrow)
                launch_ntt_lds<true, true, true>("ntt_inv", logm, rows_total << g0, s, in, out, map, c.dmods(), c.ditw(), c.dninv(), logn);
            else
                launch_ntt_lds<true, false, true>("ntt_inv", logm, rows_total << g0, s, in, out, map, c.dmods(), c.ditw(), c.dninv(), logn);
        } else if (narrow)
            launch_ntt_lds<true, true>("ntt_inv", logm, rows_total << g0, s, in, out, map, c.dmods(), c.ditw(), c.dninv(), logn);
        else
            launch_ntt_lds<true>("ntt_inv", logm, rows_total << g0, s, in, out, map, c.dmods(), c.ditw(), c.dninv(), logn);
        if (g0 == 2)
            FHE_LAUNCH("ntt_inv_global", (k::ntt_global_kernel<true, 2>), dim3(gblocks), dim3(gth), 0, s, out, out,
                       inplace, c.dmods(), c.ditw(), c.dninv(), logn);
        else
            FHE_LAUNCH("ntt_inv_global", (k::ntt_global_kernel<true, 3>), dim3(gblocks), dim3(gth), 0, s, out, out,
                       inplace, c.dmods(), c.ditw(), c.dninv(), logn);
    }
}

// Fused tensor + inverse NTT over the extended basis.  Rows that fit LDS (logn <= 14): one kernel.
// N = 32768 / 65536: the same kernel on 8192-point sub-blocks, then the global inverse stages.
// (Defined after full_map.)
inline void launch_tensor_intt(const Ctx &e, const k::TensorSrc &ts, u64 *out, size_t nb, hipStream_t s, bool reverse = false);

// all rows of [npolys][rows_in_poly][N], modulus = row index
inline k::RowMap full_map(const Ctx &c, size_t rows_in_poly) {
    k::RowMap m{};
    m.rows = (uint32_t)rows_in_poly;
    m.row_begin = 0;
    m.mod_offset = 0;
    m.src_row_fixed = -1;
    m.src_poly_stride = m.dst_poly_stride = (u64)rows_in_poly * c.n;
    m.in2 = nullptr;
    m.split = 0;
    m.reverse = 0;
    m.subst_exp = 0;
    return m;
}

inline void launch_tensor_intt_rows(const Ctx &e, const k::TensorSrc &ts, u64 *out, size_t nb, uint32_t row_begin,
                                    uint32_t lrows, bool narrow, hipStream_t s, bool reverse, int f64_hr = 0) {
    const uint32_t logn = (uint32_t)e.logn, logm = logn <= 14 ? logn : 13;
    const size_t lds = k::lds_words(1u << logm) * sizeof(u64);
    // 8 (row, pair, sub-block) combinations x 3 slots per group
    const unsigned groups = (unsigned)(((((size_t)lrows * nb) << (logn - logm)) + 7) / 8);
    if (f64_hr && logn >= 12 && logn <= 14) {   // round 6: rows below 2^50 on the F64 instances
#define FHE_TI_F64(LM, HR)                                                                                          \
    allow_big_lds((k::tensor_intt_kernel<LM, false, false, HR>), lds);                                              \
    FHE_LAUNCH("tensor_intt_f64", (k::tensor_intt_kernel<LM, false, false, HR>), dim3(groups * 24),                 \
               dim3(k::ntt_threads_c(LM)), lds, s, ts, out, e.dmods(), e.ditw_f(), e.dninv_f(), (uint32_t)e.L,      \
               (uint32_t)nb, logn, row_begin, lrows, reverse ? 1u : 0u);
#define FHE_TI_F64_HR(LM)                               \
    case LM:                                            \
        if (f64_hr == 3) { FHE_TI_F64(LM, 3) }          \
        else if (f64_hr == 4) { FHE_TI_F64(LM, 4) }     \
        else { FHE_TI_F64(LM, 5) }                      \
        break;
        switch (logn) { FHE_TI_F64_HR(12) FHE_TI_F64_HR(13) FHE_TI_F64_HR(14) }
#undef FHE_TI_F64_HR
#undef FHE_TI_F64
        return;
    }
#define FHE_TI_LAUNCH(LM, SUB, NRW)                                                                              \
    allow_big_lds((k::tensor_intt_kernel<LM, SUB, NRW>), lds);                                                   \
    FHE_LAUNCH((NRW ? "tensor_intt_narrow" : "tensor_intt"), (k::tensor_intt_kernel<LM, SUB, NRW>), dim3(groups * 24), \
               dim3(k::ntt_threads_c(LM)), lds, s, ts, out, e.dmods(), e.ditw(), e.dninv(), (uint32_t)e.L,       \
               (uint32_t)nb, logn, row_begin, lrows, reverse ? 1u : 0u);
#define FHE_TI_CASE(LM)                               \
    case LM:                                          \
        if (narrow) { FHE_TI_LAUNCH(LM, false, true) } \
        else { FHE_TI_LAUNCH(LM, false, false) }       \
        break;
    if (logn > 14) {
        if (narrow) { FHE_TI_LAUNCH(13, true, true) }
        else { FHE_TI_LAUNCH(13, true, false) }
        return;
    }
    switch (logn) {
        FHE_TI_CASE(3) FHE_TI_CASE(4) FHE_TI_CASE(5) FHE_TI_CASE(6) FHE_TI_CASE(7) FHE_TI_CASE(8)
        FHE_TI_CASE(9) FHE_TI_CASE(10) FHE_TI_CASE(11) FHE_TI_CASE(12) FHE_TI_CASE(13) FHE_TI_CASE(14)
        default: throw StatusError(E_ARG, "unsupported tensor tile size");
    }
#undef FHE_TI_CASE
#undef FHE_TI_LAUNCH
}

inline void launch_tensor_intt(const Ctx &e, const k::TensorSrc &ts, u64 *out, size_t nb, hipStream_t s, bool reverse) {
    // maximal runs of rows of the same kind (moduli below 2^60 or not): one launch each (reverse: last run first)
    bool allow = !FHE_LAB_FLAG("NO_NARROW");
    // A launch that does not fill the device anyway (its workgroups all run at once: a single ciphertext pair, a handful)
    // takes ONE workgroup's time whatever its passes cost, so two launches -- narrow rows, then general rows -- take twice
    // that: such a launch runs every row on the general passes (valid for any modulus below 2^62) in one go.
    // (C2, one pair: tensor_intt 2 x 16.4 us -> 1 x; profiles/r05_latency_breakdown.json)
    {
        const uint32_t lsub = e.logn > 14 ? (uint32_t)e.logn - 13 : 0;
        // (threshold: one round of workgroups -- two fit a CU with 8192-point tiles and smaller, one with 16384-point
        // tiles; C2 at 8 / 16 pairs: profiles/r05_tensor_one_launch_ab.jsonl)
        const size_t slots = (size_t)device_cus(e.device) * (e.logn == 14 ? 1 : 2);
        if ((((size_t)3 * e.L * nb) << lsub) <= slots && device_cus(e.device) > 8) allow = false;
    }
    struct Run {
        uint32_t r0, n;
        int kind;   // 0: general passes, 1: moduli below 2^60 (narrow passes), 2: below 2^50 (round 6: the F64 instances)
    };
    std::vector<Run> runs;
    const bool f64_rows = allow && e.logn >= 12 && e.logn <= 14 && e.root->d_tw_f.p && !f64_disabled();
    auto kind_of = [&](uint32_t r) { return !allow ? 0 : (f64_rows && (e.moduli[r] >> 50) == 0) ? 2 : (e.moduli[r] >> 60) == 0 ? 1 : 0; };
    uint32_t r0 = 0;
    while (r0 < e.L) {
        const int kd = kind_of(r0);
        uint32_t r1 = r0 + 1;
        while (r1 < e.L && kind_of(r1) == kd) r1++;
        runs.push_back(Run{r0, r1 - r0, kd});
        r0 = r1;
    }
    if (reverse) std::reverse(runs.begin(), runs.end());
    for (const Run &r : runs)
        launch_tensor_intt_rows(e, ts, out, nb, r.r0, r.n, r.kind == 1, s, reverse, r.kind == 2 ? e.f64_class(r.r0, r.n) : 0);
    const uint32_t logn = (uint32_t)e.logn;
    if (logn > 14) {  // the global inverse stages finish every row
        const uint32_t logm = 13;
        const unsigned gth = 256, gblocks = (unsigned)(nb * 3 * e.L) * ((1u << logm) / gth);
        const k::RowMap m = full_map(e, e.L);
        if (logn - logm == 2)
            FHE_LAUNCH("ntt_inv_global", (k::ntt_global_kernel<true, 2>), dim3(gblocks), dim3(gth), 0, s, out, out, m,
                       e.dmods(), e.ditw(), e.dninv(), logn);
        else
            FHE_LAUNCH("ntt_inv_global", (k::ntt_global_kernel<true, 3>), dim3(gblocks), dim3(gth), 0, s, out, out, m,
                       e.dmods(), e.ditw(), e.dninv(), logn);
    }
}

inline void ntt_polys(const Ctx &c, bool inverse, const u64 *in, u64 *out, size_t npolys, hipStream_t s) {
    c.need_device();
    launch_ntt(c, inverse, in, out, full_map(c, c.L), npolys, s);
}

inline void ew_op(const Ctx &c, u64 *a, const u64 *b, size_t npolys, uint32_t op, hipStream_t s) {
    c.need_device();
    const u64 total = (u64)npolys * c.L * c.n;
    if (!total) return;
    FHE_LAUNCH("ew", k::ew_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0, s, a, b, c.dmods(),
               (uint32_t)c.L, (uint32_t)c.logn, op, total);
}

// ------------------------------------------------------------------ Rq wire format ----
inline size_t wire_poly_bytes(const Ctx &c) {
    size_t b = 0;
    for (size_t i = 0; i < c.L; i++) b += (size_t)(64 - __builtin_clzll(c.moduli[i] - 1)) * (c.n / 8);
    return b;
}
// polys [npolys][L][N] -> bytes [npolys][wire_poly_bytes]; from_ntt: inverse NTT first (into scratch).
inline void wire_serialize(const Ctx &c, const u64 *polys, uint8_t *bytes, size_t npolys, bool from_ntt, hipStream_t s) {
    c.need_device();
    if (!npolys) return;
    const u64 pe = (u64)c.L * c.n;
    WsGuard pb(from_ntt ? npolys * pe * sizeof(u64) : 8, s);
    if (from_ntt) {
        launch_ntt(c, true, polys, pb.u(), full_map(c, c.L), npolys, s);
        polys = pb.u();
    }
    const unsigned groups = (unsigned)(c.n / 8), block = groups < 256 ? 64 : 256;
    const u64 wb = wire_poly_bytes(c);
    // rows of >= 128 coefficients (and a 16-byte aligned destination: every row then starts on a 16-byte boundary)
    // take the word-granular kernel; its grid covers the widest row (62 bits)
    const bool words = c.n >= 128 && ((uintptr_t)bytes & 15) == 0 && ((uintptr_t)polys & 15) == 0;
    for (size_t p0 = 0; p0 < npolys; p0 += 32768) {  // gridDim.z <= 65535
        const size_t np = std::min<size_t>(32768, npolys - p0);
        if (words)
            FHE_LAUNCH("wire_pack", k::wire_pack_words_kernel,
                       dim3(blocks_for((c.n >> 7) * 64, 256), (unsigned)c.L, (unsigned)np), dim3(256), 0, s, polys + p0 * pe,
                       bytes + p0 * wb, c.dmods(), (uint32_t)c.L, (uint32_t)c.logn, wb);
        else
            FHE_LAUNCH("wire_pack", k::wire_pack_kernel, dim3(blocks_for(groups, block), (unsigned)c.L, (unsigned)np),
                       dim3(block), 0, s, polys + p0 * pe, bytes + p0 * wb, c.dmods(), (uint32_t)c.L, (uint32_t)c.logn, wb);
    }
}
inline void wire_deserialize(const Ctx &c, const uint8_t *bytes, u64 *polys, size_t npolys, bool to_ntt, hipStream_t s) {
    c.need_device();
    if (!npolys) return;
    const unsigned groups = (unsigned)(c.n / 8), block = groups < 256 ? 64 : 256;
    const u64 wb = wire_poly_bytes(c), pe = (u64)c.L * c.n;
    const bool words = c.n >= 128 && ((uintptr_t)bytes & 15) == 0 && ((uintptr_t)polys & 15) == 0;   // (see wire_serialize)
    for (size_t p0 = 0; p0 < npolys; p0 += 32768) {
        const size_t np = std::min<size_t>(32768, npolys - p0);
        if (words)
            FHE_LAUNCH("wire_unpack", k::wire_unpack_words_kernel, dim3(blocks_for(c.n / 2, 256), (unsigned)c.L, (unsigned)np),
                       dim3(256), 0, s, bytes + p0 * wb, polys + p0 * pe, c.dmods(), (uint32_t)c.L, (uint32_t)c.logn, wb);
        else
            FHE_LAUNCH("wire_unpack", k::wire_unpack_kernel, dim3(blocks_for(groups, block), (unsigned)c.L, (unsigned)np),
                       dim3(block), 0, s, bytes + p0 * wb, polys + p0 * pe, c.dmods(), (uint32_t)c.L, (uint32_t)c.logn, wb);
    }
    if (to_ntt) launch_ntt(c, false, polys, polys, full_map(c, c.L), npolys, s);
}

// Poly::random_from_seed (M/rq/mod.rs:276-292): seeds [npolys][32] -> polys [npolys][L][N]
inline void polys_from_seeds(const Ctx &c, const uint8_t *seeds, u64 *polys, size_t npolys, hipStream_t s) {
    c.need_device();
    if (!npolys) return;
    FHE_LAUNCH("seed_expand", k::seed_expand_kernel, dim3((unsigned)npolys), dim3(k::SEED_THREADS), k::SEED_SMEM_BYTES, s, seeds,
               polys,
               c.dmods(), (uint32_t)c.L, (uint32_t)c.logn);
}

// ----------------------------------------------------------------------- rq::Scaler ----
struct Scaler {
    const Ctx *from = nullptr, *to = nullptr;
    size_t ncommon = 0;
    ScalerConstants c;
    DevBuf<u64> d_all;
    k::ScalerDev dev{};
};

// NF of the scale_kernel<NF> instance for `nfrom` source moduli (the column's residues live in registers)
inline size_t scale_kernel_nf(size_t nfrom) {
    // 4 / 9 / 17 / 33: the operand and product bases of BASELINE's configs (L = 4, 8, 16; K = 9, 17, 33).  Round 5 added
    // 6 / 12 / 20 for the reference's stock sets (default_parameters_128: L = 3, 5, 9 and K = 6, 10, 18), which ran on
    // the next instance up -- K = 10 on NF = 17, K = 18 on NF = 33: up to 1.8 x the term loops, all of it zero padding.
    // Round 6: the exact fits that were still padded -- 3 / 5 / 10 / 18 (stock sets: L = 3 on NF = 4, L = 5 on 6, K = 10 on
    // 12, K = 18 on 20: 10-25 % of the terms were zeros) and 8 / 16 (C3's and C5's operand bases, on 9 and 17); 14 / 23 /
    // 27 / 31 for the levels of C5's chain (L = 15 ... 10: K = 31, 29, 27, 25, 23, 21 all ran on NF = 33, up to 57 % padding).
    static constexpr size_t fits[] = {3, 4, 5, 6, 8, 9, 10, 12, 14, 16, 17, 18, 20, 23, 27, 31, 33};
    for (size_t nf : fits)
        if (nfrom <= nf) return nf;
    return 64;
}

inline void scaler_upload(Scaler &s) {
    const ScalerConstants &c = s.c;
    if (s.from->device < 0) return;
    require(c.nfrom <= 64, E_ARG, "RNS scaler supports at most 64 source moduli");
    std::vector<u64> all;
    auto push = [&](const std::vector<u64> &v) {
        size_t off = all.size();
        all.insert(all.end(), v.begin(), v.end());
        return off;
    };
    // device-side derived tables (see scale_kernel): gamma_neg and the 16-entry fold tables
    // (the kernel takes v's bits from limb 3 of the sum on: theta_garner_shift - 1 in [96, 126]; RnsScaler::new's
    // choice, scaler.rs:130-142, is 123 ... 127 for moduli below 2^62 and at most 64 of them)
    require(c.theta_garner_shift >= 97 && c.theta_garner_shift <= 127, E_ARG, "theta_garner_shift out of range");
    std::vector<u64> gneg(c.nto), vtab(c.nto * 16), wtab(c.nto * 32), c128(c.nto * 16);
    for (size_t j = 0; j < c.nto; j++) {
        const u64 q = s.to->moduli[j];
        gneg[j] = (q - c.gamma[j] % q) % q;
        const u64 two64 = (u64)((((u128)1) << 64) % q);
        const u64 two128 = mulmod(two64, two64, q);
        const u64 g64 = mulmod(two64, gneg[j], q);
        for (u64 k = 0; k < 16; k++) {
            vtab[j * 16 + k] = mulmod(k % q, g64, q);
            // +w = w_hi 2^64 + w_lo: w_hi 2^64 mod q;  -w = ~w_lo + (1 - (w_hi + 1) 2^64): that constant mod q
            wtab[j * 32 + k] = mulmod(k % q, two64, q);
            wtab[j * 32 + 16 + k] = (q + 1 % q - mulmod((k + 1) % q, two64, q)) % q;
            c128[j * 16 + k] = mulmod(k % q, two128, q);
        }
    }
    // fold_mask / fold_tab (any factor): v_lo < 2^64 and the small addends (w's low word or its complement
    // < 2^64 + q, table values < 3q) give  sum_j <= (2^64 - 1 + sum_i (p_i - 1)) (q_j - 1) + 2^65 + 4 q_j;  where that
    // is below 2^(2 k_j + 6) the bits above 2^(2 k_j) index a 64-entry table of their residues.
    std::vector<u64> fold(c.nto * 64, 0);
    u64 fold_mask = 0;
    if (c.nto <= 64) {
        BigUint sum_p = BigUint::pow2(64) - BigUint(1);
        for (size_t i = 0; i < c.nfrom; i++) sum_p = sum_p + BigUint(s.from->moduli[i] - 1);
        for (size_t j = 0; j < c.nto; j++) {
            const u64 q = s.to->moduli[j];
            const size_t k = 64 - (size_t)__builtin_clzll(q);
            const BigUint bound = sum_p * BigUint(q - 1) + BigUint::pow2(65) + BigUint(q) * BigUint(4);
            if (bound < BigUint::pow2(std::min<size_t>(2 * k + 6, 128))) {   // (< 2^128: nothing above the 128-bit sum)
                fold_mask |= (u64)1 << j;
                const u64 unit = (BigUint::pow2(2 * k) % BigUint(q)).to_u64();
                for (u64 i = 0; i < 64; i++) fold[j * 64 + i] = mulmod(i % q, unit, q);
            }
        }
    }
    // per-source tables zero-padded to the NF of the scale_kernel instance that serves this scaler (launch_scale):
    // the kernel's term loops then need no bounds checks
    const size_t nf = scale_kernel_nf(c.nfrom);
    auto padded = [&](const std::vector<u64> &v) {
        std::vector<u64> r(v);
        r.resize(nf, 0);
        return r;
    };
    std::vector<u64> omega_p(c.nto * nf, 0);
    for (size_t j = 0; j < c.nto; j++)
        std::copy(c.omega.begin() + j * c.nfrom, c.omega.begin() + (j + 1) * c.nfrom, omega_p.begin() + j * nf);
    std::vector<u64> sign64(c.theta_omega_sign.begin(), c.theta_omega_sign.end());
    // one-accumulator form of w (scale_kernel): subtracted terms enter as (~x) * theta, and the host sums what that
    // adds too much: (2^64 - 1) * theta_omega_i over the subtracted sources, plus (2^128 - 1) * theta_gamma when the
    // v * theta_gamma term is subtracted (v = v_hi 2^64 + v_lo, both words complemented); all mod 2^256
    std::vector<u64> mask64(c.nfrom, 0);
    BigUint wk(0);
    for (size_t i = 0; i < c.nfrom; i++)
        if (c.theta_omega_sign[i]) {
            mask64[i] = ~0ull;
            const u64 limbs[2] = {c.theta_omega_lo[i], c.theta_omega_hi[i]};
            wk = wk + BigUint::from_limbs(limbs, 2) * (BigUint::pow2(64) - BigUint(1));
        }
    if (!c.theta_gamma_sign) {
        const u64 limbs[2] = {c.theta_gamma_lo, c.theta_gamma_hi};
        wk = wk + BigUint::from_limbs(limbs, 2) * (BigUint::pow2(128) - BigUint(1));
    }
    wk = wk % BigUint::pow2(256);
    size_t o_fold = push(fold);
    size_t o_gn = push(gneg), o_om = push(omega_p), o_vt = push(vtab), o_wt = push(wtab), o_c128 = push(c128);
    size_t o_tol = push(padded(c.theta_omega_lo)), o_toh = push(padded(c.theta_omega_hi)), o_tos = push(padded(sign64));
    size_t o_tgl = push(padded(c.theta_garner_lo)), o_tgh = push(padded(c.theta_garner_hi));
    size_t o_tom = push(padded(mask64));
    s.d_all.upload(all);
    u64 *b = s.d_all.p;
    s.dev.gamma_neg = b + o_gn;
    s.dev.omega = b + o_om;
    s.dev.vhi_tab = b + o_vt;
    s.dev.w_tab = b + o_wt;
    s.dev.c128_tab = b + o_c128;
    s.dev.theta_omega_lo = b + o_tol;
    s.dev.theta_omega_hi = b + o_toh;
    s.dev.theta_omega_sign = b + o_tos;
    s.dev.theta_omega_mask = b + o_tom;
    for (size_t i = 0; i < 4; i++) s.dev.w_const[i] = wk.limb(i);
    s.dev.theta_garner_lo = b + o_tgl;
    s.dev.theta_garner_hi = b + o_tgh;
    s.dev.theta_gamma_lo = c.theta_gamma_lo;
    s.dev.theta_gamma_hi = c.theta_gamma_hi;
    // narrow_mask: for a factor-one scaler v = round(sum_i r_i theta_i / 2^shift) <= sum_i r_i + 2 (theta_i / 2^shift
    // ~ garner_i / Q < 1), so target j's sum  sum_i r_i omega_ij + v gamma_neg_j + (small terms < 4 q_j)  is at most
    // (2 * sum_i (q_i - 1) + 2) * (q_j - 1) + 4 q_j.  Where that is below 2^(2 k_j + 1), k_j = bits(q_j), the
    // kernel reduces with the single-word Barrett instead of the 128-bit-ratio reduction.
    s.dev.narrow_mask = 0;
    if (c.is_one && c.nto <= 64) {
        BigUint sum_q(0);
        for (size_t i = 0; i < c.nfrom; i++) sum_q = sum_q + BigUint(s.from->moduli[i] - 1);
        for (size_t j = 0; j < c.nto; j++) {
            const u64 q = s.to->moduli[j];
            const size_t k = 64 - (size_t)__builtin_clzll(q);
            const BigUint bound = (sum_q * BigUint(2) + BigUint(2)) * BigUint(q - 1) + BigUint(q) * BigUint(4);
            if (bound < BigUint::pow2(2 * k + 1)) s.dev.narrow_mask |= (u64)1 << j;
        }
    }
    // for a factor-one scaler v <= sum_i r_i + 2 (see narrow_mask above); when sum_i (q_i - 1) + 2 is below 2^64 the
    // kernel never needs v's high word
    s.dev.v_fits_64 = 0;
    if (c.is_one) {
        BigUint sum_q(2);
        for (size_t i = 0; i < c.nfrom; i++) sum_q = sum_q + BigUint(s.from->moduli[i] - 1);
        if (sum_q < BigUint::pow2(64)) s.dev.v_fits_64 = 1;
    }
    // wide_w: can |t| = |sum_i +/- r_i theta_omega_i -/+ v theta_gamma| reach 2^191?  Below that the kernel's fast path
    // (sign from bit 255, 68 bits of w) IS the reference's `t >> 191 > 0` test and 128-bit w; at or above it the
    // reference's result is defined by those bit tests and the launch takes the instance that reproduces them
    // (ADVICE r03).  Bound: r_i <= q_i - 1, v <= sum_i (q_i - 1) + 2 (theta_garner_i / 2^shift < 1).
    s.dev.wide_w = 0;
    if (!c.is_one) {
        BigUint bound(0), vmax(2);
        for (size_t i = 0; i < c.nfrom; i++) {
            const u64 limbs[2] = {c.theta_omega_lo[i], c.theta_omega_hi[i]};
            bound = bound + BigUint::from_limbs(limbs, 2) * BigUint(s.from->moduli[i] - 1);
            vmax = vmax + BigUint(s.from->moduli[i] - 1);
        }
        const u64 gl[2] = {c.theta_gamma_lo, c.theta_gamma_hi};
        bound = bound + BigUint::from_limbs(gl, 2) * vmax;
        if (!(bound < BigUint::pow2(191))) s.dev.wide_w = 1;
    }
    s.dev.fold_mask = fold_mask;
    s.dev.fold_tab = b + o_fold;
    s.dev.theta_gamma_sign = c.theta_gamma_sign ? 1 : 0;
    s.dev.is_one = c.is_one ? 1 : 0;
    s.dev.shift = (uint32_t)c.theta_garner_shift;
    s.dev.nfrom = (uint32_t)c.nfrom;
    s.dev.nto = (uint32_t)c.nto;
    s.dev.ncommon = (uint32_t)s.ncommon;
}

// Scaler::new (M/rq/scaler.rs:27-52)
inline std::unique_ptr<Scaler> scaler_create(const Ctx &from, const Ctx &to, const BigUint &num,
                                             const BigUint &den) {
    require(from.n == to.n, E_DEGREE_MISMATCH, "DegreeMismatch");
    require(from.device == to.device, E_PARAMETER_MISMATCH, "contexts live on different devices");
    auto s = std::make_unique<Scaler>();
    s->from = &from;
    s->to = &to;
    RnsContext rf(from.moduli), rt(to.moduli);
    s->c = make_scaler_constants(rf, rt, num, den);
    if (s->c.is_one) {
        size_t k = 0;
        while (k < from.L && k < to.L && from.moduli[k] == to.moduli[k]) k++;
        s->ncommon = k;
    }
    scaler_upload(*s);
    return s;
}

// RnsScaler::scale over `npolys * N` coefficient columns (register-resident residues: NF >= nfrom).
inline void launch_scale(const Scaler &sc, const u64 *in, u64 in_stride, u64 *out, u64 out_stride, size_t npolys,
                         hipStream_t s, bool ascending = false) {
    const Ctx &f = *sc.from, &t = *sc.to;
    const u64 total = (u64)npolys * f.n;
    if (!total) return;
    const dim3 grid(blocks_for(total, EW_THREADS)), block(EW_THREADS);
    // profiler label: basis extension (more output rows than input) vs down-scaling
    const char *label = t.L > f.L ? "scale_extend" : "scale_down";
    // PLAIN instances carry no w / v_hi code: factor-one scalers whose v fits one word (every basis extension of BFV)
    const bool plain = sc.dev.is_one && sc.dev.v_fits_64;
    const uint32_t asc = ascending ? 1u : 0u;
#define FHE_SCALE_CASE(NF)                                                                                   \
    if (plain)                                                                                               \
        FHE_LAUNCH(label, (k::scale_kernel<NF, true>), grid, block, 0, s, in, out, in_stride, out_stride,    \
                   sc.dev, t.dmods(), (uint32_t)f.logn, total, asc);                                         \
    else if (sc.dev.wide_w)   /* the reference's bit tests on an out-of-range t, to the letter */             \
        FHE_LAUNCH(label, (k::scale_kernel<NF, false, true>), grid, block, 0, s, in, out, in_stride,         \
                   out_stride, sc.dev, t.dmods(), (uint32_t)f.logn, total, asc);                             \
    else                                                                                                     \
        FHE_LAUNCH(label, (k::scale_kernel<NF, false>), grid, block, 0, s, in, out, in_stride, out_stride,   \
                   sc.dev, t.dmods(), (uint32_t)f.logn, total, asc)
    switch (scale_kernel_nf(f.L)) {   // (the same NF scaler_upload padded the tables to)
        case 3: FHE_SCALE_CASE(3); break;
        case 4: FHE_SCALE_CASE(4); break;
        case 5: FHE_SCALE_CASE(5); break;
        case 6: FHE_SCALE_CASE(6); break;
        case 8: FHE_SCALE_CASE(8); break;
        case 9: FHE_SCALE_CASE(9); break;
        case 10: FHE_SCALE_CASE(10); break;
        case 12: FHE_SCALE_CASE(12); break;
        case 14: FHE_SCALE_CASE(14); break;
        case 16: FHE_SCALE_CASE(16); break;
        case 17: FHE_SCALE_CASE(17); break;
        case 18: FHE_SCALE_CASE(18); break;
        case 20: FHE_SCALE_CASE(20); break;
        case 23: FHE_SCALE_CASE(23); break;
        case 27: FHE_SCALE_CASE(27); break;
        case 31: FHE_SCALE_CASE(31); break;
        case 33: FHE_SCALE_CASE(33); break;
        default:
            require(f.L <= 64, E_ARG, "RNS scaler supports at most 64 source moduli");
            FHE_SCALE_CASE(64);
    }
#undef FHE_SCALE_CASE
}

// Scaler::scale (M/rq/scaler.rs:55-127) on npolys polynomials.
// `copy_common = false` leaves rows [0, ncommon) of `out` untouched (callers that read those
// rows from the source instead, see bfv_mul).
inline void scale_polys(const Scaler &sc, const u64 *in, u64 *out, size_t npolys, bool repr_is_ntt, hipStream_t s,
                        bool copy_common = true) {
    const Ctx &f = *sc.from, &t = *sc.to;
    f.need_device();
    if (!npolys) return;
    const u64 in_stride = (u64)f.L * f.n, out_stride = (u64)t.L * t.n;
    if (sc.ncommon > 0 && copy_common) {
        const u64 per = (u64)sc.ncommon * f.n, total = per * npolys;
        FHE_LAUNCH("copy_rows", k::copy_rows_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0, s, in,
                   out, in_stride, out_stride, per, total);
    }
    if (sc.ncommon >= t.L) return;
    if (repr_is_ntt) {
        WsGuard pb(npolys * in_stride * sizeof(u64), s);
        launch_ntt(f, true, in, pb.u(), full_map(f, f.L), npolys, s);
        launch_scale(sc, pb.u(), in_stride, out, out_stride, npolys, s);
        k::RowMap m = full_map(t, t.L);
        m.rows = (uint32_t)(t.L - sc.ncommon);
        m.row_begin = (uint32_t)sc.ncommon;
        launch_ntt(t, false, out, out, m, npolys, s);
    } else {
        launch_scale(sc, in, in_stride, out, out_stride, npolys, s);
    }
}

// Scaler::scale of TWO operand arrays (Ntt form, npolys polynomials each, the same scaler) in one pass of three
// launches: out [2 * npolys][to.L][N] (in0's polynomials first).  What bfv_mul's two operand extensions are when both
// sides use the same extender (Multiplicator::default): launches twice as large, half as many of them -- each
// launch's tail (the last, partly filled round of workgroups) is paid once instead of twice.
inline void scale_polys_pair(const Scaler &sc, const u64 *in0, const u64 *in1, u64 *out, size_t npolys, hipStream_t s,
                             bool copy_common) {
    const Ctx &f = *sc.from, &t = *sc.to;
    f.need_device();
    if (!npolys) return;
    const u64 in_stride = (u64)f.L * f.n, out_stride = (u64)t.L * t.n;
    if (sc.ncommon > 0 && copy_common) {
        const u64 per = (u64)sc.ncommon * f.n, total = per * npolys;
        for (int h = 0; h < 2; h++)
            FHE_LAUNCH("copy_rows", k::copy_rows_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0, s,
                       h ? in1 : in0, out + (u64)h * npolys * out_stride, in_stride, out_stride, per, total);
    }
    if (sc.ncommon >= t.L) return;
    WsGuard pb(2 * npolys * in_stride * sizeof(u64), s);
    k::RowMap mi = full_map(f, f.L);
    mi.in2 = in1;
    mi.split = (uint32_t)npolys;
    launch_ntt(f, true, in0, pb.u(), mi, 2 * npolys, s);
    launch_scale(sc, pb.u(), in_stride, out, out_stride, 2 * npolys, s);
    k::RowMap m = full_map(t, t.L);
    m.rows = (uint32_t)(t.L - sc.ncommon);
    m.row_begin = (uint32_t)sc.ncommon;
    launch_ntt(t, false, out, out, m, 2 * npolys, s);
}

// Poly::<PowerBasis>::switch_down (M/rq/mod.rs:433-492) on npolys polynomials.
inline void switch_down_polys(const Ctx &c, const u64 *in, u64 in_stride, u64 *out, u64 out_stride, size_t npolys,
                              hipStream_t s) {
    c.need_device();
    require(c.next != nullptr, E_NO_MORE_CONTEXT, "NoMoreContext");
    const u64 total = (u64)npolys * c.n;
    if (!total) return;
    FHE_LAUNCH("switch_down", k::switch_down_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0, s, in,
               out, in_stride, out_stride, c.dmods(), c.d_inv_last.p, (uint32_t)c.L, (uint32_t)c.logn, total);
}

inline void substitute_polys(const Ctx &c, size_t exponent, const u64 *in, u64 *out, size_t npolys, bool ntt,
                             hipStream_t s) {
    c.need_device();
    const size_t e = exponent % (2 * c.n);
    require((e & 1) == 1, E_INVALID_SUBST, "InvalidSubstitutionExponent");
    const u64 total = (u64)npolys * c.L * c.n;
    if (!total) return;
    const u64 stride = (u64)c.L * c.n;
    FHE_LAUNCH("substitute", k::substitute_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0, s, in,
               out, stride, stride, c.dmods(), (uint32_t)c.L, (uint32_t)c.logn, (uint32_t)e, ntt ? 1u : 0u, total);
}

// ------------------------------------------------------------------ KeySwitchingKey ----
struct Ksk {
    const Ctx *ct_ctx = nullptr, *ksk_ctx = nullptr;
    size_t ndigits = 0, log_base = 0;
    // how far a digit can exceed the key moduli (ks_fused_kernel's lift_mode): RNS digits are residues
    // < q_i, so q_i < 2 q_j for all i, j -> 1, < 4 q_j -> 2, otherwise (and for base-2^k digits) 0
    uint32_t lift_mode() const {
        if (log_base != 0) return 0;
        u64 mx = 0, mn = ~0ull;
        for (u64 q : ct_ctx->moduli) mx = std::max(mx, q);
        for (u64 q : ksk_ctx->moduli) mn = std::min(mn, q);
        if (mx < 2 * mn) return 1;          // (moduli < 2^62: no overflow)
        if (mx < 4 * mn) return 2;
        return 0;
    }
    uint32_t digit_arg() const { return (uint32_t)log_base | (lift_mode() << 8); }
    DevBuf<u64> c0, c0s, c1, c1s;  // [ndigits][Lk][N]
    // Round 6: the key words as doubles (bit patterns) when every key modulus is below 2^50 and the digits are RNS rows
    // (ksk_fill_f64; empty otherwise): what ks_fused_kernel's F64 instances read in place of (c0, c0s, c1, c1s) -- 16 bytes
    // per coefficient and digit instead of 32: their accumulate takes its quotient from h / p, so no k / q_j twin exists.
    DevBuf<u64> c0f, c1f;
    // Execution options of this handle (fhe_ksk_set_mode; read once per call, like Mul's):
    //   mode      KS_AUTO: the engine picks per shape and launch (ks_use_unfused, key_switch_polys); KS_FUSED: ks_fused_kernel
    //             (rows larger than LDS: on 16384-point parts); KS_FUSED_SUB: rows larger than LDS on 8192-point sub-blocks
    //             (ks_fused_split_kernel), otherwise KS_FUSED;
    //             KS_UNFUSED: batched digit transforms + streaming MAC (RNS digits only; decomposition keys stay fused);
    //             KS_UNFUSED_SUB: the same with 8192-point sub-block tiles at N = 16384 as well
    //   w_budget  bytes of transformed digit rows (W) one stage-A / stage-B launch pair may have in flight; 0 = default
    std::atomic<int> mode{0};
    std::atomic<size_t> w_budget{0};
};
enum : int { KS_AUTO = 0, KS_FUSED = 1, KS_UNFUSED = 2, KS_UNFUSED_SUB = 3, KS_FUSED_SUB = 4 };

// host key words [ndigits][Lk][N] (canonical, checked by the caller) -> the F64 twins on the device
inline void ksk_fill_f64(Ksk &k_, const u64 *h0, const u64 *h1) {
    const Ctx &kc = *k_.ksk_ctx;
    if (k_.log_base != 0 || kc.device < 0 || !kc.root->d_tw_f.p) return;
    for (u64 q : kc.moduli)
        if (q >> 50) return;
    const size_t count = k_.ndigits * kc.L * kc.n;
    std::vector<u64> f(count);
    auto fill = [&](const u64 *h, DevBuf<u64> &df) {
        for (size_t x = 0; x < count; x++) {
            const double kd = (double)h[x];     // exact: canonical key words are below 2^50 here
            std::memcpy(&f[x], &kd, 8);
        }
        df.upload(f);
    };
    fill(h0, k_.c0f);
    fill(h1, k_.c1f);
}

inline void ksk_validate(const Ctx &ct_ctx, const Ctx &ksk_ctx, size_t ndigits, size_t log_base) {
    require(ct_ctx.n == ksk_ctx.n, E_DEGREE_MISMATCH, "DegreeMismatch");
    require(ct_ctx.device == ksk_ctx.device, E_PARAMETER_MISMATCH, "contexts live on different devices");
    require(ksk_ctx.niterations_to(ct_ctx) >= 0, E_CONTEXT_NOT_REACHABLE,
            "ciphertext context is not reachable from the key context");
    // the key switch adds Ntt-form rows of the two contexts (and reads the caller's Ntt-form digit rows as
    // transforms under the key moduli, `xhat`): both must evaluate the shared moduli in the same order
    require(ksk_ctx.same_tables(ct_ctx, ct_ctx.L), E_PARAMETER_MISMATCH,
            "ParameterMismatch: ciphertext and key contexts were built with different NTT tables");
    if (log_base != 0) {
        require(ksk_ctx.L == 1 && ct_ctx.L == 1, E_PARAMETER_MISMATCH,
                "decomposition keys need single-modulus contexts");
        require(log_base < 63 && ndigits * log_base < 64 + log_base, E_ARG, "bad log_base / ndigits");
    } else {
        require(ndigits == ct_ctx.L, E_PARAMETER_MISMATCH, "ndigits must equal the ciphertext context's moduli count");
        require(ksk_ctx.L >= 2, E_KEYSWITCH_UNSUPPORTED, "KeySwitchingNotSupported: single-modulus key without log_base");
    }
}

template <int LOGN>
inline void launch_ks_fused(const Ksk &k_, const u64 *p, u64 p_stride, u64 *o0, u64 *o1, u64 out_stride,
                            const u64 *a0, const u64 *a1, u64 a_stride, size_t npolys, hipStream_t s,
                            const u64 *xhat, u64 xhat_stride, uint32_t gal) {
    const Ctx &kc = *k_.ksk_ctx;
#if defined(FHE_LAB)
    if (!gal && lab_try_ks_pair<LOGN>(k_, p, p_stride, o0, o1, out_stride, a0, a1, a_stride, npolys, s)) return;   // lab/lab_engine.hpp
#endif
    const size_t lds = (k::lds_words(1u << LOGN) + (k::ks_acc1_in_lds_c(LOGN) ? (size_t)1 << LOGN : 0)) * sizeof(u64);
    // key moduli below 2^60: the transform runs without most conditional subtractions (fwd_butterfly_narrow)
    bool narrow = !FHE_LAB_FLAG("NO_NARROW");
    for (u64 q : kc.moduli) narrow = narrow && (q >> 60) == 0;
    // N = 16384 (ks_fused_kernel's GM): radix-8 while the twiddles are scalar, radix-4 after (GM_MIXED).  Lab builds:
    // FHE_LAB_KS14_PLAN = 8 radix-8 passes throughout (24 VGPRs spilled), 4 radix-4 passes throughout.
    static const int plan14 = FHE_LAB_INT("KS14_PLAN", 0);
    // N = 8192: one workgroup owns a CU (132 KiB of LDS), so the items of a launch go to one resident workgroup per
    // CU, each prefetching its next item's first row across its epilogue (lab builds, FHE_LAB_KS_PERSIST=0: one
    // workgroup per item)
    static const int ks_persist = FHE_LAB_INT("KS_PERSIST", 1);
    unsigned ks_grid = (unsigned)(npolys * kc.L);
#if defined(FHE_HOST_EMULATION)
    constexpr bool ks_persist_size = true;   // (every size, so that the emulated suite walks the item loop)
#else
    constexpr bool ks_persist_size = LOGN == 13 || (FHE_KS_PERSIST14 && LOGN == 14);
#endif
    if (ks_persist_size && ks_persist > 0) ks_grid = std::min<unsigned>(ks_grid, (unsigned)(device_cus(kc.device) * ks_persist));
    // RNS instances (N = 4096, 8192): residue-row digits of same-width moduli, see the kernel.  (Not at N = 16384: in
    // round 3 that instance spilled 52 B; it no longer does, and measured again at C3 in round 4 it changes nothing --
    // relinearise of 512: 4.09 / 4.18 / 4.18 ms against 4.17 / 4.14 / 4.13, profiles/r04_ks14_rns_ab.jsonl.)
    const bool rns = (LOGN == 12 || LOGN == 13) && k_.digit_arg() == (1u << 8);
    // Round 6: every key modulus below 2^50 and RNS digits -> the F64 instances (no lift: any residue row of the basis is
    // a representative under every key modulus, whatever the widths)
    if constexpr (LOGN >= 12 && LOGN <= 14) {
        const int hr = (k_.c0f.p && k_.log_base == 0) ? kc.f64_class(0, kc.L) : 0;
        if (hr) {
            // Geometry of the F64 instances.  N = 16384: the tile's own 1024 threads x 16 coefficients.  N = 8192 (and 4096 in
            // the same shape): 512 (256) threads x 16 coefficients, BOTH accumulator sets in registers, tile-only LDS, one
            // workgroup per item -- TWO (four) workgroups per CU, so one's barriers and LDS round trips hide behind the other's
            // arithmetic.  Rounds 3 and 6 measured this cut slower with the passes it could afford then (mixed radix-8 /
            // radix-4: profiles/r03_ks13_t512_rns_ab.txt, r06_ks13_f64_t512_ab_rejected.jsonl); with one-word per-lane twiddles
            // radix-8 passes fit (126 VGPRs, no scratch) and it is ahead: relinearise of 1,024 at the stock n = 8192 set 1.114
            // -> 1.032 ms, of 64 0.114 -> 0.106 (profiles/r06_ks13_f64_t512_radix8_ab.jsonl, same digest).
            // A launch with at most one workgroup per CU has no second workgroup to overlap with: there the tile's own 1024
            // threads (N = 8192: 1024 x 8, c1 accumulators in LDS, resident item loop) finish a single workgroup sooner --
            // 0.052-0.055 ms against 0.062-0.070 for up to 256 workgroups (profiles/r06_l_f64_ks_modes.jsonl,
            // r06_m_f64_ks_modes_grid.jsonl) -- so the 512-thread instance is taken from the second workgroup per CU on.
            constexpr int F64_TT = (LOGN == 13 || (LOGN == 12 && FHE_KS12_F64_T256)) ? (1 << LOGN) / 16 : 0;
            const bool two_per_cu = F64_TT != 0 && npolys * kc.L > (size_t)device_cus(kc.device);
#define FHE_KS_F64_G(GMV, GALV, HR, TTV)                                                                               \
    do {                                                                                                               \
        const size_t lds_f = TTV ? k::lds_words(1u << LOGN) * sizeof(u64) : lds;                                       \
        const unsigned grid_f = TTV ? (unsigned)(npolys * kc.L) : ks_grid;                                             \
        allow_big_lds((k::ks_fused_kernel<LOGN, false, GMV, TTV, true, 0, GALV, HR>), lds_f);                          \
        FHE_LAUNCH("key_switch_fused_f64", (k::ks_fused_kernel<LOGN, false, GMV, TTV, true, 0, GALV, HR>),             \
                   dim3(grid_f), dim3(k::ks_threads_tt(LOGN, TTV)), lds_f, s, p, p_stride, o0, o1, out_stride, a0, a1, \
                   a_stride, k_.c0f.p, k_.c0f.p, k_.c1f.p, k_.c1f.p, kc.dmods(), kc.dtw_f(), (uint32_t)k_.ndigits,     \
                   (uint32_t)kc.L, k_.digit_arg(), xhat, xhat_stride, (uint32_t)(npolys * kc.L), gal);                 \
    } while (0)
#define FHE_KS_F64_T(GMV, GALV, HR)                                                                                    \
    do {                                                                                                               \
        if constexpr (F64_TT != 0) {                                                                                   \
            if (two_per_cu) {                                                                                          \
                FHE_KS_F64_G(GMV, GALV, HR, F64_TT);                                                                   \
                break;                                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
        FHE_KS_F64_G(GMV, GALV, HR, 0);                                                                                \
    } while (0)
#define FHE_KS_F64(HR)                                                                                                 \
    do {                                                                                                               \
        constexpr int GMV = k::KS_GMAX;   /* radix-8 passes at every size: one-word twiddles leave the N = 16384 tile room */ \
        if (gal) {                                                                                                     \
            FHE_KS_F64_T(GMV, true, HR);                                                                               \
        } else {                                                                                                       \
            FHE_KS_F64_T(GMV, false, HR);                                                                              \
        }                                                                                                              \
    } while (0)
            if (hr == 3) FHE_KS_F64(3);
            else if (hr == 4) FHE_KS_F64(4);
            else FHE_KS_F64(5);
#undef FHE_KS_F64
#undef FHE_KS_F64_T
#undef FHE_KS_F64_G
            return;
        }
    }
#define FHE_KS_LAUNCH_G(NW, GMV, RNS, GALV)                                                                           \
    allow_big_lds((k::ks_fused_kernel<LOGN, NW, GMV, 0, RNS, 0, GALV>), lds);                                         \
    FHE_LAUNCH("key_switch_fused", (k::ks_fused_kernel<LOGN, NW, GMV, 0, RNS, 0, GALV>), dim3(ks_grid),               \
               dim3(k::ks_threads_c(LOGN)), lds, s, p, p_stride, o0, o1, out_stride, a0, a1, a_stride, k_.c0.p,       \
               k_.c0s.p, k_.c1.p, k_.c1s.p, kc.dmods(), kc.dtw(), (uint32_t)k_.ndigits, (uint32_t)kc.L,               \
               k_.digit_arg(), xhat, xhat_stride, (uint32_t)(npolys * kc.L), gal)
    // (the Galois instances exist from N = 4096 on: galois_apply folds the substitution only there)
#define FHE_KS_LAUNCH_R(NW, GMV, RNS)                                                                                 \
    do {                                                                                                              \
        if constexpr (LOGN >= 12) {                                                                                   \
            if (gal) {                                                                                                \
                FHE_KS_LAUNCH_G(NW, GMV, RNS, true);                                                                  \
            } else {                                                                                                  \
                FHE_KS_LAUNCH_G(NW, GMV, RNS, false);                                                                 \
            }                                                                                                         \
        } else {                                                                                                      \
            require(!gal, E_ARG, "folded Galois substitution below N = 4096");                                        \
            FHE_KS_LAUNCH_G(NW, GMV, RNS, false);                                                                     \
        }                                                                                                             \
    } while (0)
#define FHE_KS_LAUNCH(NW, GMV)                                                                                        \
    do {                                                                                                              \
        if constexpr (LOGN == 12 || LOGN == 13) {                                                                     \
            if (rns) {                                                                                                \
                FHE_KS_LAUNCH_R(NW, GMV, true);                                                                       \
            } else {                                                                                                  \
                FHE_KS_LAUNCH_R(NW, GMV, false);                                                                      \
            }                                                                                                         \
        } else {                                                                                                      \
            FHE_KS_LAUNCH_R(NW, GMV, false);                                                                          \
        }                                                                                                             \
    } while (0)
#if defined(FHE_LAB)
    if constexpr (LOGN == 13) {
        // FHE_LAB_KS13_T512 = 1: 512 threads x 16 coefficients, both accumulator sets in registers, tile-only LDS (two
        // workgroups per CU), mixed radix-8 / radix-4 passes; = 2: the same with radix-4 passes throughout
        static const int t512 = FHE_LAB_INT("KS13_T512", 0);
        if (t512 && !gal) {   // (the 512-thread lab instances have no gathering loader: a folded rotation takes the product launch)
            const size_t lds2 = k::lds_words(1u << LOGN) * sizeof(u64);
            const unsigned grid2 = (unsigned)(npolys * kc.L);
#define FHE_KS_T512_R(NW, GMV, RNS)                                                                                \
    allow_big_lds((k::ks_fused_kernel<LOGN, NW, GMV, 512, RNS>), lds2);                                            \
    FHE_LAUNCH("key_switch_fused", (k::ks_fused_kernel<LOGN, NW, GMV, 512, RNS>), dim3(grid2), dim3(512), lds2, s, \
               p, p_stride, o0, o1, out_stride, a0, a1, a_stride, k_.c0.p, k_.c0s.p, k_.c1.p, k_.c1s.p,            \
               kc.dmods(), kc.dtw(), (uint32_t)k_.ndigits, (uint32_t)kc.L, k_.digit_arg(), xhat, xhat_stride, grid2, gal)
#define FHE_KS_T512(NW, GMV)                                                                                       \
    do {                                                                                                           \
        if (rns) {                                                                                                 \
            FHE_KS_T512_R(NW, GMV, true);                                                                          \
        } else {                                                                                                   \
            FHE_KS_T512_R(NW, GMV, false);                                                                         \
        }                                                                                                          \
    } while (0)
            if (t512 == 2) {
                if (narrow) { FHE_KS_T512(true, 2); } else { FHE_KS_T512(false, 2); }
            } else if (t512 == 3) {
                if (narrow) { FHE_KS_T512(true, k::KS_GMAX); } else { FHE_KS_T512(false, k::KS_GMAX); }
            } else {
                if (narrow) { FHE_KS_T512(true, k::GM_MIXED); } else { FHE_KS_T512(false, k::GM_MIXED); }
            }
#undef FHE_KS_T512
#undef FHE_KS_T512_R
            return;
        }
    }
#endif
#if defined(FHE_LAB)
    if constexpr (LOGN == 14) {
        // FHE_LAB_KS14_T512 = 1: 512 threads x 32 coefficients, 256 VGPRs (two waves per SIMD), both accumulator sets in
        // registers, radix-16 passes (GM = 4), the RNS loader; = 2: the same with radix-8 passes
        static const int t512 = FHE_LAB_INT("KS14_T512", 0);
        if (t512 && !gal && k_.digit_arg() == (1u << 8)) {
            const size_t lds2 = k::lds_words(1u << LOGN) * sizeof(u64);
            const unsigned grid2 = (unsigned)(npolys * kc.L);
#define FHE_KS14_T512(NW, GMV)                                                                                         \
    allow_big_lds((k::ks_fused_kernel<LOGN, NW, GMV, 512, true>), lds2);                                               \
    FHE_LAUNCH("key_switch_fused", (k::ks_fused_kernel<LOGN, NW, GMV, 512, true>), dim3(grid2), dim3(512), lds2, s, p, \
               p_stride, o0, o1, out_stride, a0, a1, a_stride, k_.c0.p, k_.c0s.p, k_.c1.p, k_.c1s.p, kc.dmods(),       \
               kc.dtw(), (uint32_t)k_.ndigits, (uint32_t)kc.L, k_.digit_arg(), xhat, xhat_stride, grid2, gal)
            if (t512 == 2) {
                if (narrow) { FHE_KS14_T512(true, 3); } else { FHE_KS14_T512(false, 3); }
            } else {
                if (narrow) { FHE_KS14_T512(true, 4); } else { FHE_KS14_T512(false, 4); }
            }
#undef FHE_KS14_T512
            return;
        }
    }
#endif
    if constexpr (LOGN == 14) {   // (radix-4 passes at N = 8192 measured slower: 0.559 vs 0.532 ms per launch)
        if (plan14 == 4) {
            if (narrow) {
                FHE_KS_LAUNCH(true, 2);
            } else {
                FHE_KS_LAUNCH(false, 2);
            }
            return;
        }
        if (plan14 != 8) {
            if (narrow) {
                FHE_KS_LAUNCH(true, k::GM_MIXED);
            } else {
                FHE_KS_LAUNCH(false, k::GM_MIXED);
            }
            return;
        }
    }
    if (narrow) {
        FHE_KS_LAUNCH(true, k::KS_GMAX);
    } else {
        FHE_KS_LAUNCH(false, k::KS_GMAX);
    }
#undef FHE_KS_LAUNCH
#undef FHE_KS_LAUNCH_R
#undef FHE_KS_LAUNCH_G
    (void)rns;
}

// ---- unfused key switch (kernels_ks.hpp: ks_ntt_kernel + ks_mac_kernel) ----
// Default W budget.  Measured (profiles/r04_ks_unfused_ab.txt): keeping W on-die does not pay -- launch pairs cut to
// 64 ... 256 MiB of W run 8-30 % slower than one pair over the whole batch (the cut launches are small, and stage B
// streams at 3.4 TB/s either way) -- so the budget only bounds the scratch block.
constexpr size_t KS_W_BUDGET_DEFAULT = (size_t)4 << 30;
// Which shapes take the unfused path when the handle says KS_AUTO.  Measured on the MI355X, same process, alternating
// (profiles/r04_ks_unfused_ab.txt; key-switch kernel time per launch, fused vs unfused stage A + stage B):
//   C5  N = 32768, 16 moduli, 16 polynomials   1.198 ms   vs 0.809 + 0.345 = 1.154 ms   (inside the multiply: 3.25 vs 3.33 ms per step)
//   C5  the same, 64 polynomials                4.713 ms   vs 3.396 + 1.308 = 4.704 ms
//   C3  N = 16384,  8 moduli, 512 polynomials   3.545 ms   vs 2.802 + 1.492 = 4.294 ms   (8192-point tiles: 3.34 + 1.46)
//   C2  N =  8192,  4 moduli, 1024 polynomials  0.831 ms   vs 0.553 + 0.420 = 0.973 ms
// Stage A runs at the NTT kernels' rate (20 M 8192-point tiles/s with the two folded stages, 26 M without), but W
// -- L * Lk rows per polynomial, 1 GiB at C5 / 16 -- costs stage B what the transforms gained: a tie at N = 32768, a
// loss below (and since the fused kernel takes rows larger than LDS as 16384-point parts, a loss at N = 32768 too: 1.08 vs
// 1.23 ms, profiles/r04_final3_ks_modes_ab_c5.jsonl).  For launches that FILL the device KS_AUTO therefore stays on the
// fused kernels at every size.
// Launches that do not are the opposite regime (profiles/r04_ks_small_batches_all_modes.txt): a fused launch has
// batch x key moduli workgroups, each walking its digits one after the other, so its time is one workgroup's whatever the
// batch -- 0.054 ms at C2, 0.22 ms at C3, 0.52 / 0.32 ms at C5 -- while stage A of the unfused form has digits x more
// tiles and fills the device from a single ciphertext on: C3 batch 1 0.220 -> 0.055 ms, batch 16 0.225 -> 0.155; C5 batch 1
// 0.32 -> 0.12 ms; C2 batch 1 0.054 -> 0.030, batch 32 0.063 -> 0.054.  The crossover sits between 128 and 256 fused
// workgroups at every size measured (N = 4096 ... 32768): KS_AUTO takes the unfused form while 2 x workgroups <= compute
// units (at least three digits, N >= 4096: below that the second launch costs what the parallelism gains).  Round 5
// refines the rule by the fused launch's LAST round of workgroups: see the end of ks_use_unfused.
inline bool ks_use_unfused(const Ksk &k_, int mode, size_t npolys) {
    if (k_.log_base != 0) return false;                 // base-2^k digits of one row: the fused loader extracts them
    if (mode == KS_UNFUSED || mode == KS_UNFUSED_SUB) return true;
    if (mode == KS_FUSED || mode == KS_FUSED_SUB) return false;
    const Ctx &kc = *k_.ksk_ctx;
    if (kc.logn >= (size_t)FHE_LAB_INT("KS_UNFUSED_MIN_LOGN", 99)) return true;
    if (FHE_LAB_FLAG("NO_KS_SMALL_UNFUSED") || kc.logn < 12 || k_.ndigits < 3) return false;
    const size_t fused_wg = (npolys * kc.L) << (kc.logn > 14 ? kc.logn - 14 : 0);
    const size_t cus = (size_t)device_cus(kc.device);
    // (rows larger than LDS with short digit loops: the 8192-point fused sub-blocks take over earlier -- N = 32768, 4
    // moduli: 96 parts 0.093 vs 0.092 ms, 128 parts 0.112 vs 0.101; profiles/r04_final5_ks_small_launch_ab.jsonl)
    if (kc.logn > 14 && k_.ndigits < 8) return 8 * fused_wg < 3 * cus;
    // Round 5 (profiles/r05_ks_small_launch_stock.jsonl, r05_ks_rounds12_ab.jsonl): the fused kernels have ONE workgroup
    // per CU from N = 8192 on, so a fused launch's time is a step function of its rounds of workgroups, while the unfused
    // form's grows with the work.  (a) N = 4096 (two workgroups per CU, three short digits): unfused only up to a fifth of
    // the CUs' worth (3 moduli: 48 workgroups 0.0310 vs 0.0311 ms, 72: 0.0330 vs 0.0319).  (b) Less than 0.6 of one round
    // (was 0.5; 9 moduli at N = 16384, 144 workgroups: 0.217 vs 0.267 ms).  (c) One or two full rounds and a last round at
    // most half full: 288 / 320 / 384 workgroups at C2 0.095 / 0.105 / 0.115 vs 0.113 / 0.114 / 0.117 ms, at C3 0.349 /
    // 0.382 / 0.443 vs 0.461 / 0.465 / 0.479, at C5 (320 / 384 parts) 0.834 / 0.967 vs 1.040 / 1.045; 576: 0.669 vs 0.727
    // (C3), 1.440 vs 1.562 (C5); with the last round more than half full the fused kernel is ahead again (448: 0.120 vs 0.128).
    if (kc.logn == 12) return 5 * fused_wg <= cus;
    const size_t full = fused_wg / cus, rem = fused_wg % cus;
    // Round 6, the F64 instances (profiles/r06_m_f64_ks_modes_grid.jsonl, r06_n_f64_ks_modes_grid_after.jsonl,
    // r06_p_f64_ks_modes_grid_two_geometries.jsonl; stock sets, relinearise of 8 ... 1,024): their fused kernels are ahead of
    // the integer ones by more than their stage A is, so the windows after a full round shrink.  Below one round nothing
    // changes (N = 8192, 1024-thread instance: 140 workgroups 0.0542 vs 0.0567 ms, 160: 0.0574 vs 0.0571, 255: 0.0754 vs
    // 0.0639; N = 16384: 144: 0.160 vs 0.172, 180: 0.192 vs 0.173).  N = 8192 from the second workgroup per CU on (512-thread
    // instance; a CU's second workgroup costs it ~0.7 of the first, so the steps stay one per CUs' worth): unfused up to 4/9 of
    // the round after the first (320: 0.0926 vs 0.0989; 360: 0.0976 vs 0.1001; 400: 0.1104 vs 0.1018); N = 16384: up to a
    // quarter (288: 0.306 vs 0.336; 360: 0.377 vs 0.354); never after two (560 at N = 8192: 0.157 vs 0.145; 576 at N = 16384:
    // 0.598 vs 0.536).
    if (kc.logn <= 14 && k_.c0f.p && kc.f64_class(0, kc.L)) {
        if (5 * fused_wg <= 3 * cus) return true;
        return full == 1 && rem > 0 && (kc.logn == 13 ? 9 * rem <= 4 * cus : 4 * rem <= cus);
    }
    if (5 * fused_wg <= 3 * cus) return true;
    // (after two full rounds only up to 0.4 of a third: 640 workgroups at C2 0.192 vs 0.182 ms)
    return rem > 0 && ((full == 1 && 2 * rem <= cus) || (full == 2 && 5 * rem <= 2 * cus));
}
// `extra` (optional, G0 == 0): residue rows transformed in place by `extra_grid` more workgroups of the same launch
struct KsExtraFwd {
    u64 *rows = nullptr;      // [npolys][nrows][N], row r under modulus r of the key context, canonical in / out
    u64 poly_stride = 0;
    size_t npolys = 0, nrows = 0;
};
template <int LOGM, int G0>
inline void launch_ks_ntt(const Ksk &k_, bool narrow, bool rns, unsigned grid, hipStream_t s, const u64 *p, u64 p_stride,
                          u64 *w, uint32_t j0, uint32_t jg, uint32_t skip_own, const KsExtraFwd *extra = nullptr) {
    const Ctx &kc = *k_.ksk_ctx;
    const size_t lds = k::lds_words(1u << LOGM) * sizeof(u64);
    const bool with_extra = G0 == 0 && extra != nullptr && extra->rows != nullptr;
    u64 *const erows = with_extra ? extra->rows : nullptr;
    const u64 estride = with_extra ? extra->poly_stride : 0;
    const uint32_t enr = with_extra ? (uint32_t)extra->nrows : 1u;
    const unsigned egrid = with_extra ? (unsigned)(extra->npolys * extra->nrows) : 0u;
#define FHE_KSN(NW, RNS)                                                                                         \
    allow_big_lds((k::ks_ntt_kernel<LOGM, G0, NW, RNS>), lds);                                                   \
    FHE_LAUNCH("ks_digit_ntt", (k::ks_ntt_kernel<LOGM, G0, NW, RNS>), dim3(grid + egrid), dim3(k::ntt_threads_c(LOGM)), \
               lds, s, p, p_stride, w, kc.dmods(), kc.dtw(), (uint32_t)k_.ndigits, j0, jg, k_.digit_arg(), skip_own,   \
               erows, estride, enr, (uint32_t)grid)
    // Round 6: RNS digits under key moduli that are all below 2^50 -> the F64 stage A (whole-row tiles of 4096 ... 16384
    // points; W in canonical words, so stage B is unchanged)
    if constexpr (G0 == 0 && LOGM >= 12 && LOGM <= 14) {
        const int hr = k_.log_base == 0 ? kc.f64_class(0, kc.L) : 0;
#define FHE_KSN_F64(HR)                                                                                                 \
    allow_big_lds((k::ks_ntt_kernel<LOGM, 0, false, true, HR>), lds);                                                   \
    FHE_LAUNCH("ks_digit_ntt_f64", (k::ks_ntt_kernel<LOGM, 0, false, true, HR>), dim3(grid + egrid),                    \
               dim3(k::ntt_threads_c(LOGM)), lds, s, p, p_stride, w, kc.dmods(), kc.dtw_f(), (uint32_t)k_.ndigits, j0,  \
               jg, k_.digit_arg(), skip_own, erows, estride, enr, (uint32_t)grid)
        if (hr == 3) { FHE_KSN_F64(3); return; }
        if (hr == 4) { FHE_KSN_F64(4); return; }
        if (hr == 5) { FHE_KSN_F64(5); return; }
#undef FHE_KSN_F64
    }
    if constexpr (LOGM >= 12) {
        if (rns) {
            if (narrow) { FHE_KSN(true, true); } else { FHE_KSN(false, true); }
            return;
        }
    }
    if (narrow) { FHE_KSN(true, false); } else { FHE_KSN(false, false); }
#undef FHE_KSN
    (void)rns;
}
inline void key_switch_polys_unfused(const Ksk &k_, int mode, const u64 *p, u64 p_stride, u64 *o0, u64 *o1,
                                     u64 out_stride, const u64 *a0, const u64 *a1, u64 a_stride, size_t npolys,
                                     hipStream_t s, const u64 *xhat, u64 xhat_stride, uint32_t gal,
                                     const KsExtraFwd *extra = nullptr) {
    const Ctx &kc = *k_.ksk_ctx;
    const size_t N = kc.n, nd = k_.ndigits, Lk = kc.L;
    const uint32_t logn = (uint32_t)kc.logn;
    // tile geometry of stage A: whole rows up to 16384 points, 8192-point sub-blocks above (and at 16384 on request)
    // (KS_AUTO at N = 16384: sub-block tiles while the whole-row tiles of stage A would leave half the device idle --
    // one ciphertext at 8 moduli: 64 rows, 0.0527 vs 0.0556 ms; four: 256 rows, 0.0684 vs 0.0668)
    // (with extra rows to transform in the same launch -- bfv_mul's c0, c1 -- whole-row tiles: saving that launch is worth
    // more than the smaller tiles, 21 us against 3 us at one ciphertext)
    const bool sub14 = logn == 14 && (mode == KS_UNFUSED_SUB ||
                                      (mode == KS_AUTO && !extra && 2 * npolys * nd * Lk <= (size_t)device_cus(kc.device)));
    const uint32_t logm = logn > 14 ? 13 : sub14 ? 13 : logn;
    const uint32_t g0 = logn - logm;
    bool narrow = !FHE_LAB_FLAG("NO_NARROW");
    for (u64 q : kc.moduli) narrow = narrow && (q >> 60) == 0;
    const bool rns = k_.digit_arg() == (1u << 8);
    // W = [pc][nd][jg][N]: groups of jg key moduli over chunks of pc polynomials, as large as the budget allows
    // (key-modulus groups are cut first: a group's key slice is then still read once per launch pair)
    size_t budget = k_.w_budget.load(std::memory_order_relaxed);
    if (!budget) budget = KS_W_BUDGET_DEFAULT;
    const size_t row_bytes = N * sizeof(u64);
    size_t pc = npolys, jg = Lk;
    if (pc * nd * jg * row_bytes > budget) {
        jg = std::max<size_t>(1, budget / (pc * nd * row_bytes));
        if (jg > Lk) jg = Lk;
        if (pc * nd * jg * row_bytes > budget) pc = std::max<size_t>(1, budget / (nd * jg * row_bytes));
    }
    // (equal groups / chunks: no small tail launch)
    jg = (Lk + ((Lk + jg - 1) / jg) - 1) / ((Lk + jg - 1) / jg);
    pc = (npolys + ((npolys + pc - 1) / pc) - 1) / ((npolys + pc - 1) / pc);
    WsGuard w(pc * nd * jg * row_bytes, s);
    const uint32_t skip_own = xhat != nullptr ? 1u : 0u;
    // the extra rows ride on the stage-A launch only when there is exactly one (whole-row tiles, one chunk of polynomials,
    // one group of key moduli); otherwise they get their own forward transform first, as before
    const bool merge_extra = extra != nullptr && g0 == 0 && logm == logn && pc == npolys && jg == Lk;
    if (extra != nullptr && !merge_extra) {
        k::RowMap m = full_map(kc, extra->nrows);
        m.src_poly_stride = m.dst_poly_stride = extra->poly_stride;
        launch_ntt(kc, false, extra->rows, extra->rows, m, extra->npolys, s);
    }
    const KsExtraFwd *ride = merge_extra ? extra : nullptr;
    for (size_t b0 = 0; b0 < npolys; b0 += pc) {
        const size_t nb = std::min(pc, npolys - b0);
        for (size_t j0 = 0; j0 < Lk; j0 += jg) {
            const size_t njg = std::min(jg, Lk - j0);
            const unsigned grid_a = (unsigned)((nb * nd * njg) << g0);
            const u64 *pp = p + b0 * p_stride;
#define FHE_KSN_CASE(LM)                                                                                         \
    case LM: launch_ks_ntt<LM, 0>(k_, narrow, rns, grid_a, s, pp, p_stride, w.u(), (uint32_t)j0, (uint32_t)njg, skip_own, ride); break;
            if (g0 == 0) {
                switch (logm) {
                    FHE_KSN_CASE(3) FHE_KSN_CASE(4) FHE_KSN_CASE(5) FHE_KSN_CASE(6) FHE_KSN_CASE(7) FHE_KSN_CASE(8)
                    FHE_KSN_CASE(9) FHE_KSN_CASE(10) FHE_KSN_CASE(11) FHE_KSN_CASE(12) FHE_KSN_CASE(13) FHE_KSN_CASE(14)
                    default: throw StatusError(E_ARG, "unsupported key-switch row size");
                }
            } else if (g0 == 1) {
                launch_ks_ntt<13, 1>(k_, narrow, rns, grid_a, s, pp, p_stride, w.u(), (uint32_t)j0, (uint32_t)njg, skip_own);
            } else if (g0 == 2) {
                launch_ks_ntt<13, 2>(k_, narrow, rns, grid_a, s, pp, p_stride, w.u(), (uint32_t)j0, (uint32_t)njg, skip_own);
            } else {
                launch_ks_ntt<13, 3>(k_, narrow, rns, grid_a, s, pp, p_stride, w.u(), (uint32_t)j0, (uint32_t)njg, skip_own);
            }
#undef FHE_KSN_CASE
            const uint32_t cpr = N >= 512 ? (uint32_t)(N / 512) : 1u;
            const unsigned grid_b = (unsigned)((((njg * cpr) + 7) / 8) * nb * 8);
            FHE_LAUNCH("ks_mac", k::ks_mac_kernel, dim3(grid_b), dim3(256), 0, s, w.u(), o0 + b0 * out_stride,
                       o1 + b0 * out_stride, out_stride, a0 ? a0 + b0 * a_stride : nullptr, a1 ? a1 + b0 * a_stride : nullptr,
                       a_stride, k_.c0.p, k_.c1.p, kc.dmods(), (uint32_t)nd, (uint32_t)Lk, (uint32_t)j0, (uint32_t)njg, logn,
                       xhat ? xhat + b0 * xhat_stride : nullptr, xhat_stride, (uint32_t)nb, gal);
        }
    }
}

// KeySwitchingKey::key_switch (:241-320): p [npolys][L][N] PowerBasis (poly stride p_stride) ->
// o0,o1 [npolys][Lk][N] Ntt over ksk_ctx (poly stride out_stride).  If a0/a1 are given (and the
// key lives at the ciphertext level) the result is added to them on the fly.
// xhat (optional, poly stride xhat_stride): the same polynomials in Ntt form, when the caller has them (it
// produced p by an inverse transform): row j of xhat is digit j's transform under key modulus j, which the
// kernels then read instead of recomputing (L of the L * Lk transforms).
// gal != 0 (galois_apply): `xhat` and `a0` are the caller's UNPERMUTED Ntt rows, read through the substitution
// x -> x^gal inside the kernels (needs xhat; a1 must be null).
// extra (optional): residue rows over the key context to be forward-transformed IN PLACE before the sums are formed
// (bfv_mul's c0, c1, which are also a0 / a1): they ride on the unfused form's stage-A launch when that is one launch,
// and get their own launch otherwise.
inline void key_switch_polys(const Ksk &k_, const u64 *p, u64 p_stride, u64 *o0, u64 *o1, u64 out_stride,
                             const u64 *a0, const u64 *a1, u64 a_stride, size_t npolys, hipStream_t s,
                             const u64 *xhat = nullptr, u64 xhat_stride = 0, uint32_t gal = 0,
                             const KsExtraFwd *extra = nullptr) {
    if (FHE_LAB_FLAG("NO_KS_XHAT") && !gal) xhat = nullptr;
    require(!gal || (xhat != nullptr && a1 == nullptr), E_ARG, "galois key switch: needs the Ntt rows, adds to c0 only");
    const Ctx &kc = *k_.ksk_ctx;
    kc.need_device();
    if (!npolys) return;
    {
        const int mode = k_.mode.load(std::memory_order_relaxed);
        if (ks_use_unfused(k_, mode, npolys)) {
            key_switch_polys_unfused(k_, mode, p, p_stride, o0, o1, out_stride, a0, a1, a_stride, npolys, s, xhat, xhat_stride, gal,
                                     extra);
            return;
        }
    }
    if (extra != nullptr && extra->rows != nullptr) {   // (a fused launch: the extra rows' transform is its own launch)
        k::RowMap m = full_map(kc, extra->nrows);
        m.src_poly_stride = m.dst_poly_stride = extra->poly_stride;
        launch_ntt(kc, false, extra->rows, extra->rows, m, extra->npolys, s);
    }
    // N = 16384: whole-row kernel (1024 threads x 16 coefficients, 24 VGPRs spilled) or two 8192-point sub-blocks
    // with the first stage folded into the loader (FHE_KS_SPLIT14=1)
    static const bool split14 = FHE_LAB_INT("KS_SPLIT14", 0) != 0;
    // lab builds, FHE_LAB_KS_HALF13=1: N = 8192 as two 4096-point sub-blocks (512 threads, 34 KiB tile + 32 KiB of LDS
    // accumulators: two workgroups per CU), the first stage folded into the loader
    static const bool half13 = FHE_LAB_INT("KS_HALF13", 0) != 0;
    if (kc.logn <= 12 || (kc.logn == 13 && !half13) || (kc.logn == 14 && !split14)) {
#define FHE_KS_CASE(LN) \
    case LN: launch_ks_fused<LN>(k_, p, p_stride, o0, o1, out_stride, a0, a1, a_stride, npolys, s, xhat, xhat_stride, gal); break;
        switch (kc.logn) {
            FHE_KS_CASE(3) FHE_KS_CASE(4) FHE_KS_CASE(5) FHE_KS_CASE(6) FHE_KS_CASE(7) FHE_KS_CASE(8)
            FHE_KS_CASE(9) FHE_KS_CASE(10) FHE_KS_CASE(11) FHE_KS_CASE(12) FHE_KS_CASE(13) FHE_KS_CASE(14)
            default: throw StatusError(E_ARG, "unsupported key-switch row size");
        }
#undef FHE_KS_CASE
        return;
    }
    // Rows larger than LDS (N >= 32768): one workgroup per part of a row, the first stages folded into its loader.
    bool narrow = !FHE_LAB_FLAG("NO_NARROW");
    for (u64 q : kc.moduli) narrow = narrow && (q >> 60) == 0;
    // N = 32768 / 65536 as two / four 16384-point parts on the N = 16384 kernel (one / two folded stages, knobs.hpp
    // FHE_KS_HALF15: 1 the generic loader, 2 the RNS loader where the key's digits are residue rows) -- 15 % ahead of the
    // 8192-point sub-blocks below once a launch fills the device (profiles/r04_ks_half15_ab.txt).  A launch whose
    // 8192-point sub-blocks all fit the device at once (one workgroup per CU) is a different regime: its time is ONE
    // workgroup's, and a 16384-point part takes 1.7 x as long as an 8192-point one (N = 65536, 4 moduli, 8 polynomials:
    // 128 parts 0.158 ms, 256 sub-blocks 0.129 ms, profiles/r04_final3_n65536_ab.jsonl) -- KS_AUTO takes the small tiles
    // there; KS_FUSED / KS_FUSED_SUB force either form.
    static const int half15 = FHE_LAB_INT("KS_HALF15", FHE_KS_HALF15);
    const int ks_mode = k_.mode.load(std::memory_order_relaxed);
    const bool fits_at_once = ((npolys * kc.L) << (kc.logn - 13)) <= (size_t)device_cus(kc.device);
    const bool parts16k = half15 && ks_mode != KS_FUSED_SUB && (ks_mode == KS_FUSED || !fits_at_once);
    if ((kc.logn == 15 || kc.logn == 16) && parts16k) {
        const size_t lds_ = k::lds_words(1u << 14) * sizeof(u64);
        const unsigned grid = (unsigned)((npolys * kc.L) << (kc.logn - 14));
        const bool rns = half15 == 2 && k_.digit_arg() == (1u << 8);
#define FHE_KS_HALF15_LAUNCH_G(NW, RNS, G0, GALV)                                                                     \
    allow_big_lds((k::ks_fused_kernel<14, NW, k::GM_MIXED, 0, RNS, G0, GALV>), lds_);                                 \
    FHE_LAUNCH("key_switch_fused", (k::ks_fused_kernel<14, NW, k::GM_MIXED, 0, RNS, G0, GALV>), dim3(grid),           \
               dim3(k::ks_threads_c(14)), lds_, s, p, p_stride, o0, o1, out_stride, a0, a1, a_stride, k_.c0.p, k_.c0s.p, \
               k_.c1.p, k_.c1s.p, kc.dmods(), kc.dtw(), (uint32_t)k_.ndigits, (uint32_t)kc.L, k_.digit_arg(), xhat,   \
               xhat_stride, grid, gal)
#define FHE_KS_HALF15_LAUNCH(NW, RNS, G0)                                                                             \
    do {                                                                                                              \
        if (gal) {                                                                                                    \
            FHE_KS_HALF15_LAUNCH_G(NW, RNS, G0, true);                                                                \
        } else {                                                                                                      \
            FHE_KS_HALF15_LAUNCH_G(NW, RNS, G0, false);                                                               \
        }                                                                                                             \
    } while (0)
#define FHE_KS_HALF15_PICK(G0)                                                                                        \
    if (narrow) {                                                                                                     \
        if (rns) { FHE_KS_HALF15_LAUNCH(true, true, G0); } else { FHE_KS_HALF15_LAUNCH(true, false, G0); }            \
    } else {                                                                                                          \
        if (rns) { FHE_KS_HALF15_LAUNCH(false, true, G0); } else { FHE_KS_HALF15_LAUNCH(false, false, G0); }          \
    }
        if (kc.logn == 15) { FHE_KS_HALF15_PICK(1) } else { FHE_KS_HALF15_PICK(2) }
#undef FHE_KS_HALF15_PICK
#undef FHE_KS_HALF15_LAUNCH
#undef FHE_KS_HALF15_LAUNCH_G
        return;
    }
    // 8192-point sub-blocks, logn - 13 folded stages (ks_fused_split_kernel; rounds 1-3: every launch)
#define FHE_KS_SPLIT_LAUNCH_M(G0, LM, NW)                                                                          \
    do {                                                                                                           \
        const size_t lds_ = (k::lds_words(1u << LM) + ((size_t)1 << LM)) * sizeof(u64);                            \
        if (k_.digit_arg() == (1u << 8)) {   /* RNS instance: residue-row digits of same-width moduli */          \
            allow_big_lds((k::ks_fused_split_kernel<G0, LM, NW, true>), lds_);                                     \
            FHE_LAUNCH("key_switch_fused_sub", (k::ks_fused_split_kernel<G0, LM, NW, true>),                           \
                       dim3((unsigned)((npolys * kc.L) << G0)), dim3((1u << LM) / 8), lds_, s, p, p_stride, o0,    \
                       o1, out_stride, a0, a1, a_stride, k_.c0.p, k_.c0s.p, k_.c1.p, k_.c1s.p, kc.dmods(),         \
                       kc.dtw(), (uint32_t)k_.ndigits, (uint32_t)kc.L, k_.digit_arg(), xhat, xhat_stride, gal);    \
        } else {                                                                                                   \
            allow_big_lds((k::ks_fused_split_kernel<G0, LM, NW, false>), lds_);                                    \
            FHE_LAUNCH("key_switch_fused_sub", (k::ks_fused_split_kernel<G0, LM, NW, false>),                          \
                       dim3((unsigned)((npolys * kc.L) << G0)), dim3((1u << LM) / 8), lds_, s, p, p_stride, o0,    \
                       o1, out_stride, a0, a1, a_stride, k_.c0.p, k_.c0s.p, k_.c1.p, k_.c1s.p, kc.dmods(),         \
                       kc.dtw(), (uint32_t)k_.ndigits, (uint32_t)kc.L, k_.digit_arg(), xhat, xhat_stride, gal);    \
        }                                                                                                          \
    } while (0)
#define FHE_KS_SPLIT_LAUNCH(G0, NW) FHE_KS_SPLIT_LAUNCH_M(G0, 13, NW)
#if defined(FHE_LAB)
    if (kc.logn == 13) {
        if (narrow) {
            FHE_KS_SPLIT_LAUNCH_M(1, 12, true);
        } else {
            FHE_KS_SPLIT_LAUNCH_M(1, 12, false);
        }
        return;
    }
#endif
#define FHE_KS_SPLIT_CASE(G0)                                                                                      \
    case 13 + G0:                                                                                                  \
        if (narrow) {                                                                                              \
            FHE_KS_SPLIT_LAUNCH(G0, true);                                                                         \
        } else {                                                                                                   \
            FHE_KS_SPLIT_LAUNCH(G0, false);                                                                        \
        }                                                                                                          \
        break;
    switch (kc.logn) {
        FHE_KS_SPLIT_CASE(1) FHE_KS_SPLIT_CASE(2) FHE_KS_SPLIT_CASE(3)
        default: throw StatusError(E_ARG, "unsupported key-switch row size");
    }
#undef FHE_KS_SPLIT_CASE
#undef FHE_KS_SPLIT_LAUNCH
#undef FHE_KS_SPLIT_LAUNCH_M
}

// Poly::<PowerBasis>::switch_down_to (M/rq/mod.rs:498-507): `iters` applications of switch_down.
// in [npolys][from.L][N] (poly stride in_stride) -> out [npolys][from.L - iters][N] (poly stride out_stride), both
// PowerBasis.  Intermediate levels live in one scratch block at a constant stride, updated in place (a lane reads its
// column's rows before it writes them), so `iters` levels cost one scratch block whatever `iters` is.
inline void switch_down_to_pb(const Ctx &from, size_t iters, const u64 *in, u64 in_stride, u64 *out, u64 out_stride,
                              size_t npolys, hipStream_t s) {
    from.need_device();
    require(from.at_level(iters) != nullptr, E_NO_MORE_CONTEXT, "NoMoreContext");
    if (!npolys) return;
    if (iters == 0) {
        const u64 per = (u64)from.L * from.n, total = per * npolys;
        FHE_LAUNCH("copy_rows", k::copy_rows_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0, s, in, out,
                   in_stride, out_stride, per, total);
        return;
    }
    if (iters == 1) {
        switch_down_polys(from, in, in_stride, out, out_stride, npolys, s);
        return;
    }
    const u64 ts = (u64)(from.L - 1) * from.n;
    WsGuard tmp(npolys * ts * sizeof(u64), s);
    const Ctx *c = &from;
    for (size_t i = 0; i < iters; i++, c = c->next.get()) {
        const bool first = i == 0, last = i + 1 == iters;
        switch_down_polys(*c, first ? in : tmp.u(), first ? in_stride : ts, last ? out : tmp.u(), last ? out_stride : ts,
                          npolys, s);
    }
}

// The same for Ntt polys living over `from` (Ciphertext::switch_down / switch_to_level, F/bfv/ciphertext.rs:148-183,
// one part at a time there): in [npolys][from.L][N] Ntt -> out [npolys][from.L-iters][N] Ntt.  The reference goes
// PowerBasis -> switch_down -> Ntt once per level; the Ntt -> PowerBasis -> Ntt round trips between levels are the
// identity on canonical residues, so one inverse transform, `iters` switch_downs and one forward transform give
// the same values.
inline void switch_down_to_ntt(const Ctx &from, size_t iters, const u64 *in, u64 *out, size_t npolys, hipStream_t s) {
    require(from.at_level(iters) != nullptr, E_NO_MORE_CONTEXT, "NoMoreContext");
    if (!npolys) return;
    const Ctx &to = *from.at_level(iters);
    const u64 stride = (u64)from.L * from.n, ostride = (u64)to.L * to.n;
    WsGuard a(npolys * stride * sizeof(u64), s);
    launch_ntt(from, true, in, a.u(), full_map(from, from.L), npolys, s);
    switch_down_to_pb(from, iters, a.u(), stride, out, ostride, npolys, s);
    launch_ntt(to, false, out, out, full_map(to, to.L), npolys, s);
}

// key switch followed by the level fix-up and "+= (a0, a1)" used by relinearise / rotate:
// out0/out1 [npolys][Lct][N] (poly stride out_stride) = a + switch_down_to(key_switch(p), ct_ctx)
inline void key_switch_add(const Ksk &k_, const u64 *p, u64 p_stride, const u64 *a0, const u64 *a1, u64 a_stride,
                           u64 *out0, u64 *out1, u64 out_stride, size_t npolys, hipStream_t s,
                           const u64 *xhat = nullptr, u64 xhat_stride = 0, uint32_t gal = 0,
                           const KsExtraFwd *extra = nullptr) {
    const Ctx &kc = *k_.ksk_ctx, &cc = *k_.ct_ctx;
    const long iters = kc.niterations_to(cc);
    if (iters == 0) {
        key_switch_polys(k_, p, p_stride, out0, out1, out_stride, a0, a1, a_stride, npolys, s, xhat, xhat_stride, gal, extra);
        return;
    }
    require(extra == nullptr, E_ARG, "extra rows ride only on a key switch at the ciphertext's level");
    require(!gal, E_ARG, "the folded Galois substitution needs the key at the ciphertext's level");
    const u64 kstride = (u64)kc.L * kc.n, cstride = (u64)cc.L * cc.n;
    WsGuard r0(npolys * kstride * sizeof(u64), s), r1(npolys * kstride * sizeof(u64), s);
    WsGuard d0(npolys * cstride * sizeof(u64), s), d1(npolys * cstride * sizeof(u64), s);
    key_switch_polys(k_, p, p_stride, r0.u(), r1.u(), kstride, nullptr, nullptr, 0, npolys, s, xhat, xhat_stride);
    switch_down_to_ntt(kc, (size_t)iters, r0.u(), d0.u(), npolys, s);
    switch_down_to_ntt(kc, (size_t)iters, r1.u(), d1.u(), npolys, s);
    // out = a + d  (strided copy of a -- or of d when there is no addend -- then add)
    const u64 per = cstride, total = per * npolys;
    struct {
        const u64 *a, *d;
        u64 *o;
    } jobs[2] = {{a0, d0.u(), out0}, {a1, d1.u(), out1}};
    for (auto &jb : jobs) {
        if (jb.a) {
            FHE_LAUNCH("copy_rows", k::copy_rows_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0, s,
                       jb.a, jb.o, a_stride, out_stride, per, total);
            for (size_t i = 0; i < npolys; i++) ew_op(cc, jb.o + i * out_stride, jb.d + i * cstride, 1, k::EW_ADD, s);
        } else {
            FHE_LAUNCH("copy_rows", k::copy_rows_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0, s,
                       jb.d, jb.o, cstride, out_stride, per, total);
        }
    }
}

// --------------------------------------------------- PIR / RGSW / inner sum ----
// dot_product_scalar (F/bfv/ops/dot_product.rs:54-180) / rq::dot_product (M/rq/ops.rs:449-570)
inline void dot_product_scalar(const Ctx &c, size_t nparts, size_t count, const u64 *cts, bool cts_shared,
                               const u64 *pts, bool pts_shared, u64 *out, size_t batch, hipStream_t s) {
    c.need_device();
    require(count > 0, E_EMPTY_DOT, "EmptyDotProduct: no operands");
    if (!batch || !nparts) return;
    const u64 pl = (u64)c.L * c.n;
    require(batch <= 65535 && nparts <= 65535, E_ARG, "dot product: batch / parts exceed the grid limits");
    const u64 cstride = cts_shared ? (u64)0 : (u64)count * nparts * pl, pstride = pts_shared ? (u64)0 : (u64)count * pl;
    const dim3 block(EW_THREADS);
    if (nparts == 1) {
        FHE_LAUNCH("dot_product", (k::dot_kernel<1>), dim3(blocks_for(pl / 2, EW_THREADS), 1, (unsigned)batch), block, 0, s,
                   cts, cstride, pts, pstride, out, c.dmods(), c.dpow2(), (uint32_t)nparts, (uint32_t)count,
                   (uint32_t)c.logn, pl);
    } else {  // two parts per lane (a ciphertext), further parts in extra grid rows
        FHE_LAUNCH("dot_product", (k::dot_kernel<2>), dim3(blocks_for(pl / 2, EW_THREADS), (unsigned)((nparts + 1) / 2),
                                                          (unsigned)batch),
                   block, 0, s, cts, cstride, pts, pstride, out, c.dmods(), c.dpow2(), (uint32_t)nparts, (uint32_t)count,
                   (uint32_t)c.logn, pl);
    }
}

// `Ciphertext * Plaintext` (F/bfv/ops/mod.rs:229-257)
inline void mul_plain(const Ctx &c, size_t nparts, const u64 *ct, const u64 *pt, bool pt_shared, u64 *out,
                      size_t batch, hipStream_t s) {
    c.need_device();
    if (!batch || !nparts) return;
    const u64 pl = (u64)c.L * c.n;
    require(batch <= 65535 && nparts <= 65535, E_ARG, "mul_plain: batch / parts exceed the grid limits");
    // (the kernel moves 16 bytes per lane: every device allocator hands out 256-byte aligned blocks and rows are whole
    // multiples of 64 bytes, so only a pointer into the middle of a coefficient pair can fail this)
    require((((uintptr_t)ct | (uintptr_t)pt | (uintptr_t)out) & 15) == 0, E_ARG, "mul_plain: buffers must be 16-byte aligned");
    FHE_LAUNCH("mul_plain", k::mul_plain_kernel, dim3(blocks_for(pl / 2, EW_THREADS), 1, (unsigned)batch),
               dim3(EW_THREADS), 0, s, ct, pt, pt_shared ? (u64)0 : pl, out, c.dmods(), (uint32_t)nparts,
               (uint32_t)c.logn, pl);
}

// GaloisKey::relinearize (F/bfv/keys/galois_key.rs:63-86): ct, out [batch][2][L][N] Ntt
inline void galois_apply(const Ksk &ks, size_t exponent, const u64 *ct, u64 *out, size_t batch, hipStream_t s) {
    const Ctx &cc = *ks.ct_ctx;
    const u64 PL = (u64)cc.L * cc.n;
    if (!batch) return;
    const size_t e = exponent % (2 * cc.n);
    require((e & 1) == 1, E_INVALID_SUBST, "InvalidSubstitutionExponent");
    // Round 5: no separate permutation pass from N = 4096 on when the key sits at the ciphertext's level (every
    // EvaluationKey the reference builds by default; smaller rows keep the copying path and its kernel instances) -- the Ntt-domain substitution is a gather, and its three consumers read through it:
    //   c2 = PowerBasis(substitute(c1))     the inverse transform's loader            (ntt_kernel<true, ..., GATHER>)
    //   digit j under key modulus j         = row j of substitute(c1): the key switch's own-row read
    //   out0 = key_switch0 + substitute(c0) the key switch's addend read              (galois_key.rs:66-79)
    // (before: substitute_kernel wrote both permuted parts -- 2 polynomials out, 3 row sets read back: 9 % of a rotation)
    // (the consumers gather from `ct` while `out` is being written: a caller that rotates in place takes the copying path)
    const bool overlap = out < ct + batch * 2 * PL && ct < out + batch * 2 * PL;
    if (cc.logn >= 12 && ks.ksk_ctx->niterations_to(cc) == 0 && !overlap && !FHE_LAB_FLAG("NO_GALOIS_FOLD")) {
        WsGuard c2(batch * PL * sizeof(u64), s);
        k::RowMap m = full_map(cc, cc.L);
        m.src_poly_stride = 2 * PL;
        m.dst_poly_stride = PL;
        m.subst_exp = (uint32_t)e;
        launch_ntt(cc, true, ct + PL, c2.u(), m, batch, s);
        key_switch_add(ks, c2.u(), PL, ct, nullptr, 2 * PL, out, out + PL, 2 * PL, batch, s, ct + PL, 2 * PL, (uint32_t)e);
        return;
    }
    // substitute both parts at once: sub [b][2][L][N]; c2 = PowerBasis(substitute(c1))
    WsGuard sub(batch * 2 * PL * sizeof(u64), s), c2(batch * PL * sizeof(u64), s);
    substitute_polys(cc, exponent, ct, sub.u(), batch * 2, true, s);
    k::RowMap m = full_map(cc, cc.L);
    m.src_poly_stride = 2 * PL;
    m.dst_poly_stride = PL;
    launch_ntt(cc, true, sub.u() + PL, c2.u(), m, batch, s);
    // out0 = key_switch0 + substitute(c0) ; out1 = key_switch1   (galois_key.rs:66-79)
    // (substitute(c1) in Ntt form doubles as the transforms of digit j under key modulus j)
    key_switch_add(ks, c2.u(), PL, sub.u(), nullptr, 2 * PL, out, out + PL, 2 * PL, batch, s, sub.u() + PL, 2 * PL);
}

// `&Ciphertext * &RGSWCiphertext` (F/bfv/rgsw_ciphertext.rs:122-156)
inline void rgsw_mul(const Ksk &k0, const Ksk &k1, const u64 *ct, u64 *out, size_t batch, hipStream_t s) {
    const Ctx &cc = *k0.ct_ctx;
    require(k0.ct_ctx->same_ring(*k1.ct_ctx) && k0.ksk_ctx->same_ring(*k0.ct_ctx) &&
                k1.ksk_ctx->same_ring(*k1.ct_ctx),
            E_PARAMETER_MISMATCH, "RGSW: both key-switching keys must live at the ciphertext level");
    cc.need_device();
    if (!batch) return;
    const u64 PL = (u64)cc.L * cc.n;
    WsGuard pb(batch * 2 * PL * sizeof(u64), s), t(batch * 2 * PL * sizeof(u64), s);
    launch_ntt(cc, true, ct, pb.u(), full_map(cc, cc.L), batch * 2, s);   // ct0, ct1 -> PowerBasis
    // (c0, c1) = ksk0.key_switch(ct0);  out = (c0, c1) + ksk1.key_switch(ct1)
    key_switch_polys(k0, pb.u(), 2 * PL, t.u(), t.u() + PL, 2 * PL, nullptr, nullptr, 0, batch, s, ct, 2 * PL);
    key_switch_polys(k1, pb.u() + PL, 2 * PL, out, out + PL, 2 * PL, t.u(), t.u() + PL, 2 * PL, batch, s, ct + PL,
                     2 * PL);
}

// EvaluationKey::computes_inner_sum (F/bfv/keys/evaluation_key.rs:56-100)
inline void inner_sum(const Ksk *const *gks, const size_t *exps, size_t ngk, const u64 *ct, u64 *out, size_t batch,
                      hipStream_t s) {
    require(ngk > 0, E_ARG, "inner sum needs at least one Galois key");
    const Ctx &cc = *gks[0]->ct_ctx;
    cc.need_device();
    if (!batch) return;
    const u64 total = (u64)batch * 2 * cc.L * cc.n;
    WsGuard tmp(total * sizeof(u64), s);
    FHE_HIP_CHECK(hipMemcpyAsync(out, ct, total * sizeof(u64), hipMemcpyDeviceToDevice, s));
    for (size_t i = 0; i < ngk; i++) {
        require(gks[i]->ct_ctx->same_ring(cc), E_PARAMETER_MISMATCH, "inner sum: Galois keys of different levels");
        galois_apply(*gks[i], exps[i], out, tmp.u(), batch, s);
        ew_op(cc, out, tmp.u(), batch * 2, k::EW_ADD, s);
    }
}

// SecretKey::try_decrypt, small-plaintext branch (F/bfv/keys/secret_key.rs:198-247):
// ct [batch][nparts][L][N] Ntt, s_ntt [L][N] -> out [batch][N] in [0, t).
inline void decrypt(const Scaler &sc, u64 t, const u64 *s_ntt, const u64 *ct, size_t nparts, u64 *out, size_t batch,
                    hipStream_t s) {
    const Ctx &cc = *sc.from, &pc = *sc.to;
    cc.need_device();
    require(nparts >= 1, E_ARG, "a ciphertext has at least one part");
    // (deep levels may be shorter than the plaintext context; only q_0 has to agree, :232-234)
    require(pc.L >= 1 && pc.moduli[0] == cc.moduli[0], E_PARAMETER_MISMATCH,
            "the plaintext context must start with the first ciphertext modulus");
    if (!batch) return;
    const ModConsts tm = make_mod_consts(t);
    const u64 PL = (u64)cc.L * cc.n;

    // phase and scaled plaintext are secret-dependent: cleared before the blocks go back to the pool
    WsGuard ph(batch * PL * sizeof(u64), s, true), d(batch * pc.L * cc.n * sizeof(u64), s, true);
    FHE_LAUNCH("phase", k::phase_kernel, dim3(blocks_for(PL, EW_THREADS), (unsigned)batch), dim3(EW_THREADS), 0, s, ct,
               s_ntt, ph.u(), cc.dmods(), (uint32_t)nparts, (uint32_t)cc.logn, PL);
    launch_ntt(cc, true, ph.u(), ph.u(), full_map(cc, cc.L), batch, s);
    scale_polys(sc, ph.u(), d.u(), batch, false, s);
    DevMod q0, tmd;
    static_assert(sizeof(DevMod) == sizeof(ModConsts), "DevMod layout");
    std::memcpy(&q0, &cc.root->mods[0], sizeof(DevMod));  // (host tables live on the chain's root)
    std::memcpy(&tmd, &tm, sizeof(DevMod));
    const u64 total = (u64)batch * cc.n;
    FHE_LAUNCH("decrypt_tail", k::decrypt_tail_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0, s, d.u(),
               (u64)pc.L * cc.n, out, q0, tmd, (uint32_t)cc.logn, total);
}

// EvaluationKey::expands (F/bfv/keys/evaluation_key.rs:192-256): ct [batch][2][L][N] ->
// out [size][batch][2][L][N] (slot-major, so that the 2^l * batch ciphertexts a level works on
// are contiguous and go through one batched Galois key switch).
inline void expand(const Ksk *const *gks, size_t nlevels, const u64 *ct, u64 *out, size_t size, size_t batch,
                   hipStream_t s) {
    require(nlevels > 0, E_ARG, "expansion needs at least one Galois key");
    const Ctx &cc = *gks[0]->ct_ctx;
    cc.need_device();
    if (size < 1 || size > cc.n)
        throw StatusError(E_EXPANSION_SIZE,
                          "InvalidExpansionSize: size " + std::to_string(size) + ", degree " + std::to_string(cc.n));
    size_t level = 0;
    while (((size_t)1 << level) < size) level++;
    if (level > nlevels)
        throw StatusError(E_EXPANSION_UNSUPPORTED, "expansion to level " + std::to_string(level) +
                                                       " needs the Galois keys of (N >> l) + 1, l < level");
    if (!batch) return;
    const u64 PL = (u64)cc.L * cc.n, CT = 2 * PL;
    FHE_HIP_CHECK(hipMemcpyAsync(out, ct, batch * CT * sizeof(u64), hipMemcpyDeviceToDevice, s));
    if (level == 0) return;
    for (size_t l = 0; l < level; l++)
        require(gks[l]->ct_ctx->same_ring(cc), E_PARAMETER_MISMATCH, "expansion: Galois keys of different levels");
    // monomials -x^(N - 2^l) in Ntt form (the reference keeps them in the EvaluationKey, :467-474)
    WsGuard mono(level * PL * sizeof(u64), s), sub(((size_t)1 << (level - 1)) * batch * CT * sizeof(u64), s);
    FHE_HIP_CHECK(hipMemsetAsync(mono.u(), 0, level * PL * sizeof(u64), s));
    FHE_LAUNCH("monomial", k::monomial_kernel, dim3(blocks_for(level * cc.L, 64)), dim3(64), 0, s, mono.u(), cc.dmods(),
               (uint32_t)level, (uint32_t)cc.L, (uint32_t)cc.logn);
    launch_ntt(cc, false, mono.u(), mono.u(), full_map(cc, cc.L), level, s);
    for (size_t l = 0; l < level; l++) {
        const size_t step = (size_t)1 << l, cnt = step * batch;
        const size_t nhigh = std::min(step, size - step) * batch;
        galois_apply(*gks[l], (cc.n >> l) + 1, out, sub.u(), cnt, s);
        FHE_LAUNCH("expand_step", k::expand_step_kernel, dim3(blocks_for(PL, EW_THREADS), (unsigned)(cnt * 2)),
                   dim3(EW_THREADS), 0, s, out, sub.u(), out + step * batch * CT, mono.u() + l * PL, cc.dmods(),
                   (uint32_t)cc.logn, PL, (uint32_t)(nhigh * 2));
    }
}

// --------------------------------------------------------------------- Multiplicator ----
// Execution options live on the handle (not in process globals): concurrent callers that share a Mul all see the
// same, atomically read values, and a call reads them exactly once on entry.
//   chunk   ciphertext pairs per pipeline pass; 0 = default (equal chunks under a workspace budget)
//   streams 2 (default): the chunks of a multiply alternate between the caller's stream and an internal one
//           (fork / join through events): while one chunk's launch drains, the other chunk's kernels fill the
//           idle workgroup slots, which makes small, cache-friendly chunks affordable (+4.5 % at C2).
//           1: the whole pipeline on the caller's stream (exact per-kernel attribution for profiling).
struct Mul {
    const Scaler *ext_lhs = nullptr, *ext_rhs = nullptr, *down = nullptr;
    const Ksk *rk = nullptr;
    bool mod_switch = false;
    const Ctx *base = nullptr, *mulc = nullptr;
    std::atomic<size_t> chunk{0}, streams{2};
    size_t out_parts() const { return rk ? 2 : 3; }
    size_t out_rows() const { return mod_switch ? base->L - 1 : base->L; }
};
// The internal second stream of a (device, caller stream) pair.  The fork / join events are taken from a pool PER CALL
// (an event that two concurrent callers record and wait on would tie their orderings together); concurrent callers on
// the SAME user stream still share its one internal stream and therefore serialise there, which is all stream order
// promises them anyway.  fhe_stream_destroy / fhe_workspace_trim drop what belongs to a stream that is gone.
class AuxStreams {
public:
    static AuxStreams &get() {
        static AuxStreams a;
        return a;
    }
    // The internal stream that shadows `user` on `device`; the caller holds it until done(device, user).
    // `tag_key`: `user` is not a stream but the address of a static tag (host_sliced's three internal streams).
    // At most CAP user streams keep an internal one: beyond that the least recently used idle entry goes (stream
    // synchronised and destroyed, its scratch blocks freed) -- which is also what eventually collects the entries of
    // user streams that no longer exist (a host's own streams: the engine is never told, and cannot ask, see Workspace).
    hipStream_t stream_for(int device, hipStream_t user, bool tag_key = false) {
        std::vector<std::pair<int, hipStream_t>> victims;
        hipStream_t out;
        {
            std::lock_guard<std::mutex> lk(mu);
            const std::pair<int, hipStream_t> key{device, user};
            Entry &a = reg[key];
            if (!a.aux) {
                // A parked stream of this device first (retire): the runtime multiplexes a process's streams onto a few
                // hardware queues (four by default), assigned round-robin at CREATION -- after a dozen streams have come and
                // gone a new one can land on the caller's own queue, and the two lanes of a multiply then run one after the
                // other (bench.py's legs trim the workspace between configs: after its microbenchmarks' sixteen short-lived
                // streams the stock n = 8192 mul_and_relin fell from 182 k to 165 k ops/s,
                // profiles/r06_x_aux_priority_ab_rejected.jsonl, rows of the `before` build).  A stream that is kept keeps its queue.
                for (auto it = parked.begin(); it != parked.end(); ++it)
                    if (it->first == device) {
                        a.aux = it->second;
                        parked.erase(it);
                        break;
                    }
            }
            if (!a.aux) {
                hipStream_t made = nullptr;
                const hipError_t err = hipStreamCreateWithFlags(&made, hipStreamNonBlocking);
                if (err != hipSuccess) {   // (no half-made entry stays behind: a later eviction would retire a NULL stream)
                    reg.erase(key);
                    throw StatusError(E_HIP, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(err));
                }
                a.aux = made;
            }
            a.tag_key = tag_key;
            if (!tag_key) a.users++;      // (tag keys are never evicted: their holders do not report back)
            a.last_use = ++tick;
            out = a.aux;
            while (reg.size() > CAP) {
                auto lru = reg.end();
                for (auto it = reg.begin(); it != reg.end(); ++it)
                    if (!it->second.tag_key && it->second.users == 0 && (lru == reg.end() || it->second.last_use < lru->second.last_use))
                        lru = it;
                if (lru == reg.end()) break;
                victims.emplace_back(lru->first.first, lru->second.aux);
                reg.erase(lru);
            }
        }
        retire(victims);
        return out;
    }
    void done(int device, hipStream_t user) {
        std::lock_guard<std::mutex> lk(mu);
        auto it = reg.find({device, user});
        if (it != reg.end() && it->second.users > 0) it->second.users--;
    }
    hipEvent_t take_event() {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!pool.empty()) {
                hipEvent_t e = pool.back();
                pool.pop_back();
                return e;
            }
        }
        hipEvent_t e;
        FHE_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return e;
    }
    void give_event(hipEvent_t e) {
        std::lock_guard<std::mutex> lk(mu);
        pool.push_back(e);
    }
    // the internal stream that shadows `user` on `device` (device < 0: on any device; all: every internal stream and the
    // pooled events as well) is synchronised and destroyed, and the scratch blocks keyed to it are freed (ADVICE r03:
    // they used to stay in the pool under a handle that no longer existed)
    void drop(int device, hipStream_t user, bool all) {
        std::vector<std::pair<int, hipStream_t>> victims;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto it = reg.begin(); it != reg.end();) {
                if (all || ((device < 0 || it->first.first == device) && it->first.second == user)) {
                    if (it->second.aux) victims.emplace_back(it->first.first, it->second.aux);
                    it = reg.erase(it);
                } else {
                    ++it;
                }
            }
            if (all) {
                for (hipEvent_t e : pool) (void)hipEventDestroy(e);
                pool.clear();
            }
        }
        retire(victims);
    }
    size_t count() {
        std::lock_guard<std::mutex> lk(mu);
        return reg.size();
    }
    static constexpr size_t CAP = 32;

private:
    struct Entry {
        hipStream_t aux = nullptr;
        bool tag_key = false;
        unsigned users = 0;
        uint64_t last_use = 0;
    };
    // An internal stream that is no longer wanted: synchronised, its scratch blocks freed -- and the handle PARKED for the
    // next stream_for on its device (at most PARK of them; the rest are destroyed).  A stream holds no device memory.
    void retire(const std::vector<std::pair<int, hipStream_t>> &victims) {
        for (const auto &v : victims) {
            hipStream_t aux = v.second;
            if (!aux) continue;
            (void)hipStreamSynchronize(aux);
            Workspace::get().drop_internal_stream(aux);   // (its work is over: plain frees)
            bool keep = false;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (parked.size() < PARK) {
                    parked.emplace_back(v.first, aux);
                    keep = true;
                }
            }
            if (!keep) (void)hipStreamDestroy(aux);
        }
    }
    static constexpr size_t PARK = 8;
    std::vector<std::pair<int, hipStream_t>> parked;
    std::mutex mu;
    std::map<std::pair<int, hipStream_t>, Entry> reg;
    std::vector<hipEvent_t> pool;
    uint64_t tick = 0;
};

inline void *Workspace::acquire(size_t bytes, hipStream_t s) {
    int dev = 0;
    FHE_HIP_CHECK(hipGetDevice(&dev));
    std::unique_lock<std::mutex> lk(mu);
    Block *best = nullptr;
    for (auto &b : blocks)
        if (!b.in_use && b.device == dev && b.stream == s && b.bytes >= bytes && (!best || b.bytes < best->bytes))
            best = &b;
    if (!best) {
        // growing.  Idle blocks of this (device, stream) that are too small go back to the pool in stream order (the
        // allocation below may reuse their memory, ordered behind whatever still reads them), the limits are
        // enforced, then the new block is taken from the device's pool on this stream.
        for (auto &b : blocks)
            if (!b.in_use && b.device == dev && b.stream == s && b.ptr) free_block_locked(b, true);
        compact_locked();
        enforce_limits_locked(s, dev, bytes);
        Block nb;
        sync_threshold_locked(dev);
        nb.ptr = DevPools::get().alloc(dev, bytes, s, DevPools::SCRATCH);
        nb.bytes = bytes ? bytes : 8;
        nb.stream = s;
        nb.device = dev;
        blocks.push_back(nb);
        best = &blocks.back();
    }
    best->in_use = true;
    best->last_use = ++tick;
    void *const ptr = best->ptr;
    flush_trims(lk);   // (after the new block is taken: it may reuse what the eviction just returned to the pool)
    return ptr;
}

// How a batch is cut into chunks, and whether the chunks alternate between two streams.
// One stream: every launch should cover >> 512 workgroup slots (64-pair chunks are 15 % slower at C2), but beyond
// ~3 GiB of workspace nothing is gained and the step-to-step reuse in the 256 MiB Infinity Cache is lost (chunks of
// 256-512 pairs measured 1.5 % ahead of one 1024-pair chunk): equal chunks under that budget.
// Two streams: while one chunk's launch drains, the other chunk's kernels fill the idle workgroup slots, so small,
// cache-friendly chunks pay: 384 MiB of workspace per chunk (64 pairs at C2: 168.1 k against 163.4-166.9 k ops/s
// for 128-pair chunks, same box; 48 pairs 161.8-163.2 k), but at least 8 pairs (C5, 90 MiB per pair: chunks of 8-12
// 4.46-4.51 k ops/s, chunks of 4 4.31-4.34 k).  Fewer than four such chunks do not overlap enough to pay for the
// smaller launches (C5 at batch 16: 4.04 k in two chunks of 8, 4.14-4.20 k in one chunk): the batch then takes the
// one-stream cut.  (profiles/r02_chunk_sweeps.txt)
struct ChunkPlan {
    size_t chunk;
    bool dual;
};
inline ChunkPlan plan_chunks(const Ctx &base, const Ctx &mulc, size_t batch, size_t chunk_opt, size_t streams_opt) {
    auto nchunks = [&](size_t c) { return (batch + c - 1) / c; };
    if (chunk_opt) {
        const size_t c = std::min(batch, chunk_opt);
        return ChunkPlan{c, streams_opt >= 2 && nchunks(c) >= 2};
    }
    const size_t per_ct = (7 * mulc.L + 7 * base.L) * mulc.n * sizeof(u64);
    auto equal_chunks = [&](size_t budget, size_t min_pairs) {
        const size_t cap = std::max<size_t>(min_pairs, std::min<size_t>(budget / per_ct, 4096));
        const size_t nc = (batch + cap - 1) / cap;
        return (batch + nc - 1) / nc;
    };
    // Round 5 (profiles/r05_plan14_ab.jsonl: same box, alternating builds; r05_chunk_sweep_n16384.jsonl, r05_chunk_sweep_c5.jsonl):
    // at N = 16384 the small-chunk plan below is skipped -- two or three large chunks alternating between the two streams
    // (the 3 GiB cut at the end) run the reference's stock n = 16384 set 6.7 % (256 pairs) / 8.4 % (1,024 pairs) faster and
    // a 12-moduli basis 3.6 %, at the price of 1.0-1.6 % on 4- and 8-moduli bases: at this size every NTT-type kernel is one
    // 1024-thread workgroup per CU and so is the key switch, so small launches are mostly tail (9 digits x 32 pairs = 288
    // key-switch workgroups: one round and an eighth).  N = 8192 (C2, stock n = 8192: unchanged in the same A/B) and
    // N = 32768 (C5 at 32 / 64 / 128 pairs, 8 moduli at 128 pairs: the default plan is the best cell of the sweep) keep it.
    if (streams_opt >= 2 && base.logn != 14) {
        // Two streams: chunks small enough that a chunk's intermediates mostly stay in the 256 MiB Infinity Cache, at
        // least four of them.  Round 3 (profiles/r03_chunk_sweep_event_free.jsonl, C2): 64 ... 256 pairs per chunk are
        // within 1 % of each other, 96 - 128 best at 8,192 pairs -- but 66 pairs (what the 384 MB budget gave there)
        // lost 5 %: 66 * 16 rows is two full rounds of the 512 resident workgroups plus a nearly empty third.  Chunks
        // of 64 pairs or more are therefore cut to a multiple of 32.
        for (const size_t budget : {(size_t)768 << 20, (size_t)384 << 20}) {
            size_t c = equal_chunks(budget, 8);
            if (c >= 64) {
                c = c / 32 * 32;
                // (round 5: rounding down can leave a small ragged last chunk -- 1,024 pairs of the stock n = 4096 set: 342 ->
                // 320 = 320 + 320 + 320 + 64; equal chunks over the same count instead: 4 x 256, profiles/r05_plan_ragged_ab.jsonl)
                const size_t nc = nchunks(c), ce = (batch + nc - 1) / nc;
                if (ce >= 64 && ce % 32 == 0) c = ce;
            }
            if (nchunks(c) >= 4) return ChunkPlan{c, true};
        }
    }
    // (the large cut is taken on up to two lanes, so it is clamped to a quarter of the device's workspace bound
    // per lane -- a host with a tight fhe_workspace_set_limit, or a small device, must not be planned past its own limit)
    size_t total_limit = 0;
    Workspace::get().get_limits(nullptr, &total_limit);
    size_t big = (size_t)3 << 30;
    if (total_limit) big = std::min(big, std::max<size_t>(per_ct, total_limit / 4));
    const size_t c = equal_chunks(big, 1);
    return ChunkPlan{c, streams_opt >= 2 && nchunks(c) >= 2};
}

// Ciphertext::switch_down (F/bfv/ciphertext.rs:148-161): ct [b][nparts][L][N] Ntt -> [b][nparts][L-1][N] Ntt
inline void bfv_switch_down(const Ctx &c, size_t nparts, const u64 *ct, u64 *out, size_t batch, hipStream_t s) {
    c.need_device();
    require(c.next != nullptr, E_NO_MORE_CONTEXT, "NoMoreContext");
    switch_down_to_ntt(c, 1, ct, out, batch * nparts, s);
}

// Multiplicator::multiply (F/bfv/ops/mul.rs:165-243) on `batch` ciphertext pairs.
// `&ct * &ct` with any number of parts (F/bfv/ops/mod.rs:259-358), the generic (unfused) pipeline:
// lhs [batch][la][L][N], rhs [batch][lb][L][N] -> out [batch][la+lb-1][L][N], all Ntt.
inline void bfv_tensor(const Mul &m, size_t la, size_t lb, const u64 *lhs, const u64 *rhs, u64 *out, size_t batch,
                       hipStream_t s) {
    const Ctx &b = *m.base, &e = *m.mulc;
    b.need_device();
    require(la >= 1 && lb >= 1, E_MUL_POLY_COUNT, "a ciphertext has at least one part");
    if (!batch) return;
    const u64 PL = (u64)b.L * b.n, PK = (u64)e.L * e.n;
    const size_t lo = la + lb - 1;
    // bound the workspace like bfv_mul does: chunks of ciphertext pairs
    const size_t per_pair = (la + lb + lo) * PK * sizeof(u64);
    const size_t chunk = std::max<size_t>(1, std::min<size_t>({batch, (size_t)32768, ((size_t)8 << 30) / per_pair}));
    WsGuard ea(chunk * la * PK * sizeof(u64), s), eb(chunk * lb * PK * sizeof(u64), s), ten(chunk * lo * PK * sizeof(u64), s);
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = std::min(chunk, batch - b0);
        scale_polys(*m.ext_lhs, lhs + b0 * la * PL, ea.u(), nb * la, true, s);
        scale_polys(*m.ext_rhs, rhs + b0 * lb * PL, eb.u(), nb * lb, true, s);
        FHE_LAUNCH("tensor_general", k::tensor_general_kernel, dim3(blocks_for(PK, EW_THREADS), (unsigned)lo, (unsigned)nb),
                   dim3(EW_THREADS), 0, s, ea.u(), eb.u(), ten.u(), e.dmods(), (uint32_t)la, (uint32_t)lb,
                   (uint32_t)e.logn, PK);
        scale_polys(*m.down, ten.u(), out + b0 * lo * PL, nb * lo, true, s);
    }
}

// Workspace per chunk of nb pairs (all slot-major so that every step is ONE launch over
// nb * {2,3} polynomials):  extL/extR [nb][2][K][N] extended operands,  ten [3][nb][K][N]
// tensor,  d [3][nb][L][N] down-scaled parts (c0, c1 Ntt; c2 PowerBasis for the key switch).
inline void bfv_mul(const Mul &m, const u64 *lhs, const u64 *rhs, u64 *out, size_t batch, hipStream_t s0) {
    const Ctx &b = *m.base, &e = *m.mulc;
    b.need_device();
    const size_t L = b.L, K = e.L, N = b.n;
    const u64 PL = (u64)L * N, PK = (u64)K * N;
    const size_t parts = m.out_parts();
    if (!batch) return;
    const ChunkPlan plan = plan_chunks(b, e, batch, m.chunk.load(std::memory_order_relaxed),
                                       m.streams.load(std::memory_order_relaxed));
    const size_t chunk = plan.chunk;
    // the extenders copy the shared prefix rows verbatim; when both share all L rows the tensor
    // kernel reads those rows from the inputs directly and the copy is skipped
    const bool skip_copy = m.ext_lhs->ncommon == L && m.ext_rhs->ncommon == L && !FHE_LAB_FLAG("NO_SKIP_COPY");
    struct ChunkWs {   // (ext: the extended lhs parts of the chunk, then the extended rhs parts)
        WsGuard ext, ten, d, pre;
        ChunkWs(size_t chunk, u64 PK, u64 PL, size_t pre_bytes, hipStream_t st)
            : ext(chunk * 4 * PK * sizeof(u64), st), ten(chunk * 3 * PK * sizeof(u64), st),
              d(chunk * 3 * PL * sizeof(u64), st), pre(pre_bytes, st) {}
    };
    const size_t pre_bytes = m.mod_switch ? chunk * parts * PL * sizeof(u64) : 8;
    const bool dual = plan.dual;
    hipStream_t lanes[2] = {s0, s0};
    struct Join {  // the internal stream always rejoins the caller's, also on an error path
        hipStream_t aux = nullptr, to = nullptr;
        hipEvent_t fork = nullptr, join = nullptr;
        int device = 0;
        ~Join() {
            if (aux && hipEventRecord(join, aux) == hipSuccess) (void)hipStreamWaitEvent(to, join, 0);
            if (aux) AuxStreams::get().done(device, to);
            // (a wait captures the event's state when it is enqueued: both events may be reused right away)
            if (fork) AuxStreams::get().give_event(fork);
            if (join) AuxStreams::get().give_event(join);
        }
    } join;
    // A batch that is ONE chunk has nothing to alternate: there the two operand extensions (independent chains of
    // five launches each: inverse NTT, scaler, forward NTT of the new rows) run side by side, lhs on the caller's
    // stream and rhs on the internal one, and meet again in front of the tensor kernel -- the tail of one chain's
    // launch is filled by the other's (C5 at batch 16: launches of 4.25 waves; profiles/r03_split_ext_ab.txt).
    // Squaring (`&ct * &ct` with both operands the same buffer, the reference's bench ID "square", F/bfv/ops/mod.rs:
    // 259-358): the rhs extension would recompute the lhs one bit for bit, so it is skipped and the tensor kernel
    // reads the one extended operand twice (10 launches -> 7 per chunk; same values by construction).
    const bool square = lhs == rhs && m.ext_lhs == m.ext_rhs;
    // Both operands through the same extender (Multiplicator::default; not the second HPS strategy): the two
    // extensions are ONE pass of three launches over 4 nb polynomials (scale_polys_pair) -- round 4,
    // profiles/r04_merged_ext_ab.txt.  Otherwise, for a batch that is one chunk, the two extension chains run side by
    // side on the caller's and the internal stream (round 3's split_ext).
    const bool merged_ext = !square && m.ext_lhs == m.ext_rhs && FHE_LAB_INT("MUL_MERGED_EXT", FHE_MUL_MERGED_EXT) != 0;
    // Every kernel of the chain starts on what its producer wrote last (knobs.hpp FHE_MUL_DIRFLAGS): the extension ends
    // ascending, so the tensor kernel runs descending, the down-scaler ascending, the transform of c0 / c1 descending
    // and the key switch ascending again.  (Rows larger than LDS keep round 3's order: their transforms are two
    // kernels each and the global halves run ascending.)
    const bool dirflags = b.logn <= 14 && FHE_LAB_INT("MUL_DIRFLAGS", FHE_MUL_DIRFLAGS) != 0;
    const bool split_ext = !dual && !square && !merged_ext && m.streams.load(std::memory_order_relaxed) >= 2 &&
                           batch <= chunk && FHE_LAB_INT("MUL_SPLIT_EXT", 1) != 0;
    hipEvent_t ext_done = nullptr;
    struct EventBack {
        hipEvent_t &e;
        ~EventBack() {
            if (e) AuxStreams::get().give_event(e);
        }
    } ext_done_back{ext_done};
    if (dual || split_ext) {
        AuxStreams &ax = AuxStreams::get();
        join.to = s0;
        join.fork = ax.take_event();
        join.join = ax.take_event();
        join.device = b.device;
        hipStream_t aux = ax.stream_for(b.device, s0);
        join.aux = aux;   // (from here on ~Join reports the stream back, whatever throws below)
        if (dual) lanes[1] = aux;
        if (split_ext) ext_done = ax.take_event();
        FHE_HIP_CHECK(hipEventRecord(join.fork, s0));
        FHE_HIP_CHECK(hipStreamWaitEvent(aux, join.fork, 0));
    }
    ChunkWs ws0(chunk, PK, PL, pre_bytes, lanes[0]);
    std::unique_ptr<ChunkWs> ws1;
    if (dual) ws1 = std::make_unique<ChunkWs>(chunk, PK, PL, pre_bytes, lanes[1]);
    size_t ci = 0;
    for (size_t b0 = 0; b0 < batch; b0 += chunk, ci++) {
        const size_t nb = std::min(chunk, batch - b0);
        const hipStream_t s = lanes[dual ? ci & 1 : 0];
        ChunkWs &w = (dual && (ci & 1)) ? *ws1 : ws0;
        WsGuard &ten = w.ten, &d = w.d, &pre = w.pre;
        u64 *const extL = w.ext.u(), *const extR = w.ext.u() + nb * 2 * PK;
        const u64 *l = lhs + b0 * 2 * PL, *r = rhs + b0 * 2 * PL;
        // EXTEND (mul.rs:192-195): both parts of every lhs (rhs) ciphertext in one go
        if (merged_ext) {
            scale_polys_pair(*m.ext_lhs, l, r, extL, nb * 2, s, !skip_copy);
        } else {
            scale_polys(*m.ext_lhs, l, extL, nb * 2, true, s, !skip_copy);
            if (square) {
                // (nothing: extR is never read)
            } else if (split_ext) {   // (one chunk: the fork above put the internal stream behind the caller's earlier work)
                scale_polys(*m.ext_rhs, r, extR, nb * 2, true, join.aux, !skip_copy);
                FHE_HIP_CHECK(hipEventRecord(ext_done, join.aux));
                FHE_HIP_CHECK(hipStreamWaitEvent(s, ext_done, 0));
            } else {
                scale_polys(*m.ext_rhs, r, extR, nb * 2, true, s, !skip_copy);
            }
        }
        // TENSOR (mul.rs:198-201) + the inverse NTT of the down-scaler (M/rq/scaler.rs:69-79):
        // ten [3][nb][K][N] ends up in PowerBasis.  Rows that fit LDS: one fused kernel;
        // larger rows: element-wise tensor kernel, then the two-kernel inverse NTT.
        const bool fused_tensor = e.logn <= 16 && !FHE_LAB_FLAG("NO_TENSOR_FUSION");
        if (fused_tensor) {
            require(nb <= 32768, E_ARG, "chunk too large for the fused tensor kernel");  // 3*K*nb blocks in a 1-D grid
            k::TensorSrc ts{extL, square ? extL : extR, skip_copy ? l : nullptr, skip_copy ? r : nullptr,
                            (uint32_t)L, (uint32_t)L};
            launch_tensor_intt(e, ts, ten.u(), nb, s, dirflags);
        } else {
            for (size_t t0 = 0; t0 < nb; t0 += 32768) {  // grid.y limit
                const size_t tn = std::min<size_t>(32768, nb - t0);
                FHE_LAUNCH("tensor", k::tensor_kernel, dim3(blocks_for(PK, EW_THREADS), (unsigned)tn), dim3(EW_THREADS), 0,
                           s, extL + t0 * 2 * PK, (square ? extL : extR) + t0 * 2 * PK, skip_copy ? l + t0 * 2 * PL : nullptr,
                           skip_copy ? r + t0 * 2 * PL : nullptr, ten.u() + t0 * PK, e.dmods(), (uint32_t)K, (uint32_t)L,
                           (uint32_t)L, (uint32_t)e.logn, (u64)nb, 0u);
            }
        }
        if (!fused_tensor) launch_ntt(e, true, ten.u(), ten.u(), full_map(e, K), nb * 3, s);
        u64 *dst = m.mod_switch ? pre.u() : out + b0 * parts * PL;
        // DOWN-SCALE (mul.rs:204-206) to PowerBasis rows of d [3][nb][L][N]
        // (dirflags: the tensor kernel ran backwards and ended on the first polynomials: the scaler starts there)
        launch_scale(*m.down, ten.u(), PK, d.u(), PL, nb * 3, s, dirflags && fused_tensor);
        k::RowMap back = full_map(b, L);
        back.reverse = (dirflags && fused_tensor) ? 1u : 0u;   // ... and the forward transform where the scaler ended
        if (m.rk) {
            // c0, c1 go back to Ntt; c2 stays in PowerBasis for the key switch (the reference
            // transforms c2 forward and, at mul.rs:212, back again; iNTT(NTT(x)) = x exactly).
            // (Transforming c2 as well and handing it to the key switch as `xhat` was measured at C2: the key
            // switch gains 1.0 ms per 10 steps, the larger forward launch costs 1.6: profiles/r02_mul_xhat_ab.txt.)
            // Round 5: when the key switch of this chunk runs unfused (a launch that does not fill the device) and the key
            // sits at the ciphertext's level, the forward transform of (c0, c1) rides on its stage-A launch -- one launch
            // less on the critical path of a small call (C2, one pair: 0.088 -> 0.077 ms, profiles/r05_merged_fwd_ab.jsonl)
            const bool ride = m.rk->ksk_ctx->niterations_to(*m.rk->ct_ctx) == 0 && b.logn <= 14 &&
                              ks_use_unfused(*m.rk, m.rk->mode.load(std::memory_order_relaxed), nb) &&
                              !FHE_LAB_FLAG("NO_MERGED_FWD");
            KsExtraFwd ex;
            ex.rows = d.u();
            ex.poly_stride = PL;
            ex.npolys = nb * 2;
            ex.nrows = L;
            if (!ride) launch_ntt(b, false, d.u(), d.u(), back, nb * 2, s);
            // RELINEARIZE (mul.rs:211-227): (c0, c1) += key_switch(c2), written to the output layout
            key_switch_add(*m.rk, d.u() + 2 * nb * PL, PL, d.u(), d.u() + nb * PL, PL, dst, dst + PL, 2 * PL, nb, s, nullptr, 0, 0,
                           ride ? &ex : nullptr);
        } else {
            // no relinearisation: three Ntt parts, slot-major scratch -> [b][3][L][N]
            launch_ntt(b, false, d.u(), d.u(), back, nb * 3, s);
            for (size_t slot = 0; slot < 3; slot++) {
                const u64 total = (u64)nb * PL;
                FHE_LAUNCH("copy_rows", k::copy_rows_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0,
                           s, d.u() + slot * nb * PL, dst + slot * PL, PL, 3 * PL, PL, total);
            }
        }
        if (m.mod_switch) bfv_switch_down(b, parts, pre.u(), out + b0 * parts * (PL - N), nb, s);
    }
}

}  // namespace fhe

#if defined(FHE_LAB)
#include "lab/lab_engine.hpp"
#endif
