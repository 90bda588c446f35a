// fhe_hip.cpp -- the extern "C" boundary declared in include/fhe_hip.h.
// Compiled as HIP for gfx950 (`hipcc -x hip --offload-arch=gfx950`, see __graft_entry__.build()).
// Every entry point: validate -> run engine code -> translate exceptions to status codes;
// nothing throws across the boundary.
#include "../../include/fhe_hip.h"

#include <cstdio>
#include <map>

#include <cxxabi.h>

#include "engine.hpp"
#include "ubench.hpp"

using namespace fhe;

struct fhe_ctx {
    std::unique_ptr<Ctx> owned;  // set on handles returned by fhe_ctx_create
    const Ctx *c = nullptr;
    std::vector<std::unique_ptr<fhe_ctx>> level_handles;  // borrowed-handle storage for fhe_ctx_at_level
    const fhe_ctx *root_handle = nullptr;                 // set on borrowed handles
    size_t level_in_root = 0;
};
struct fhe_scaler {
    std::unique_ptr<Scaler> s;
};
struct fhe_ksk {
    std::unique_ptr<Ksk> k;
};
struct fhe_mul {
    std::unique_ptr<Mul> m;
    // scalers created by fhe_mul_create_default are owned here
    std::vector<std::unique_ptr<fhe_ctx>> ctxs;
    std::vector<std::unique_ptr<fhe_scaler>> scalers;
};
struct HostTables {  // one NttOperator's tables as the host's callback filled them
    std::vector<u64> om, oms, zi, zis;
    u64 si = 0, sis = 0;
};
struct fhe_params {
    int device = -1;
    bool host_tables = false;               // created with the host's NTT tables (callback)
    std::map<u64, HostTables> table_cache;  // modulus -> tables, filled while the callback was alive
    size_t degree = 0;
    u64 plaintext = 0;
    std::vector<u64> moduli;
    std::vector<size_t> moduli_sizes;
    std::unique_ptr<fhe_ctx> top;                       // level-0 context (+ chain)
    std::vector<std::unique_ptr<fhe_ctx>> mul_ctx;      // per level
    std::vector<std::unique_ptr<fhe_scaler>> extender;  // per level
    std::vector<std::unique_ptr<fhe_scaler>> down;      // per level
};

namespace {
thread_local std::string g_last_error;

template <class F>
fhe_status guard(F &&f) {
    try {
        f();
        return FHE_OK;
    } catch (const StatusError &e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        g_last_error = "out of host memory";
        return FHE_E_ARG;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return FHE_E_ARG;
    }
}
void need(const void *p, const char *what) {
    if (!p) throw StatusError(FHE_E_ARG, std::string("null argument: ") + what);
}
std::unique_ptr<fhe_ctx> wrap_ctx(std::unique_ptr<Ctx> c) {
    auto h = std::make_unique<fhe_ctx>();
    h->c = c.get();
    h->owned = std::move(c);
    // one borrowed handle per level of the chain
    for (const Ctx *l = h->c->next.get(); l; l = l->next.get()) {
        auto b = std::make_unique<fhe_ctx>();
        b->c = l;
        b->root_handle = h.get();
        b->level_in_root = h->level_handles.size() + 1;
        h->level_handles.push_back(std::move(b));
    }
    return h;
}
const fhe_ctx *ctx_level_handle(const fhe_ctx *h, size_t level) {
    if (level == 0) return h;
    if (h->root_handle) return ctx_level_handle(h->root_handle, h->level_in_root + level);
    if (level - 1 < h->level_handles.size()) return h->level_handles[level - 1].get();
    return nullptr;
}
void set_device(const Ctx &c) {
    if (c.device >= 0) FHE_HIP_CHECK(hipSetDevice(c.device));
}

// Host-pointer convenience: copy in, run `body(device_ptrs...)` on the null stream, copy out.
// Staging blocks come from the engine's workspace pool (null stream), not from hipMalloc / hipFree per call: a host
// call is synchronous, so a block released here is idle by the time the next call takes it.
struct HostIO {
    std::vector<void *> bufs;
    std::vector<std::pair<void *, size_t>> secrets;  // staging copies cleared before they are released
    ~HostIO() {
        for (auto &s : secrets) (void)hipMemset(s.first, 0, s.second);
        if (!secrets.empty()) (void)hipStreamSynchronize(nullptr);
        for (void *p : bufs) Workspace::get().release(p);
    }
    u64 *secret(u64 *d, size_t count) {
        secrets.emplace_back(d, std::max<size_t>(count, 1) * sizeof(u64));
        return d;
    }
    u64 *in(const u64 *h, size_t count) {
        u64 *d = out(count);
        if (count) FHE_HIP_CHECK(hipMemcpy(d, h, count * sizeof(u64), hipMemcpyHostToDevice));
        return d;
    }
    u64 *out(size_t count) {
        void *d = Workspace::get().acquire(std::max<size_t>(count, 1) * sizeof(u64), nullptr);
        bufs.push_back(d);
        return (u64 *)d;
    }
    void back(u64 *h, const u64 *d, size_t count) {
        FHE_HIP_CHECK(hipStreamSynchronize(nullptr));
        if (count) FHE_HIP_CHECK(hipMemcpy(h, d, count * sizeof(u64), hipMemcpyDeviceToHost));
    }
};


// Host-pointer calls on a large batch of independent items: the batch goes through in slices whose upload, kernels and
// download overlap (three internal streams: the link is full duplex and the GPU works on slice i while slice i + 1
// arrives and slice i - 1 leaves).  Values are those of the one-shot path.  `ins`: host arrays with their words per
// item (equal pointers share one device copy: squaring); body(device inputs of the slice, device output of the slice,
// items, stream).  Returns false -- nothing done -- when the batch is smaller than three slices of >= 32 MiB.
struct HostIn {
    const u64 *h;
    size_t words;
};
template <class Body>
static bool host_sliced(int dev, size_t batch, const std::vector<HostIn> &ins, u64 *out, size_t out_words, Body body) {
    size_t widest = out_words;
    for (const HostIn &in : ins) widest = std::max(widest, in.words);
    if (!widest) return false;
#if defined(FHE_HOST_EMULATION)
    const size_t slice = 2;   // (so that the emulated suite walks the sliced path: batches of 6 and more)
#else
    const size_t slice = std::max<size_t>(8, (((size_t)32 << 20) / (widest * sizeof(u64)) + 7) / 8 * 8);
#endif
    if (batch < 3 * slice) return false;
    static char tag_up, tag_comp, tag_down;   // keys of the three internal streams (no user stream has these addresses)
    AuxStreams &ax = AuxStreams::get();
    hipStream_t s_up = ax.stream_for(dev, (hipStream_t)&tag_up, true), s_comp = ax.stream_for(dev, (hipStream_t)&tag_comp, true),
                s_down = ax.stream_for(dev, (hipStream_t)&tag_down, true);
    // The staging blocks below come from the pool's NULL-stream blocks, which count as idle as soon as the call that
    // used them has returned -- but an asynchronous `_dev` call on the null stream may still be running in them, and
    // the three internal streams are non-blocking: they do not order against the null stream by themselves.
    FHE_HIP_CHECK(hipStreamSynchronize(nullptr));
    HostIO io;
    std::vector<u64 *> din(ins.size(), nullptr);
    for (size_t k = 0; k < ins.size(); k++) {
        for (size_t j = 0; j < k; j++)
            if (ins[j].h == ins[k].h && ins[j].words == ins[k].words) din[k] = din[j];
        if (!din[k]) din[k] = io.out(batch * ins[k].words);
    }
    u64 *dout = io.out(batch * out_words);
    const size_t nsl = (batch + slice - 1) / slice;
    std::vector<hipEvent_t> up(nsl, nullptr), done(nsl, nullptr);
    auto finish = [&] {   // nothing may still use the buffers when HostIO gives them back
        (void)hipStreamSynchronize(s_up);
        (void)hipStreamSynchronize(s_comp);
        (void)hipStreamSynchronize(s_down);
        for (auto *v : {&up, &done})
            for (hipEvent_t &e : *v)
                if (e) ax.give_event(e), e = nullptr;
    };
    try {
        for (size_t i = 0; i < nsl + 2; i++) {
            if (i < nsl) {
                const size_t o = i * slice, n = std::min(slice, batch - o);
                for (size_t k = 0; k < ins.size(); k++) {
                    bool first = true;
                    for (size_t j = 0; j < k; j++) first = first && din[j] != din[k];
                    if (first)
                        FHE_HIP_CHECK(hipMemcpyAsync(din[k] + o * ins[k].words, ins[k].h + o * ins[k].words,
                                                     n * ins[k].words * sizeof(u64), hipMemcpyHostToDevice, s_up));
                }
                up[i] = ax.take_event();
                FHE_HIP_CHECK(hipEventRecord(up[i], s_up));
            }
            if (i >= 1 && i - 1 < nsl) {
                const size_t c = i - 1, o = c * slice, n = std::min(slice, batch - o);
                FHE_HIP_CHECK(hipStreamWaitEvent(s_comp, up[c], 0));
                std::vector<const u64 *> ds(ins.size());
                for (size_t k = 0; k < ins.size(); k++) ds[k] = din[k] + o * ins[k].words;
                body(ds, dout + o * out_words, n, s_comp);
                done[c] = ax.take_event();
                FHE_HIP_CHECK(hipEventRecord(done[c], s_comp));
            }
            if (i >= 2) {
                const size_t c = i - 2, o = c * slice, n = std::min(slice, batch - o);
                FHE_HIP_CHECK(hipStreamWaitEvent(s_down, done[c], 0));
                FHE_HIP_CHECK(hipMemcpyAsync(out + o * out_words, dout + o * out_words, n * out_words * sizeof(u64),
                                             hipMemcpyDeviceToHost, s_down));
            }
        }
    } catch (...) {
        finish();
        throw;
    }
    finish();
    return true;
}

std::unique_ptr<Ksk> make_ksk(const Ctx &ct, const Ctx &kc, size_t ndigits, size_t log_base) {
    ksk_validate(ct, kc, ndigits, log_base);
    kc.need_device();
    auto k_ = std::make_unique<Ksk>();
    k_->ct_ctx = &ct;
    k_->ksk_ctx = &kc;
    k_->ndigits = ndigits;
    k_->log_base = log_base;
    return k_;
}
// Context over `moduli` whose NTT tables come from the host's callback (NULL: the engine's own psi).  Every table
// is cached per modulus, so handles made later from the parameter set (a level-specific multiplication basis) never
// call back into the host: `fn` is only invoked while fhe_params_create_with_tables runs.
std::unique_ptr<Ctx> ctx_create_cb(int device, size_t degree, const std::vector<u64> &moduli, fhe_ntt_tables_fn fn,
                                   void *user, bool host_tables, std::map<u64, HostTables> &cache) {
    if (!host_tables) return ctx_create(device, degree, moduli, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    const size_t L = moduli.size();
    std::vector<u64> om(L * degree), oms(L * degree), zi(L * degree), zis(L * degree), si(L), sis(L);
    for (size_t i = 0; i < L; i++) {
        auto it = cache.find(moduli[i]);
        if (it == cache.end()) {
            if (!fn) throw StatusError(FHE_E_NTT_UNAVAILABLE, "NttOperatorUnavailable: no host table cached for this modulus");
            HostTables t;
            t.om.resize(degree), t.oms.resize(degree), t.zi.resize(degree), t.zis.resize(degree);
            if (fn(user, moduli[i], degree, t.om.data(), t.oms.data(), t.zi.data(), t.zis.data(), &t.si, &t.sis) != 0)
                throw StatusError(FHE_E_NTT_UNAVAILABLE, "NttOperatorUnavailable: the host's table callback failed");
            it = cache.emplace(moduli[i], std::move(t)).first;
        }
        const HostTables &t = it->second;
        std::copy(t.om.begin(), t.om.end(), om.begin() + i * degree);
        std::copy(t.oms.begin(), t.oms.end(), oms.begin() + i * degree);
        std::copy(t.zi.begin(), t.zi.end(), zi.begin() + i * degree);
        std::copy(t.zis.begin(), t.zis.end(), zis.begin() + i * degree);
        si[i] = t.si, sis[i] = t.sis;
    }
    return ctx_create(device, degree, moduli, om.data(), oms.data(), zi.data(), zis.data(), si.data(), sis.data());
}
}  // namespace

extern "C" {

const char *fhe_last_error(void) { return g_last_error.c_str(); }
const char *fhe_version(void) { return "fhe.rs_amd 0.1 (gfx950)"; }
int fhe_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------ device memory and streams ----
fhe_status fhe_buf_alloc(int device, size_t bytes, void **out) {
    return guard([&] {
        need(out, "out");
        *out = nullptr;
        int ndev = 0;
        FHE_HIP_CHECK(hipGetDeviceCount(&ndev));
        require(device >= 0 && device < ndev, E_NO_DEVICE, "no such HIP device");
        FHE_HIP_CHECK(hipSetDevice(device));
        FHE_HIP_CHECK(hipMalloc(out, std::max<size_t>(bytes, 1)));
    });
}
fhe_status fhe_buf_free(void *buf) {
    return guard([&] {
        if (buf) FHE_HIP_CHECK(hipFree(buf));
    });
}
// fhe_buf_alloc_async takes its memory from the device's PRIVATE stream-ordered pool (DevPools, engine.hpp), which is
// told to keep what is freed into it -- a host that allocates its results per call does not go back to the driver
// each time, and no other hipMallocAsync user of the process sees a changed default pool.
fhe_status fhe_buf_alloc_async(int device, size_t bytes, void *stream, void **out) {
    return guard([&] {
        need(out, "out");
        *out = nullptr;
        int ndev = 0;
        FHE_HIP_CHECK(hipGetDeviceCount(&ndev));
        require(device >= 0 && device < ndev, E_NO_DEVICE, "no such HIP device");
        FHE_HIP_CHECK(hipSetDevice(device));
        *out = DevPools::get().alloc(device, std::max<size_t>(bytes, 1), as_stream(stream), DevPools::BUFFERS);
    });
}
fhe_status fhe_buf_free_async(void *buf, void *stream) {
    return guard([&] {
        if (buf) FHE_HIP_CHECK(hipFreeAsync(buf, as_stream(stream)));
    });
}
static void copy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, void *stream, bool wait) {
    if (bytes) {
        need(dst, "dst");
        need(src, "src");
        FHE_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, as_stream(stream)));
    }
    if (wait) FHE_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
}
fhe_status fhe_buf_upload_async(void *dst_dev, const void *src_host, size_t bytes, void *stream) {
    return guard([&] { copy_async(dst_dev, src_host, bytes, hipMemcpyHostToDevice, stream, false); });
}
fhe_status fhe_buf_upload(void *dst_dev, const void *src_host, size_t bytes, void *stream) {
    return guard([&] { copy_async(dst_dev, src_host, bytes, hipMemcpyHostToDevice, stream, true); });
}
fhe_status fhe_buf_download_async(void *dst_host, const void *src_dev, size_t bytes, void *stream) {
    return guard([&] { copy_async(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, stream, false); });
}
fhe_status fhe_buf_download(void *dst_host, const void *src_dev, size_t bytes, void *stream) {
    return guard([&] { copy_async(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, stream, true); });
}
fhe_status fhe_buf_copy_async(void *dst_dev, const void *src_dev, size_t bytes, void *stream) {
    return guard([&] { copy_async(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, stream, false); });
}
fhe_status fhe_buf_zero_async(void *buf, size_t bytes, void *stream) {
    return guard([&] {
        if (!bytes) return;
        need(buf, "buf");
        FHE_HIP_CHECK(hipMemsetAsync(buf, 0, bytes, as_stream(stream)));
    });
}
fhe_status fhe_host_alloc(size_t bytes, void **out) {
    return guard([&] {
        need(out, "out");
        *out = nullptr;
        FHE_HIP_CHECK(hipHostMalloc(out, std::max<size_t>(bytes, 1), 0));
    });
}
fhe_status fhe_host_free(void *p) {
    return guard([&] {
        if (p) FHE_HIP_CHECK(hipHostFree(p));
    });
}
fhe_status fhe_stream_create(int device, void **out) {
    return guard([&] {
        need(out, "out");
        *out = nullptr;
        int ndev = 0;
        FHE_HIP_CHECK(hipGetDeviceCount(&ndev));
        require(device >= 0 && device < ndev, E_NO_DEVICE, "no such HIP device");
        FHE_HIP_CHECK(hipSetDevice(device));
        hipStream_t s = nullptr;
        FHE_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        *out = (void *)s;
    });
}
fhe_status fhe_stream_sync(void *stream) {
    return guard([&] { FHE_HIP_CHECK(hipStreamSynchronize(as_stream(stream))); });
}
fhe_status fhe_stream_destroy(void *stream) {
    return guard([&] {
        if (!stream) return;   // the null stream is not ours to destroy
        FHE_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
        // what the engine keeps per caller stream: its internal second stream and its idle scratch blocks
        // (the internal stream's own blocks too -- a ChunkWs and the split-extension scratch are keyed to it: ADVICE r03)
        AuxStreams::get().drop(-1, as_stream(stream), false);
        Workspace::get().drop_stream(as_stream(stream));
        FHE_HIP_CHECK(hipStreamDestroy(as_stream(stream)));
    });
}
fhe_status fhe_device_sync(int device) {
    return guard([&] {
        int ndev = 0;
        FHE_HIP_CHECK(hipGetDeviceCount(&ndev));
        require(device >= 0 && device < ndev, E_NO_DEVICE, "no such HIP device");
        FHE_HIP_CHECK(hipSetDevice(device));
        FHE_HIP_CHECK(hipDeviceSynchronize());
    });
}
fhe_status fhe_device_mem_info(int device, size_t *free_bytes, size_t *total_bytes) {
    return guard([&] {
        int ndev = 0;
        FHE_HIP_CHECK(hipGetDeviceCount(&ndev));
        require(device >= 0 && device < ndev, E_NO_DEVICE, "no such HIP device");
        FHE_HIP_CHECK(hipSetDevice(device));
        size_t f = 0, t = 0;
        FHE_HIP_CHECK(hipMemGetInfo(&f, &t));
        if (free_bytes) *free_bytes = f;
        if (total_bytes) *total_bytes = t;
    });
}

// ----------------------------------------------------------------------------- ctx ----
fhe_status fhe_ctx_create(int device, size_t degree, size_t nmoduli, const uint64_t *moduli, const uint64_t *omegas,
                          const uint64_t *omegas_shoup, const uint64_t *zetas_inv, const uint64_t *zetas_inv_shoup,
                          const uint64_t *size_inv, const uint64_t *size_inv_shoup, fhe_ctx **out) {
    return guard([&] {
        need(out, "out");
        *out = nullptr;
        if (nmoduli == 0) throw StatusError(FHE_E_EMPTY_MODULI, "EmptyModuli");
        need(moduli, "moduli");
        std::vector<u64> m(moduli, moduli + nmoduli);
        *out = wrap_ctx(ctx_create(device, degree, m, omegas, omegas_shoup, zetas_inv, zetas_inv_shoup, size_inv,
                                   size_inv_shoup))
                   .release();
    });
}
void fhe_ctx_destroy(fhe_ctx *ctx) {
    if (ctx && ctx->owned) delete ctx;
}
fhe_status fhe_ctx_at_level(const fhe_ctx *ctx, size_t level, const fhe_ctx **out) {
    return guard([&] {
        need(ctx, "ctx");
        need(out, "out");
        // a borrowed level handle can itself be asked for deeper levels: resolve on the Ctx chain
        const Ctx *target = ctx->c->at_level(level);
        if (!target) throw StatusError(FHE_E_INVALID_LEVEL, "InvalidContextLevel");
        if (level == 0) {
            *out = ctx;
            return;
        }
        const fhe_ctx *h = ctx_level_handle(ctx, level);
        if (!h) throw StatusError(FHE_E_ARG, "fhe_ctx_at_level must be called on a handle from fhe_ctx_create");
        *out = h;
    });
}
fhe_status fhe_ctx_niterations_to(const fhe_ctx *from, const fhe_ctx *to, size_t *out) {
    return guard([&] {
        need(from, "from");
        need(to, "to");
        need(out, "out");
        long it = from->c->niterations_to(*to->c);
        if (it < 0) throw StatusError(FHE_E_CONTEXT_NOT_REACHABLE, "ContextNotReachable");
        *out = (size_t)it;
    });
}
size_t fhe_ctx_degree(const fhe_ctx *ctx) { return ctx ? ctx->c->n : 0; }
size_t fhe_ctx_nmoduli(const fhe_ctx *ctx) { return ctx ? ctx->c->L : 0; }
int fhe_ctx_device(const fhe_ctx *ctx) { return ctx ? ctx->c->device : -1; }
fhe_status fhe_ctx_moduli(const fhe_ctx *ctx, uint64_t *out) {
    return guard([&] {
        need(ctx, "ctx");
        need(out, "out");
        std::copy(ctx->c->moduli.begin(), ctx->c->moduli.end(), out);
    });
}
fhe_status fhe_ctx_get_table(const fhe_ctx *ctx, int which, uint64_t *out) {
    return guard([&] {
        need(ctx, "ctx");
        need(out, "out");
        const Ctx &c = *ctx->c;
        for (size_t i = 0; i < c.L; i++) {
            const NttTables &t = c.tab(i);
            switch (which) {
                case 0: std::copy(t.omegas.begin(), t.omegas.end(), out + i * c.n); break;
                case 1: std::copy(t.omegas_shoup.begin(), t.omegas_shoup.end(), out + i * c.n); break;
                case 2: std::copy(t.zetas_inv.begin(), t.zetas_inv.end(), out + i * c.n); break;
                case 3: std::copy(t.zetas_inv_shoup.begin(), t.zetas_inv_shoup.end(), out + i * c.n); break;
                case 4: out[i] = t.size_inv; break;
                case 5: out[i] = t.size_inv_shoup; break;
                case 6: if (i + 1 < c.L) out[i] = c.inv_last[i]; break;
                case 7: if (i + 1 < c.L) out[i] = c.inv_last_shoup[i]; break;
                default: throw StatusError(FHE_E_ARG, "unknown table selector");
            }
        }
    });
}

#define FHE_POLY_IO_PROLOGUE(ctxh)      \
    need(ctxh, "ctx");                  \
    const Ctx &c = *(ctxh)->c;          \
    c.need_device();                    \
    set_device(c);                      \
    const size_t pe = c.L * c.n

fhe_status fhe_ntt_forward_dev(const fhe_ctx *ctx, uint64_t *polys, size_t batch, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        if (batch) need(polys, "polys");
        ntt_polys(c, false, polys, polys, batch, as_stream(stream));
    });
}
fhe_status fhe_ntt_backward_dev(const fhe_ctx *ctx, uint64_t *polys, size_t batch, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        if (batch) need(polys, "polys");
        ntt_polys(c, true, polys, polys, batch, as_stream(stream));
    });
}
fhe_status fhe_ntt_forward(const fhe_ctx *ctx, uint64_t *polys, size_t batch) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (batch) need(polys, "polys");
        HostIO io;
        u64 *d = io.in(polys, batch * pe);
        ntt_polys(c, false, d, d, batch, nullptr);
        io.back(polys, d, batch * pe);
    });
}
fhe_status fhe_ntt_backward(const fhe_ctx *ctx, uint64_t *polys, size_t batch) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (batch) need(polys, "polys");
        HostIO io;
        u64 *d = io.in(polys, batch * pe);
        ntt_polys(c, true, d, d, batch, nullptr);
        io.back(polys, d, batch * pe);
    });
}

static fhe_status ew_host(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t batch, uint32_t op) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (batch) {
            need(a, "a");
            if (op != k::EW_NEG) need(b, "b");
        }
        HostIO io;
        u64 *da = io.in(a, batch * pe);
        u64 *db = op != k::EW_NEG ? io.in(b, batch * pe) : nullptr;
        ew_op(c, da, db, batch, op, nullptr);
        io.back(a, da, batch * pe);
    });
}
static fhe_status ew_dev(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t batch, uint32_t op, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        if (batch) {
            need(a, "a");
            if (op != k::EW_NEG) need(b, "b");
        }
        ew_op(c, a, b, batch, op, as_stream(stream));
    });
}
fhe_status fhe_poly_add(const fhe_ctx *c, uint64_t *a, const uint64_t *b, size_t n) { return ew_host(c, a, b, n, k::EW_ADD); }
fhe_status fhe_poly_sub(const fhe_ctx *c, uint64_t *a, const uint64_t *b, size_t n) { return ew_host(c, a, b, n, k::EW_SUB); }
fhe_status fhe_poly_mul(const fhe_ctx *c, uint64_t *a, const uint64_t *b, size_t n) { return ew_host(c, a, b, n, k::EW_MUL); }
fhe_status fhe_poly_neg(const fhe_ctx *c, uint64_t *a, size_t n) { return ew_host(c, a, nullptr, n, k::EW_NEG); }
fhe_status fhe_poly_add_dev(const fhe_ctx *c, uint64_t *a, const uint64_t *b, size_t n, void *s) { return ew_dev(c, a, b, n, k::EW_ADD, s); }
fhe_status fhe_poly_sub_dev(const fhe_ctx *c, uint64_t *a, const uint64_t *b, size_t n, void *s) { return ew_dev(c, a, b, n, k::EW_SUB, s); }
fhe_status fhe_poly_mul_dev(const fhe_ctx *c, uint64_t *a, const uint64_t *b, size_t n, void *s) { return ew_dev(c, a, b, n, k::EW_MUL, s); }
fhe_status fhe_poly_neg_dev(const fhe_ctx *c, uint64_t *a, size_t n, void *s) { return ew_dev(c, a, nullptr, n, k::EW_NEG, s); }

static void mul_shoup_run(const Ctx &c, u64 *a, const u64 *b, const u64 *bs, size_t batch, hipStream_t s) {
    const u64 total = (u64)batch * c.L * c.n;
    if (!total) return;
    FHE_LAUNCH("mul_shoup", k::mul_shoup_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0, s, a, b, bs,
               c.dmods(), (uint32_t)c.L, (uint32_t)c.logn, total);
}
fhe_status fhe_poly_mul_shoup_dev(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, const uint64_t *b_shoup,
                                  size_t batch, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        if (batch) {
            need(a, "a");
            need(b, "b");
            need(b_shoup, "b_shoup");
        }
        mul_shoup_run(c, a, b, b_shoup, batch, as_stream(stream));
    });
}
fhe_status fhe_poly_mul_shoup(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, const uint64_t *b_shoup,
                              size_t batch) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (batch) {
            need(a, "a");
            need(b, "b");
            need(b_shoup, "b_shoup");
        }
        HostIO io;
        u64 *da = io.in(a, batch * pe), *db = io.in(b, batch * pe), *dbs = io.in(b_shoup, batch * pe);
        mul_shoup_run(c, da, db, dbs, batch, nullptr);
        io.back(a, da, batch * pe);
    });
}
fhe_status fhe_poly_shoup(const fhe_ctx *ctx, const uint64_t *a, uint64_t *a_shoup, size_t batch) {
    return guard([&] {
        need(ctx, "ctx");
        const Ctx &c = *ctx->c;
        if (batch) {
            need(a, "a");
            need(a_shoup, "a_shoup");
        }
        // setup-time host computation (128/64-bit division), M/zq/mod.rs:195-199
        for (size_t b = 0; b < batch; b++)
            for (size_t r = 0; r < c.L; r++)
                for (size_t j = 0; j < c.n; j++) {
                    const size_t i = (b * c.L + r) * c.n + j;
                    a_shoup[i] = shoup(a[i], c.moduli[r]);
                }
    });
}

fhe_status fhe_poly_substitute_dev(const fhe_ctx *ctx, size_t exponent, const uint64_t *in, uint64_t *out,
                                   size_t batch, int repr_is_ntt, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        if (batch) {
            need(in, "in");
            need(out, "out");
        }
        if (in == out) throw StatusError(FHE_E_ARG, "substitute is not in-place");
        substitute_polys(c, exponent, in, out, batch, repr_is_ntt != 0, as_stream(stream));
    });
}
fhe_status fhe_poly_substitute(const fhe_ctx *ctx, size_t exponent, const uint64_t *in, uint64_t *out, size_t batch,
                               int repr_is_ntt) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (batch) {
            need(in, "in");
            need(out, "out");
        }
        HostIO io;
        u64 *di = io.in(in, batch * pe), *dout = io.out(batch * pe);
        substitute_polys(c, exponent, di, dout, batch, repr_is_ntt != 0, nullptr);
        io.back(out, dout, batch * pe);
    });
}

size_t fhe_poly_serialized_size(const fhe_ctx *ctx) { return ctx ? wire_poly_bytes(*ctx->c) : 0; }
fhe_status fhe_poly_serialize_dev(const fhe_ctx *ctx, const uint64_t *polys, uint8_t *bytes, size_t batch, int from_ntt,
                                  void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        if (batch) {
            need(polys, "polys");
            need(bytes, "bytes");
        }
        wire_serialize(c, polys, bytes, batch, from_ntt != 0, as_stream(stream));
    });
}
fhe_status fhe_poly_serialize(const fhe_ctx *ctx, const uint64_t *polys, uint8_t *bytes, size_t batch, int from_ntt) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (batch) {
            need(polys, "polys");
            need(bytes, "bytes");
        }
        const size_t wb = wire_poly_bytes(c);
        HostIO io;
        u64 *di = io.in(polys, batch * pe);
        uint8_t *db = reinterpret_cast<uint8_t *>(io.out((batch * wb + 7) / 8));
        wire_serialize(c, di, db, batch, from_ntt != 0, nullptr);
        FHE_HIP_CHECK(hipStreamSynchronize(nullptr));
        if (batch) FHE_HIP_CHECK(hipMemcpy(bytes, db, batch * wb, hipMemcpyDeviceToHost));
    });
}
fhe_status fhe_poly_deserialize_dev(const fhe_ctx *ctx, const uint8_t *bytes, uint64_t *polys, size_t batch, int to_ntt,
                                    void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        if (batch) {
            need(polys, "polys");
            need(bytes, "bytes");
        }
        wire_deserialize(c, bytes, polys, batch, to_ntt != 0, as_stream(stream));
    });
}
fhe_status fhe_poly_deserialize(const fhe_ctx *ctx, const uint8_t *bytes, uint64_t *polys, size_t batch, int to_ntt) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (batch) {
            need(polys, "polys");
            need(bytes, "bytes");
        }
        const size_t wb = wire_poly_bytes(c);
        HostIO io;
        uint8_t *db = reinterpret_cast<uint8_t *>(io.out((batch * wb + 7) / 8));
        if (batch) FHE_HIP_CHECK(hipMemcpy(db, bytes, batch * wb, hipMemcpyHostToDevice));
        u64 *dout = io.out(batch * pe);
        wire_deserialize(c, db, dout, batch, to_ntt != 0, nullptr);
        io.back(polys, dout, batch * pe);
    });
}

fhe_status fhe_poly_switch_down_dev(const fhe_ctx *ctx, const uint64_t *in, uint64_t *out, size_t batch, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (!c.next) throw StatusError(FHE_E_NO_MORE_CONTEXT, "NoMoreContext");
        if (batch) {
            need(in, "in");
            need(out, "out");
        }
        switch_down_polys(c, in, pe, out, pe - c.n, batch, as_stream(stream));
    });
}
fhe_status fhe_poly_switch_down(const fhe_ctx *ctx, const uint64_t *in, uint64_t *out, size_t batch) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (batch) {
            need(in, "in");
            need(out, "out");
        }
        if (!c.next) throw StatusError(FHE_E_NO_MORE_CONTEXT, "NoMoreContext");
        HostIO io;
        u64 *di = io.in(in, batch * pe), *dout = io.out(batch * (pe - c.n));
        switch_down_polys(c, di, pe, dout, pe - c.n, batch, nullptr);
        io.back(out, dout, batch * (pe - c.n));
    });
}

static size_t iterations_between(const Ctx &from, const Ctx &to) {
    const long it = from.niterations_to(to);
    if (it < 0) throw StatusError(FHE_E_CONTEXT_NOT_REACHABLE, "ContextNotReachable");
    return (size_t)it;
}
fhe_status fhe_poly_switch_down_to_dev(const fhe_ctx *from, const fhe_ctx *to, const uint64_t *in, uint64_t *out,
                                       size_t batch, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(from);
        need(to, "to");
        const size_t iters = iterations_between(c, *to->c);
        if (batch) {
            need(in, "in");
            need(out, "out");
        }
        switch_down_to_pb(c, iters, in, pe, out, pe - iters * c.n, batch, as_stream(stream));
    });
}
fhe_status fhe_poly_switch_down_to(const fhe_ctx *from, const fhe_ctx *to, const uint64_t *in, uint64_t *out,
                                   size_t batch) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(from);
        need(to, "to");
        const size_t iters = iterations_between(c, *to->c);
        if (batch) {
            need(in, "in");
            need(out, "out");
        }
        const size_t po = pe - iters * c.n;
        HostIO io;
        u64 *di = io.in(in, batch * pe), *dout = io.out(batch * po);
        switch_down_to_pb(c, iters, di, pe, dout, po, batch, nullptr);
        io.back(out, dout, batch * po);
    });
}

// -------------------------------------------------------------------------- scaler ----
fhe_status fhe_scaler_create(const fhe_ctx *from, const fhe_ctx *to, const uint64_t *numerator, size_t numerator_limbs,
                             const uint64_t *denominator, size_t denominator_limbs, fhe_scaler **out) {
    return guard([&] {
        need(from, "from");
        need(to, "to");
        need(out, "out");
        need(numerator, "numerator");
        need(denominator, "denominator");
        *out = nullptr;
        set_device(*from->c);
        BigUint num = BigUint::from_limbs(numerator, numerator_limbs);
        BigUint den = BigUint::from_limbs(denominator, denominator_limbs);
        auto h = std::make_unique<fhe_scaler>();
        h->s = scaler_create(*from->c, *to->c, num, den);
        *out = h.release();
    });
}
fhe_status fhe_scaler_create_from_constants(const fhe_ctx *from, const fhe_ctx *to, size_t number_common_moduli,
                                            int is_one, const uint64_t *gamma, const uint64_t *gamma_shoup,
                                            const uint64_t *omega, const uint64_t *omega_shoup, uint64_t theta_gamma_lo,
                                            uint64_t theta_gamma_hi, int theta_gamma_sign,
                                            const uint64_t *theta_omega_lo, const uint64_t *theta_omega_hi,
                                            const uint8_t *theta_omega_sign, const uint64_t *theta_garner_lo,
                                            const uint64_t *theta_garner_hi, size_t theta_garner_shift,
                                            fhe_scaler **out) {
    return guard([&] {
        need(from, "from");
        need(to, "to");
        need(out, "out");
        *out = nullptr;
        need(gamma, "gamma");
        need(gamma_shoup, "gamma_shoup");
        need(omega, "omega");
        need(omega_shoup, "omega_shoup");
        need(theta_omega_lo, "theta_omega_lo");
        need(theta_omega_hi, "theta_omega_hi");
        need(theta_omega_sign, "theta_omega_sign");
        need(theta_garner_lo, "theta_garner_lo");
        need(theta_garner_hi, "theta_garner_hi");
        const Ctx &f = *from->c, &t = *to->c;
        require(f.n == t.n, E_DEGREE_MISMATCH, "DegreeMismatch");
        require(f.device == t.device, E_PARAMETER_MISMATCH, "contexts live on different devices");
        require(theta_garner_shift >= 2 && theta_garner_shift <= 127, E_ARG, "theta_garner_shift out of range");
        require(number_common_moduli <= std::min(f.L, t.L), E_ARG, "number_common_moduli too large");
        set_device(f);
        auto h = std::make_unique<fhe_scaler>();
        h->s = std::make_unique<Scaler>();
        Scaler &s = *h->s;
        s.from = &f;
        s.to = &t;
        s.ncommon = number_common_moduli;
        ScalerConstants &c = s.c;
        c.nfrom = f.L;
        c.nto = t.L;
        c.is_one = is_one != 0;
        c.gamma.assign(gamma, gamma + t.L);
        c.gamma_shoup.assign(gamma_shoup, gamma_shoup + t.L);
        c.omega.assign(omega, omega + t.L * f.L);
        c.omega_shoup.assign(omega_shoup, omega_shoup + t.L * f.L);
        c.theta_gamma_lo = theta_gamma_lo;
        c.theta_gamma_hi = theta_gamma_hi;
        c.theta_gamma_sign = theta_gamma_sign != 0;
        c.theta_omega_lo.assign(theta_omega_lo, theta_omega_lo + f.L);
        c.theta_omega_hi.assign(theta_omega_hi, theta_omega_hi + f.L);
        c.theta_omega_sign.assign(theta_omega_sign, theta_omega_sign + f.L);
        c.theta_garner_lo.assign(theta_garner_lo, theta_garner_lo + f.L);
        c.theta_garner_hi.assign(theta_garner_hi, theta_garner_hi + f.L);
        c.theta_garner_shift = theta_garner_shift;
        scaler_upload(s);
        *out = h.release();
    });
}
fhe_status fhe_switcher_create(const fhe_ctx *from, const fhe_ctx *to, fhe_scaler **out) {
    return guard([&] {
        need(from, "from");
        need(to, "to");
        need(out, "out");
        *out = nullptr;
        set_device(*from->c);
        RnsContext rf(from->c->moduli), rt(to->c->moduli);
        auto h = std::make_unique<fhe_scaler>();
        h->s = scaler_create(*from->c, *to->c, rt.product, rf.product);
        *out = h.release();
    });
}
void fhe_scaler_destroy(fhe_scaler *s) { delete s; }
size_t fhe_scaler_number_common_moduli(const fhe_scaler *s) { return s ? s->s->ncommon : 0; }
fhe_status fhe_scaler_get_constants(const fhe_scaler *s, int which, uint64_t *out) {
    return guard([&] {
        need(s, "scaler");
        need(out, "out");
        const ScalerConstants &c = s->s->c;
        auto cp = [&](const std::vector<u64> &v) { std::copy(v.begin(), v.end(), out); };
        switch (which) {
            case 0: cp(c.gamma); break;
            case 1: cp(c.gamma_shoup); break;
            case 2: cp(c.omega); break;
            case 3: cp(c.omega_shoup); break;
            case 4: cp(c.theta_omega_lo); break;
            case 5: cp(c.theta_omega_hi); break;
            case 6: for (size_t i = 0; i < c.nfrom; i++) out[i] = c.theta_omega_sign[i]; break;
            case 7: cp(c.theta_garner_lo); break;
            case 8: cp(c.theta_garner_hi); break;
            case 9:
                out[0] = c.theta_gamma_lo;
                out[1] = c.theta_gamma_hi;
                out[2] = c.theta_gamma_sign ? 1 : 0;
                out[3] = c.theta_garner_shift;
                out[4] = c.is_one ? 1 : 0;
                break;
            default: throw StatusError(FHE_E_ARG, "unknown constant selector");
        }
    });
}
fhe_status fhe_poly_scale_dev(const fhe_scaler *s, const uint64_t *in, uint64_t *out, size_t batch, int repr_is_ntt,
                              void *stream) {
    return guard([&] {
        need(s, "scaler");
        if (batch) {
            need(in, "in");
            need(out, "out");
        }
        set_device(*s->s->from);
        scale_polys(*s->s, in, out, batch, repr_is_ntt != 0, as_stream(stream));
    });
}
fhe_status fhe_poly_scale(const fhe_scaler *s, const uint64_t *in, uint64_t *out, size_t batch, int repr_is_ntt) {
    return guard([&] {
        need(s, "scaler");
        if (batch) {
            need(in, "in");
            need(out, "out");
        }
        const Scaler &sc = *s->s;
        sc.from->need_device();
        set_device(*sc.from);
        HostIO io;
        const size_t ie = sc.from->L * sc.from->n, oe = sc.to->L * sc.to->n;
        u64 *di = io.in(in, batch * ie), *dout = io.out(batch * oe);
        FHE_HIP_CHECK(hipMemsetAsync(dout, 0, std::max<size_t>(batch * oe, 1) * sizeof(u64), nullptr));
        scale_polys(sc, di, dout, batch, repr_is_ntt != 0, nullptr);
        io.back(out, dout, batch * oe);
    });
}

// ----------------------------------------------------------------------------- ksk ----
fhe_status fhe_ksk_create(const fhe_ctx *ct_ctx, const fhe_ctx *ksk_ctx, size_t ndigits, const uint64_t *c0,
                          const uint64_t *c0_shoup, const uint64_t *c1, const uint64_t *c1_shoup, size_t log_base,
                          fhe_ksk **out) {
    return guard([&] {
        need(ct_ctx, "ct_ctx");
        need(ksk_ctx, "ksk_ctx");
        need(out, "out");
        *out = nullptr;
        need(c0, "c0");
        need(c1, "c1");
        const Ctx &kc = *ksk_ctx->c;
        set_device(kc);
        auto h = std::make_unique<fhe_ksk>();
        h->k = make_ksk(*ct_ctx->c, kc, ndigits, log_base);
        const size_t count = ndigits * kc.L * kc.n;
        auto up = [&](DevBuf<u64> &d, const u64 *src) {
            d.alloc(count);
            FHE_HIP_CHECK(hipMemcpy(d.p, src, count * sizeof(u64), hipMemcpyHostToDevice));
        };
        // every coefficient must be a canonical residue (the Shoup quotient of an unreduced value does not fit
        // 64 bits and the key would silently give wrong results); supplied twins must be the twins
        auto shoup_of = [&](const u64 *src, const u64 *given) {
            std::vector<u64> v(count);
            for (size_t i = 0; i < ndigits; i++)
                for (size_t r = 0; r < kc.L; r++)
                    for (size_t j = 0; j < kc.n; j++) {
                        const size_t x = (i * kc.L + r) * kc.n + j;
                        if (src[x] >= kc.moduli[r]) throw StatusError(FHE_E_ARG, "key coefficient not reduced");
                        v[x] = shoup(src[x], kc.moduli[r]);
                        if (given && given[x] != v[x])
                            throw StatusError(FHE_E_ARG, "Shoup twin is not floor(c * 2^64 / q)");
                    }
            return v;
        };
        up(h->k->c0, c0);
        up(h->k->c1, c1);
        {
            auto v = shoup_of(c0, c0_shoup);
            up(h->k->c0s, v.data());
        }
        {
            auto v = shoup_of(c1, c1_shoup);
            up(h->k->c1s, v.data());
        }
        ksk_fill_f64(*h->k, c0, c1);
        *out = h.release();
    });
}
fhe_status fhe_ksk_create_dev(const fhe_ctx *ct_ctx, const fhe_ctx *ksk_ctx, size_t ndigits, const uint64_t *c0,
                              const uint64_t *c1, size_t log_base, void *stream, fhe_ksk **out) {
    return guard([&] {
        need(ct_ctx, "ct_ctx");
        need(ksk_ctx, "ksk_ctx");
        need(out, "out");
        *out = nullptr;
        need(c0, "c0");
        need(c1, "c1");
        const Ctx &kc = *ksk_ctx->c;
        set_device(kc);
        auto h = std::make_unique<fhe_ksk>();
        h->k = make_ksk(*ct_ctx->c, kc, ndigits, log_base);
        const size_t count = ndigits * kc.L * kc.n;
        hipStream_t s = as_stream(stream);
        FHE_HIP_CHECK(hipStreamSynchronize(s));
        std::vector<u64> h0(count), h1(count), s0(count), s1(count);
        FHE_HIP_CHECK(hipMemcpy(h0.data(), c0, count * sizeof(u64), hipMemcpyDeviceToHost));
        FHE_HIP_CHECK(hipMemcpy(h1.data(), c1, count * sizeof(u64), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ndigits; i++)
            for (size_t r = 0; r < kc.L; r++)
                for (size_t j = 0; j < kc.n; j++) {
                    const size_t x = (i * kc.L + r) * kc.n + j;
                    if (h0[x] >= kc.moduli[r] || h1[x] >= kc.moduli[r])
                        throw StatusError(FHE_E_ARG, "key coefficient not reduced");
                    s0[x] = shoup(h0[x], kc.moduli[r]);
                    s1[x] = shoup(h1[x], kc.moduli[r]);
                }
        h->k->c0.upload(h0);
        h->k->c1.upload(h1);
        h->k->c0s.upload(s0);
        h->k->c1s.upload(s1);
        ksk_fill_f64(*h->k, h0.data(), h1.data());
        *out = h.release();
    });
}
void fhe_ksk_destroy(fhe_ksk *k_) { delete k_; }
fhe_status fhe_ksk_set_mode(fhe_ksk *k_, int mode, size_t w_budget) {
    return guard([&] {
        need(k_, "ksk");
        require(mode >= KS_AUTO && mode <= KS_FUSED_SUB, E_ARG, "mode must be 0 (auto), 1 (fused), 2 (unfused), 3 (unfused on sub-block tiles) or 4 (fused on sub-block tiles)");
        k_->k->mode.store(mode, std::memory_order_relaxed);
        k_->k->w_budget.store(w_budget, std::memory_order_relaxed);
    });
}
fhe_status fhe_ksk_get_mode(const fhe_ksk *k_, int *mode, size_t *w_budget) {
    return guard([&] {
        need(k_, "ksk");
        if (mode) *mode = k_->k->mode.load(std::memory_order_relaxed);
        if (w_budget) *w_budget = k_->k->w_budget.load(std::memory_order_relaxed);
    });
}

fhe_status fhe_key_switch_dev(const fhe_ksk *k_, const uint64_t *p, uint64_t *c0_out, uint64_t *c1_out, size_t batch,
                              void *stream) {
    return guard([&] {
        need(k_, "ksk");
        if (batch) {
            need(p, "p");
            need(c0_out, "c0_out");
            need(c1_out, "c1_out");
        }
        const Ksk &ks = *k_->k;
        set_device(*ks.ksk_ctx);
        key_switch_polys(ks, p, (u64)ks.ct_ctx->L * ks.ct_ctx->n, c0_out, c1_out, (u64)ks.ksk_ctx->L * ks.ksk_ctx->n,
                         nullptr, nullptr, 0, batch, as_stream(stream));
    });
}
fhe_status fhe_key_switch(const fhe_ksk *k_, const uint64_t *p, uint64_t *c0_out, uint64_t *c1_out, size_t batch) {
    return guard([&] {
        need(k_, "ksk");
        if (batch) {
            need(p, "p");
            need(c0_out, "c0_out");
            need(c1_out, "c1_out");
        }
        const Ksk &ks = *k_->k;
        set_device(*ks.ksk_ctx);
        HostIO io;
        const size_t ie = ks.ct_ctx->L * ks.ct_ctx->n, oe = ks.ksk_ctx->L * ks.ksk_ctx->n;
        u64 *dp = io.in(p, batch * ie), *d0 = io.out(batch * oe), *d1 = io.out(batch * oe);
        key_switch_polys(ks, dp, ie, d0, d1, oe, nullptr, nullptr, 0, batch, nullptr);
        io.back(c0_out, d0, batch * oe);
        io.back(c1_out, d1, batch * oe);
    });
}

// relinearizes: ct3 [b][3][L][N] -> out [b][2][L][N]
static void relinearize_run(const Ksk &ks, const u64 *ct3, u64 *out, size_t batch, hipStream_t s) {
    const Ctx &cc = *ks.ct_ctx;
    const u64 PL = (u64)cc.L * cc.n;
    if (!batch) return;
    WsGuard c2(batch * PL * sizeof(u64), s);
    k::RowMap m = full_map(cc, cc.L);
    m.src_poly_stride = 3 * PL;
    m.dst_poly_stride = PL;
    launch_ntt(cc, true, ct3 + 2 * PL, c2.u(), m, batch, s);
    // (c2 in Ntt form doubles as the transforms of digit j under key modulus j)
    key_switch_add(ks, c2.u(), PL, ct3, ct3 + PL, 3 * PL, out, out + PL, 2 * PL, batch, s, ct3 + 2 * PL, 3 * PL);
}
fhe_status fhe_bfv_relinearize_dev(const fhe_ksk *rk, const uint64_t *ct3, uint64_t *out, size_t batch, void *stream) {
    return guard([&] {
        need(rk, "rk");
        if (batch) {
            need(ct3, "ct3");
            need(out, "out");
        }
        set_device(*rk->k->ksk_ctx);
        relinearize_run(*rk->k, ct3, out, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_relinearize(const fhe_ksk *rk, const uint64_t *ct3, uint64_t *out, size_t batch) {
    return guard([&] {
        need(rk, "rk");
        if (batch) {
            need(ct3, "ct3");
            need(out, "out");
        }
        const Ksk &ks = *rk->k;
        set_device(*ks.ksk_ctx);
        const size_t pe = ks.ct_ctx->L * ks.ct_ctx->n;
        if (host_sliced(ks.ksk_ctx->device, batch, {{ct3, 3 * pe}}, out, 2 * pe,
                        [&](const std::vector<const u64 *> &d, u64 *o, size_t n, hipStream_t st) {
                            relinearize_run(ks, d[0], o, n, st);
                        }))
            return;
        HostIO io;
        u64 *di = io.in(ct3, batch * 3 * pe), *dout = io.out(batch * 2 * pe);
        relinearize_run(ks, di, dout, batch, nullptr);
        io.back(out, dout, batch * 2 * pe);
    });
}

static void galois_run(const Ksk &ks, size_t exponent, const u64 *ct, u64 *out, size_t batch, hipStream_t s) {
    galois_apply(ks, exponent, ct, out, batch, s);
}
fhe_status fhe_bfv_galois_dev(const fhe_ksk *gk, size_t exponent, const uint64_t *ct, uint64_t *out, size_t batch,
                              void *stream) {
    return guard([&] {
        need(gk, "gk");
        if (batch) {
            need(ct, "ct");
            need(out, "out");
        }
        set_device(*gk->k->ksk_ctx);
        galois_run(*gk->k, exponent, ct, out, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_galois(const fhe_ksk *gk, size_t exponent, const uint64_t *ct, uint64_t *out, size_t batch) {
    return guard([&] {
        need(gk, "gk");
        if (batch) {
            need(ct, "ct");
            need(out, "out");
        }
        const Ksk &ks = *gk->k;
        set_device(*ks.ksk_ctx);
        const size_t pe = ks.ct_ctx->L * ks.ct_ctx->n;
        if (host_sliced(ks.ksk_ctx->device, batch, {{ct, 2 * pe}}, out, 2 * pe,
                        [&](const std::vector<const u64 *> &d, u64 *o, size_t n, hipStream_t st) {
                            galois_run(ks, exponent, d[0], o, n, st);
                        }))
            return;
        HostIO io;
        u64 *di = io.in(ct, batch * 2 * pe), *dout = io.out(batch * 2 * pe);
        galois_run(ks, exponent, di, dout, batch, nullptr);
        io.back(out, dout, batch * 2 * pe);
    });
}

fhe_status fhe_bfv_switch_down_dev(const fhe_ctx *ctx, size_t nparts, const uint64_t *ct, uint64_t *out, size_t batch,
                                   void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        if (!c.next) throw StatusError(FHE_E_NO_MORE_CONTEXT, "NoMoreContext");
        if (batch && nparts) {
            need(ct, "ct");
            need(out, "out");
        }
        bfv_switch_down(c, nparts, ct, out, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_switch_down(const fhe_ctx *ctx, size_t nparts, const uint64_t *ct, uint64_t *out, size_t batch) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (batch && nparts) {
            need(ct, "ct");
            need(out, "out");
        }
        if (!c.next) throw StatusError(FHE_E_NO_MORE_CONTEXT, "NoMoreContext");
        HostIO io;
        u64 *di = io.in(ct, batch * nparts * pe), *dout = io.out(batch * nparts * (pe - c.n));
        bfv_switch_down(c, nparts, di, dout, batch, nullptr);
        io.back(out, dout, batch * nparts * (pe - c.n));
    });
}

// Ciphertext::switch_to_level (F/bfv/ciphertext.rs:164-183), `levels` = target_level - level
static void check_levels(const Ctx &c, size_t levels) {
    if (!c.at_level(levels))
        throw StatusError(FHE_E_INVALID_LEVEL, "InvalidLevel: the context chain has " + std::to_string(c.L - 1) +
                                                   " levels below this one, asked for " + std::to_string(levels));
}
fhe_status fhe_bfv_switch_to_level_dev(const fhe_ctx *ctx, size_t levels, size_t nparts, const uint64_t *ct,
                                       uint64_t *out, size_t batch, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        check_levels(c, levels);
        if (batch && nparts) {
            need(ct, "ct");
            need(out, "out");
        }
        switch_down_to_ntt(c, levels, ct, out, batch * nparts, as_stream(stream));
    });
}
fhe_status fhe_bfv_switch_to_level(const fhe_ctx *ctx, size_t levels, size_t nparts, const uint64_t *ct, uint64_t *out,
                                   size_t batch) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        check_levels(c, levels);
        if (batch && nparts) {
            need(ct, "ct");
            need(out, "out");
        }
        const size_t po = pe - levels * c.n;
        HostIO io;
        u64 *di = io.in(ct, batch * nparts * pe), *dout = io.out(batch * nparts * po);
        switch_down_to_ntt(c, levels, di, dout, batch * nparts, nullptr);
        io.back(out, dout, batch * nparts * po);
    });
}

// ------------------------------------------------------ PIR / RGSW / inner sum ----
fhe_status fhe_bfv_dot_product_scalar_dev(const fhe_ctx *ctx, size_t nparts, size_t count, const uint64_t *cts,
                                          int cts_shared, const uint64_t *pts, int pts_shared, uint64_t *out,
                                          size_t batch, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        if (batch && nparts && count) {
            need(cts, "cts");
            need(pts, "pts");
            need(out, "out");
        }
        dot_product_scalar(c, nparts, count, cts, cts_shared != 0, pts, pts_shared != 0, out, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_dot_product_scalar(const fhe_ctx *ctx, size_t nparts, size_t count, const uint64_t *cts,
                                      int cts_shared, const uint64_t *pts, int pts_shared, uint64_t *out,
                                      size_t batch) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (count == 0) throw StatusError(FHE_E_EMPTY_DOT_PRODUCT, "EmptyDotProduct: no operands");
        if (batch && nparts) {
            need(cts, "cts");
            need(pts, "pts");
            need(out, "out");
        }
        HostIO io;
        u64 *dc = io.in(cts, (cts_shared ? 1 : batch) * count * nparts * pe);
        u64 *dp = io.in(pts, (pts_shared ? 1 : batch) * count * pe);
        u64 *dout = io.out(batch * nparts * pe);
        dot_product_scalar(c, nparts, count, dc, cts_shared != 0, dp, pts_shared != 0, dout, batch, nullptr);
        io.back(out, dout, batch * nparts * pe);
    });
}
fhe_status fhe_bfv_mul_plain_dev(const fhe_ctx *ctx, size_t nparts, const uint64_t *ct, const uint64_t *pt,
                                 int pt_shared, uint64_t *out, size_t batch, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        (void)pe;
        if (batch && nparts) {
            need(ct, "ct");
            need(pt, "pt");
            need(out, "out");
        }
        mul_plain(c, nparts, ct, pt, pt_shared != 0, out, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_mul_plain(const fhe_ctx *ctx, size_t nparts, const uint64_t *ct, const uint64_t *pt, int pt_shared,
                             uint64_t *out, size_t batch) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        if (batch && nparts) {
            need(ct, "ct");
            need(pt, "pt");
            need(out, "out");
        }
        HostIO io;
        u64 *dc = io.in(ct, batch * nparts * pe), *dp = io.in(pt, (pt_shared ? 1 : batch) * pe);
        u64 *dout = io.out(batch * nparts * pe);
        mul_plain(c, nparts, dc, dp, pt_shared != 0, dout, batch, nullptr);
        io.back(out, dout, batch * nparts * pe);
    });
}
fhe_status fhe_bfv_rgsw_mul_dev(const fhe_ksk *ksk0, const fhe_ksk *ksk1, const uint64_t *ct, uint64_t *out,
                                size_t batch, void *stream) {
    return guard([&] {
        need(ksk0, "ksk0");
        need(ksk1, "ksk1");
        if (batch) {
            need(ct, "ct");
            need(out, "out");
        }
        set_device(*ksk0->k->ksk_ctx);
        rgsw_mul(*ksk0->k, *ksk1->k, ct, out, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_rgsw_mul(const fhe_ksk *ksk0, const fhe_ksk *ksk1, const uint64_t *ct, uint64_t *out, size_t batch) {
    return guard([&] {
        need(ksk0, "ksk0");
        need(ksk1, "ksk1");
        if (batch) {
            need(ct, "ct");
            need(out, "out");
        }
        const Ksk &k0 = *ksk0->k;
        set_device(*k0.ksk_ctx);
        const size_t pe = k0.ct_ctx->L * k0.ct_ctx->n;
        HostIO io;
        u64 *di = io.in(ct, batch * 2 * pe), *dout = io.out(batch * 2 * pe);
        rgsw_mul(k0, *ksk1->k, di, dout, batch, nullptr);
        io.back(out, dout, batch * 2 * pe);
    });
}
static void inner_sum_args(const fhe_ksk *const *gks, const size_t *exponents, size_t ngk, std::vector<const Ksk *> &v) {
    need(gks, "gks");
    need(exponents, "exponents");
    if (ngk == 0) throw StatusError(FHE_E_ARG, "inner sum needs at least one Galois key");
    for (size_t i = 0; i < ngk; i++) {
        need(gks[i], "gks[i]");
        v.push_back(gks[i]->k.get());
    }
}
fhe_status fhe_bfv_inner_sum_dev(const fhe_ksk *const *gks, const size_t *exponents, size_t ngk, const uint64_t *ct,
                                 uint64_t *out, size_t batch, void *stream) {
    return guard([&] {
        std::vector<const Ksk *> v;
        inner_sum_args(gks, exponents, ngk, v);
        if (batch) {
            need(ct, "ct");
            need(out, "out");
        }
        set_device(*v[0]->ksk_ctx);
        inner_sum(v.data(), exponents, ngk, ct, out, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_inner_sum(const fhe_ksk *const *gks, const size_t *exponents, size_t ngk, const uint64_t *ct,
                             uint64_t *out, size_t batch) {
    return guard([&] {
        std::vector<const Ksk *> v;
        inner_sum_args(gks, exponents, ngk, v);
        if (batch) {
            need(ct, "ct");
            need(out, "out");
        }
        set_device(*v[0]->ksk_ctx);
        const size_t pe = v[0]->ct_ctx->L * v[0]->ct_ctx->n;
        HostIO io;
        u64 *di = io.in(ct, batch * 2 * pe), *dout = io.out(batch * 2 * pe);
        inner_sum(v.data(), exponents, ngk, di, dout, batch, nullptr);
        io.back(out, dout, batch * 2 * pe);
    });
}

fhe_status fhe_poly_from_seed_dev(const fhe_ctx *ctx, const uint8_t *seeds, uint64_t *out, size_t batch, void *stream) {
    return guard([&] {
        need(ctx, "ctx");
        if (batch) {
            need(seeds, "seeds");
            need(out, "out");
        }
        set_device(*ctx->c);
        polys_from_seeds(*ctx->c, seeds, out, batch, as_stream(stream));
    });
}
fhe_status fhe_poly_from_seed(const fhe_ctx *ctx, const uint8_t *seeds, uint64_t *out, size_t batch) {
    return guard([&] {
        need(ctx, "ctx");
        if (batch) {
            need(seeds, "seeds");
            need(out, "out");
        }
        const Ctx &c = *ctx->c;
        c.need_device();
        set_device(c);
        const size_t pe = c.L * c.n;
        HostIO io;
        u64 *ds = io.out((batch * 32 + 7) / 8), *dout = io.out(batch * pe);
        if (batch) FHE_HIP_CHECK(hipMemcpy(ds, seeds, batch * 32, hipMemcpyHostToDevice));
        polys_from_seeds(c, reinterpret_cast<const uint8_t *>(ds), dout, batch, nullptr);
        io.back(out, dout, batch * pe);
    });
}
fhe_status fhe_bfv_decrypt_dev(const fhe_scaler *sc, uint64_t t, const uint64_t *s_ntt, const uint64_t *ct, size_t nparts,
                               uint64_t *out, size_t batch, void *stream) {
    return guard([&] {
        need(sc, "cipher_plain_scaler");
        need(s_ntt, "s_ntt");
        if (batch) {
            need(ct, "ct");
            need(out, "out");
        }
        set_device(*sc->s->from);
        decrypt(*sc->s, t, s_ntt, ct, nparts, out, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_decrypt(const fhe_scaler *sc, uint64_t t, const uint64_t *s_ntt, const uint64_t *ct, size_t nparts,
                           uint64_t *out, size_t batch) {
    return guard([&] {
        need(sc, "cipher_plain_scaler");
        need(s_ntt, "s_ntt");
        if (batch) {
            need(ct, "ct");
            need(out, "out");
        }
        const Ctx &cc = *sc->s->from;
        cc.need_device();
        set_device(cc);
        const size_t pe = cc.L * cc.n;
        HostIO io;
        // the staged secret key and the plaintext are cleared on the device before their buffers are freed
        // (the reference wraps s, c and the result in Zeroizing, F/bfv/keys/secret_key.rs:198-226)
        u64 *ds = io.secret(io.in(s_ntt, pe), pe), *di = io.in(ct, batch * nparts * pe);
        u64 *dout = io.secret(io.out(batch * cc.n), batch * cc.n);
        decrypt(*sc->s, t, ds, di, nparts, dout, batch, nullptr);
        io.back(out, dout, batch * cc.n);
    });
}

static const Ctx &expand_args(const fhe_ksk *const *gks, size_t nlevels, std::vector<const Ksk *> &v) {
    need(gks, "gks");
    if (nlevels == 0) throw StatusError(FHE_E_ARG, "expansion needs at least one Galois key");
    for (size_t i = 0; i < nlevels; i++) {
        need(gks[i], "gks[i]");
        v.push_back(gks[i]->k.get());
    }
    return *v[0]->ct_ctx;
}
fhe_status fhe_bfv_expand_dev(const fhe_ksk *const *gks, size_t nlevels, const uint64_t *ct, uint64_t *out, size_t size,
                              size_t batch, void *stream) {
    return guard([&] {
        std::vector<const Ksk *> v;
        expand_args(gks, nlevels, v);
        if (batch && size) {
            need(ct, "ct");
            need(out, "out");
        }
        set_device(*v[0]->ksk_ctx);
        expand(v.data(), nlevels, ct, out, size, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_expand(const fhe_ksk *const *gks, size_t nlevels, const uint64_t *ct, uint64_t *out, size_t size,
                          size_t batch) {
    return guard([&] {
        std::vector<const Ksk *> v;
        const Ctx &cc = expand_args(gks, nlevels, v);
        if (batch && size) {
            need(ct, "ct");
            need(out, "out");
        }
        set_device(*v[0]->ksk_ctx);
        const size_t ce = 2 * cc.L * cc.n;
        const bool ok = size >= 1 && size <= cc.n;  // (the engine reports the error; do not size buffers from it)
        HostIO io;
        u64 *di = io.in(ct, batch * ce), *dout = io.out(ok ? size * batch * ce : 0);
        expand(v.data(), nlevels, di, dout, size, batch, nullptr);
        io.back(out, dout, size * batch * ce);
    });
}

// ----------------------------------------------------------------------------- mul ----
static std::unique_ptr<Mul> make_mul(const Scaler *el, const Scaler *er, const Scaler *dn, const Ksk *rk,
                                     bool mod_switch) {
    require(el->from->same_ring(*er->from) && el->to->same_ring(*er->to), E_PARAMETER_MISMATCH,
            "extenders disagree on contexts");
    require(dn->from->same_ring(*el->to) && dn->to->same_ring(*el->from), E_PARAMETER_MISMATCH,
            "down scaler does not invert the extenders' contexts");
    auto m = std::make_unique<Mul>();
    m->ext_lhs = el;
    m->ext_rhs = er;
    m->down = dn;
    m->rk = rk;
    m->mod_switch = mod_switch;
    m->base = el->from;
    m->mulc = el->to;
    // the extenders hand their common rows over in Ntt form (M/rq/scaler.rs:61-65): those moduli must be evaluated in
    // the same order on both sides, i.e. both contexts were built from the same NTT tables
    require(m->base->same_tables(*m->mulc, std::min(el->ncommon, er->ncommon)) &&
                el->from->same_tables(*er->from, el->from->L) && el->to->same_tables(*er->to, el->to->L) &&
                dn->from->same_tables(*el->to, el->to->L) && dn->to->same_tables(*el->from, el->from->L),
            E_PARAMETER_MISMATCH, "ParameterMismatch: the multiplicator's contexts were built with different NTT tables");
    if (rk) {  // enable_relinearization (mul.rs:141-151)
        require(rk->ct_ctx->same_ring(*m->base), E_PARAMETER_MISMATCH,
                "ParameterMismatch: relinearization key level != multiplicator level");
        require(rk->ct_ctx->same_tables(*m->base, m->base->L), E_PARAMETER_MISMATCH,
                "ParameterMismatch: relinearization key and multiplicator were built with different NTT tables");
    }
    if (mod_switch) require(m->base->next != nullptr, E_NO_MORE_CONTEXT, "NoMoreContext");  // mul.rs:155-162
    return m;
}
fhe_status fhe_mul_create(const fhe_scaler *extender_lhs, const fhe_scaler *extender_rhs, const fhe_scaler *down_scaler,
                          const fhe_ksk *rk_or_null, int mod_switch, fhe_mul **out) {
    return guard([&] {
        need(extender_lhs, "extender_lhs");
        need(extender_rhs, "extender_rhs");
        need(down_scaler, "down_scaler");
        need(out, "out");
        *out = nullptr;
        auto h = std::make_unique<fhe_mul>();
        h->m = make_mul(extender_lhs->s.get(), extender_rhs->s.get(), down_scaler->s.get(),
                        rk_or_null ? rk_or_null->k.get() : nullptr, mod_switch != 0);
        *out = h.release();
    });
}
void fhe_mul_destroy(fhe_mul *m) { delete m; }
fhe_status fhe_mul_out_shape(const fhe_mul *m, size_t *parts, size_t *rows) {
    return guard([&] {
        need(m, "mul");
        if (parts) *parts = m->m->out_parts();
        if (rows) *rows = m->m->out_rows();
    });
}
fhe_status fhe_mul_basis(const fhe_mul *m, size_t *count, uint64_t *moduli) {
    return guard([&] {
        need(m, "mul");
        need(count, "count");
        *count = m->m->mulc->L;
        if (moduli) std::copy(m->m->mulc->moduli.begin(), m->m->mulc->moduli.end(), moduli);
    });
}
fhe_status fhe_mul_set_chunk(fhe_mul *m, size_t chunk) {
    return guard([&] {
        need(m, "mul");
        m->m->chunk.store(chunk, std::memory_order_relaxed);
    });
}
fhe_status fhe_mul_set_streams(fhe_mul *m, size_t streams) {
    return guard([&] {
        need(m, "mul");
        require(streams == 1 || streams == 2, E_ARG, "streams must be 1 or 2");
        m->m->streams.store(streams, std::memory_order_relaxed);
    });
}
fhe_status fhe_mul_get_options(const fhe_mul *m, size_t *chunk, size_t *streams) {
    return guard([&] {
        need(m, "mul");
        if (chunk) *chunk = m->m->chunk.load(std::memory_order_relaxed);
        if (streams) *streams = m->m->streams.load(std::memory_order_relaxed);
    });
}
fhe_status fhe_bfv_mul_dev(const fhe_mul *m, const uint64_t *lhs, const uint64_t *rhs, uint64_t *out, size_t batch,
                           void *stream) {
    return guard([&] {
        need(m, "mul");
        if (batch) {
            need(lhs, "lhs");
            need(rhs, "rhs");
            need(out, "out");
        }
        set_device(*m->m->base);
        bfv_mul(*m->m, lhs, rhs, out, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_mul(const fhe_mul *m, const uint64_t *lhs, const uint64_t *rhs, uint64_t *out, size_t batch) {
    return guard([&] {
        need(m, "mul");
        if (batch) {
            need(lhs, "lhs");
            need(rhs, "rhs");
            need(out, "out");
        }
        const Mul &mm = *m->m;
        mm.base->need_device();
        set_device(*mm.base);
        const size_t ie = 2 * mm.base->L * mm.base->n, oe = mm.out_parts() * mm.out_rows() * mm.base->n;
        // a large batch goes through in slices whose copies and pipeline overlap (host_sliced)
        if (host_sliced(mm.base->device, batch, {{lhs, ie}, {rhs, ie}}, out, oe,
                        [&](const std::vector<const u64 *> &d, u64 *o, size_t n, hipStream_t st) {
                            bfv_mul(mm, d[0], d[1], o, n, st);
                        }))
            return;
        HostIO io;
        u64 *dl = io.in(lhs, batch * ie), *dr = lhs == rhs ? dl : io.in(rhs, batch * ie), *dout = io.out(batch * oe);
        bfv_mul(mm, dl, dr, dout, batch, nullptr);   // (lhs == rhs: one upload, and the squaring shortcut of bfv_mul)
        io.back(out, dout, batch * oe);
    });
}

fhe_status fhe_bfv_tensor_dev(const fhe_mul *m, size_t lhs_parts, size_t rhs_parts, const uint64_t *lhs,
                              const uint64_t *rhs, uint64_t *out, size_t batch, void *stream) {
    return guard([&] {
        need(m, "mul");
        if (batch) {
            need(lhs, "lhs");
            need(rhs, "rhs");
            need(out, "out");
        }
        const Mul &mm = *m->m;
        mm.base->need_device();
        set_device(*mm.base);
        bfv_tensor(mm, lhs_parts, rhs_parts, lhs, rhs, out, batch, as_stream(stream));
    });
}
fhe_status fhe_bfv_tensor(const fhe_mul *m, size_t lhs_parts, size_t rhs_parts, const uint64_t *lhs, const uint64_t *rhs,
                          uint64_t *out, size_t batch) {
    return guard([&] {
        need(m, "mul");
        if (batch) {
            need(lhs, "lhs");
            need(rhs, "rhs");
            need(out, "out");
        }
        if (lhs_parts < 1 || rhs_parts < 1)
            throw StatusError(FHE_E_MUL_POLY_COUNT, "a ciphertext has at least one part");
        const Mul &mm = *m->m;
        mm.base->need_device();
        set_device(*mm.base);
        const size_t pe = mm.base->L * mm.base->n, lo = lhs_parts + rhs_parts - 1;
        HostIO io;
        u64 *dl = io.in(lhs, batch * lhs_parts * pe), *dr = io.in(rhs, batch * rhs_parts * pe);
        u64 *dout = io.out(batch * lo * pe);
        bfv_tensor(mm, lhs_parts, rhs_parts, dl, dr, dout, batch, nullptr);
        io.back(out, dout, batch * lo * pe);
    });
}

// -------------------------------------------------------------------------- params ----
fhe_status fhe_params_create_with_tables(int device, size_t degree, size_t nmoduli, const uint64_t *moduli,
                                         uint64_t plaintext_modulus, fhe_ntt_tables_fn tables, void *user,
                                         fhe_params **out) {
    return guard([&] {
        need(out, "out");
        *out = nullptr;
        if (nmoduli == 0) throw StatusError(FHE_E_EMPTY_MODULI, "EmptyModuli");
        need(moduli, "moduli");
        require(plaintext_modulus >= 2, E_ARG, "plaintext modulus must be >= 2");
        auto p = std::make_unique<fhe_params>();
        p->device = device;
        p->host_tables = tables != nullptr;
        p->degree = degree;
        p->plaintext = plaintext_modulus;
        p->moduli.assign(moduli, moduli + nmoduli);
        for (u64 q : p->moduli) p->moduli_sizes.push_back(64 - (size_t)__builtin_clzll(q | 1));
        p->top = wrap_ctx(ctx_create_cb(device, degree, p->moduli, tables, user, p->host_tables, p->table_cache));
        // extended basis: n+1 primes of 62 bits (parameters.rs:660-676)
        std::vector<u64> ext = extended_basis_primes(degree, p->moduli, nmoduli + 1);
        BigUint t(plaintext_modulus);
        for (size_t level = 0; level < nmoduli; level++) {
            const size_t nl = nmoduli - level;
            size_t modulus_size = 0;
            for (size_t i = 0; i < nl; i++) modulus_size += p->moduli_sizes[i];
            const size_t n_moduli = (modulus_size + 60 + 61) / 62;  // div_ceil
            std::vector<u64> mm(p->moduli.begin(), p->moduli.begin() + nl);
            mm.insert(mm.end(), ext.begin(), ext.begin() + n_moduli);
            auto mc = wrap_ctx(ctx_create_cb(device, degree, mm, tables, user, p->host_tables, p->table_cache));
            const Ctx &base = *p->top->c->at_level(level);
            RnsContext rb(base.moduli);
            auto e = std::make_unique<fhe_scaler>();
            e->s = scaler_create(base, *mc->c, BigUint(1), BigUint(1));
            auto d = std::make_unique<fhe_scaler>();
            d->s = scaler_create(*mc->c, base, t, rb.product);
            p->mul_ctx.push_back(std::move(mc));
            p->extender.push_back(std::move(e));
            p->down.push_back(std::move(d));
        }
        *out = p.release();
    });
}
fhe_status fhe_params_create(int device, size_t degree, size_t nmoduli, const uint64_t *moduli,
                             uint64_t plaintext_modulus, fhe_params **out) {
    return fhe_params_create_with_tables(device, degree, nmoduli, moduli, plaintext_modulus, nullptr, nullptr, out);
}
void fhe_params_destroy(fhe_params *p) { delete p; }
size_t fhe_params_max_level(const fhe_params *p) { return p ? p->moduli.size() - 1 : 0; }
#define FHE_PARAMS_LEVEL_CHECK()                                                          \
    need(p, "params");                                                                    \
    need(out, "out");                                                                     \
    if (level >= p->moduli.size()) throw StatusError(FHE_E_INVALID_LEVEL, "InvalidLevel")
fhe_status fhe_params_ctx(const fhe_params *p, size_t level, const fhe_ctx **out) {
    return guard([&] {
        FHE_PARAMS_LEVEL_CHECK();
        *out = ctx_level_handle(p->top.get(), level);
    });
}
fhe_status fhe_params_mul_ctx(const fhe_params *p, size_t level, const fhe_ctx **out) {
    return guard([&] {
        FHE_PARAMS_LEVEL_CHECK();
        *out = p->mul_ctx[level].get();
    });
}
fhe_status fhe_params_extender(const fhe_params *p, size_t level, const fhe_scaler **out) {
    return guard([&] {
        FHE_PARAMS_LEVEL_CHECK();
        *out = p->extender[level].get();
    });
}
fhe_status fhe_params_down_scaler(const fhe_params *p, size_t level, const fhe_scaler **out) {
    return guard([&] {
        FHE_PARAMS_LEVEL_CHECK();
        *out = p->down[level].get();
    });
}
fhe_status fhe_mul_create_default(const fhe_params *p, size_t level, const fhe_ksk *rk_or_null, int mod_switch,
                                  fhe_mul **out) {
    return guard([&] {
        FHE_PARAMS_LEVEL_CHECK();
        *out = nullptr;
        auto h = std::make_unique<fhe_mul>();
        const Scaler *ext = p->extender[level]->s.get(), *down = p->down[level]->s.get();
        if (rk_or_null) {
            // Multiplicator::default (mul.rs:101-138): the extension primes skip only the moduli of rk's level
            // (parameters.rs:660-676 skips every top-level modulus).  Where the two bases differ -- a dropped
            // top-level modulus is itself one of the first 62-bit NTT primes -- the handle gets its own
            // multiplication context and scalers, exactly the reference's new_leveled_internal.
            const Ctx &base = *p->top->c->at_level(level);
            size_t modulus_size = 0;
            for (size_t i = 0; i < base.L; i++) modulus_size += p->moduli_sizes[i];
            const size_t n_moduli = (modulus_size + 60 + 61) / 62;
            std::vector<u64> mm = base.moduli;
            const std::vector<u64> extp = extended_basis_primes(p->degree, base.moduli, n_moduli);
            mm.insert(mm.end(), extp.begin(), extp.end());
            if (mm != p->mul_ctx[level]->c->moduli) {
                // (every prime of this basis is a top-level modulus or one of the cached extension primes)
                auto mc = wrap_ctx(ctx_create_cb(p->device, p->degree, mm, nullptr, nullptr, p->host_tables,
                                                 const_cast<fhe_params *>(p)->table_cache));
                RnsContext rb(base.moduli);
                auto e = std::make_unique<fhe_scaler>();
                e->s = scaler_create(base, *mc->c, BigUint(1), BigUint(1));
                auto d = std::make_unique<fhe_scaler>();
                d->s = scaler_create(*mc->c, base, BigUint(p->plaintext), rb.product);
                ext = e->s.get();
                down = d->s.get();
                h->ctxs.push_back(std::move(mc));
                h->scalers.push_back(std::move(e));
                h->scalers.push_back(std::move(d));
            }
        }
        h->m = make_mul(ext, ext, down, rk_or_null ? rk_or_null->k.get() : nullptr, mod_switch != 0);
        *out = h.release();
    });
}

// --------------------------------------------------------------------------- primes ----
uint64_t fhe_generate_prime(size_t num_bits, uint64_t modulo, uint64_t upper_bound) {
    return generate_prime(num_bits, modulo, upper_bound);
}
int fhe_supports_opt(uint64_t p) { return supports_opt(p) ? 1 : 0; }
int fhe_is_prime(uint64_t p) { return is_prime_u64(p) ? 1 : 0; }
fhe_status fhe_generate_moduli(const size_t *sizes, size_t count, size_t degree, uint64_t *out) {
    return guard([&] {
        need(sizes, "sizes");
        need(out, "out");
        std::vector<size_t> sz(sizes, sizes + count);
        auto m = generate_moduli(sz, degree);
        std::copy(m.begin(), m.end(), out);
    });
}

// ---------------------------------------------------------------------- bench / test ----
fhe_status fhe_synth_uniform_dev(const fhe_ctx *ctx, uint64_t seed, uint64_t ct0, uint64_t part0, size_t nparts,
                                 uint64_t *out, size_t batch, void *stream) {
    return guard([&] {
        FHE_POLY_IO_PROLOGUE(ctx);
        const u64 total = (u64)batch * nparts * pe;
        if (!total) return;
        need(out, "out");
        FHE_LAUNCH("synth", k::synth_kernel, dim3(blocks_for(total, EW_THREADS)), dim3(EW_THREADS), 0,
                   as_stream(stream), out, c.dmods(), (uint32_t)c.L, (uint32_t)c.logn, (uint32_t)nparts, seed, ct0,
                   part0, total);
    });
}
size_t fhe_workspace_trim(void) {
    AuxStreams::get().drop(-1, nullptr, true);   // internal streams, their blocks and the pooled events go too
    const size_t freed = Workspace::get().trim();
    DevPools::get().trim();                      // what the private pools kept (scratch blocks, fhe_buf_alloc_async)
    return freed;
}
fhe_status fhe_workspace_set_limit(size_t per_stream_bytes, size_t total_bytes) {
    return guard([&] { Workspace::get().set_limits(per_stream_bytes, total_bytes); });
}
fhe_status fhe_workspace_get_limit(size_t *per_stream_bytes, size_t *total_bytes) {
    return guard([&] { Workspace::get().get_limits(per_stream_bytes, total_bytes); });
}
fhe_status fhe_workspace_stats(size_t *held_bytes, size_t *in_use_bytes, size_t *blocks, size_t *owners,
                               size_t *internal_streams) {
    return guard([&] {
        Workspace::get().stats(held_bytes, in_use_bytes, blocks, owners);
        if (internal_streams) *internal_streams = AuxStreams::get().count();
    });
}
fhe_status fhe_workspace_pool_stats(int device, size_t *scratch_reserved_bytes, size_t *scratch_used_bytes,
                                    size_t *buffers_reserved_bytes, size_t *buffers_used_bytes) {
    return guard([&] {
        size_t v[4] = {0, 0, 0, 0};
        DevPools::get().stats(device, v);
        if (scratch_reserved_bytes) *scratch_reserved_bytes = v[0];
        if (scratch_used_bytes) *scratch_used_bytes = v[1];
        if (buffers_reserved_bytes) *buffers_reserved_bytes = v[2];
        if (buffers_used_bytes) *buffers_used_bytes = v[3];
    });
}
fhe_status fhe_ubench_int(int device, int which, double min_seconds, double *ops_per_s) {
    return guard([&] {
        need(ops_per_s, "ops_per_s");
        *ops_per_s = 0;
        int ndev = 0;
        FHE_HIP_CHECK(hipGetDeviceCount(&ndev));
        require(device >= 0 && device < ndev, E_NO_DEVICE, "no such HIP device");
        *ops_per_s = ub::run(device, which, min_seconds);
    });
}
fhe_status fhe_ubench_scaler(const fhe_scaler *scaler, double min_seconds, double *columns_per_s) {
    return guard([&] {
        need(scaler, "scaler");
        need(columns_per_s, "columns_per_s");
        *columns_per_s = ub::run_scaler(*scaler->s, min_seconds);
    });
}
fhe_status fhe_ubench_copy(int device, size_t bytes, double min_seconds, double *bytes_per_s) {
    return guard([&] {
        need(bytes_per_s, "bytes_per_s");
        *bytes_per_s = 0;
        int ndev = 0;
        FHE_HIP_CHECK(hipGetDeviceCount(&ndev));
        require(device >= 0 && device < ndev, E_NO_DEVICE, "no such HIP device");
        *bytes_per_s = ub::run_copy(device, bytes, min_seconds);
    });
}
void fhe_engine_set_f64(int on) { f64_enabled_flag().store(on != 0, std::memory_order_relaxed); }
int fhe_engine_get_f64(void) { return f64_enabled_flag().load(std::memory_order_relaxed) ? 1 : 0; }
void fhe_prof_enable(int on) { Profiler::get().enabled = on != 0; }
void fhe_prof_reset(void) {
    try {
        Profiler::get().reset();
    } catch (...) {
    }
}
size_t fhe_prof_count(void) {
    try {
        Profiler::get().drain();
    } catch (...) {
    }
    return Profiler::get().entries.size();
}
fhe_status fhe_prof_get(size_t index, char *name, size_t name_cap, uint64_t *launches, double *total_ms) {
    return guard([&] {
        Profiler &p = Profiler::get();
        p.drain();
        if (index >= p.entries.size()) throw StatusError(FHE_E_ARG, "profile index out of range");
        const auto &e = p.entries[index];
        if (name && name_cap) std::snprintf(name, name_cap, "%s", e.name.c_str());
        if (launches) *launches = e.launches;
        if (total_ms) *total_ms = e.ms;
    });
}
fhe_status fhe_prof_get_symbol(size_t index, char *symbol, size_t symbol_cap) {
    return guard([&] {
        Profiler &p = Profiler::get();
        std::lock_guard<std::mutex> lk(p.mu);
        if (index >= p.entries.size()) throw StatusError(FHE_E_ARG, "profile index out of range");
        if (!symbol || !symbol_cap) throw StatusError(FHE_E_ARG, "symbol buffer is null");
        const auto &e = p.entries[index];
        const char *mangled = e.fn ? hipKernelNameRefByPtr(e.fn, nullptr) : nullptr;
        std::string text = e.name;                     // (no symbol known to the runtime: the label)
        if (mangled && *mangled) {
            int st = 0;
            char *dm = abi::__cxa_demangle(mangled, nullptr, nullptr, &st);
            text = (st == 0 && dm) ? dm : mangled;
            std::free(dm);
        }
        std::snprintf(symbol, symbol_cap, "%s", text.c_str());
    });
}

#if defined(FHE_LAB) && defined(FHE_PHASE_TIMING)
// Lab builds only (not in the header): reads and clears the phase-timing slots of kernels.hpp.
fhe_status fhe_debug_phase_timing(uint64_t *out, size_t n) {
    return guard([&] {
        static unsigned long long h[FHE_TS_TOTAL_SLOTS], z[FHE_TS_TOTAL_SLOTS];
        FHE_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(fhe::k::g_phase_ts), sizeof(h)));
        FHE_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(fhe::k::g_phase_ts), z, sizeof(z)));
        for (size_t i = 0; i < n && i < FHE_TS_TOTAL_SLOTS; i++) out[i] = h[i];
    });
}
#endif

}  // extern "C"
