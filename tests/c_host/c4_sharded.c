/* c4_sharded.c -- BASELINE config C4's split as a host WITHOUT Python, torch or a process per GPU does it:
 * ONE process, G devices, G host threads, each with its own handles (parameter set, key, multiplicator), stream and
 * buffers on its device, each multiplying its contiguous block of the batch (fhe.rs_amd/shard.py's shard_bounds:
 * the first total % G shards get one pair more).  No collective on the data path: the shards never talk.
 * This is what the Rust shim's `HipMul` / `DeviceCiphertexts` do behind RAII when a Rust host shards a batch over
 * the GPUs of a node (rust/fhe-math-hip/src/lib.rs; fhe.rs itself is single-threaded and has no such notion).
 *
 *   c4_sharded <seed> <plaintext_modulus> <total_batch> <devices (0 = all visible)> [<degree, default 8192>]
 *
 * Check: the same total batch is also multiplied in ONE call on device 0, and every shard's output must equal the
 * corresponding slice of that result byte for byte (the reference result itself is pinned by tests/c_host/c2_golden.c
 * and the parity suites).  Prints the wall-clock rate of the sharded phase.  Exit code 0 = all shards identical.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fhe_hip.h"

enum { L = 4 };
static size_t N = 8192;

typedef struct {
    int device, ndev, rc;
    uint64_t seed, t;
    size_t total, begin, end;
    uint64_t *host_out;   /* [end - begin][2][L][N], filled by the worker */
    double seconds;       /* the multiply alone: enqueue to stream-sync */
    char err[256];
} shard_t;

#define WCHECK(call)                                                                              \
    do {                                                                                          \
        fhe_status st_ = (call);                                                                  \
        if (st_ != FHE_OK) {                                                                      \
            snprintf(sh->err, sizeof sh->err, "%s -> %d (%s)", #call, (int)st_, fhe_last_error()); \
            sh->rc = 2;                                                                           \
            return NULL;                                                                          \
        }                                                                                         \
    } while (0)

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* everything a shard needs lives on ITS device: handles are per device (fhe_params_create(device, ...)) */
static void *worker(void *arg) {
    shard_t *sh = (shard_t *)arg;
    const int d = sh->device;
    const size_t batch = sh->end - sh->begin;
    size_t sizes[L] = {60, 60, 60, 60};
    uint64_t q[L];
    WCHECK(fhe_generate_moduli(sizes, L, N, q));
    fhe_params *par = NULL;
    WCHECK(fhe_params_create(d, N, L, q, sh->t, &par));
    const fhe_ctx *ctx = NULL;
    WCHECK(fhe_params_ctx(par, 0, &ctx));
    void *stream = NULL;
    WCHECK(fhe_stream_create(d, &stream));
    const size_t poly = (size_t)L * N * sizeof(uint64_t), ct = 2 * poly;
    uint64_t *kraw = NULL, *c0 = NULL, *c1 = NULL, *lhs = NULL, *rhs = NULL, *out = NULL;
    WCHECK(fhe_buf_alloc(d, 2 * L * poly, (void **)&kraw));
    WCHECK(fhe_buf_alloc(d, L * poly, (void **)&c0));
    WCHECK(fhe_buf_alloc(d, L * poly, (void **)&c1));
    WCHECK(fhe_buf_alloc(d, (batch ? batch : 1) * ct, (void **)&lhs));
    WCHECK(fhe_buf_alloc(d, (batch ? batch : 1) * ct, (void **)&rhs));
    WCHECK(fhe_buf_alloc(d, (batch ? batch : 1) * ct, (void **)&out));
    /* the same synthetic relinearisation key on every device (read-only, replicated: SURVEY 8e) */
    WCHECK(fhe_synth_uniform_dev(ctx, sh->seed, 0, 8, 2 * L, kraw, 1, stream));
    for (size_t i = 0; i < L; i++) {
        WCHECK(fhe_buf_copy_async(c0 + i * L * N, kraw + (2 * i) * L * N, poly, stream));
        WCHECK(fhe_buf_copy_async(c1 + i * L * N, kraw + (2 * i + 1) * L * N, poly, stream));
    }
    fhe_ksk *rk = NULL;
    WCHECK(fhe_ksk_create_dev(ctx, ctx, L, c0, c1, 0, stream, &rk));
    fhe_mul *mul = NULL;
    WCHECK(fhe_mul_create_default(par, 0, rk, 0, &mul));
    /* this shard's block of the global synthetic stream: ciphertexts [begin, end) */
    WCHECK(fhe_synth_uniform_dev(ctx, sh->seed, sh->begin, 0, 2, lhs, batch, stream));
    WCHECK(fhe_synth_uniform_dev(ctx, sh->seed, sh->begin, 2, 2, rhs, batch, stream));
    WCHECK(fhe_bfv_mul_dev(mul, lhs, rhs, out, batch, stream));   /* (first call: scratch allocation, code load) */
    WCHECK(fhe_stream_sync(stream));
    const double t0 = now();
    WCHECK(fhe_bfv_mul_dev(mul, lhs, rhs, out, batch, stream));
    WCHECK(fhe_stream_sync(stream));
    sh->seconds = now() - t0;
    if (batch) WCHECK(fhe_buf_download(sh->host_out, out, batch * ct, stream));
    fhe_mul_destroy(mul);
    fhe_ksk_destroy(rk);
    WCHECK(fhe_buf_free(kraw));
    WCHECK(fhe_buf_free(c0));
    WCHECK(fhe_buf_free(c1));
    WCHECK(fhe_buf_free(lhs));
    WCHECK(fhe_buf_free(rhs));
    WCHECK(fhe_buf_free(out));
    WCHECK(fhe_stream_destroy(stream));
    fhe_params_destroy(par);
    sh->rc = 0;
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <seed> <plaintext> <total_batch> <devices (0 = all)> [<degree>]\n", argv[0]);
        return 2;
    }
    const uint64_t seed = strtoull(argv[1], NULL, 10), t = strtoull(argv[2], NULL, 10);
    const size_t total = (size_t)strtoull(argv[3], NULL, 10);
    int ndev = atoi(argv[4]);
    if (argc > 5) N = (size_t)strtoull(argv[5], NULL, 10);
    const int visible = fhe_device_count();
    if (visible < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    if (ndev <= 0 || ndev > visible) ndev = visible;
    const size_t ct_words = 2 * (size_t)L * N;
    shard_t *sh = (shard_t *)calloc((size_t)ndev + 1, sizeof *sh);
    pthread_t *th = (pthread_t *)calloc((size_t)ndev, sizeof *th);
    if (!sh || !th) return 2;
    /* the unsharded reference: the whole batch in one call on device 0 (shard ndev of the table) */
    shard_t *ref = &sh[ndev];
    ref->device = 0, ref->ndev = 1, ref->seed = seed, ref->t = t, ref->total = total, ref->begin = 0, ref->end = total;
    ref->host_out = (uint64_t *)malloc((total ? total : 1) * ct_words * sizeof(uint64_t));
    if (!ref->host_out) return 2;
    worker(ref);
    if (ref->rc) {
        fprintf(stderr, "reference run failed: %s\n", ref->err);
        return 2;
    }
    const size_t base = total / (size_t)ndev, extra = total % (size_t)ndev;
    for (int d = 0; d < ndev; d++) {
        const size_t r = (size_t)d;
        sh[d].device = d, sh[d].ndev = ndev, sh[d].seed = seed, sh[d].t = t, sh[d].total = total;
        sh[d].begin = r * base + (r < extra ? r : extra);
        sh[d].end = sh[d].begin + base + (r < extra ? 1 : 0);
        sh[d].host_out = (uint64_t *)malloc(((sh[d].end - sh[d].begin) ? (sh[d].end - sh[d].begin) : 1) * ct_words * sizeof(uint64_t));
        if (!sh[d].host_out) return 2;
    }
    const double t0 = now();
    for (int d = 0; d < ndev; d++)
        if (pthread_create(&th[d], NULL, worker, &sh[d])) return 2;
    for (int d = 0; d < ndev; d++) pthread_join(th[d], NULL);
    const double wall = now() - t0;
    int bad = 0;
    double slowest = 0;
    for (int d = 0; d < ndev; d++) {
        if (sh[d].rc) {
            fprintf(stderr, "device %d: %s\n", d, sh[d].err);
            bad++;
            continue;
        }
        const size_t nb = sh[d].end - sh[d].begin;
        const int same = !memcmp(sh[d].host_out, ref->host_out + sh[d].begin * ct_words, nb * ct_words * sizeof(uint64_t));
        printf("device %d: ciphertexts [%zu, %zu) multiply %.3f ms  %s\n", d, sh[d].begin, sh[d].end, sh[d].seconds * 1e3,
               same ? "identical to the unsharded result" : "MISMATCH");
        bad += !same;
        if (sh[d].seconds > slowest) slowest = sh[d].seconds;
        free(sh[d].host_out);
    }
    printf("devices %d of %d visible, total batch %zu, one-call reference %.3f ms, sharded multiply (slowest shard) %.3f ms = %.1f ops/s, "
           "wall incl. per-device setup %.1f ms, data-path collectives 0\n",
           ndev, visible, total, ref->seconds * 1e3, slowest * 1e3, slowest > 0 ? (double)total / slowest : 0.0, wall * 1e3);
    free(ref->host_out);
    free(sh);
    free(th);
    printf("%s\n", bad ? "FAILED" : "ALL OK");
    return bad ? 1 : 0;
}
