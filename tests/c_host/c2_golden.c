/* c2_golden.c -- a C99 host with no Python, PyTorch or HIP headers between main() and the kernels.
 *
 * BASELINE config C2 (N = 8192, 4 x 60-bit moduli, ct x ct + relinearise) driven exclusively through
 * include/fhe_hip.h: device buffers and the stream come from the ABI's own fhe_buf_* / fhe_stream_* entry
 * points, inputs and the relinearisation key from the ABI's counter-based generator, the multiply runs on
 * `fhe_bfv_mul_dev`, and whole ciphertexts are downloaded and SHA-256'd against tests/golden/c2_digest.json
 * (digests passed on the command line by tests/test_c_host.py, which only compiles and launches this file).
 *
 *   c2_golden <seed> <plaintext_modulus> <batch> <key_sha256> [<ct>:<input_sha256>:<output_sha256> ...]
 *
 * Exit code 0 = every digest matched.  This is what a Rust / Go / C host of the library does (the Rust
 * shim's DeviceCiphertext is this program's buffer handling behind RAII, rust/fhe-math-hip/src/lib.rs).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fhe_hip.h"

/* ---- SHA-256 (FIPS 180-4), enough for hashing a few MiB ---- */
typedef struct {
    uint32_t h[8];
    uint64_t len;
    unsigned char buf[64];
    size_t fill;
} sha256_t;
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(sha256_t *s, const unsigned char *p) {
    uint32_t w[64], a, b, c, d, e, f, g, h;
    int i;
    for (i = 0; i < 16; i++)
        w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (i = 16; i < 64; i++) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    a = s->h[0], b = s->h[1], c = s->h[2], d = s->h[3], e = s->h[4], f = s->h[5], g = s->h[6], h = s->h[7];
    for (i = 0; i < 64; i++) {
        uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    s->h[0] += a, s->h[1] += b, s->h[2] += c, s->h[3] += d, s->h[4] += e, s->h[5] += f, s->h[6] += g, s->h[7] += h;
}
static void sha256_init(sha256_t *s) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(s->h, iv, sizeof iv);
    s->len = 0, s->fill = 0;
}
static void sha256_update(sha256_t *s, const void *data, size_t n) {
    const unsigned char *p = (const unsigned char *)data;
    s->len += n;
    while (n) {
        size_t take = 64 - s->fill < n ? 64 - s->fill : n;
        memcpy(s->buf + s->fill, p, take);
        s->fill += take, p += take, n -= take;
        if (s->fill == 64) sha256_block(s, s->buf), s->fill = 0;
    }
}
static void sha256_hex(sha256_t *s, char out[65]) {
    uint64_t bits = s->len * 8;
    unsigned char pad[72] = {0x80}, lenb[8];
    size_t padn = (s->fill < 56 ? 56 : 120) - s->fill;
    int i;
    for (i = 0; i < 8; i++) lenb[i] = (unsigned char)(bits >> (56 - 8 * i));
    sha256_update(s, pad, padn);
    sha256_update(s, lenb, 8);
    for (i = 0; i < 8; i++) sprintf(out + 8 * i, "%08x", (unsigned)s->h[i]);
}

#define CHECK(call)                                                                        \
    do {                                                                                   \
        fhe_status st_ = (call);                                                           \
        if (st_ != FHE_OK) {                                                               \
            fprintf(stderr, "%s -> %d (%s)\n", #call, (int)st_, fhe_last_error());         \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

enum { N = 8192, L = 4 };

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <seed> <plaintext> <batch> <key_sha256> [<ct>:<in_sha>:<out_sha> ...]\n", argv[0]);
        return 2;
    }
    const uint64_t seed = strtoull(argv[1], NULL, 10), t = strtoull(argv[2], NULL, 10);
    const size_t batch = (size_t)strtoull(argv[3], NULL, 10);
    const char *key_sha = argv[4];
    if (fhe_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    size_t sizes[L] = {60, 60, 60, 60};
    uint64_t q[L];
    CHECK(fhe_generate_moduli(sizes, L, N, q));
    fhe_params *par = NULL;
    CHECK(fhe_params_create(0, N, L, q, t, &par));
    const fhe_ctx *ctx = NULL;
    CHECK(fhe_params_ctx(par, 0, &ctx));

    void *stream = NULL;
    CHECK(fhe_stream_create(0, &stream));
    const size_t row = (size_t)N * sizeof(uint64_t), poly = L * row, ct = 2 * poly;
    uint64_t *kraw = NULL, *c0 = NULL, *c1 = NULL, *lhs = NULL, *rhs = NULL, *out = NULL;
    CHECK(fhe_buf_alloc(0, 2 * L * poly, (void **)&kraw));
    CHECK(fhe_buf_alloc(0, L * poly, (void **)&c0));
    CHECK(fhe_buf_alloc(0, L * poly, (void **)&c1));
    CHECK(fhe_buf_alloc(0, batch * ct, (void **)&lhs));
    CHECK(fhe_buf_alloc(0, batch * ct, (void **)&rhs));
    CHECK(fhe_buf_alloc_async(0, batch * ct, stream, (void **)&out));   /* the result: stream-ordered, as a chained host would */

    /* synthetic relinearisation key: digit i = generator parts 8 + 2i (c0) and 9 + 2i (c1) of ciphertext 0 */
    CHECK(fhe_synth_uniform_dev(ctx, seed, 0, 8, 2 * L, kraw, 1, stream));
    for (size_t i = 0; i < L; i++) {
        CHECK(fhe_buf_copy_async(c0 + i * L * N, kraw + (2 * i) * L * N, poly, stream));
        CHECK(fhe_buf_copy_async(c1 + i * L * N, kraw + (2 * i + 1) * L * N, poly, stream));
    }
    uint64_t *hk = (uint64_t *)malloc(2 * L * poly), *hbuf = (uint64_t *)malloc(2 * ct);
    if (!hk || !hbuf) return 2;
    CHECK(fhe_buf_download(hk, c0, L * poly, stream));
    CHECK(fhe_buf_download(hk + (size_t)L * L * N, c1, L * poly, stream));
    char hex[65];
    sha256_t s;
    int bad = 0;
    sha256_init(&s), sha256_update(&s, hk, 2 * L * poly), sha256_hex(&s, hex);
    printf("key %s %s\n", hex, strcmp(hex, key_sha) ? "MISMATCH" : "ok");
    bad += strcmp(hex, key_sha) != 0;

    fhe_ksk *rk = NULL;
    CHECK(fhe_ksk_create_dev(ctx, ctx, L, c0, c1, 0, stream, &rk));
    fhe_mul *mul = NULL;
    CHECK(fhe_mul_create_default(par, 0, rk, 0, &mul));
    size_t parts = 0, rows = 0;
    CHECK(fhe_mul_out_shape(mul, &parts, &rows));
    if (parts != 2 || rows != L) return 2;

    CHECK(fhe_synth_uniform_dev(ctx, seed, 0, 0, 2, lhs, batch, stream));
    CHECK(fhe_synth_uniform_dev(ctx, seed, 0, 2, 2, rhs, batch, stream));
    CHECK(fhe_bfv_mul_dev(mul, lhs, rhs, out, batch, stream));
    CHECK(fhe_stream_sync(stream));

    for (int a = 5; a < argc; a++) {
        char in_sha[65], out_sha[65];
        unsigned long idx = 0;
        if (sscanf(argv[a], "%lu:%64[0-9a-f]:%64[0-9a-f]", &idx, in_sha, out_sha) != 3 || idx >= batch) {
            fprintf(stderr, "bad digest argument %s\n", argv[a]);
            return 2;
        }
        CHECK(fhe_buf_download(hbuf, lhs + idx * 2 * L * N, ct, stream));
        CHECK(fhe_buf_download(hbuf + 2 * L * N, rhs + idx * 2 * L * N, ct, stream));
        sha256_init(&s), sha256_update(&s, hbuf, 2 * ct), sha256_hex(&s, hex);
        const int in_ok = !strcmp(hex, in_sha);
        CHECK(fhe_buf_download(hbuf, out + idx * 2 * L * N, ct, stream));
        sha256_init(&s), sha256_update(&s, hbuf, ct), sha256_hex(&s, hex);
        const int out_ok = !strcmp(hex, out_sha);
        printf("ct %lu input %s output %s %s\n", idx, in_ok ? "ok" : "MISMATCH", hex, out_ok ? "ok" : "MISMATCH");
        bad += !in_ok + !out_ok;
    }
    free(hk), free(hbuf);
    fhe_mul_destroy(mul);
    fhe_ksk_destroy(rk);
    CHECK(fhe_buf_free(kraw));
    CHECK(fhe_buf_free(c0));
    CHECK(fhe_buf_free(c1));
    CHECK(fhe_buf_free(lhs));
    CHECK(fhe_buf_free(rhs));
    CHECK(fhe_buf_free_async(out, stream));
    CHECK(fhe_stream_destroy(stream));
    fhe_params_destroy(par);
    printf("%s\n", bad ? "FAILED" : "ALL OK");
    return bad ? 1 : 0;
}
