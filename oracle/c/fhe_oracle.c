/* TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.
 *
 * Plain-C CPU restatement of the fhe.rs BFV hot path, same algorithms and the
 * same pass structure as the reference's single-threaded Rust:
 *   zq    : crates/fhe-math/src/zq/mod.rs        (Barrett / NFLlib-opt / Shoup)
 *   ntt   : crates/fhe-math/src/ntt/native.rs    (Harvey lazy CT / GS butterflies)
 *   rns   : crates/fhe-math/src/rns/scaler.rs    (RnsScaler::scale, U256 fixed point)
 *   rq    : crates/fhe-math/src/rq/{mod,ops,scaler}.rs
 *   bfv   : crates/fhe/src/bfv/ops/mul.rs, keys/key_switching_key.rs,
 *           keys/galois_key.rs, ciphertext.rs
 * Constants (NTT tables, scaler thetas, keys) are supplied by the Python oracle
 * (oracle/fhe_oracle), which is pinned by the reference's KATs / closed forms;
 * tests/test_oracle_c.py checks this file bit-for-bit against the Python one.
 *
 * Uses: (1) parity checker for the HIP engine at full sizes, (2) the timed
 * `cpu_baseline` of bench.py (kind "port": the Rust reference cannot be built
 * in this image).  PARITY UNPINNED only w.r.t. the choice of psi (see ntt.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <malloc.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* Scratch memory: a per-thread bump arena when one is installed (the timed loops install
 * one so that the CPU baseline does not pay malloc/page-fault costs per call), else malloc. */
static __thread char *g_arena = 0;
static __thread size_t g_arena_off = 0, g_arena_cap = 0;
static void *scratch_alloc(size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    if (g_arena && g_arena_off + bytes <= g_arena_cap) {
        void *p = g_arena + g_arena_off;
        g_arena_off += bytes;
        return p;
    }
    return malloc(bytes);
}
static void scratch_free(void *p) {
    if (g_arena && (char *)p >= g_arena && (char *)p < g_arena + g_arena_cap) return; /* released by mark */
    free(p);
}
static size_t scratch_mark(void) { return g_arena_off; }
static void scratch_release(size_t mark) { if (g_arena) g_arena_off = mark; }

/* ---------------------------------------------------------------- zq ---- */
typedef struct {
    u64 p, barrett_hi, barrett_lo;
    unsigned leading_zeros;
    int supports_opt;
} orc_mod;

static int supports_opt_u64(u64 p) {
    /* primes.rs:10-24: (2^(3s)+1)*2^64 < 2^(3s)*(2^s+1)*p, s = leading zeros.
     * Evaluated exactly with a small fixed-width big integer (<= 320 bits). */
    unsigned s = (unsigned)__builtin_clzll(p);
    if (s < 1) return 0;
    /* left  = (2^(3s)+1) << 64 ; right = (2^(3s) * (2^s+1)) * p
     * compare right - left > 0  <=>  2^(3s) * ((2^s+1)*p - 2^64) > 2^64 */
    u128 t = (u128)(((u128)1 << s) + 1) * p; /* < 2^(s+1) * 2^(64-s) = 2^65 */
    u128 two64 = (u128)1 << 64;
    if (t <= two64) return 0;
    u128 diff = t - two64; /* >= 1 */
    /* need diff * 2^(3s) > 2^64 */
    if (3 * s >= 65) return 1;
    if (diff >> (64 - 3 * s) > 1) return 1;
    if (diff >> (64 - 3 * s) == 1) return (diff & ((((u128)1) << (64 - 3 * s)) - 1)) != 0;
    return 0;
}

static void mod_init(orc_mod *m, u64 p) {
    /* mod.rs:83-98: barrett = floor(2^128 / p) */
    u128 all = ~(u128)0;
    u128 q = all / p, r = all % p;
    if (r == (u128)(p - 1)) q += 1;
    m->p = p;
    m->barrett_hi = (u64)(q >> 64);
    m->barrett_lo = (u64)q;
    m->leading_zeros = (unsigned)__builtin_clzll(p);
    m->supports_opt = supports_opt_u64(p);
}

static inline u64 reduce1(u64 x, u64 p) { return x >= p ? x - p : x; } /* mod.rs:659 */

static inline u64 lazy_reduce_u128(const orc_mod *m, u128 a) { /* mod.rs:693-707 */
    u64 a_lo = (u64)a, a_hi = (u64)(a >> 64);
    u128 p_lo_lo = ((u128)a_lo * m->barrett_lo) >> 64;
    u128 p_hi_lo = (u128)a_hi * m->barrett_lo;
    u128 p_lo_hi = (u128)a_lo * m->barrett_hi;
    u128 q = ((p_lo_hi + p_hi_lo + p_lo_lo) >> 64) + (u128)a_hi * m->barrett_hi;
    return (u64)(a - q * (u128)m->p);
}
static inline u64 reduce_u128(const orc_mod *m, u128 a) { return reduce1(lazy_reduce_u128(m, a), m->p); }

static inline u64 lazy_reduce(const orc_mod *m, u64 a) { /* mod.rs:712-723 */
    u128 p_lo_lo = ((u128)a * m->barrett_lo) >> 64;
    u128 p_lo_hi = (u128)a * m->barrett_hi;
    u128 q = (p_lo_hi + p_lo_lo) >> 64;
    return (u64)((u128)a - q * (u128)m->p);
}
static inline u64 reduce_u64(const orc_mod *m, u64 a) { return reduce1(lazy_reduce(m, a), m->p); }

static inline u64 lazy_reduce_opt_u128(const orc_mod *m, u128 a) { /* mod.rs:730-740 */
    u128 q = (((u128)m->barrett_lo * (a >> 64)) + (a << m->leading_zeros)) >> 64;
    return (u64)(a - q * (u128)m->p);
}
static inline u64 lazy_reduce_opt(const orc_mod *m, u64 a) { /* mod.rs:744-752 */
    u64 q = a >> (64 - m->leading_zeros);
    return (u64)((u128)a - (u128)q * m->p);
}
static inline u64 mod_mul(const orc_mod *m, u64 a, u64 b) { /* mul_vec: mod.rs:332-344 */
    return m->supports_opt ? reduce1(lazy_reduce_opt_u128(m, (u128)a * b), m->p)
                           : reduce_u128(m, (u128)a * b);
}
static inline u64 lazy_mul_shoup(u64 p, u64 a, u64 b, u64 b_shoup) { /* mod.rs:224-234 */
    u64 q = (u64)(((u128)a * b_shoup) >> 64);
    return (u64)((u128)a * b - (u128)q * p);
}
static inline u64 mul_shoup(u64 p, u64 a, u64 b, u64 bs) { return reduce1(lazy_mul_shoup(p, a, b, bs), p); }
static inline u64 mod_add(u64 p, u64 a, u64 b) { return reduce1(a + b, p); }
static inline u64 mod_sub(u64 p, u64 a, u64 b) { return reduce1(a + p - b, p); }

/* --------------------------------------------------------------- ctx ---- */
typedef struct {
    u64 n, nmod;
    const u64 *moduli;                                              /* [nmod]      */
    const u64 *omegas, *omegas_shoup, *zetas_inv, *zetas_inv_shoup; /* [nmod][n]   */
    const u64 *size_inv, *size_inv_shoup;                           /* [nmod]      */
    const u64 *inv_last, *inv_last_shoup;                           /* [nmod-1]    */
} orc_ctx;

/* --------------------------------------------------------------- ntt ---- */
static void ntt_forward_lazy(const orc_ctx *c, u64 mi, u64 *a) { /* native.rs:142-175 */
    const u64 n = c->n, p = c->moduli[mi], p2 = 2 * p;
    const u64 *om = c->omegas + mi * n, *oms = c->omegas_shoup + mi * n;
    u64 l = n >> 1, m = 1, k = 1;
    while (l > 0) {
        for (u64 i = 0; i < m; i++) {
            u64 w = om[k], ws = oms[k];
            k++;
            u64 s = 2 * i * l;
            for (u64 j = s; j < s + l; j++) { /* butterfly, native.rs:256-269 */
                u64 x = reduce1(a[j], p2);
                u64 t = lazy_mul_shoup(p, a[j + l], w, ws);
                a[j + l] = x + p2 - t;
                a[j] = x + t;
            }
        }
        l >>= 1;
        m <<= 1;
    }
}
void orc_ntt_forward(const orc_ctx *c, u64 mi, u64 *a, int lazy) {
    ntt_forward_lazy(c, mi, a);
    if (!lazy) { /* reduce3, native.rs:238-246 */
        const u64 p = c->moduli[mi];
        for (u64 j = 0; j < c->n; j++) a[j] = reduce1(reduce1(a[j], 2 * p), p);
    }
}
void orc_ntt_backward(const orc_ctx *c, u64 mi, u64 *a) { /* native.rs:197-233 */
    const u64 n = c->n, p = c->moduli[mi], p2 = 2 * p;
    const u64 *zt = c->zetas_inv + mi * n, *zts = c->zetas_inv_shoup + mi * n;
    u64 k = 0, m = n >> 1, l = 1;
    while (m > 0) {
        for (u64 i = 0; i < m; i++) {
            u64 s = 2 * i * l, z = zt[k], zs = zts[k];
            k++;
            for (u64 j = s; j < s + l; j++) { /* inv_butterfly, native.rs:288-300 */
                u64 t = a[j], y = a[j + l];
                a[j] = reduce1(y + t, p2);
                a[j + l] = lazy_mul_shoup(p, p2 + t - y, z, zs);
            }
        }
        l <<= 1;
        m >>= 1;
    }
    const u64 si = c->size_inv[mi], sis = c->size_inv_shoup[mi];
    for (u64 j = 0; j < n; j++) a[j] = mul_shoup(p, a[j], si, sis);
}
void orc_poly_ntt_forward(const orc_ctx *c, u64 *poly) {
    for (u64 r = 0; r < c->nmod; r++) orc_ntt_forward(c, r, poly + r * c->n, 0);
}
void orc_poly_ntt_backward(const orc_ctx *c, u64 *poly) {
    for (u64 r = 0; r < c->nmod; r++) orc_ntt_backward(c, r, poly + r * c->n);
}

/* --------------------------------------------------------- poly ops ---- */
void orc_poly_add(const orc_ctx *c, u64 *a, const u64 *b) { /* ops.rs:10-118 */
    for (u64 r = 0; r < c->nmod; r++)
        for (u64 j = 0; j < c->n; j++) a[r * c->n + j] = mod_add(c->moduli[r], a[r * c->n + j], b[r * c->n + j]);
}
void orc_poly_sub(const orc_ctx *c, u64 *a, const u64 *b) {
    for (u64 r = 0; r < c->nmod; r++)
        for (u64 j = 0; j < c->n; j++) a[r * c->n + j] = mod_sub(c->moduli[r], a[r * c->n + j], b[r * c->n + j]);
}
void orc_poly_neg(const orc_ctx *c, u64 *a) {
    for (u64 r = 0; r < c->nmod; r++)
        for (u64 j = 0; j < c->n; j++) a[r * c->n + j] = reduce1(c->moduli[r] - a[r * c->n + j], c->moduli[r]);
}
void orc_poly_mul(const orc_ctx *c, u64 *a, const u64 *b) { /* ops.rs:174-206 */
    for (u64 r = 0; r < c->nmod; r++) {
        orc_mod m;
        mod_init(&m, c->moduli[r]);
        for (u64 j = 0; j < c->n; j++) a[r * c->n + j] = mod_mul(&m, a[r * c->n + j], b[r * c->n + j]);
    }
}
void orc_poly_mul_shoup(const orc_ctx *c, u64 *a, const u64 *b, const u64 *bs) { /* ops.rs:208-245 */
    for (u64 r = 0; r < c->nmod; r++)
        for (u64 j = 0; j < c->n; j++)
            a[r * c->n + j] = mul_shoup(c->moduli[r], a[r * c->n + j], b[r * c->n + j], bs[r * c->n + j]);
}
void orc_shoup_vec(u64 p, const u64 *a, u64 *out, u64 n) { /* mod.rs:195-199 */
    for (u64 j = 0; j < n; j++) out[j] = (u64)((((u128)a[j]) << 64) / p);
}

/* ------------------------------------------------------------ scaler ---- */
typedef struct { u64 w[4]; } u256;
static inline void u256_add_mul_64x128(u256 *acc, u64 r, u64 lo, u64 hi, int negate) {
    /* acc +/-= r * (lo | hi<<64)  (mod 2^256) -- ethnum U256 wrapping ops */
    u128 p0 = (u128)r * lo, p1 = (u128)r * hi;
    u64 t0 = (u64)p0;
    u128 mid = (p0 >> 64) + (u64)p1;
    u64 t1 = (u64)mid;
    u64 t2 = (u64)((mid >> 64) + (p1 >> 64));
    u64 term[4] = {t0, t1, t2, 0};
    if (!negate) {
        u128 cy = 0;
        for (int i = 0; i < 4; i++) { cy += (u128)acc->w[i] + term[i]; acc->w[i] = (u64)cy; cy >>= 64; }
    } else {
        u64 borrow = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)acc->w[i] - term[i] - borrow;
            acc->w[i] = (u64)d;
            borrow = (u64)(d >> 64) & 1;
        }
    }
}
static inline void u256_shr(u256 *a, unsigned s) {
    unsigned ws = s / 64, bs = s % 64;
    u64 r[4] = {0, 0, 0, 0};
    for (unsigned i = 0; i + ws < 4; i++) {
        r[i] = a->w[i + ws] >> bs;
        if (bs && i + ws + 1 < 4) r[i] |= a->w[i + ws + 1] << (64 - bs);
    }
    memcpy(a->w, r, sizeof r);
}
static inline u128 u256_as_u128(const u256 *a) { return ((u128)a->w[1] << 64) | a->w[0]; }

typedef struct {
    u64 nfrom, nto, ncommon, is_one, shift;
    const u64 *gamma, *gamma_shoup;                                /* [nto]        */
    const u64 *omega, *omega_shoup;                                /* [nto][nfrom] */
    u64 theta_gamma_lo, theta_gamma_hi, theta_gamma_sign;
    const u64 *theta_omega_lo, *theta_omega_hi, *theta_omega_sign; /* [nfrom]      */
    const u64 *theta_garner_lo, *theta_garner_hi;                  /* [nfrom]      */
} orc_scaler;

/* scaler.rs:249-352, one coefficient column; `rests` strided by `rstride`,
 * outputs strided by `ostride` (the reference walks ndarray columns too). */
static void rns_scale(const orc_scaler *s, const orc_mod *to_mods, const u64 *rests, u64 rstride,
                      u64 *out, u64 ostride, u64 size, u64 starting_index) {
    u256 sum = {{0, 0, 0, 0}};
    for (u64 i = 0; i < s->nfrom; i++)
        u256_add_mul_64x128(&sum, rests[i * rstride], s->theta_garner_lo[i], s->theta_garner_hi[i], 0);
    u256_shr(&sum, (unsigned)s->shift - 1);
    u128 v = u256_as_u128(&sum);
    v = (v >> 1) + (v & 1); /* div_ceil(2) */

    int w_sign = 0;
    u128 w = 0;
    if (!s->is_one) {
        u256 t = {{0, 0, 0, 0}};
        for (u64 i = 0; i < s->nfrom; i++)
            u256_add_mul_64x128(&t, rests[i * rstride], s->theta_omega_lo[i], s->theta_omega_hi[i],
                                (int)s->theta_omega_sign[i]);
        /* v * theta_gamma (128 x 128 -> 256, wrapping) */
        u256 vt = {{0, 0, 0, 0}};
        u256_add_mul_64x128(&vt, (u64)v, s->theta_gamma_lo, s->theta_gamma_hi, 0);
        {
            u256 hi = {{0, 0, 0, 0}};
            u256_add_mul_64x128(&hi, (u64)(v >> 64), s->theta_gamma_lo, s->theta_gamma_hi, 0);
            /* vt += hi << 64 */
            u128 cy = 0;
            for (int i = 1; i < 4; i++) { cy += (u128)vt.w[i] + hi.w[i - 1]; vt.w[i] = (u64)cy; cy >>= 64; }
        }
        if (s->theta_gamma_sign) {
            u128 cy = 0;
            for (int i = 0; i < 4; i++) { cy += (u128)t.w[i] + vt.w[i]; t.w[i] = (u64)cy; cy >>= 64; }
        } else {
            u64 borrow = 0;
            for (int i = 0; i < 4; i++) {
                u128 d = (u128)t.w[i] - vt.w[i] - borrow;
                t.w[i] = (u64)d;
                borrow = (u64)(d >> 64) & 1;
            }
        }
        u256 sg = t;
        u256_shr(&sg, 191);
        w_sign = (sg.w[0] | sg.w[1] | sg.w[2] | sg.w[3]) != 0;
        if (w_sign) {
            u256 nt = {{~t.w[0], ~t.w[1], ~t.w[2], ~t.w[3]}};
            u256_shr(&nt, 126);
            w = u256_as_u128(&nt) + 1;
            w /= 2;
        } else {
            u256_shr(&t, 126);
            w = u256_as_u128(&t);
            w = (w >> 1) + (w & 1);
        }
    }
    for (u64 i = 0; i < size; i++) {
        const orc_mod *qi = &to_mods[starting_index + i];
        const u64 *om = s->omega + (starting_index + i) * s->nfrom;
        const u64 *oms = s->omega_shoup + (starting_index + i) * s->nfrom;
        u128 yi = (u128)(qi->p * 2 - lazy_mul_shoup(qi->p, reduce_u128(qi, v), s->gamma[starting_index + i],
                                                    s->gamma_shoup[starting_index + i]));
        if (!s->is_one) {
            u64 wi = lazy_reduce_u128(qi, w);
            yi += w_sign ? (qi->p * 2 - wi) : wi;
        }
        for (u64 j = 0; j < s->nfrom; j++) yi += lazy_mul_shoup(qi->p, rests[j * rstride], om[j], oms[j]);
        out[i * ostride] = reduce_u128(qi, yi);
    }
}

/* rq/scaler.rs:55-127.  in: [nfrom][n], out: [nto][n]. */
void orc_poly_scale(const orc_scaler *s, const orc_ctx *from, const orc_ctx *to, const u64 *in, u64 *out,
                    int repr_is_ntt) {
    const u64 n = from->n;
    orc_mod *to_mods = (orc_mod *)scratch_alloc(sizeof(orc_mod) * to->nmod);
    for (u64 i = 0; i < to->nmod; i++) mod_init(&to_mods[i], to->moduli[i]);
    memset(out, 0, sizeof(u64) * to->nmod * n);
    if (s->ncommon > 0) memcpy(out, in, sizeof(u64) * s->ncommon * n);
    if (s->ncommon < to->nmod) {
        u64 *pb = (u64 *)in;
        if (repr_is_ntt) {
            pb = (u64 *)scratch_alloc(sizeof(u64) * from->nmod * n);
            memcpy(pb, in, sizeof(u64) * from->nmod * n);
            orc_poly_ntt_backward(from, pb);
        }
        for (u64 col = 0; col < n; col++)
            rns_scale(s, to_mods, pb + col, n, out + s->ncommon * n + col, n, to->nmod - s->ncommon, s->ncommon);
        if (repr_is_ntt) {
            for (u64 r = s->ncommon; r < to->nmod; r++) orc_ntt_forward(to, r, out + r * n, 0);
            scratch_free(pb);
        }
    }
    scratch_free(to_mods);
}

/* One column through RnsScaler::scale (for fine-grained tests). */
void orc_rns_scale(const orc_scaler *s, const u64 *to_moduli, const u64 *rests, u64 *out, u64 size,
                   u64 starting_index) {
    orc_mod *to_mods = (orc_mod *)scratch_alloc(sizeof(orc_mod) * s->nto);
    for (u64 i = 0; i < s->nto; i++) mod_init(&to_mods[i], to_moduli[i]);
    rns_scale(s, to_mods, rests, 1, out, 1, size, starting_index);
    scratch_free(to_mods);
}

/* ------------------------------------------------------ switch_down ---- */
/* rq/mod.rs:433-492: poly [L][n] PowerBasis -> [L-1][n] in place (first L-1 rows). */
void orc_poly_switch_down(const orc_ctx *c, u64 *poly) {
    const u64 n = c->n, L = c->nmod;
    const u64 q_last = c->moduli[L - 1], q_last_div_2 = q_last / 2;
    u64 *last = poly + (L - 1) * n;
    for (u64 j = 0; j < n; j++) last[j] = mod_add(q_last, last[j], q_last_div_2);
    for (u64 r = 0; r + 1 < L; r++) {
        orc_mod qi;
        mod_init(&qi, c->moduli[r]);
        const u64 inv = c->inv_last[r], invs = c->inv_last_shoup[r];
        const u64 q_last_div_2_mod_qi = qi.p - reduce_u64(&qi, q_last_div_2);
        u64 *row = poly + r * n;
        for (u64 j = 0; j < n; j++) {
            u64 tmp = lazy_reduce(&qi, last[j]) + q_last_div_2_mod_qi;
            u64 cf = row[j] + 3 * qi.p - tmp;
            row[j] = mul_shoup(qi.p, cf, inv, invs);
        }
    }
}

/* -------------------------------------------------------- substitute ---- */
static inline u64 bitrev_u64(u64 x, unsigned logn) {
    u64 r = 0;
    for (unsigned i = 0; i < logn; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
/* rq/mod.rs:360-412.  repr_is_ntt: NTT-domain gather; else PowerBasis signed scatter. */
void orc_poly_substitute(const orc_ctx *c, u64 exponent, const u64 *in, u64 *out, int repr_is_ntt) {
    const u64 n = c->n, L = c->nmod;
    unsigned logn = 0;
    while (((u64)1 << logn) < n) logn++;
    exponent %= 2 * n;
    if (repr_is_ntt) {
        u64 power = (exponent - 1) / 2, mask = n - 1;
        for (u64 j = 0; j < n; j++) {
            u64 dst = bitrev_u64(j, logn), src = bitrev_u64(power & mask, logn);
            for (u64 r = 0; r < L; r++) out[r * n + dst] = in[r * n + src];
            power += exponent;
        }
    } else {
        memset(out, 0, sizeof(u64) * L * n);
        u64 power = 0, mask = n - 1;
        for (u64 j = 0; j < n; j++) {
            for (u64 r = 0; r < L; r++) {
                u64 p = c->moduli[r], *q = &out[r * n + (power & mask)];
                *q = (power & n) ? mod_sub(p, *q, in[r * n + j]) : mod_add(p, *q, in[r * n + j]);
            }
            power += exponent;
        }
    }
}

/* -------------------------------------------------------- key switch ---- */
typedef struct {
    u64 ndigits;                                  /* = L of the ciphertext ctx       */
    const u64 *c0, *c0_shoup, *c1, *c1_shoup;     /* [ndigits][Lk][n] NttShoup polys  */
} orc_ksk;

/* key_switching_key.rs:241-270 + rq/mod.rs:563-586 (lazy lift + forward_vt_lazy).
 * p: [L][n] PowerBasis over ct_ctx.  out0/out1: [Lk][n] Ntt over ksk_ctx. */
void orc_key_switch(const orc_ctx *ct_ctx, const orc_ctx *ksk_ctx, const orc_ksk *k, const u64 *p, u64 *out0,
                    u64 *out1) {
    const u64 n = ksk_ctx->n, Lk = ksk_ctx->nmod;
    memset(out0, 0, sizeof(u64) * Lk * n);
    memset(out1, 0, sizeof(u64) * Lk * n);
    u64 *c2 = (u64 *)scratch_alloc(sizeof(u64) * Lk * n);
    orc_mod *mods = (orc_mod *)scratch_alloc(sizeof(orc_mod) * Lk);
    for (u64 j = 0; j < Lk; j++) mod_init(&mods[j], ksk_ctx->moduli[j]);
    for (u64 i = 0; i < k->ndigits; i++) {
        const u64 *row = p + i * n;
        for (u64 j = 0; j < Lk; j++) {
            u64 *d = c2 + j * n;
            if (mods[j].supports_opt)
                for (u64 x = 0; x < n; x++) d[x] = lazy_reduce_opt(&mods[j], row[x]);
            else
                for (u64 x = 0; x < n; x++) d[x] = lazy_reduce(&mods[j], row[x]);
            ntt_forward_lazy(ksk_ctx, j, d);
        }
        const u64 *k0 = k->c0 + i * Lk * n, *k0s = k->c0_shoup + i * Lk * n;
        const u64 *k1 = k->c1 + i * Lk * n, *k1s = k->c1_shoup + i * Lk * n;
        for (u64 j = 0; j < Lk; j++) {
            const u64 pj = ksk_ctx->moduli[j];
            for (u64 x = 0; x < n; x++) {
                u64 idx = j * n + x, cv = c2[idx];
                out0[idx] = mod_add(pj, out0[idx], mul_shoup(pj, cv, k0[idx], k0s[idx]));
                out1[idx] = mod_add(pj, out1[idx], mul_shoup(pj, cv, k1[idx], k1s[idx]));
            }
        }
    }
    scratch_free(mods);
    scratch_free(c2);
    (void)ct_ctx;
}

/* ----------------------------------------------- Multiplicator::multiply ---- */
typedef struct {
    const orc_ctx *base_ctx, *mul_ctx;
    const orc_scaler *extender_lhs, *extender_rhs, *down_scaler;
    const orc_ksk *rk;        /* NULL: no relinearisation; key at the SAME level as base_ctx */
    u64 mod_switch;
} orc_mul;

/* ops/mul.rs:165-243.  lhs, rhs: [2][L][n] Ntt.  out: [2 or 3][L or L-1][n] Ntt. */
void orc_bfv_multiply(const orc_mul *m, const u64 *lhs, const u64 *rhs, u64 *out) {
    const size_t mark = scratch_mark();
    const orc_ctx *b = m->base_ctx, *e = m->mul_ctx;
    const u64 n = b->n, L = b->nmod, K = e->nmod, PL = L * n, PK = K * n;
    u64 *ext = (u64 *)scratch_alloc(sizeof(u64) * 4 * PK);
    orc_poly_scale(m->extender_lhs, b, e, lhs, ext, 1);
    orc_poly_scale(m->extender_lhs, b, e, lhs + PL, ext + PK, 1);
    orc_poly_scale(m->extender_rhs, b, e, rhs, ext + 2 * PK, 1);
    orc_poly_scale(m->extender_rhs, b, e, rhs + PL, ext + 3 * PK, 1);
    u64 *c00 = ext, *c01 = ext + PK, *c10 = ext + 2 * PK, *c11 = ext + 3 * PK;
    u64 *t = (u64 *)scratch_alloc(sizeof(u64) * 4 * PK);
    u64 *c0 = t, *c1 = t + PK, *c2 = t + 2 * PK, *tmp = t + 3 * PK;
    memcpy(c0, c00, sizeof(u64) * PK); orc_poly_mul(e, c0, c10);
    memcpy(c1, c00, sizeof(u64) * PK); orc_poly_mul(e, c1, c11);
    memcpy(tmp, c01, sizeof(u64) * PK); orc_poly_mul(e, tmp, c10);
    orc_poly_add(e, c1, tmp);
    memcpy(c2, c01, sizeof(u64) * PK); orc_poly_mul(e, c2, c11);
    u64 *d = (u64 *)scratch_alloc(sizeof(u64) * 3 * PL);
    orc_poly_scale(m->down_scaler, e, b, c0, d, 1);
    orc_poly_scale(m->down_scaler, e, b, c1, d + PL, 1);
    orc_poly_scale(m->down_scaler, e, b, c2, d + 2 * PL, 1);
    u64 nparts = 3;
    if (m->rk) {
        u64 *c2pb = (u64 *)scratch_alloc(sizeof(u64) * PL);
        memcpy(c2pb, d + 2 * PL, sizeof(u64) * PL);
        orc_poly_ntt_backward(b, c2pb);
        u64 *r = (u64 *)scratch_alloc(sizeof(u64) * 2 * PL);
        orc_key_switch(b, b, m->rk, c2pb, r, r + PL);
        orc_poly_add(b, d, r);
        orc_poly_add(b, d + PL, r + PL);
        scratch_free(r);
        scratch_free(c2pb);
        nparts = 2;
    }
    if (m->mod_switch) { /* ciphertext.rs:148-161 */
        for (u64 q = 0; q < nparts; q++) {
            u64 *poly = d + q * PL;
            orc_poly_ntt_backward(b, poly);
            orc_poly_switch_down(b, poly);
            for (u64 r = 0; r + 1 < L; r++) orc_ntt_forward(b, r, poly + r * n, 0);
            memcpy(out + q * (L - 1) * n, poly, sizeof(u64) * (L - 1) * n);
        }
    } else {
        memcpy(out, d, sizeof(u64) * nparts * PL);
    }
    scratch_free(d);
    scratch_free(t);
    scratch_free(ext);
    scratch_release(mark);
}

/* galois_key.rs:63-86 with the key at the ciphertext level.  ct: [2][L][n] Ntt. */
void orc_galois_relinearize(const orc_ctx *c, const orc_ksk *k, u64 exponent, const u64 *ct, u64 *out) {
    const u64 PL = c->nmod * c->n;
    u64 *c2 = (u64 *)scratch_alloc(sizeof(u64) * PL), *s0 = (u64 *)scratch_alloc(sizeof(u64) * PL);
    orc_poly_substitute(c, exponent, ct + PL, c2, 1);
    orc_poly_ntt_backward(c, c2);
    orc_key_switch(c, c, k, c2, out, out + PL);
    orc_poly_substitute(c, exponent, ct, s0, 1);
    orc_poly_add(c, out, s0);
    scratch_free(s0);
    scratch_free(c2);
}

/* -------------------------------------------------------------- timing ---- */
/* Keep the multi-MiB scratch buffers of orc_bfv_multiply on the heap instead of
 * fresh mmap()s per call: otherwise page faults (and the kernel mm lock, when
 * threaded) dominate and the CPU baseline would be unfairly slow. */
__attribute__((constructor)) static void orc_init_malloc(void) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_ARENA_MAX, 64);
}
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
/* Times `count` multiplies of the given (lhs, rhs) pairs spread over `threads`
 * OpenMP threads (batch-parallel, one ciphertext pair per task; the reference
 * itself is single-threaded).  lhs/rhs: [npairs][2][L][n]; pairs are cycled.
 * Returns elapsed seconds. */
double orc_time_multiply(const orc_mul *m, const u64 *lhs, const u64 *rhs, u64 npairs, u64 count, int threads,
                         u64 *out_last) {
    const u64 L = m->base_ctx->nmod, n = m->base_ctx->n, CT = 2 * L * n;
    double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        const size_t cap = (size_t)40 * m->mul_ctx->nmod * n * sizeof(u64) + (1 << 20);
        char *arena = (char *)malloc(cap);
        memset(arena, 0, cap);
        g_arena = arena;
        g_arena_off = 0;
        g_arena_cap = cap;
        u64 *out = (u64 *)scratch_alloc(sizeof(u64) * 3 * L * n);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (long long it = 0; it < (long long)count; it++) {
            u64 pi = (u64)it % npairs;
            orc_bfv_multiply(m, lhs + pi * CT, rhs + pi * CT, out);
            if ((u64)it == count - 1 && out_last) memcpy(out_last, out, sizeof(u64) * 2 * L * n);
        }
        g_arena = 0;
        g_arena_cap = g_arena_off = 0;
        free(arena);
    }
    (void)threads;
    return now_s() - t0;
}
double orc_time_ntt_forward(const orc_ctx *c, u64 *poly, u64 count) {
    double t0 = now_s();
    for (u64 it = 0; it < count; it++) orc_poly_ntt_forward(c, poly);
    return now_s() - t0;
}
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* synthetic inputs: splitmix64 counter generator */
static inline u64 splitmix64(u64 x) {
    u64 z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void orc_synth_poly(u64 seed, u64 ct, u64 part, const u64 *moduli, u64 nmod, u64 n, u64 *out) {
    for (u64 r = 0; r < nmod; r++)
        for (u64 c = 0; c < n; c++)
            out[r * n + c] = splitmix64(seed ^ (ct << 40) ^ (part << 36) ^ (r << 28) ^ c) % moduli[r];
}
int orc_supports_opt(u64 p) { return supports_opt_u64(p); }
u64 orc_mod_mul(u64 p, u64 a, u64 b) { orc_mod m; mod_init(&m, p); return mod_mul(&m, a, b); }
u64 orc_reduce_u128(u64 p, u64 lo, u64 hi) { orc_mod m; mod_init(&m, p); return reduce_u128(&m, ((u128)hi << 64) | lo); }
