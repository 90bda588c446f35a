// hostmath.hpp -- setup-time host mathematics: Z_q helpers, prime generation, NTT tables,
// RNS context and RnsScaler constants.  Mirrors the reference's one-off precomputation
// (M/zq/mod.rs:83-98, M/zq/primes.rs, M/ntt/native.rs:35-73, M/rns/mod.rs:52-116,
// M/rns/scaler.rs:79-229, M/rq/context.rs:42-92).  Runs once per parameter set; everything
// produced here is uploaded to the GPU as read-only tables.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "bigint.hpp"

namespace fhe {

using u64 = uint64_t;
using u128 = unsigned __int128;

struct StatusError : std::runtime_error {
    int code;
    StatusError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline u64 mulmod(u64 a, u64 b, u64 p) { return (u64)((u128)a * b % p); }
inline u64 powmod(u64 a, u64 e, u64 p) {
    u64 r = 1 % p;
    a %= p;
    while (e) {
        if (e & 1) r = mulmod(r, a, p);
        a = mulmod(a, a, p);
        e >>= 1;
    }
    return r;
}

// Exact primality on u64 (fhe-util/src/lib.rs:16-18 uses a deterministic BPSW; any exact
// test yields the same prime lists): Miller-Rabin with the first 12 prime bases.
inline bool is_prime_u64(u64 n) {
    if (n < 2) return false;
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (u64 q : small)
        if (n % q == 0) return n == q;
    u64 d = n - 1;
    int s = 0;
    while ((d & 1) == 0) {
        d >>= 1;
        s++;
    }
    for (u64 a : small) {
        u64 x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool composite = true;
        for (int i = 1; i < s; i++) {
            x = mulmod(x, x, n);
            if (x == n - 1) {
                composite = false;
                break;
            }
        }
        if (composite) return false;
    }
    return true;
}

// M/zq/primes.rs:10-24 (NFLlib equation 1), evaluated with BigUint like the reference.
inline bool supports_opt(u64 p) {
    if (p == 0) return false;
    unsigned lz = (unsigned)__builtin_clzll(p);
    if (lz < 1) return false;
    BigUint middle = BigUint::pow2(3 * lz);
    BigUint left = (middle + BigUint(1)) << 64;
    middle = middle * (BigUint::pow2(lz) + BigUint(1));
    middle = middle * BigUint(p);
    return left < middle;
}

// M/zq/primes.rs:30-59.  Returns 0 for None.
inline u64 generate_prime(size_t num_bits, u64 modulo, u64 upper_bound) {
    if (num_bits < 10 || num_bits > 62 || modulo == 0) return 0;
    if (((u64)1 << num_bits) < upper_bound) return 0;  // debug_assert in the reference
    const unsigned lz = (unsigned)(64 - num_bits);
    auto leading = [](u64 x) -> unsigned { return x ? (unsigned)__builtin_clzll(x) : 64u; };
    u64 t = upper_bound - 1;
    while (t % modulo != 1 && leading(t) == lz) t -= 1;
    while (leading(t) == lz && !is_prime_u64(t) && t >= modulo) t -= modulo;
    if (leading(t) == lz && is_prime_u64(t)) return t;
    return 0;
}

// F/bfv/parameters.rs:391-431.
inline std::vector<u64> generate_moduli(const std::vector<size_t> &sizes, size_t degree) {
    std::vector<u64> moduli;
    for (size_t size : sizes) {
        if (size > 62 || size < 10) throw StatusError(-3, "InvalidModulusSize");
        u64 upper = (u64)1 << size;
        for (;;) {
            u64 prime = generate_prime(size, 2 * (u64)degree, upper);
            if (!prime) throw StatusError(-16, "NotEnoughPrimes");
            if (std::find(moduli.begin(), moduli.end(), prime) == moduli.end()) {
                moduli.push_back(prime);
                break;
            }
            upper = prime;
        }
    }
    return moduli;
}

// F/bfv/parameters.rs:660-676 == F/bfv/ops/mul.rs:109-125: 62-bit primes descending from
// 2^62 that are not in `existing`.
inline std::vector<u64> extended_basis_primes(size_t degree, const std::vector<u64> &existing, size_t count) {
    std::vector<u64> ext;
    u64 upper = (u64)1 << 62;
    while (ext.size() != count) {
        upper = generate_prime(62, 2 * (u64)degree, upper);
        if (!upper) throw StatusError(-16, "NotEnoughPrimes");
        if (std::find(ext.begin(), ext.end(), upper) == ext.end() &&
            std::find(existing.begin(), existing.end(), upper) == existing.end())
            ext.push_back(upper);
    }
    return ext;
}

inline u64 shoup(u64 a, u64 p) { return (u64)(((u128)a << 64) / p); }  // M/zq/mod.rs:195-199

inline u64 bitrev(u64 x, unsigned logn) {
    u64 r = 0;
    for (unsigned i = 0; i < logn; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

// Per-modulus constants used by the device code (see zq_dev.hpp).
struct ModConsts {
    u64 p;
    u64 p2;          // 2p
    u64 mu;          // floor(2^(2k) / p) << (63 - k), k = bit length of p  (single-word Barrett)
    u64 brt_hi;      // floor(2^128 / p) high/low words (M/zq/mod.rs:87-91) for full u128 reduction
    u64 brt_lo;
    uint32_t k;      // bit length of p
    uint32_t pad;
    u64 np;          // 2^64 - p and 2^64 - 2p: the device adds these instead of subtracting p / 2p
    u64 np2;         // (gfx950 has a one-instruction 64-bit add but no one-instruction 64-bit subtract)
};

inline ModConsts make_mod_consts(u64 p) {
    if (p < 2 || (p >> 62) != 0) throw StatusError(-3, "InvalidModulus(" + std::to_string(p) + ")");
    ModConsts m{};
    m.p = p;
    m.p2 = 2 * p;
    m.np = 0 - p;
    m.np2 = 0 - 2 * p;
    m.k = 64 - (uint32_t)__builtin_clzll(p);
    BigUint mu = BigUint::pow2(2 * m.k) / BigUint(p);  // < 2^(k+1)
    m.mu = (mu << (63 - m.k)).to_u64();
    BigUint brt = BigUint::pow2(128) / BigUint(p);
    m.brt_lo = brt.limb(0);
    m.brt_hi = brt.limb(1);
    return m;
}

// One modulus' NTT tables (M/ntt/native.rs:16-26).
struct NttTables {
    std::vector<u64> omegas, omegas_shoup, zetas_inv, zetas_inv_shoup;
    u64 size_inv = 0, size_inv_shoup = 0;
    u64 psi = 0;
};

inline bool supports_ntt(u64 p, size_t n) { return p % (2 * (u64)n) == 1 && is_prime_u64(p); }

// Engine's deterministic primitive 2N-th root when the host supplies no tables
// (the reference draws it from ChaCha8Rng::seed_from_u64(0), M/ntt/native.rs:320-336 --
// the single unpinned point; a Rust host passes its own tables instead).
inline u64 default_primitive_root(size_t n, u64 p) {
    u64 lambda = (p - 1) / (2 * (u64)n);
    for (u64 g = 2;; g++) {
        u64 root = powmod(g, lambda, p);
        if (powmod(root, 2 * (u64)n, p) == 1 && powmod(root, (u64)n, p) != 1) return root;
    }
}

inline NttTables make_ntt_tables(u64 p, size_t n, u64 psi = 0) {
    if (!supports_ntt(p, n))
        throw StatusError(-5, "NttOperatorUnavailable(modulus " + std::to_string(p) + ", degree " + std::to_string(n) + ")");
    NttTables t;
    unsigned logn = 0;
    while (((size_t)1 << logn) < n) logn++;
    t.psi = psi ? psi : default_primitive_root(n, p);
    u64 psi_inv = powmod(t.psi, p - 2, p);
    t.size_inv = powmod((u64)n % p, p - 2, p);
    t.size_inv_shoup = shoup(t.size_inv, p);
    std::vector<u64> powers(n), powers_inv(n);
    powers[0] = 1;
    powers_inv[0] = psi_inv;
    for (size_t i = 1; i < n; i++) {
        powers[i] = mulmod(powers[i - 1], t.psi, p);
        powers_inv[i] = mulmod(powers_inv[i - 1], psi_inv, p);
    }
    t.omegas.resize(n);
    t.zetas_inv.resize(n);
    t.omegas_shoup.resize(n);
    t.zetas_inv_shoup.resize(n);
    for (size_t i = 0; i < n; i++) {
        size_t j = (size_t)bitrev(i, logn);
        t.omegas[i] = powers[j];
        t.zetas_inv[i] = powers_inv[j];
        t.omegas_shoup[i] = shoup(t.omegas[i], p);
        t.zetas_inv_shoup[i] = shoup(t.zetas_inv[i], p);
    }
    return t;
}

inline u64 gcd_u64(u64 a, u64 b) {
    while (b) {
        u64 t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// modular inverse of a mod m for coprime a, m (m need not be prime: the reference's RNS tests
// use moduli 4 and 15).  Extended Euclid on signed 128-bit.
inline u64 inv_mod_general(u64 a, u64 m) {
    __int128 t = 0, newt = 1;
    __int128 r = m, newr = a % m;
    while (newr != 0) {
        __int128 q = r / newr;
        __int128 tmp = t - q * newt;
        t = newt;
        newt = tmp;
        tmp = r - q * newr;
        r = newr;
        newr = tmp;
    }
    if (r != 1) throw StatusError(-15, "NonCoprimeModuli");
    if (t < 0) t += m;
    return (u64)t;
}

// M/rns/mod.rs:23-116.
struct RnsContext {
    std::vector<u64> moduli;
    BigUint product;
    std::vector<BigUint> q_star, garner;
    std::vector<u64> q_tilde;

    explicit RnsContext(const std::vector<u64> &m) : moduli(m) {
        if (m.empty()) throw StatusError(-14, "EmptyModuli");
        for (size_t i = 0; i < m.size(); i++)
            for (size_t j = 0; j < m.size(); j++)
                if (i != j && gcd_u64(m[i], m[j]) != 1) throw StatusError(-15, "NonCoprimeModuli");
        product = BigUint(1);
        for (u64 q : m) {
            if (q < 2 || (q >> 62) != 0) throw StatusError(-3, "InvalidModulus");
            product = product * BigUint(q);
        }
        for (u64 q : m) {
            BigUint qs = product / BigUint(q);
            u64 qt = inv_mod_general(qs.mod_u64(q), q);
            q_star.push_back(qs);
            q_tilde.push_back(qt);
            garner.push_back(qs * BigUint(qt));  // not reduced mod product (mod.rs:97)
        }
    }
};

// M/rns/scaler.rs:52-72.
struct ScalerConstants {
    size_t nfrom = 0, nto = 0;
    bool is_one = false;
    std::vector<u64> gamma, gamma_shoup;          // [to]
    std::vector<u64> omega, omega_shoup;          // [to][from]
    u64 theta_gamma_lo = 0, theta_gamma_hi = 0;
    bool theta_gamma_sign = false;
    std::vector<u64> theta_omega_lo, theta_omega_hi;
    std::vector<uint8_t> theta_omega_sign;        // [from]
    std::vector<u64> theta_garner_lo, theta_garner_hi;
    size_t theta_garner_shift = 0;
};

// M/rns/scaler.rs:183-229.
inline void extract_projection_and_theta(const RnsContext &ctx, const BigUint &input, const BigUint &num,
                                         const BigUint &den, bool round_up, std::vector<u64> &projected,
                                         u64 &theta_lo, u64 &theta_hi, bool &theta_sign) {
    BigUint ni = num * input;
    BigUint gamma, theta;
    BigUint::divmod(ni + (den >> 1), den, gamma, theta);
    projected.clear();
    for (u64 q : ctx.moduli) projected.push_back(gamma.mod_u64(q));
    theta = ni % den;
    theta_sign = false;
    if (den > BigUint(1)) {
        BigUint half = den >> 1;
        if (den.is_odd()) {
            if (theta > half) {
                theta_sign = true;
                theta = den - theta;
            }
        } else if (theta >= half) {
            theta_sign = true;
            theta = den - theta;
        }
    }
    bool ceil = round_up ? !theta_sign : theta_sign;
    BigUint scaled = theta << 127;
    if (ceil) scaled = scaled + den - BigUint(1);
    scaled = scaled / den;
    theta_lo = scaled.limb(0);
    theta_hi = scaled.limb(1);
}

// M/rns/scaler.rs:79-175.
inline ScalerConstants make_scaler_constants(const RnsContext &from, const RnsContext &to, const BigUint &num,
                                             const BigUint &den) {
    if (den.is_zero()) throw StatusError(-1, "ScalingFactor: zero denominator");
    ScalerConstants c;
    c.nfrom = from.moduli.size();
    c.nto = to.moduli.size();
    c.is_one = (num == den);
    extract_projection_and_theta(to, from.product, num, den, false, c.gamma, c.theta_gamma_lo, c.theta_gamma_hi,
                                 c.theta_gamma_sign);
    c.gamma_shoup.resize(c.nto);
    for (size_t j = 0; j < c.nto; j++) c.gamma_shoup[j] = shoup(c.gamma[j], to.moduli[j]);
    c.omega.assign(c.nto * c.nfrom, 0);
    c.omega_shoup.assign(c.nto * c.nfrom, 0);
    c.theta_omega_lo.resize(c.nfrom);
    c.theta_omega_hi.resize(c.nfrom);
    c.theta_omega_sign.resize(c.nfrom);
    for (size_t i = 0; i < c.nfrom; i++) {
        std::vector<u64> proj;
        bool sign;
        extract_projection_and_theta(to, from.garner[i], num, den, true, proj, c.theta_omega_lo[i],
                                     c.theta_omega_hi[i], sign);
        c.theta_omega_sign[i] = sign ? 1 : 0;
        for (size_t j = 0; j < c.nto; j++) {
            c.omega[j * c.nfrom + i] = proj[j] % to.moduli[j];
            c.omega_shoup[j * c.nfrom + i] = shoup(c.omega[j * c.nfrom + i], to.moduli[j]);
        }
    }
    // (shift + 1) + log(q * n) <= 192  (scaler.rs:130-142)
    size_t shift = 127;
    for (u64 q : from.moduli) {
        u128 v = (u128)q * (u128)c.nfrom;
        unsigned lg = 0;  // next_power_of_two().ilog2()
        while (((u128)1 << lg) < v) lg++;
        size_t cand = 192 - 1 - lg;
        if (cand < shift) shift = cand;
    }
    c.theta_garner_shift = shift;
    c.theta_garner_lo.resize(c.nfrom);
    c.theta_garner_hi.resize(c.nfrom);
    for (size_t i = 0; i < c.nfrom; i++) {
        BigUint theta = ((from.garner[i] << shift) + (from.product >> 1)) / from.product;
        c.theta_garner_lo[i] = theta.limb(0);
        c.theta_garner_hi[i] = theta.limb(1);
    }
    return c;
}

}  // namespace fhe
