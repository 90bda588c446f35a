// bigint.hpp -- minimal unsigned big integer for SETUP-TIME precomputation on the host.
//
// The reference does all of this with num-bigint's BigUint (M/rns/mod.rs:52-116,
// M/rns/scaler.rs:79-229, F/bfv/parameters.rs:560-738).  It runs once per parameter set,
// so clarity beats speed: schoolbook multiply, shift-subtract division.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace fhe {

class BigUint {
public:
    std::vector<uint64_t> w;  // little endian, no trailing zero limbs

    BigUint() = default;
    BigUint(uint64_t v) {  // NOLINT(google-explicit-constructor)
        if (v) w.push_back(v);
    }
    static BigUint from_limbs(const uint64_t *limbs, size_t n) {
        BigUint r;
        r.w.assign(limbs, limbs + n);
        r.trim();
        return r;
    }
    static BigUint from_u128(unsigned __int128 v) {
        BigUint r;
        r.w = {(uint64_t)v, (uint64_t)(v >> 64)};
        r.trim();
        return r;
    }
    static BigUint pow2(size_t k) {
        BigUint r;
        r.w.assign(k / 64 + 1, 0);
        r.w[k / 64] = 1ull << (k % 64);
        return r;
    }
    void trim() {
        while (!w.empty() && w.back() == 0) w.pop_back();
    }
    bool is_zero() const { return w.empty(); }
    bool is_one() const { return w.size() == 1 && w[0] == 1; }
    bool is_odd() const { return !w.empty() && (w[0] & 1); }
    size_t bits() const {
        if (w.empty()) return 0;
        return 64 * (w.size() - 1) + (64 - (size_t)__builtin_clzll(w.back()));
    }
    uint64_t limb(size_t i) const { return i < w.size() ? w[i] : 0; }
    uint64_t to_u64() const { return limb(0); }
    bool fits_u64() const { return w.size() <= 1; }

    static int cmp(const BigUint &a, const BigUint &b) {
        if (a.w.size() != b.w.size()) return a.w.size() < b.w.size() ? -1 : 1;
        for (size_t i = a.w.size(); i-- > 0;)
            if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
        return 0;
    }
    friend bool operator==(const BigUint &a, const BigUint &b) { return cmp(a, b) == 0; }
    friend bool operator!=(const BigUint &a, const BigUint &b) { return cmp(a, b) != 0; }
    friend bool operator<(const BigUint &a, const BigUint &b) { return cmp(a, b) < 0; }
    friend bool operator<=(const BigUint &a, const BigUint &b) { return cmp(a, b) <= 0; }
    friend bool operator>(const BigUint &a, const BigUint &b) { return cmp(a, b) > 0; }
    friend bool operator>=(const BigUint &a, const BigUint &b) { return cmp(a, b) >= 0; }

    friend BigUint operator+(const BigUint &a, const BigUint &b) {
        BigUint r;
        size_t n = std::max(a.w.size(), b.w.size());
        r.w.resize(n + 1);
        unsigned __int128 c = 0;
        for (size_t i = 0; i < n; i++) {
            c += (unsigned __int128)a.limb(i) + b.limb(i);
            r.w[i] = (uint64_t)c;
            c >>= 64;
        }
        r.w[n] = (uint64_t)c;
        r.trim();
        return r;
    }
    // a - b, requires a >= b
    friend BigUint operator-(const BigUint &a, const BigUint &b) {
        BigUint r;
        r.w.resize(a.w.size());
        uint64_t borrow = 0;
        for (size_t i = 0; i < a.w.size(); i++) {
            unsigned __int128 d = (unsigned __int128)a.w[i] - b.limb(i) - borrow;
            r.w[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        r.trim();
        return r;
    }
    friend BigUint operator*(const BigUint &a, const BigUint &b) {
        BigUint r;
        if (a.is_zero() || b.is_zero()) return r;
        r.w.assign(a.w.size() + b.w.size(), 0);
        for (size_t i = 0; i < a.w.size(); i++) {
            unsigned __int128 c = 0;
            for (size_t j = 0; j < b.w.size(); j++) {
                c += (unsigned __int128)a.w[i] * b.w[j] + r.w[i + j];
                r.w[i + j] = (uint64_t)c;
                c >>= 64;
            }
            r.w[i + b.w.size()] = (uint64_t)c;
        }
        r.trim();
        return r;
    }
    friend BigUint operator<<(const BigUint &a, size_t s) {
        if (a.is_zero()) return a;
        BigUint r;
        size_t ws = s / 64, bs = s % 64;
        r.w.assign(a.w.size() + ws + 1, 0);
        for (size_t i = 0; i < a.w.size(); i++) {
            r.w[i + ws] |= a.w[i] << bs;
            if (bs) r.w[i + ws + 1] |= a.w[i] >> (64 - bs);
        }
        r.trim();
        return r;
    }
    friend BigUint operator>>(const BigUint &a, size_t s) {
        BigUint r;
        size_t ws = s / 64, bs = s % 64;
        if (ws >= a.w.size()) return r;
        r.w.assign(a.w.size() - ws, 0);
        for (size_t i = 0; i < r.w.size(); i++) {
            r.w[i] = a.w[i + ws] >> bs;
            if (bs && i + ws + 1 < a.w.size()) r.w[i] |= a.w[i + ws + 1] << (64 - bs);
        }
        r.trim();
        return r;
    }
    bool bit(size_t i) const { return (limb(i / 64) >> (i % 64)) & 1; }

    // quotient and remainder by shift-subtract (setup-time only).
    static void divmod(const BigUint &a, const BigUint &b, BigUint &q, BigUint &r) {
        q = BigUint();
        r = BigUint();
        if (cmp(a, b) < 0) {
            r = a;
            return;
        }
        if (b.fits_u64()) {
            uint64_t d = b.to_u64();
            q.w.assign(a.w.size(), 0);
            unsigned __int128 rem = 0;
            for (size_t i = a.w.size(); i-- > 0;) {
                rem = (rem << 64) | a.w[i];
                q.w[i] = (uint64_t)(rem / d);
                rem %= d;
            }
            q.trim();
            r = BigUint((uint64_t)rem);
            return;
        }
        size_t nb = a.bits();
        q.w.assign(a.w.size(), 0);
        for (size_t i = nb; i-- > 0;) {
            r = r << 1;
            if (a.bit(i)) {
                if (r.w.empty()) r.w.push_back(1);
                else r.w[0] |= 1;
            }
            if (cmp(r, b) >= 0) {
                r = r - b;
                q.w[i / 64] |= 1ull << (i % 64);
            }
        }
        q.trim();
    }
    friend BigUint operator/(const BigUint &a, const BigUint &b) {
        BigUint q, r;
        divmod(a, b, q, r);
        return q;
    }
    friend BigUint operator%(const BigUint &a, const BigUint &b) {
        BigUint q, r;
        divmod(a, b, q, r);
        return r;
    }
    uint64_t mod_u64(uint64_t d) const {
        unsigned __int128 rem = 0;
        for (size_t i = w.size(); i-- > 0;) rem = ((rem << 64) | w[i]) % d;
        return (uint64_t)rem;
    }
};

}  // namespace fhe
