"""TEST INFRASTRUCTURE (oracle): seeded polynomial expansion, the `c1` of a fresh secret-key ciphertext on the wire.

Restates crates/fhe-math/src/rq/mod.rs:276-292 (`Poly::random_from_seed`: SHA-256 of the 32-byte seed becomes the
key of a ChaCha8 generator, every residue row draws `degree` values from `Uniform::new(0, q_i)`) and
crates/fhe/src/bfv/ciphertext.rs:287-302 (the received ciphertext appends that polynomial, taken as Ntt form).

What is public and pinned by known-answer tests here: SHA-256 (hashlib) and the ChaCha block function
(RFC 7539 section 2.3.2 for 20 rounds; the same code with 8 rounds).  What comes from third-party crates that are
NOT in /root/reference (rand 0.10.2, rand_chacha 0.10.0, rand_core) and is therefore PARITY-UNPINNED, exactly like
the NTT's psi: (1) the generator's word layout -- 64-bit block counter in state words 12-13 starting at 0, 64-bit
stream id 0 in words 14-15, output words consumed in order, `next_u64` = two consecutive little-endian words, low
word first; (2) `Uniform<u64>` sampling -- Lemire's widening-multiply rejection: draw x, (hi, lo) = x * range, accept
hi when lo >= thresh with thresh = (2^64 - range) mod range.  Both are restated from the crates' published
algorithms; no golden vector of the reference fixes them.
"""
import hashlib
import struct

MASK32 = 0xFFFFFFFF
MASK64 = (1 << 64) - 1
SIGMA = (0x61707865, 0x3320646E, 0x79622D32, 0x6B206574)   # "expand 32-byte k"


def _rotl(v, c):
    return ((v << c) & MASK32) | (v >> (32 - c))


def _quarter(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & MASK32; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & MASK32; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & MASK32; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & MASK32; s[b] = _rotl(s[b] ^ s[c], 7)


def chacha_block(key_words, tail_words, rounds):
    """One 64-byte block: state = sigma | key[8] | tail[4] (tail = words 12..15), `rounds` rounds, feed-forward."""
    init = list(SIGMA) + list(key_words) + list(tail_words)
    s = list(init)
    for _ in range(rounds // 2):
        _quarter(s, 0, 4, 8, 12); _quarter(s, 1, 5, 9, 13); _quarter(s, 2, 6, 10, 14); _quarter(s, 3, 7, 11, 15)
        _quarter(s, 0, 5, 10, 15); _quarter(s, 1, 6, 11, 12); _quarter(s, 2, 7, 8, 13); _quarter(s, 3, 4, 9, 14)
    return [(a + b) & MASK32 for a, b in zip(s, init)]


class ChaCha8Rng:
    """rand_chacha::ChaCha8Rng::from_seed(seed): see the module docstring for what is restated."""

    def __init__(self, seed32: bytes, rounds=8):
        assert len(seed32) == 32
        self.key = struct.unpack("<8I", seed32)
        self.rounds = rounds
        self.pos = 0          # index of the next u64 of the stream
        self._blk, self._words = None, None

    def next_u64(self):
        blk, k = self.pos >> 3, self.pos & 7
        if blk != self._blk:
            self._blk = blk
            self._words = chacha_block(self.key, (blk & MASK32, blk >> 32, 0, 0), self.rounds)
        self.pos += 1
        return self._words[2 * k] | (self._words[2 * k + 1] << 32)


def uniform_below(rng, p):
    """rand::distr::Uniform::new(0, p).sample(rng) for u64 (Lemire's method with a precomputed threshold)."""
    thresh = ((1 << 64) - p) % p
    while True:
        m = rng.next_u64() * p
        if (m & MASK64) >= thresh:
            return m >> 64


def random_from_seed(moduli, degree, seed32: bytes):
    """rq/mod.rs:276-292: rows [L][degree] of uniform residues, one shared stream, row after row."""
    rng = ChaCha8Rng(hashlib.sha256(seed32).digest())
    return [[uniform_below(rng, q) for _ in range(degree)] for q in moduli]
