"""Pins the public building blocks of the seeded-polynomial oracle (oracle/fhe_oracle/seeded.py): the ChaCha block
function against RFC 7539 (20 rounds) and the classic all-zero test vector of the 8-round variant, the stream
layout's internal consistency, and the sampler's range / rejection rule.  The generator layout and the sampling
rule themselves come from un-vendored crates: parity-unpinned, see the oracle's docstring."""
import hashlib
import struct

from fhe_oracle import seeded


def test_chacha20_block_rfc7539_section_2_3_2():
    key = struct.unpack("<8I", bytes(range(32)))
    tail = (1, 0x09000000, 0x4A000000, 0)          # counter 1, nonce 00 00 00 09 00 00 00 4a 00 00 00 00
    out = struct.pack("<16I", *seeded.chacha_block(key, tail, 20))
    assert out.hex() == ("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                         "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


def test_chacha8_zero_key_vector():
    """8-round keystream of the all-zero key / IV (block 0), the vector of the eSTREAM-era test suites."""
    out = struct.pack("<16I", *seeded.chacha_block((0,) * 8, (0, 0, 0, 0), 8))
    assert out.hex().startswith("3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e")


def test_stream_layout_and_u64_assembly():
    seed = bytes(range(32))
    rng = seeded.ChaCha8Rng(seed)
    words = []
    for blk in range(3):
        words += seeded.chacha_block(struct.unpack("<8I", seed), (blk, 0, 0, 0), 8)
    got = [rng.next_u64() for _ in range(24)]
    assert got == [words[2 * i] | (words[2 * i + 1] << 32) for i in range(24)]


def test_uniform_rejection_rule():
    class Fixed:
        def __init__(self, xs):
            self.xs = list(xs)

        def next_u64(self):
            return self.xs.pop(0)
    p = 0x1400000000000001 | 1          # about 1.25 * 2^60: 2^64 mod p is a sizeable fraction of p
    thresh = ((1 << 64) - p) % p
    lo_of = lambda x: (x * p) & ((1 << 64) - 1)
    rejected = next(x for x in range(1, 1 << 20) if lo_of(x) < thresh)
    accepted = next(x for x in range(1, 1 << 20) if lo_of(x) >= thresh)
    assert seeded.uniform_below(Fixed([rejected, accepted]), p) == (accepted * p) >> 64
    assert seeded.uniform_below(Fixed([accepted]), p) < p


def test_random_from_seed_shape_range_determinism():
    moduli, n = [1152921504606830593, 0x1400000000000001 | 1, 65537], 64
    a = seeded.random_from_seed(moduli, n, b"\x07" * 32)
    assert a == seeded.random_from_seed(moduli, n, b"\x07" * 32)
    assert a != seeded.random_from_seed(moduli, n, b"\x08" * 32)
    assert all(len(r) == n and all(0 <= v < q for v in r) for r, q in zip(a, moduli))
    # the key of the generator is the SHA-256 of the seed (rq/mod.rs:279-282)
    rng = seeded.ChaCha8Rng(hashlib.sha256(b"\x07" * 32).digest())
    assert a[0][0] == seeded.uniform_below(rng, moduli[0])
