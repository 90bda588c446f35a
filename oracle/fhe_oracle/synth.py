"""TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.

Counter-based synthetic inputs shared by the oracle, the C restatement and the
HIP engine (SURVEY.md §8d / BASELINE.md §2): uniform residues
    x = splitmix64(seed ^ (ct<<40) ^ (part<<36) ^ (row<<28) ^ coeff) mod q_row
What fresh/evaluated BFV ciphertexts look like in Ntt form is i.i.d. uniform
per row, so this is a faithful stand-in for throughput and parity runs.
"""

M64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    z = (x + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def synth_value(seed, ct, part, row, coeff, q):
    return splitmix64(seed ^ (ct << 40) ^ (part << 36) ^ (row << 28) ^ coeff) % q


def synth_rows(seed, ct, part, moduli, degree):
    """[len(moduli)][degree] residues of one polynomial."""
    return [[synth_value(seed, ct, part, r, c, q) for c in range(degree)]
            for r, q in enumerate(moduli)]


def seed_for_config(cfg: int) -> int:
    return 0xF4E50000 + cfg
