"""TEST INFRASTRUCTURE -- CPU oracle for the fhe.rs BFV hot path.

This package restates the reference's algorithm (tlepoint/fhe.rs, crates
fhe-math 0.2.0 / fhe 0.2.0) with Python integers.  It is the CHECKER used by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only; nothing
in the shipped product path (fhe.rs_amd/) imports it.

Parity status: every primitive is pinned by the reference's own KATs and
closed-form test oracles (see tests/test_oracle_*.py, each citing the
reference test it restates).  PARITY UNPINNED at one point only: the choice of
the primitive root psi (third-party rand/rand_chacha draw, ntt/native.rs:325)
-- see ntt.py.  The reference is Rust and cannot be built in this image
(no cargo/rustc, ~100 un-vendored crates), so there is no oracle/_ref.
"""
from . import zq, ntt, rns, rq, bfv, synth  # noqa: F401
