"""The wave-local LDS exchanges of the radix passes (fhe.rs_amd/csrc/kernels.hpp: wave_local_exchange, fwd_pass,
inv_pass, wave_sync) rest on an index invariant: when the predicate holds for two consecutive passes, every
wavefront reads in the second pass exactly the elements it wrote in the first.  This test restates the group ->
element maps of both pass kinds in Python and checks the invariant by enumeration for every tile size, thread
count and pass plan the kernels instantiate -- so a future change of the plans or of the maps cannot silently
turn a wave-local exchange into a race (the GPU soak would only catch that sporadically)."""
import itertools

GMAX, KS_GMAX = 4, 3


def ntt_threads(logm):
    return min(max((1 << logm) // 16, 64), 1024)


def ks_threads(logn):
    return min(max((1 << logn) // 8, 64), 1024)


def plan(logm, gmax, late):
    np_ = (logm + gmax - 1) // gmax
    base, rem = logm // np_, logm % np_
    return [base + (1 if ((p >= np_ - rem) if late else (p < rem)) else 0) for p in range(np_)]


def wave_local_exchange(skip_a, g_a, skip_b, g_b):
    return g_a == g_b and skip_a <= 6 and skip_b <= 6


def fwd_elements(logm, s0, g, t, tid):
    """fwd_pass: groups {g0 + tid}; lo_bits = logm - s0 - g; base = (grp >> lo_bits) << (logm - s0) | lo."""
    lo_bits, ngroups = logm - s0 - g, 1 << (logm - g)
    out = set()
    for g0 in range(0, ngroups, t):
        grp = g0 + tid
        if grp >= ngroups:
            break
        base = ((grp >> lo_bits) << (logm - s0)) + (grp & ((1 << lo_bits) - 1))
        out.update(base + (e << lo_bits) for e in range(1 << g))
    return out


def inv_elements(logm, v0, g, t, tid):
    """inv_pass: lo = grp & (2^v0 - 1); base = (grp >> v0) << (v0 + g) | lo; elements base + (e << v0)."""
    ngroups = 1 << (logm - g)
    out = set()
    for g0 in range(0, ngroups, t):
        grp = g0 + tid
        if grp >= ngroups:
            break
        base = ((grp >> v0) << (v0 + g)) + (grp & ((1 << v0) - 1))
        out.update(base + (e << v0) for e in range(1 << g))
    return out


def wave_sets(fn, logm, skip_or_v0, g, t):
    waves = {}
    for tid in range(t):
        waves.setdefault(tid >> 6, set()).update(fn(logm, skip_or_v0, g, t, tid))
    return waves


def check_forward(logm, t, gmax, late):
    gs, s0, nlocal = plan(logm, gmax, late), 0, 0
    for p in range(len(gs) - 1):
        g, gn = gs[p], gs[p + 1]
        if wave_local_exchange(logm - s0 - g, g, logm - s0 - g - gn, gn):
            a = wave_sets(fwd_elements, logm, s0, g, t)
            b = wave_sets(fwd_elements, logm, s0 + g, gn, t)
            assert a == b, (logm, t, gs, p)
            nlocal += 1
        s0 += g
    return nlocal


def check_inverse(logm, t):
    np_ = (logm + GMAX - 1) // GMAX
    base, rem = logm // np_, logm % np_
    gs = [base + (1 if p >= np_ - rem else 0) for p in range(np_)]   # inv_plan_g
    v0, nlocal = 0, 0
    for p in range(len(gs) - 1):
        g, gn = gs[p], gs[p + 1]
        if wave_local_exchange(v0, g, v0 + g, gn):
            assert wave_sets(inv_elements, logm, v0, g, t) == wave_sets(inv_elements, logm, v0 + g, gn, t), (logm, t, gs, p)
            nlocal += 1
        v0 += g
    return nlocal


def test_forward_ntt_kernels():
    counts = {logm: check_forward(logm, ntt_threads(logm), GMAX, False) for logm in range(3, 15)}
    assert counts[13] == 2            # 4+3+3+3: the exchanges after passes 1 and 2 are wave-local


def test_key_switch_transforms():
    counts = {logn: check_forward(logn, ks_threads(logn), KS_GMAX, True) for logn in range(3, 15)}
    assert counts[13] == 2            # 2+2+3+3+3
    # the split kernel runs 8192-point sub-blocks with the same 1024 threads
    assert check_forward(13, 1024, KS_GMAX, True) == 2


def test_inverse_ntt_kernels():
    counts = {logm: check_inverse(logm, ntt_threads(logm)) for logm in range(3, 15)}
    assert counts[13] == 2            # 3+3+3+4: passes 0 -> 1 -> 2


def test_predicate_is_not_vacuous():
    """A pair of passes the predicate rejects really does cross wavefronts (first exchange at N = 8192)."""
    a = wave_sets(fwd_elements, 13, 0, 4, 512)
    b = wave_sets(fwd_elements, 13, 4, 3, 512)
    assert not wave_local_exchange(13 - 0 - 4, 4, 13 - 4 - 3, 3) and a != b
