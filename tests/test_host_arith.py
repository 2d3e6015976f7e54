"""The DEVICE arithmetic (poseidon252_amd/csrc/fr29.hpp, hades29.hpp, tables.hpp) compiled for the
host and checked limb-for-limb against the oracle.  Same source the kernels run; no GPU needed."""
import ctypes
import os
import random

import numpy as np
import pytest

import pymodel

P = pymodel.P
HERE = os.path.dirname(os.path.abspath(__file__))
u64p = ctypes.POINTER(ctypes.c_uint64)


def p(a):
    return a.ctypes.data_as(u64p)


def edge_scalars(oracle_mod):
    vals = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, (1 << 255) % P, (1 << 254), (1 << 29) - 1, 1 << 29,
            (1 << 232) - 1, 1 << 232, pow(2, -256, P), pow(2, 256, P), pow(3, 200, P)]
    mont = [oracle_mod.mont_from_int(v) for v in vals]
    # also raw limb patterns near the modulus (any value < p is a valid Montgomery residue)
    mont += [oracle_mod.int_to_limbs(P - 1), oracle_mod.int_to_limbs(P - (1 << 200)), oracle_mod.int_to_limbs((1 << 254) + 12345)]
    return np.stack(mont)


def test_roundtrip_and_canonical(oracle_mod, hosttest_lib):
    a = np.concatenate([edge_scalars(oracle_mod), oracle_mod.fill_random(11, 3000)])
    out = np.empty_like(a)
    hosttest_lib.ht_roundtrip29(p(a), p(out), a.shape[0])
    assert np.array_equal(a, out)


def test_mul_matches_oracle(oracle_mod, hosttest_lib):
    e = edge_scalars(oracle_mod)
    a = np.concatenate([np.repeat(e, len(e), axis=0), oracle_mod.fill_random(12, 4000)])
    b = np.concatenate([np.tile(e, (len(e), 1)), oracle_mod.fill_random(13, 4000)])
    exp = np.empty_like(a)
    for i in range(a.shape[0]):
        oracle_mod.lib().p252o_mul(p(a[i]), p(b[i]), p(exp[i]))
    out = np.empty_like(a)
    hosttest_lib.ht_mul29(p(a), p(b), p(out), a.shape[0])
    assert np.array_equal(exp, out)


def test_sbox_matches_bigint(oracle_mod, hosttest_lib):
    a = np.concatenate([edge_scalars(oracle_mod), oracle_mod.fill_random(14, 300)])
    exp = np.stack([oracle_mod.mont_from_int(pow(oracle_mod.int_from_mont(v), 5, P)) for v in a])
    out = np.empty_like(a)
    hosttest_lib.ht_sbox29(p(a), p(out), a.shape[0])
    assert np.array_equal(exp, out)


def test_permutation_matches_oracle(oracle_mod, hosttest_lib):
    e = edge_scalars(oracle_mod)
    rng = random.Random(3)
    special = np.stack([np.stack([e[rng.randrange(len(e))] for _ in range(5)]) for _ in range(64)])
    st = np.concatenate([special, oracle_mod.fill_random(15, 5 * 1500).reshape(1500, 5, 4)])
    out = np.empty_like(st)
    hosttest_lib.ht_permute29(p(st), p(out), st.shape[0])
    assert np.array_equal(out, oracle_mod.permute_batch(st))


def test_tables_match_independent_derivation(oracle_mod, hosttest_lib):
    """csrc/tables.hpp (C++) vs tests/pymodel.py (big ints): same sparse matrices and folded constants"""
    C, M = pymodel.load_constants()
    T = pymodel.derive_optimised(C, M)
    n = 5 + 40 + 25 + 60 * 10 + 4
    raw = np.empty((n, 4), dtype=np.uint64)
    hosttest_lib.ht_tables_raw.restype = ctypes.c_size_t
    assert hosttest_lib.ht_tables_raw(p(raw)) == n
    got = [oracle_mod.int_from_mont(v) for v in raw]
    exp = list(T["c_first"])
    full_add = {0: C[1], 1: C[2], 2: C[3], 3: T["pre_add"], 4: C[65], 5: C[66], 6: C[67], 7: [0] * 5}
    for f in range(8):
        exp += list(full_add[f])
    exp += [T["m_pre"][i][j] for i in range(5) for j in range(5)]
    for q in range(60):
        s = T["sparse"][q]
        exp += list(s["w"]) + [s["d"]] + list(s["b"]) + [s["add"][4]]
    exp += list(T["sparse"][59]["add"][:4])
    assert got == exp


def test_table_digit_bounds(hosttest_lib):
    """every encoded constant digit is balanced (|d| <= 2^28): the column-overflow argument in
    DESIGN.md (45 products < 2^57 each per column, plus < 2^61 from the reduction) relies on it"""
    hosttest_lib.ht_tables29_total.restype = ctypes.c_int
    n = hosttest_lib.ht_tables29_total()
    tab = np.empty(n, dtype=np.int32)
    hosttest_lib.ht_tables29(tab.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    assert n % 9 == 0 and np.abs(tab.astype(np.int64)).max() <= (1 << 28)
    digits = tab.reshape(-1, 9).astype(object)
    vals = [sum(int(d) << (29 * i) for i, d in enumerate(row)) for row in digits]
    plain_ints = {5, 6} | set(range(47, 50)) | set(range(177, 185))  # rows of one-digit integers (INT_N, AI_AB, AI_EX_N), not 9-digit residues
    assert all(abs(v) <= P // 2 + 1 for k, v in enumerate(vals) if k not in plain_ints)


def test_all_schedules_match_oracle(oracle_mod, hosttest_lib):
    """integer-ARMA (kernels), sparse and integer-MDS schedules are three independent derivations; all must equal the oracle"""
    st = oracle_mod.fill_random(21, 5 * 600).reshape(600, 5, 4)
    exp = oracle_mod.permute_batch(st)
    for sched in (0, 1, 2):
        out = np.empty_like(st)
        hosttest_lib.ht_permute29_sched(p(st), p(out), st.shape[0], sched)
        assert np.array_equal(out, exp), "schedule %d" % sched


def test_arma_bigint_model_equals_reference():
    C, M = pymodel.load_constants()
    AR = pymodel.derive_arma(C, M)
    rng = random.Random(9)
    for _ in range(3):
        x = [rng.randrange(P) for _ in range(5)]
        assert pymodel.perm_arma(x, C, M, AR) == pymodel.perm_reference(x, C, M)


def test_static_column_bound(hosttest_lib):
    """worst-case |column| of every lazy accumulation of the three schedules, for the actual constants,
    stays below 2^63: int64 columns cannot overflow for ANY input"""
    import math
    hosttest_lib.ht_max_column_bound29.restype = ctypes.c_double
    b = hosttest_lib.ht_max_column_bound29()
    assert 60 < math.log2(b) < 62.9, math.log2(b)


def _canon(raw):
    return [sum(int(v[i]) << (64 * i) for i in range(4)) for v in raw]


def test_integer_mds_tables_match_bigint_derivation(hosttest_lib):
    """csrc/tables.hpp (B) vs tests/pymodel.py::derive_int, and the big-int model of that schedule vs the reference"""
    C, M = pymodel.load_constants()
    T = pymodel.derive_int(C, M)
    n = 340 + 60 + 1
    raw = np.empty((n, 4), dtype=np.uint64)
    hosttest_lib.ht_tables_int_raw.restype = ctypes.c_size_t
    assert hosttest_lib.ht_tables_int_raw(p(raw)) == n
    exp = [T["kappa"][k][i] for k in range(68) for i in range(5)] + [T["G"][k] for k in range(4, 64)] + [T["F"]]
    assert _canon(raw) == exp
    rng = random.Random(5)
    x = [rng.randrange(P) for _ in range(5)]
    assert pymodel.perm_int([v * pymodel.RM % P for v in x], C, M, T) == [v * pymodel.RM % P for v in pymodel.perm_reference(x, C, M)]


def test_integer_arma_tables_match_bigint_derivation(hosttest_lib):
    """csrc/tables.hpp (C) — what the kernels run — vs tests/pymodel.py::derive_armaint (independent derivation:
    exact fractions for the integer coefficients, big ints for the residues), and the big-int model vs the reference"""
    C, M = pymodel.load_constants()
    T = pymodel.derive_armaint(C, M)
    n = 40 + 3 + 4 + 120 + 4 + 4 + 1
    raw = np.empty((n, 4), dtype=np.uint64)
    hosttest_lib.ht_tables_armaint_raw.restype = ctypes.c_size_t
    assert hosttest_lib.ht_tables_armaint_raw(p(raw)) == n
    fr = dict(T["fr_kappa"])
    exp = [fr[k][i] for k in (0, 1, 2, 3, 64, 65, 66, 67) for i in range(5)]
    exp += list(T["ent_fix"]) + list(T["ent_add"])
    for q in range(1, 61):
        exp += [T["K"][q + 1], T["G"][q]]
    exp += list(T["ex_fix"]) + list(T["ex_add"]) + [T["F"]]
    assert _canon(raw) == exp
    # the integer coefficients the device table carries are the ones exact rational arithmetic gives
    hosttest_lib.ht_tables29_total.restype = ctypes.c_int
    tab = np.empty(hosttest_lib.ht_tables29_total(), dtype=np.int32)
    hosttest_lib.ht_tables29(tab.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    assert list(tab[45:54]) == [pymodel.L_INT // (d + 5) for d in range(9)]
    assert list(tab[54:63]) == pymodel.A_INT + pymodel.B_INT
    assert max(abs(v) for v in pymodel.A_INT + pymodel.B_INT) < 1 << 25
    base = 63 + 8 * 45 + 3 * 9 + 3 * 9 + 4 * 9 + 60 * 18  # Tab29Layout::AI_EX_N
    for i, (den, ny, nv) in enumerate(pymodel.EXIT_INT):  # two-digit exit coefficients, balanced low digit
        row = [int(v) for v in tab[base + 18 * i: base + 18 * i + 16]]
        assert [row[2 * t] + (row[2 * t + 1] << 29) for t in range(8)] == ny + nv
        assert all(abs(row[2 * t]) <= 1 << 28 and abs(row[2 * t + 1]) < 1 << 18 for t in range(8))
    for i, nums in enumerate(pymodel.ENTRY_INT):  # entry coefficients (lane 4 does not enter)
        row = [int(v) for v in tab[63 + 8 * 45 + 9 * i: 63 + 8 * 45 + 9 * i + 8]]
        assert [row[2 * t] + (row[2 * t + 1] << 29) for t in range(4)] == nums[:4] and nums[4] == 0
    rng = random.Random(6)
    for _ in range(2):
        x = [rng.randrange(P) for _ in range(5)]
        assert pymodel.perm_armaint([v * pymodel.RM % P for v in x], C, M, T) == [v * pymodel.RM % P for v in pymodel.perm_reference(x, C, M)]


def test_integer_schedules_refuse_a_non_cauchy_mds(hosttest_lib):
    """the kernels' schedule exists only because mds.bin is R/(i+j+5); derive_tables re-checks that (and the hard-coded
    integer coefficients against the field values computed from the file) and the library refuses anything else"""
    assert hosttest_lib.ht_int_ok(0) == 1
    for entry in (1, 7, 13, 25):
        assert hosttest_lib.ht_int_ok(entry) == 0


def test_adversarial_noncanonical_limbs(oracle_mod, hosttest_lib):
    """inputs the reference would never produce (limbs >= p, all-ones, digit-saturating patterns): the device
    code treats the 256-bit pattern V as an integer, so the result must be the permutation of V * R^-1 mod p —
    exercised to probe the column-bound / lazy-range arguments with the largest possible digits"""
    C, M = pymodel.load_constants()
    Rinv = pow(1 << 256, -1, P)
    pats = [(1 << 256) - 1, (1 << 255) + 12345, P, P + 1, 2 * P - 1, 2 * P + 7, int("55" * 32, 16), int("aa" * 32, 16),
            sum(((1 << 29) - 1) << (29 * i) for i in range(8)) | (((1 << 24) - 1) << 232), (1 << 256) - (1 << 200)]
    rng = random.Random(4)
    states = [[rng.choice(pats) for _ in range(5)] for _ in range(40)] + [[pats[0]] * 5, [pats[8]] * 5]
    st = np.array([[oracle_mod.int_to_limbs(v) for v in s] for s in states], dtype=np.uint64)
    out = np.empty_like(st)
    for sched in (0, 1, 2):
        hosttest_lib.ht_permute29_sched(p(st), p(out), st.shape[0], sched)
        for s, o in zip(states, out):
            exp = pymodel.perm_reference([v * Rinv % P for v in s], C, M)
            assert [oracle_mod.int_from_mont(x) for x in o] == exp
            assert all(oracle_mod.lib().p252o_is_reduced(p(x)) for x in o)  # outputs are always canonical


def test_dynamic_bounds(oracle_mod, hosttest_lib):
    """Instrumented host build of the kernels' schedule: the largest |column| any reduction meets and the largest
    |top digit| any reduction produces, over random, edge and adversarial (non-canonical, digit-saturating) states,
    stay inside what the static analysis (max_column_bound29) assumes: columns < 2^63, top digits < 2^27 (the
    wide Montgomery step lets values drift to several p; W_0 = 28 X_4 is reduced from the top like every term of the
    aligned recurrence since round 3, so no value is special any more)."""
    import math
    for f in (hosttest_lib.ht_bounds_max_col, hosttest_lib.ht_bounds_max_top, hosttest_lib.ht_bounds_max_top1):
        f.restype = ctypes.c_double
    hosttest_lib.ht_bounds_reset()
    if hosttest_lib.ht_bounds_max_col() < 0:
        pytest.skip("hosttest library built without -DP252_TRACK_BOUNDS")
    pats = [(1 << 256) - 1, (1 << 255) + 12345, P, P - 1, 0, 1, 2 * P - 1, 4 * P + 3, int("55" * 32, 16), int("aa" * 32, 16),
            sum(((1 << 29) - 1) << (29 * i) for i in range(8)) | (((1 << 24) - 1) << 232), (1 << 256) - (1 << 200)]
    rng = random.Random(11)
    states = [[rng.choice(pats) for _ in range(5)] for _ in range(300)] + [[v] * 5 for v in pats]
    st = np.array([[oracle_mod.int_to_limbs(v) for v in s] for s in states], dtype=np.uint64)
    st = np.concatenate([st, oracle_mod.fill_random(77, 5 * 3000).reshape(3000, 5, 4)])
    out = np.empty_like(st)
    hosttest_lib.ht_permute29_sched(p(st), p(out), st.shape[0], 0)
    col, top, top1 = hosttest_lib.ht_bounds_max_col(), hosttest_lib.ht_bounds_max_top(), hosttest_lib.ht_bounds_max_top1()
    assert 2 ** 58 < col < 2 ** 62.6, math.log2(col)
    assert top < 2 ** 27, math.log2(top)     # what max_column_bound29 assumes for the multiplicands of generic products
    assert top1 == 0  # (round 2 tracked W_0 = 28 X_4 apart: up to 2^30; it is an ordinary lazy residue now)
    print("max |column| 2^%.2f, max |top digit| 2^%.2f" % (math.log2(col), math.log2(top)))


def test_fold_top_quotient_bounds():
    """fr29.hpp fold_top (the aligned recurrence's reduction from the top) in big ints: for nine one-digit multiples of
    lazy residues up to 5.3 p — signs chosen adversarially for the actual coefficients — the shifted top of the value fits
    the int32 the quotient estimate multiplies, the quotient stays below 2^29, every column below 2^62, and what is left
    is below 1.3 p.  The constants are the header's."""
    import re
    src = open(os.path.join(os.path.dirname(HERE), "poseidon252_amd", "csrc", "fr29.hpp")).read()
    M = int(re.search(r"#define P252_FOLD_M (\d+)", src).group(1))
    SH = int(re.search(r"#define P252_FOLD_SHIFT (\d+)", src).group(1))
    assert M == round(2 ** (232 + SH + 32) / P)
    pb = [1] + [int(re.search(r"#define P252_PB_%d \((-?\d+)\)" % k, src).group(1)) for k in range(1, 9)]
    assert sum(d << (29 * k) for k, d in enumerate(pb)) == P
    coef = pymodel.A_INT + pymodel.B_INT

    def digits(v):
        d = []
        for _ in range(8):
            d.append(v & ((1 << 29) - 1))
            v >>= 29
        return d + [v]
    def wide_digits(v, push):
        """the same value in WIDE digits (fr29.hpp redc_w<true>: a column's whole low register read as a signed number,
        |d| <= 2^31) — what the kernels feed the recurrence for the W terms.  push = +1 / -1 drives every digit to the
        positive / negative end of its range (the adversarial extreme for a coefficient of that sign), 0 = random transfers."""
        d = digits(v)
        for k in range(8):
            lo, hi = -((2 ** 31 + d[k]) >> 29), (2 ** 31 - 1 - d[k]) >> 29  # transfers that keep d[k] + t 2^29 inside [-2^31, 2^31)
            t = {1: hi, -1: lo}.get(push, rng.randint(lo, hi))
            d[k] += t << 29
            d[k + 1] -= t
        assert sum(x << (29 * k) for k, x in enumerate(d)) == v and all(-2 ** 31 <= x < 2 ** 31 for x in d[:8])
        return d
    rng = random.Random(5)
    lim = int(5.3 * P)
    n_a = len(pymodel.A_INT)  # the first four terms are U values (carried digits), the other five W values (wide digits)
    for wide in (False, True):
        worst, worst_col = 0.0, 0
        for trial in range(20000):
            cols = [0] * 9
            for j, c in enumerate(coef):
                if trial % 3 == 0:
                    x, push = (lim if c > 0 else -lim), (1 if c > 0 else -1)
                elif trial % 3 == 1:
                    x, push = (-lim if c > 0 else lim), (1 if c > 0 else -1)
                else:
                    x, push = rng.choice([lim, -lim, rng.randrange(-lim, lim)]), 0
                dx = wide_digits(x, push) if (wide and j >= n_a) else digits(x)
                for k, d in enumerate(dx):
                    cols[k] += d * c
            for k, d in enumerate(digits(rng.randrange(P))):
                cols[k] += d
            V = sum(c << (29 * k) for k, c in enumerate(cols))
            th = (cols[8] + (cols[7] >> 29)) >> SH
            assert -2 ** 31 <= th < 2 ** 31
            q = (th * M) >> 32
            worst_col = max(worst_col, max(abs(c) for c in cols))
            assert abs(q) < 2 ** 29 and worst_col < 2 ** 62
            worst = max(worst, abs(V - q * P) / P)
        assert worst < 1.3, (wide, worst)
        if wide:  # the representation that actually runs: wide digits do reach further into the columns
            assert worst_col > 2 * carried_col, (worst_col, carried_col)
        carried_col = worst_col


def test_digest_path_with_hoisted_tag_sbox(oracle_mod, hosttest_lib):
    """k_merkle4's specialisation on the host: lane 0's first S-box precomputed from the tag (hades_pre0), last layer
    reduced to the squeezed row — against Hash::digest(Merkle4, ..) of the oracle, random / edge / non-canonical tags"""
    n = 500
    x = oracle_mod.fill_random(1234, 4 * n).reshape(n, 4, 4)
    tags = [oracle_mod.tag(0, [4], 1), oracle_mod.fill_random(5, 1)[0], np.zeros(4, dtype=np.uint64),
            np.array(oracle_mod.int_to_limbs(P - 1), dtype=np.uint64)]
    for tag in tags:
        tag = np.ascontiguousarray(tag, dtype=np.uint64)
        out = np.empty((n, 4), dtype=np.uint64)
        hosttest_lib.ht_merkle4_digest29(p(tag), p(x), p(out), n)
        assert np.array_equal(out, oracle_mod.hash_batch(tag, x, 4, 1).reshape(n, 4))
    # a non-canonical tag pattern (the device reads 256 bits as an integer): same residue class as tag mod p
    raw = np.array(oracle_mod.int_to_limbs((1 << 256) - 5), dtype=np.uint64)
    red = np.array(oracle_mod.int_to_limbs(((1 << 256) - 5) % P), dtype=np.uint64)
    a, b = np.empty((n, 4), dtype=np.uint64), np.empty((n, 4), dtype=np.uint64)
    hosttest_lib.ht_merkle4_digest29(p(raw), p(x), p(a), n)
    hosttest_lib.ht_merkle4_digest29(p(red), p(x), p(b), n)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("lanes", [8, 4])
def test_cooperative_digest_lane_groups(oracle_mod, hosttest_lib, lanes):
    """coop29.hpp — one permutation computed by a group of eight (or four) lanes (the low-latency kernels), with the lanes
    played by host threads and the cross-lane exchange by a barrier: the digest form (k_merkle4_coop<8|4>) and the full
    five-element permutation equal the oracle's for random, edge and non-canonical inputs, and the reductions stay inside
    the same column / digit bounds."""
    import math
    for f in (hosttest_lib.ht_bounds_max_col, hosttest_lib.ht_bounds_max_top, hosttest_lib.ht_bounds_max_top1):
        f.restype = ctypes.c_double
    hosttest_lib.ht_bounds_reset()
    n = 120
    x = oracle_mod.fill_random(4321, 4 * n).reshape(n, 4, 4)
    edge = edge_scalars(oracle_mod)
    x[:edge.shape[0] - 3, 0] = edge[:-3]
    x[:edge.shape[0] - 3, 3] = edge[3:][::-1]
    for tag in (oracle_mod.tag(0, [4], 1), np.zeros(4, dtype=np.uint64), oracle_mod.fill_random(6, 1)[0]):
        tag = np.ascontiguousarray(tag, dtype=np.uint64)
        out = np.empty((n, 4), dtype=np.uint64)
        assert hosttest_lib.ht_merkle4_digest_coop(p(tag), p(x), p(out), n, lanes) == 0
        assert np.array_equal(out, oracle_mod.hash_batch(tag, x, 4, 1).reshape(n, 4))
    # saturated 256-bit patterns (not residues below p): the same class as their reduction, on every lane
    pats = [(1 << 256) - 1, (1 << 255) + 12345, P, 2 * P - 1, int("aa" * 32, 16), (1 << 256) - (1 << 200)]
    raw = np.array([[oracle_mod.int_to_limbs(pats[(i + k) % len(pats)]) for k in range(4)] for i in range(12)], dtype=np.uint64)
    red = np.array([[oracle_mod.int_to_limbs(pats[(i + k) % len(pats)] % P) for k in range(4)] for i in range(12)], dtype=np.uint64)
    tag = np.ascontiguousarray(oracle_mod.tag(0, [4], 1), dtype=np.uint64)
    a, b = np.empty((12, 4), dtype=np.uint64), np.empty((12, 4), dtype=np.uint64)
    hosttest_lib.ht_merkle4_digest_coop(p(tag), p(raw), p(a), 12, lanes)
    hosttest_lib.ht_merkle4_digest_coop(p(tag), p(red), p(b), 12, lanes)
    assert np.array_equal(a, b) and np.array_equal(b, oracle_mod.hash_batch(tag, red, 4, 1).reshape(12, 4))
    # the full permutation: lane i returns element i (4 lanes: element 4 rides on every lane)
    st = np.concatenate([oracle_mod.fill_random(99, 5 * 60).reshape(60, 5, 4), edge[:15].reshape(3, 5, 4),
                         np.array([[oracle_mod.int_to_limbs(pats[(i + k) % len(pats)] % P) for k in range(5)] for i in range(6)], dtype=np.uint64)])
    got = np.empty_like(st)
    assert hosttest_lib.ht_permute_coop(p(st), p(got), st.shape[0], lanes) == 0
    assert np.array_equal(got, oracle_mod.permute_batch(st))
    col, top = hosttest_lib.ht_bounds_max_col(), hosttest_lib.ht_bounds_max_top()
    if col >= 0:
        assert 2 ** 58 < col < 2 ** 62.6, math.log2(col)
        assert top < 2 ** 27, math.log2(top)
        print("cooperative schedule, %d lanes: max |column| 2^%.2f, max |top digit| 2^%.2f" % (lanes, math.log2(col), math.log2(top)))


def test_openings_fast_division_is_exact_over_the_32_bit_range(hosttest_lib):
    """openings.hip's FAST extraction kernel turns the lane's record index into (opening, level) with fastdiv.hpp's reciprocal
    product.  ADVICE r5: round 5's 40-bit multiply-shift wrapped from opening 2^24 on (record >= 2^24 * depth) and nothing
    tested it.  Boundary records of every depth 1..64, a dense sweep around the old failure point, and the old formula as the
    control that this test would have caught it."""
    u32p = ctypes.POINTER(ctypes.c_uint32)
    L = hosttest_lib
    L.ht_fastdiv_check.restype = ctypes.c_size_t
    L.ht_fastdiv_check.argtypes = [u32p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int, u32p]
    L.ht_fastdiv_sweep.restype = ctypes.c_size_t
    L.ht_fastdiv_sweep.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint]
    rng = np.random.default_rng(5)
    for depth in range(1, 65):
        q = np.concatenate([np.arange(0, 4), (1 << np.arange(1, 32)), rng.integers(0, (1 << 32) // depth, 4096)]).astype(np.uint64)
        recs = np.concatenate([q * depth + d for d in (0, 1, depth - 1)] + [q * depth - 1])
        recs = np.concatenate([recs[(recs >= 0) & (recs < (1 << 32))], [(1 << 32) - 1, (1 << 32) - 2, (1 << 32) - depth]])
        recs = np.ascontiguousarray(recs.astype(np.uint32))
        first = ctypes.c_uint32(0)
        bad = L.ht_fastdiv_check(recs.ctypes.data_as(u32p), recs.size, depth, 0, ctypes.byref(first))
        assert bad == 0, "depth %d: %d wrong quotients, first at record %d" % (depth, bad, first.value)
    # the records round 5's kernel got wrong: opening 2^24 at depths 10 / 12 / 20 (the ADVICE's numbers), swept densely
    for depth in (10, 12, 20):
        lo = (1 << 24) * depth - 1000
        assert L.ht_fastdiv_sweep(lo, lo + 2_000_000, depth) == 0
        recs = np.arange(lo, lo + 4096, dtype=np.uint32)
        assert L.ht_fastdiv_check(recs.ctypes.data_as(u32p), recs.size, depth, 1, None) > 0  # the old multiply-shift fails here
    # the top of the 32-bit range, densely, for the depths of the configs (2^24 leaves: 12; 2^20: 10) and the extremes
    for depth in (1, 2, 3, 7, 10, 12, 13, 32, 63, 64):
        assert L.ht_fastdiv_sweep((1 << 32) - 3_000_000, 1 << 32, depth) == 0
