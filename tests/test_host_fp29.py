"""CPU: the S29 field form (pos_evolution_amd/csrc/fp381_s29.h: 14 signed limbs of 29 bits, lazy Montgomery with
R' = 2^406) and the XYZZ accumulation over it (g1_s29.h), compiled for the HOST from the very source the gfx950 kernels
use (tests/native/fp29_host.cpp) and held against Python integers and oracle/g1.py.  No GPU; the kernels that use the
form are checked by the -m gpu tests through the C ABI like every other path."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import g1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = g1.P
B, N = 29, 14
MASK = (1 << B) - 1
RP = 1 << (B * N)
R32 = 1 << 384


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("fp29") / "libfp29.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-Wno-unknown-pragmas", "-shared",
                           "-fPIC", os.path.join(ROOT, "tests", "native", "fp29_host.cpp"), "-o", str(out)])
    return C.CDLL(str(out))


def limbs_of(v):
    """Canonical limbs: 0..12 in [0, 2^29), the top one takes the rest (signed)."""
    out = []
    for _ in range(N - 1):
        out.append(v & MASK)
        v >>= B
    out.append(v)
    return np.array(out, dtype=np.int32)


def value_of(l):
    return sum(int(x) << (B * i) for i, x in enumerate(l))


def loose(rng, v):
    """A redundant representation of v with limbs of both signs, |limb| <= 2^29 + 16 (what limb-wise subtractions leave):
    here and there limb i goes down by 2^29 and limb i + 1 up by one (or the other way round where limb i is tiny)."""
    l = [int(x) for x in limbs_of(v)]
    for i in range(N - 1):
        if rng.random() < 0.5 and l[i] - (1 << B) >= -(1 << B) - 16:
            l[i] -= 1 << B
            l[i + 1] += 1
        elif l[i] <= 16 and rng.random() < 0.5:
            l[i] += 1 << B
            l[i + 1] -= 1
    assert value_of(l) == v and all(abs(x) <= (1 << B) + 16 for x in l[:-1])
    return np.array(l, dtype=np.int32)


def ptr(a, t=C.c_int32):
    return a.ctypes.data_as(C.POINTER(t))


def test_product_and_square_against_python_integers(lib):
    rng = random.Random(29)
    out = np.zeros(N, dtype=np.int32)
    cases = [(0, 0), (1, 1), (P - 1, P - 1), (-(P - 1), P - 1), (7 * P - 3, -5 * P + 11)]
    for _ in range(3000):
        cases.append((rng.randrange(-8 * P, 8 * P), rng.randrange(-8 * P, 8 * P)))
    for a, b in cases:
        for la, lb in ((limbs_of(a), limbs_of(b)), (loose(rng, a), loose(rng, b))):
            lib.fq29_mul(ptr(la), ptr(lb), ptr(out))
            r = value_of(out)
            assert (r * RP - a * b) % P == 0                      # r = a b / R' mod p ...
            assert a * b // RP - 1 <= r <= a * b // RP + P + 1     # ... lazily: within one p above a b / R'
            assert all(-(1 << 28) <= int(x) < (1 << 28) for x in out[:-1]) and abs(int(out[-1])) < 64   # balanced digits
        la = loose(rng, a)
        lib.fq29_sqr(ptr(la), ptr(out))
        r = value_of(out)
        assert (r * RP - a * a) % P == 0 and 0 <= r <= a * a // RP + P + 1
        assert all(-(1 << 28) <= int(x) < (1 << 28) for x in out[:-1])


def test_worst_case_limbs_do_not_overflow_the_accumulator(lib):
    """Every limb at the bound the formulas guarantee (|limb| <= 2^29 + 16, all signs alike): the 28-term column sum must
    still be exact -- a wrapped 64-bit accumulator would break the congruence."""
    out = np.zeros(N, dtype=np.int32)
    top = (1 << B) + 16
    for sa in (1, -1):
        for sb in (1, -1):
            la = np.full(N, sa * top, dtype=np.int32)
            lb = np.full(N, sb * top, dtype=np.int32)
            la[-1], lb[-1] = sa * 40, sb * 40   # top limbs stay small: they hold the value's excess over 2^377
            a, b = value_of(la), value_of(lb)
            lib.fq29_mul(ptr(la), ptr(lb), ptr(out))
            assert (value_of(out) * RP - a * b) % P == 0
            lib.fq29_sqr(ptr(la), ptr(out))
            assert (value_of(out) * RP - a * a) % P == 0


def test_carry_pass_canonical_forms_and_the_zero_test(lib):
    rng = random.Random(5)
    out = np.zeros(N, dtype=np.int32)
    filt = C.c_int(0)
    for _ in range(2000):
        v = rng.randrange(-7 * P, 8 * P)
        l = np.array([rng.randrange(-(3 << B) // 2, 1 << B) for _ in range(N - 1)] + [rng.randrange(-50, 50)], dtype=np.int32)
        lib.fq29_norm(ptr(l), ptr(out))
        assert value_of(out) == value_of(l) and all(abs(int(x)) <= (1 << 28) + 4 for x in out[:-1])
        lv = loose(rng, v)
        lib.fq29_canonical(ptr(lv), ptr(out), 0)
        assert value_of(out) == v % P and np.array_equal(out, limbs_of(v % P))
        w = rng.randrange(-P + 1, 2 * P)
        lib.fq29_canonical(ptr(loose(rng, w)), ptr(out), 1)
        assert np.array_equal(out, limbs_of(w % P))
        assert lib.fq29_is_zero_modp(ptr(lv), C.byref(filt)) == (1 if v % P == 0 else 0)
    for k in range(-8, 17):                      # every multiple of p a lazily reduced value can be
        assert lib.fq29_is_zero_modp(ptr(loose(rng, k * P)), C.byref(filt)) == 1 and filt.value == 1
        assert lib.fq29_is_zero_modp(ptr(loose(rng, k * P + 1)), C.byref(filt)) == 0
    # the filter passes a non-multiple about 25 times in 2^29: it must then be caught by the exact test
    v = 3 * P + (1 << B) * 12345          # same low 29 bits as 3p, not a multiple of p
    assert lib.fq29_is_zero_modp(ptr(limbs_of(v)), C.byref(filt)) == 0 and filt.value == 1


def test_hand_over_between_the_two_montgomery_forms(lib):
    rng = random.Random(11)
    w = np.zeros(12, dtype=np.uint32)
    back = np.zeros(12, dtype=np.uint32)
    out = np.zeros(N, dtype=np.int32)
    for _ in range(500):
        x = rng.randrange(P)
        m32 = x * R32 % P
        w[:] = [(m32 >> (32 * j)) & 0xFFFFFFFF for j in range(12)]
        lib.fq29_words(ptr(w, C.c_uint32), ptr(out), ptr(back, C.c_uint32))
        assert value_of(out) == m32 and np.array_equal(back, w)          # pure re-packing, both ways
        lib.fq29_from_mont32(ptr(w, C.c_uint32), ptr(out))
        assert np.array_equal(out, limbs_of(x * RP % P))                 # x 2^384 -> x R', canonical
        lib.fq29_to_mont32(ptr(loose(rng, x * RP % P + rng.randrange(-3, 4) * P)), ptr(back, C.c_uint32))
        assert np.array_equal(back, w)                                   # and back, from a lazy value


def test_the_square_root_exponentiation_against_python_integers(lib):
    """fq_pow_pm3d4 between the two hand-overs, as fp_sqrt.h's fp_pow_pm3d4 runs it under every square root of the G1 / G2
    decompression kernels: x 2^384 in, x^((p-3)/4) 2^384 out (canonical words) -- residues, non-residues, 0, 1, p - 1; and
    the identities the G2 decompression builds on (x w = sqrt(x) and w = 1 / sqrt(x) for a residue; (x w)^2 = -x otherwise)."""
    rng = random.Random(17)
    E = (P - 3) // 4
    w_in = np.zeros(12, dtype=np.uint32)
    w_out = np.zeros(12, dtype=np.uint32)
    worst = C.c_int32(0)
    inv32 = pow(R32, -1, P)
    for x in [0, 1, 2, P - 1, P - 2, (P - 1) // 2] + [rng.randrange(P) for _ in range(60)]:
        m = x * R32 % P
        w_in[:] = [(m >> (32 * j)) & 0xFFFFFFFF for j in range(12)]
        lib.fq29_pow_pm3d4_words(ptr(w_in, C.c_uint32), ptr(w_out, C.c_uint32), C.byref(worst))
        got = sum(int(w_out[j]) << (32 * j) for j in range(12))
        assert got < P and got * inv32 % P == pow(x, E, P), hex(x)
        assert worst.value <= (1 << (B - 1)) + 64
        w = got * inv32 % P
        if x and pow(x, (P - 1) // 2, P) == 1:
            assert (x * w) ** 2 % P == x and x * w * w % P == 1
        elif x:
            assert (x * w) ** 2 % P == P - x and x * w * w % P == P - 1


def _row(pt):
    """A registry row of the 32-bit form: x, y as 12-word Montgomery values; None -> all zero."""
    if pt is None:
        return [0] * 24
    out = []
    for c in pt:
        m = c * R32 % P
        out += [(m >> (32 * j)) & 0xFFFFFFFF for j in range(12)]
    return out


def _point_of(words48):
    x, y, zz, zzz = (sum(int(words48[12 * c + j]) << (32 * j) for j in range(12)) for c in range(4))
    if zz == 0:
        return None
    inv = pow(R32, -1, P)
    x, y, zz, zzz = (v * inv % P for v in (x, y, zz, zzz))
    assert pow(zz, 3, P) == pow(zzz, 2, P)
    return (x * pow(zz, -1, P) % P, y * pow(zzz, -1, P) % P)


def _run(lib, pts):
    rows = np.array([w for pt in pts for w in _row(pt)], dtype=np.uint32)
    out = np.zeros(48, dtype=np.uint32)
    worst = C.c_int32(0)
    lib.g1q_run(ptr(rows, C.c_uint32), len(pts), ptr(out, C.c_uint32), C.byref(worst))
    assert worst.value <= (1 << B), worst.value          # a row's canonical limbs at most; everything computed is balanced
    assert all(int(v) < P for v in [sum(int(out[12 * c + j]) << (32 * j) for j in range(12)) for c in range(4)])
    return _point_of(out)


def _run_kernel_way(lib, pts):
    rows = np.array([w for pt in pts for w in _row(pt)], dtype=np.uint32)
    out = np.zeros(48, dtype=np.uint32)
    worst, slow = C.c_int32(0), C.c_int(0)
    lib.g1q_run_kernel_way(ptr(rows, C.c_uint32), len(pts), ptr(out, C.c_uint32), C.byref(worst), C.byref(slow))
    assert worst.value <= (1 << B) or slow.value, worst.value   # the constant one's canonical limbs at most (zz = zzz of a first point)
    return _point_of(out), bool(slow.value)


def test_the_kernels_run_general_body_only_and_redo_on_same_x(lib):
    """k_g1_accumulate's lane logic (round 4): first point taken as it is, the general body for every add, same-x
    cases detected by the filter and the run redone by the complete add.  Random runs never leave the fast path; every
    edge case of the group law does, and comes out exact."""
    rng = random.Random(31)
    base = [g1.mul(rng.randrange(1, g1.R_ORDER), g1.G) for _ in range(40)]
    for n in (1, 2, 3, 8, 16, 40):
        got, slow = _run_kernel_way(lib, base[:n])
        assert got == g1.sum_points(base[:n]) and not slow
    pts = [None, base[0], None, None, base[1], base[2], None]
    got, slow = _run_kernel_way(lib, pts)
    assert got == g1.sum_points([p for p in pts if p]) and not slow
    assert _run_kernel_way(lib, [None, None]) == (None, False) and _run_kernel_way(lib, []) == (None, False)
    A, Bp = base[0], base[1]
    for pts, want in (([A, A], g1.double(A)), ([A, g1.neg(A)], None), ([A, g1.neg(A), Bp], Bp),
                      ([A, Bp, g1.add(A, Bp)], g1.double(g1.add(A, Bp))), ([A, Bp, g1.neg(g1.add(A, Bp))], None),
                      ([A, A, A, A], g1.mul(4, A)), ([g1.G] * 9, g1.mul(9, g1.G)),
                      ([g1.mul(i + 1, g1.G) for i in range(12)], g1.mul(78, g1.G))):
        got, slow = _run_kernel_way(lib, pts)
        assert got == want and slow, (pts, got, want, slow)


def test_accumulation_against_the_oracle(lib):
    rng = random.Random(3)
    base = [g1.mul(rng.randrange(1, g1.R_ORDER), g1.G) for _ in range(24)]
    for n in (1, 2, 3, 8, 24):
        pts = base[:n]
        assert _run(lib, pts) == g1.sum_points(pts)
    # rows that hold no point are skipped wherever they stand
    pts = [None, base[0], None, base[1], base[2], None]
    assert _run(lib, pts) == g1.sum_points([p for p in pts if p])
    assert _run(lib, [None, None]) is None and _run(lib, []) is None


def test_accumulation_edge_cases_of_the_group_law(lib):
    rng = random.Random(4)
    A = g1.mul(rng.randrange(1, g1.R_ORDER), g1.G)
    Bp = g1.mul(rng.randrange(1, g1.R_ORDER), g1.G)
    assert _run(lib, [A, A]) == g1.double(A)                         # second point equals the (affine) accumulator
    assert _run(lib, [A, g1.neg(A)]) is None                         # ... or its negative
    assert _run(lib, [A, g1.neg(A), Bp]) == Bp                       # and the run goes on from infinity
    assert _run(lib, [A, Bp, g1.add(A, Bp)]) == g1.double(g1.add(A, Bp))     # XYZZ accumulator meets an equal point
    assert _run(lib, [A, Bp, g1.neg(g1.add(A, Bp))]) is None
    assert _run(lib, [A, A, A, A]) == g1.mul(4, A)                   # double, then madd, then double again (3A + A)
    seq = [g1.mul(i + 1, g1.G) for i in range(12)]                   # the structured keys of the synthetic registry
    assert _run(lib, seq) == g1.mul(78, g1.G)
    assert _run(lib, [g1.G] * 9) == g1.mul(9, g1.G)


def _run_tree(lib, pts, k):
    rows = np.array([w for pt in pts for w in _row(pt)], dtype=np.uint32) if pts else np.zeros(24, dtype=np.uint32)
    out = np.zeros(48, dtype=np.uint32)
    worst = C.c_int32(0)
    lib.g1q_tree_run(ptr(rows, C.c_uint32), len(pts), k, ptr(out, C.c_uint32), C.byref(worst))
    assert worst.value <= (1 << B), worst.value   # a table row's canonical limbs at most (a lane of one point passed up); sums are balanced
    assert all(int(v) < P for v in [sum(int(out[12 * c + j]) << (32 * j) for j in range(12)) for c in range(4)])
    return _point_of(out)


def test_the_trees_adds_over_the_lanes_accumulators(lib):
    """k_g1_tree since round 6: the lanes' accumulators arrive in the accumulation's own lazy form (X, Y carry-passed or a table
    row, ZZ, ZZZ products or the constant one, all limbs zero = infinity) and are added in S29 with the complete add of
    g1_s29.h (g1q_add: what the cooperative two- and four-lane adds of g1_kernels.hip spread over lanes and fall back to).
    Lanes of k points reduced pairwise, level by level, against the oracle's sum: random points, every lane count up to a
    workgroup's, empty lanes, and every special case of the group law BETWEEN lanes."""
    rng = random.Random(77)
    base = [g1.mul(rng.randrange(1, g1.R_ORDER), g1.G) for _ in range(96)]
    for n, k in ((1, 4), (2, 1), (3, 1), (7, 2), (16, 4), (33, 4), (64, 1), (96, 16), (96, 5)):
        assert _run_tree(lib, base[:n], k) == g1.sum_points(base[:n]), (n, k)
    assert _run_tree(lib, [], 4) is None
    A, Bp, Cp = base[0], base[1], base[2]
    N = g1.neg
    # whole lanes without a point (all-zero rows): infinity operands on either side, at the first level and later ones
    assert _run_tree(lib, [None, None, A, Bp], 2) == g1.add(A, Bp)
    assert _run_tree(lib, [A, Bp, None, None], 2) == g1.add(A, Bp)
    assert _run_tree(lib, [None] * 8 + [A] + [None] * 7, 4) == A
    assert _run_tree(lib, [None] * 16, 4) is None
    # two lanes hold the same point / opposite points: the doubling and the cancellation inside the tree
    assert _run_tree(lib, [A, A], 1) == g1.double(A)                              # affine + affine lanes
    assert _run_tree(lib, [A, N(A)], 1) is None
    assert _run_tree(lib, [A, Bp, A, Bp], 2) == g1.double(g1.add(A, Bp))          # XYZZ + XYZZ, equal
    assert _run_tree(lib, [A, Bp, N(A), N(Bp)], 2) is None                        # ... and opposite
    assert _run_tree(lib, [A, Bp, N(Bp), N(A), Cp], 2) == Cp                      # infinity met at the second level
    assert _run_tree(lib, [A, Bp, g1.add(A, Bp), None], 2) == g1.double(g1.add(A, Bp))   # XYZZ lane + a lane of one (affine) point
    assert _run_tree(lib, [A] * 64, 1) == g1.mul(64, A)                           # a doubling at every level
    assert _run_tree(lib, [A] * 64, 4) == g1.mul(64, A)
    seq = [g1.mul(i + 1, g1.G) for i in range(48)]                                # the synthetic registry's structured keys
    assert _run_tree(lib, seq, 4) == g1.mul(48 * 49 // 2, g1.G)


def test_generated_constants_are_current():
    """fp381_s29_consts.inc is what tools/gen_fp29_consts.py prints (everything in it follows from the prime)."""
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_fp29_consts.py")], capture_output=True, text=True,
                         check=True).stdout
    assert out == open(os.path.join(ROOT, "pos_evolution_amd", "csrc", "fp381_s29_consts.inc")).read()
    n0 = int(out.split("FQ_N0INV = ")[1].split("u;")[0])
    assert (n0 * P + 1) % (1 << B) == 0
