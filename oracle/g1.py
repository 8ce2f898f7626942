"""BLS12-381 G1 arithmetic on Python ints -- ORACLE / TEST INFRASTRUCTURE ONLY.

This file is part of ``oracle/``: a CPU restatement used by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as the
checker.  The product path (``pos_evolution_amd/``) never imports it.

PARITY UNPINNED for this file: the reference (``/root/reference/pos-evolution.md``)
contains no BLS arithmetic at all -- its only ``bls.`` call is ``bls.Verify`` in
``process_deposit`` (pe:165) and the aggregation step exists as prose only
(pe:474, pe:659, pe:715, pe:1536).  Upstream the arithmetic lives in ``py_ecc``
(pyspec default; no version pinned by the reference, none on disk, no network).
What is restated here is the published curve (draft-irtf-cfrg-pairing-friendly-
curves, BLS12-381): E/Fp: y^2 = x^3 + 4.  What pins it instead are self-derived
known answers (generator on curve, r*G = infinity, closed-form sums, three
independent implementations agreeing) -- see tests/test_oracle_g1.py.

Affine points are ``(x, y)`` tuples of ints in [0, p); the point at infinity is
``None``.  Everything is exact integer arithmetic, so the "tolerance" for the
projective->affine output is 0.
"""

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
B_COEFF = 4
GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G = (GX, GY)
INF = None


def is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B_COEFF) % P == 0


def neg(pt):
    if pt is None:
        return None
    x, y = pt
    return (x, (-y) % P)


def double(pt):
    if pt is None:
        return None
    x, y = pt
    if y == 0:
        return None
    lam = (3 * x * x) * pow(2 * y, -1, P) % P
    x3 = (lam * lam - 2 * x) % P
    y3 = (lam * (x - x3) - y) % P
    return (x3, y3)


def add(p1, p2):
    """Affine group law with every edge case (inf, P == Q, P == -Q)."""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        return double(p1)
    lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def mul(k, pt):
    """Double-and-add scalar multiplication (independent check for closed forms)."""
    k %= R_ORDER
    acc = None
    base = pt
    while k:
        if k & 1:
            acc = add(acc, base)
        base = double(base)
        k >>= 1
    return acc


def mul_unreduced(k, pt):
    """k * pt WITHOUT reducing k mod r: the curve E(Fp) has points outside the prime-order subgroup (cofactor
    0x396c8c005555e1568c00aaab0000aaab), for which r * pt != infinity."""
    acc = None
    base = pt
    while k:
        if k & 1:
            acc = add(acc, base)
        base = double(base)
        k >>= 1
    return acc


def in_subgroup(pt) -> bool:
    """The subgroup part of KeyValidate (IETF BLS draft 2.5): r * P == infinity."""
    return mul_unreduced(R_ORDER, pt) is None


def key_validate(pt) -> bool:
    """KeyValidate: a curve point, not the identity, in the subgroup."""
    return pt is not None and is_on_curve(pt) and in_subgroup(pt)


def curve_point_from_x(x0: int):
    """The first x >= x0 with x^3 + 4 a square, and the smaller root y (p = 3 mod 4): almost surely OUTSIDE the
    subgroup -- test input for KeyValidate."""
    x = x0 % P
    while True:
        rhs = (x * x * x + B_COEFF) % P
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P == rhs:
            return (x, min(y, P - y))
        x += 1


def sum_points(points):
    acc = None
    for pt in points:
        acc = add(acc, pt)
    return acc


# ---- serialisation -------------------------------------------------------
# Uncompressed 96-byte form: big-endian x (48 B) || big-endian y (48 B); the top
# three bits of byte 0 are flags (bit 7 compressed = 0 here, bit 6 infinity,
# bit 5 sign -- unused when uncompressed).  Infinity = 0x40 followed by zeros.
# Compressed 48-byte form: big-endian x with bit 7 = 1, bit 6 = infinity,
# bit 5 = (y > (p-1)/2).   [UPSTREAM-MEMORY: ZCash/IETF BLS12-381 encoding]

def to_bytes96(pt):
    if pt is None:
        return bytes([0x40]) + bytes(95)
    x, y = pt
    return x.to_bytes(48, "big") + y.to_bytes(48, "big")


def from_bytes96(b):
    assert len(b) == 96
    if b[0] & 0x40:
        return None
    x = int.from_bytes(b[:48], "big") & ((1 << 381) - 1)
    y = int.from_bytes(b[48:], "big")
    return (x, y)


def compress(pt):
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    flag = 0x80 | (0x20 if y > (P - 1) // 2 else 0)
    raw = bytearray(x.to_bytes(48, "big"))
    raw[0] |= flag
    return bytes(raw)


def decompress(b):
    """48-byte compressed form -> affine point; raises ValueError on a malformed or off-curve encoding."""
    assert len(b) == 48
    c, inf, sign = b[0] & 0x80, b[0] & 0x40, b[0] & 0x20
    x = int.from_bytes(b, "big") & ((1 << 381) - 1)
    if not c:
        raise ValueError("not a compressed encoding")
    if inf:
        if sign or x:
            raise ValueError("malformed infinity")
        return None
    if x >= P:
        raise ValueError("x not canonical")
    rhs = (x * x * x + B_COEFF) % P
    y = pow(rhs, (P + 1) // 4, P)
    if y * y % P != rhs:
        raise ValueError("not on the curve")
    if (y > (P - 1) // 2) != bool(sign):
        y = P - y
    return (x, y)


def synthetic_points(n, a_scalar, b_scalar):
    """P_i = A + i*B built incrementally (one add each), A = a*G, B = b*G.

    Closed form used by the tests:
        sum_{i in S} P_i = |S|*A + (sum_{i in S} i)*B
    """
    A = mul(a_scalar, G)
    B = mul(b_scalar, G)
    out = []
    cur = A
    for _ in range(n):
        out.append(cur)
        cur = add(cur, B)
    return out
