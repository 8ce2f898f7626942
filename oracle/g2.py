"""BLS12-381 G2 arithmetic on Python ints -- ORACLE / TEST INFRASTRUCTURE ONLY.

Part of ``oracle/`` (see oracle/g1.py for the rules: only tests/, smoke() and
bench.py's cpu_baseline leg may import it; the product path never does).

PARITY UNPINNED, as for G1: the reference holds no BLS arithmetic (its signature
type is ``BLSSignature`` = Bytes96, pe:37, pe:717; aggregation is prose, pe:659,
pe:1536).  Restated from the published curve (draft-irtf-cfrg-pairing-friendly-
curves, BLS12-381): the sextic twist E'/Fp2: y^2 = x^3 + 4(1 + u), Fp2 = Fp[u]/(u^2 + 1).
Pinned by self-derived known answers: the generator satisfies the curve equation and
r * G2 = infinity (a wrong constant or a wrong group law fails both), the compressed
generator encodes to the well known ``93e02b60...`` prefix, closed-form sums.

Fp2 elements are ``(c0, c1)`` int tuples (c0 + c1 u); affine points are ``(x, y)``
tuples of Fp2 elements; infinity is ``None``.  Wire format (uncompressed, 192 bytes,
ZCash convention): x.c1 || x.c0 || y.c1 || y.c0, each 48 bytes big-endian; bit 6 of
byte 0 flags infinity (all other bytes zero).
"""
from oracle.g1 import P, R_ORDER

B2 = (4, 4)
G2X = (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
       0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E)
G2Y = (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
       0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)
G2 = (G2X, G2Y)
INF = None


# ---- Fp2
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return (-a[0] % P, -a[1] % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return f2_mul(a, a)


def f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * n % P, -a[1] * n % P)


def f2_scalar(a, k):
    return (a[0] * k % P, a[1] * k % P)


F2_ZERO = (0, 0)
F2_ONE = (1, 0)


# ---- group law (affine, exact)
def is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B2)) == F2_ZERO


def neg(pt):
    if pt is None:
        return None
    return (pt[0], f2_neg(pt[1]))


def double(pt):
    if pt is None:
        return None
    x, y = pt
    if y == F2_ZERO:
        return None
    lam = f2_mul(f2_scalar(f2_sqr(x), 3), f2_inv(f2_scalar(y, 2)))
    x3 = f2_sub(f2_sqr(lam), f2_scalar(x, 2))
    return (x3, f2_sub(f2_mul(lam, f2_sub(x, x3)), y))


def add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    if p1[0] == p2[0]:
        return double(p1) if p1[1] == p2[1] else None
    lam = f2_mul(f2_sub(p2[1], p1[1]), f2_inv(f2_sub(p2[0], p1[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), p1[0]), p2[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(p1[0], x3)), p1[1]))


# Jacobian internals keep scalar multiplication and long sums free of per-step inversions.
def _jac_double(X, Y, Z):
    if Z == F2_ZERO or Y == F2_ZERO:
        return (F2_ONE, F2_ONE, F2_ZERO)
    A = f2_sqr(X)
    B = f2_sqr(Y)
    C = f2_sqr(B)
    D = f2_scalar(f2_sub(f2_sub(f2_sqr(f2_add(X, B)), A), C), 2)
    E = f2_scalar(A, 3)
    X3 = f2_sub(f2_sqr(E), f2_scalar(D, 2))
    Y3 = f2_sub(f2_mul(E, f2_sub(D, X3)), f2_scalar(C, 8))
    return (X3, Y3, f2_scalar(f2_mul(Y, Z), 2))


def _jac_add(p, q):
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    if Z1 == F2_ZERO:
        return q
    if Z2 == F2_ZERO:
        return p
    Z1Z1, Z2Z2 = f2_sqr(Z1), f2_sqr(Z2)
    U1, U2 = f2_mul(X1, Z2Z2), f2_mul(X2, Z1Z1)
    S1, S2 = f2_mul(f2_mul(Y1, Z2), Z2Z2), f2_mul(f2_mul(Y2, Z1), Z1Z1)
    if U1 == U2:
        return _jac_double(X1, Y1, Z1) if S1 == S2 else (F2_ONE, F2_ONE, F2_ZERO)
    H, Rr = f2_sub(U2, U1), f2_sub(S2, S1)
    HH = f2_sqr(H)
    HHH = f2_mul(H, HH)
    V = f2_mul(U1, HH)
    X3 = f2_sub(f2_sub(f2_sqr(Rr), HHH), f2_scalar(V, 2))
    Y3 = f2_sub(f2_mul(Rr, f2_sub(V, X3)), f2_mul(S1, HHH))
    return (X3, Y3, f2_mul(f2_mul(Z1, Z2), H))


def _to_jac(pt):
    return (F2_ONE, F2_ONE, F2_ZERO) if pt is None else (pt[0], pt[1], F2_ONE)


def _from_jac(j):
    X, Y, Z = j
    if Z == F2_ZERO:
        return None
    zi = f2_inv(Z)
    zi2 = f2_sqr(zi)
    return (f2_mul(X, zi2), f2_mul(Y, f2_mul(zi2, zi)))


def mul(k, pt):
    k %= R_ORDER * 8  # keep cofactor-free scalars as given; reduce only absurd sizes
    acc = _to_jac(None)
    base = _to_jac(pt)
    while k:
        if k & 1:
            acc = _jac_add(acc, base)
        base = _jac_double(*base)
        k >>= 1
    return _from_jac(acc)


def sum_points(points):
    acc = _to_jac(None)
    for p in points:
        acc = _jac_add(acc, _to_jac(p))
    return _from_jac(acc)


# ---- wire format
def to_bytes192(pt):
    if pt is None:
        return bytes([0x40]) + bytes(191)
    (x0, x1), (y0, y1) = pt
    return x1.to_bytes(48, "big") + x0.to_bytes(48, "big") + y1.to_bytes(48, "big") + y0.to_bytes(48, "big")


def from_bytes192(b):
    assert len(b) == 192
    if b[0] & 0x40:
        return None
    x1 = int.from_bytes(b[0:48], "big") & ((1 << 381) - 1)
    x0 = int.from_bytes(b[48:96], "big")
    y1 = int.from_bytes(b[96:144], "big")
    y0 = int.from_bytes(b[144:192], "big")
    return ((x0, x1), (y0, y1))


def compress(pt):
    """96-byte compressed form (the BLSSignature wire type, pe:37): x.c1 || x.c0 with flag bits in byte 0."""
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    (x0, x1), (y0, y1) = pt
    # sign: lexicographically largest of y / -y, compared on (c1, c0)
    ny0, ny1 = -y0 % P, -y1 % P
    largest = (y1, y0) > (ny1, ny0)
    out = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    out[0] |= 0x80 | (0x20 if largest else 0)
    return bytes(out)


def f2_sqrt(a):
    """A square root of a in Fp2 or None (p = 3 mod 4, "complex method": two Fp square roots and one inversion)."""
    a0, a1 = a
    def fp_sqrt(v):
        r = pow(v, (P + 1) // 4, P)
        return r if r * r % P == v % P else None
    if a1 == 0:
        r = fp_sqrt(a0)
        if r is not None:
            return (r, 0)
        t = fp_sqrt(-a0 % P)                    # -1 is a non-residue: exactly one of a0, -a0 is a square
        return (0, t)
    s = fp_sqrt((a0 * a0 + a1 * a1) % P)       # the norm must be a square in Fp
    if s is None:
        return None
    inv2 = (P + 1) // 2
    d = (a0 + s) * inv2 % P
    y0 = fp_sqrt(d)
    if y0 is None:
        d = (a0 - s) * inv2 % P
        y0 = fp_sqrt(d)
    y1 = a1 * pow(2 * y0, -1, P) % P
    return (y0, y1)


def decompress(b):
    """96-byte compressed BLSSignature (pe:37) -> affine point; raises ValueError on malformed / off-curve input."""
    assert len(b) == 96
    c, inf, sign = b[0] & 0x80, b[0] & 0x40, b[0] & 0x20
    x1 = int.from_bytes(b[:48], "big") & ((1 << 381) - 1)
    x0 = int.from_bytes(b[48:], "big")
    if not c:
        raise ValueError("not a compressed encoding")
    if inf:
        if sign or x0 or x1:
            raise ValueError("malformed infinity")
        return None
    if x0 >= P or x1 >= P:
        raise ValueError("x not canonical")
    x = (x0, x1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), B2))
    if y is None:
        raise ValueError("not on the curve")
    ny = f2_neg(y)
    if ((y[1], y[0]) > (ny[1], ny[0])) != bool(sign):
        y = ny
    return (x, y)


def synthetic_points(n, a_scalar, b_scalar):
    """[(a + i*b) * G2 for i in range(n)] by repeated addition (exact), mirroring oracle.g1.synthetic_points."""
    cur = _to_jac(mul(a_scalar, G2))
    step = _to_jac(mul(b_scalar, G2))
    out = []
    for _ in range(n):
        out.append(cur)
        cur = _jac_add(cur, step)
    return _batch_from_jac(out)


def _batch_from_jac(js):
    """Montgomery's trick over the Z coordinates: one Fp2 inversion for the whole list."""
    prefix, acc = [], F2_ONE
    for (_, _, Z) in js:
        prefix.append(acc)
        if Z != F2_ZERO:
            acc = f2_mul(acc, Z)
    inv = f2_inv(acc)
    out = [None] * len(js)
    for i in range(len(js) - 1, -1, -1):
        X, Y, Z = js[i]
        if Z == F2_ZERO:
            continue
        zi = f2_mul(inv, prefix[i])
        inv = f2_mul(inv, Z)
        zi2 = f2_sqr(zi)
        out[i] = (f2_mul(X, zi2), f2_mul(Y, f2_mul(zi2, zi)))
    return out
