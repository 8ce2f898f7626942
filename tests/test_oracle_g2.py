"""oracle/g2.py pins (CPU): the curve constants, the group law and the wire format."""
import json
import os

from oracle import g1, g2

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "g2_vectors.json")


def test_generator_and_order():
    assert g2.is_on_curve(g2.G2)
    assert g2.mul(g1.R_ORDER, g2.G2) is None              # wrong constants or a wrong group law fail here
    assert g2.mul(g1.R_ORDER - 1, g2.G2) == g2.neg(g2.G2)


def test_external_known_answers():
    # compressed generator and 2*generator as published with the curve (zkcrypto/bls12_381 test vectors)
    assert g2.compress(g2.G2).hex().startswith("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049")
    assert g2.compress(g2.double(g2.G2)).hex().startswith("aa4edef9c1ed7f729f520e47730a124fd70662a904ba1074728114d1031e1572")
    assert g2.compress(None) == bytes([0xC0]) + bytes(95)
    # round 5: the whole 96 bytes of 2 * G2 and the leading 16 of 3 * G2, typed in from the published
    # encodings BEFORE this oracle's output was looked at (no network: memory of the IETF / zkcrypto vectors is the only
    # external source there is); the rest of each point is then held by the structure no wrong point has:
    two, three = g2.double(g2.G2), g2.mul(3, g2.G2)
    assert g2.compress(two).hex() == (
        "aa4edef9c1ed7f729f520e47730a124fd70662a904ba1074728114d1031e1572c6c886f6b57ec72a6178288c47c33577"
        "1638533957d540a9d2370f17cc7ed5863bc0b995b8825e0ee1ea1e1e4d00dbae81f14b0bf3611b78c952aacab827a053")
    assert g2.compress(three).hex().startswith("89380275bbc8e5dcea7dc4dd7e0550ff")
    for p in (two, three):
        assert g2.is_on_curve(p) and g2.mul(g1.R_ORDER, p) is None          # r * P = infinity
        assert g2.decompress(g2.compress(p)) == p
        from tests.test_oracle_g2_psi import Z, psi                          # psi(P) = [z] P on the subgroup
        assert psi(p) == g2.mul(Z % g1.R_ORDER, p)


def test_group_law_consistency():
    two = g2.double(g2.G2)
    assert two == g2.add(g2.G2, g2.G2) == g2.mul(2, g2.G2)
    assert g2.add(two, g2.G2) == g2.mul(3, g2.G2)
    assert g2.add(g2.G2, g2.neg(g2.G2)) is None
    assert g2.add(None, two) == two and g2.add(two, None) == two
    pts = g2.synthetic_points(40, 5, 3)
    assert all(g2.is_on_curve(p) for p in pts)
    assert pts[7] == g2.mul(5 + 21, g2.G2)
    assert g2.sum_points(pts) == g2.mul(sum(5 + 3 * i for i in range(40)), g2.G2)
    for p in (two, pts[3], None):
        assert g2.from_bytes192(g2.to_bytes192(p)) == p


def test_fp2_field_axioms():
    a, b = (3, 5), (g1.P - 7, 11)
    assert g2.f2_mul(a, g2.f2_inv(a)) == g2.F2_ONE
    assert g2.f2_mul(a, b) == g2.f2_mul(b, a)
    assert g2.f2_sqr((0, 1)) == (g1.P - 1, 0)             # u^2 = -1


def test_golden_vectors_match_oracle():
    with open(GOLDEN) as f:
        vec = json.load(f)
    for case in vec["sums"]:
        pts = [g2.from_bytes192(bytes.fromhex(h)) for h in case["points"]]
        assert g2.to_bytes192(g2.sum_points(pts)).hex() == case["sum"], case["name"]


def test_compress_decompress_roundtrip():
    import pytest
    assert g2.decompress(g2.compress(g2.G2)) == g2.G2
    signs = set()
    for p in g2.synthetic_points(60, 7, 11):
        c = g2.compress(p)
        signs.add(c[0] & 0x20)
        assert g2.decompress(c) == p
        assert g2.decompress(bytes([c[0] ^ 0x20]) + c[1:]) == g2.neg(p)
    assert signs == {0, 0x20}
    assert g2.decompress(g2.compress(None)) is None
    for a in [(4, 0), (g1.P - 4, 0), (9, 0), (5, 0), (g1.P - 5, 0)]:        # the a1 == 0 branches of the Fp2 root
        r = g2.f2_sqrt(a)
        assert g2.f2_sqr(r) == a
    for bad in (bytes(96), bytes([0xE0]) + bytes(95), bytes([0x9F]) + b"\xff" * 95):
        with pytest.raises(ValueError):
            g2.decompress(bad)


def test_c_abi_g2_compress_matches_oracle():
    import ctypes as C
    import numpy as np
    from pos_evolution_amd import _abi
    lib = _abi.load()
    pts = g2.synthetic_points(20, 3, 5) + [None, g2.G2, g2.neg(g2.G2)]
    raw = np.frombuffer(b"".join(g2.to_bytes192(p) for p in pts), dtype=np.uint8).copy()
    out = np.zeros(96 * len(pts), dtype=np.uint8)
    assert lib.pe_g2_compress(raw.ctypes.data_as(C.POINTER(C.c_uint8)), len(pts), out.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    assert out.tobytes() == b"".join(g2.compress(p) for p in pts)
