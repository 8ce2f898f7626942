"""CPU: the endomorphism test k_g2_subgroup_check uses (round 4) -- P in G2 <=> psi(P) = [z]P -- held against oracle/g2.py:
the constants c_x, c_y follow from their definition, the Montgomery limbs in g2_kernels.hip are theirs, the criterion agrees
with r * P == infinity on points inside and outside the subgroup."""
import os
import random
import re

from oracle import g1, g2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, R = g1.P, g1.R_ORDER
Z = -0xd201000000010000     # the BLS12-381 parameter


def _f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = g2.f2_mul(r, a)
        a = g2.f2_sqr(a)
        e >>= 1
    return r


CX = g2.f2_inv(_f2_pow((1, 1), (P - 1) // 3))
CY = g2.f2_inv(_f2_pow((1, 1), (P - 1) // 2))


def psi(pt):
    (x0, x1), (y0, y1) = pt
    return (g2.f2_mul((x0, (-x1) % P), CX), g2.f2_mul((y0, (-y1) % P), CY))


def test_constants_in_the_kernel_source_are_the_definitions():
    src = open(os.path.join(ROOT, "pos_evolution_amd", "csrc", "g2_kernels.hip")).read()

    def limbs(name):
        body = src[src.index(f"{name}(int j)"):]
        body = body[:body.index("}")]
        ws = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{8})u", body)]
        assert len(ws) == 12, name
        return sum(w << (32 * j) for j, w in enumerate(ws))

    mont = lambda v: v * (1 << 384) % P
    assert CX[0] == 0
    assert limbs("psi_cx1_limb") == mont(CX[1])
    assert limbs("psi_cy0_limb") == mont(CY[0]) and limbs("psi_cy1_limb") == mont(CY[1])
    assert "0xd2010000u" in src and "0x00010000u" in src          # |z|, high and low word


def test_psi_is_an_endomorphism_acting_as_z_on_the_subgroup():
    rng = random.Random(7)
    for _ in range(4):
        pt = g2.mul(rng.randrange(1, R), g2.G2)
        assert g2.is_on_curve(psi(pt))
        assert psi(pt) == g2.mul(Z % R, pt)


def test_the_criterion_equals_r_times_p_is_infinity_off_the_subgroup():
    rng = random.Random(8)
    seen = 0
    while seen < 4:
        x = (rng.randrange(P), rng.randrange(P))
        y = g2.f2_sqrt(g2.f2_add(g2.f2_mul(g2.f2_sqr(x), x), (4, 4)))
        if y is None:
            continue
        pt = (x, y)
        in_g2 = g2.mul(R, pt) is None
        q = g2.mul(-Z, pt)                                  # no reduction mod r: the point has another order
        crit = q is not None and psi(pt) == g2.neg(q)
        assert crit == in_g2
        seen += 0 if in_g2 else 1
