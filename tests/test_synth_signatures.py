"""CPU: the synthetic BLSSignatures of bench.py's `with_signatures` leg (pos_evolution_amd/synth.py: a small Fp2 curve
arithmetic of its own, so that the workload is built without the oracle) against oracle/g2.py -- the points, the 192-byte
wire order and the closed form of a subset sum the bench and the -m gpu test compare aggregate signatures with."""
from oracle import g2
from pos_evolution_amd import synth


def test_generator_progression_wire_order_and_closed_form():
    a, b = 0xABCDEF12345, 0x1357
    assert synth._G2 == g2.G2 and g2.is_on_curve(synth._G2)
    A, B = synth._ec2_mul(a, synth._G2), synth._ec2_mul(b, synth._G2)
    assert A == g2.mul(a, g2.G2) and B == g2.mul(b, g2.G2)
    pts = g2.synthetic_points(300, a, b)
    for i in (0, 1, 127, 128, 299):
        assert synth._ec2_add(A, synth._ec2_mul(i, B)) == pts[i]
        assert synth._enc192(pts[i]) == g2.to_bytes192(pts[i])
    assert synth._enc192(None) == g2.to_bytes192(None)
    assert synth._ec2_add(pts[5], g2.neg(pts[5])) is None and synth._ec2_add(pts[5], pts[5]) == g2.double(pts[5])
    rows = [3, 7, 128, 299]
    assert synth.signature_closed_form(rows, a, b) == g2.sum_points([pts[i] for i in rows])
    assert synth.signature_closed_form([], a, b) is None
