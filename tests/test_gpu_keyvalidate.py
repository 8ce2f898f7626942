"""-m gpu: KeyValidate on the GPU (pe_g1_key_validate: r * P == infinity per key, identity rejected) against the oracle
(oracle/g1.py key_validate) -- SURVEY A.7: FastAggregateVerify validates every pubkey before summing them."""
import numpy as np
import pytest

import pos_evolution_amd.synth as synth
from oracle import g1

pytestmark = pytest.mark.gpu


def test_key_validate_mixed_points(engine_factory):
    e = engine_factory()
    rng = np.random.default_rng(4)
    pts, want = [], []
    for k in (1, 2, 3, 0xDEADBEEF, g1.R_ORDER - 1):
        pts.append(g1.mul(k, g1.G)); want.append(0)
    for _ in range(20):
        pts.append(g1.mul(int(rng.integers(1, 2**62)) * int(rng.integers(1, 2**62)), g1.G)); want.append(0)
    pts.append(None); want.append(4)                                   # identity: KeyValidate fails
    for x0 in (1, 2, 3, 1000, 2**200, 2**380):
        p = g1.curve_point_from_x(x0)
        assert not g1.in_subgroup(p)
        pts.append(p); want.append(3)                                  # on the curve, outside the subgroup
        pts.append(g1.neg(p)); want.append(3)
    h_cof = 0x396C8C005555E1568C00AAAB0000AAAB
    pts.append(g1.mul_unreduced(h_cof, g1.curve_point_from_x(7))); want.append(0)   # cofactor-cleared: inside
    buf = np.frombuffer(b"".join(g1.to_bytes96(p) for p in pts), dtype=np.uint8).reshape(-1, 96)
    got = e.g1_key_validate(buf)
    assert list(got) == want
    assert [0 if g1.key_validate(p) else (4 if p is None else 3) for p in pts] == want


def test_key_validate_registry(engine_factory):
    """The registry as loaded (points96 = None): the synthetic keys A + v*B are subgroup points; one forged row is found."""
    e = engine_factory()
    n = 5000
    pts = synth.registry_points(e, n).copy()
    bad = g1.curve_point_from_x(12345)
    pts[1234] = np.frombuffer(g1.to_bytes96(bad), dtype=np.uint8)
    e.set_validators(synth.balances(n, 1), np.ones(n, dtype=np.uint8), pts)
    st = e.g1_key_validate()
    assert st[1234] == 3 and (np.delete(st, 1234) == 0).all()
