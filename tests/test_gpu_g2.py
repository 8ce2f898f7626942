"""G2 signature aggregation through the C ABI vs oracle/g2.py: exact affine bytes."""
import json
import os

import numpy as np
import pytest

from oracle import g1, g2

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "g2_vectors.json")


def _rows(points):
    return np.stack([np.frombuffer(g2.to_bytes192(p), dtype=np.uint8) for p in points]) if points else \
        np.zeros((0, 192), dtype=np.uint8)


def test_g2_golden_vectors(engine_factory):
    e = engine_factory()
    with open(GOLDEN) as f:
        vec = json.load(f)
    # every case as its own group of ONE call (ragged, with an empty group) ...
    pts, offsets = [], [0]
    for case in vec["sums"]:
        pts += [np.frombuffer(bytes.fromhex(h), dtype=np.uint8) for h in case["points"]]
        offsets.append(len(pts))
    out = e.g2_sum(np.stack(pts), offsets)
    for case, got in zip(vec["sums"], out):
        assert got.tobytes().hex() == case["sum"], case["name"]
    # ... and the published compressed 2*G2 through the doubling branch
    two = e.g2_sum(_rows([g2.G2]), [0, 2], index=[0, 0])[0].tobytes()
    assert g2.compress(g2.from_bytes192(two)).hex() == vec["two_g_compressed"]
    assert vec["two_g_compressed"].startswith("aa4edef9c1ed7f72")


def test_g2_edge_cases(engine_factory):
    e = engine_factory()
    A = g2.mul(7, g2.G2)
    pts = _rows([A, g2.neg(A), None, A, g2.G2])
    out = e.g2_sum(pts, [0, 2, 4, 4, 5, 5], index=None)
    assert out[0][0] == 0x40 and not out[0][1:].any()                 # P + (-P)
    assert out[1].tobytes() == g2.to_bytes192(A)                      # infinity inputs are skipped
    assert out[2][0] == 0x40 and not out[2][1:].any()                 # empty group
    assert out[3].tobytes() == g2.to_bytes192(g2.G2)
    # (i+1)*G2: the running sum meets an equal point (1G + 2G = 3G, + 3G)
    prog = g2.synthetic_points(64, 1, 1)
    got = e.g2_sum(_rows(prog), [0, 64])[0].tobytes()
    assert got == g2.to_bytes192(g2.mul(64 * 65 // 2, g2.G2))
    assert e.g2_sum(np.zeros((0, 192), dtype=np.uint8), [0, 0])[0][0] == 0x40


@pytest.mark.parametrize("n,groups", [(600, 7), (3000, 200), (12000, 3)])
def test_g2_sum_vs_oracle(engine_factory, n, groups):
    """Random ragged groups over an index list; closed form (a*len + b*sum(idx)) * G2 per group."""
    e = engine_factory()
    a, b = 0x1234567 + n, 0x89ABCDE
    pts = _rows(g2.synthetic_points(n, a, b))
    rng = np.random.default_rng(n)
    index = rng.integers(0, n, size=n, dtype=np.uint32)
    cuts = np.sort(rng.integers(0, n + 1, size=groups - 1))
    offsets = np.concatenate([[0], cuts, [n]]).astype(np.uint32)
    got = e.g2_sum(pts, offsets, index=index)
    for g in range(groups):
        idx = index[offsets[g]:offsets[g + 1]].astype(object)
        k = (len(idx) * a + int(idx.sum()) * b) % g1.R_ORDER if len(idx) else 0
        assert got[g].tobytes() == g2.to_bytes192(g2.mul(k, g2.G2)), g


def test_g2_single_wide_group(engine_factory):
    """One group spanning many workgroups (n_tasks > 128): per-workgroup partials summed in k_g2_finish."""
    e = engine_factory()
    n, a, b = 20000, 3, 5
    pts = _rows(g2.synthetic_points(n, a, b))
    got = e.g2_sum(pts, [0, n])[0].tobytes()
    assert got == g2.to_bytes192(g2.mul((n * a + b * (n * (n - 1) // 2)) % g1.R_ORDER, g2.G2))


def test_g2_rejects_bad_arguments(engine_factory):
    from pos_evolution_amd import EngineError
    e = engine_factory()
    pts = _rows([g2.G2])
    with pytest.raises(EngineError):
        e.g2_sum(pts, [0, 2])                      # offsets exceed the points
    with pytest.raises(EngineError):
        e.g2_sum(pts, [0, 1], index=[3])           # index out of range


def test_aggregate_with_g2_signatures(engine_factory):
    """pe_aggregate's grouping + pe_g2_sum: Attestation.signature aggregated as real G2 points (pe:717, pe:1536)."""
    import pos_evolution_amd.synth as synth
    from tests import helpers as H
    e = engine_factory()
    n_val, spe = 4000, 32
    tree = synth.random_tree(40, 9, "branchy")
    H.load_tree(e, tree)
    e.set_validators(synth.balances(n_val, 9), synth.validator_flags(n_val, 9))
    comm = synth.random_committees(n_val, 64, 9)
    e.set_committees(1, comm.offsets, comm.members)
    atts, arena, bit_rows = synth.epoch_attestations(comm, tree, 1, spe, seed=9, density=0.9, parts=5)
    perm = np.random.default_rng(9).permutation(len(atts))
    atts, bit_rows = atts[perm], [bit_rows[i] for i in perm]
    arena, offs, nb = synth.pack_bit_rows(bit_rows)
    atts["bits_offset"], atts["n_bits"] = offs, nb
    a, b = 77, 1001
    table = g2.synthetic_points(len(atts), a, b)             # signature i = (a + i*b) * G2
    sigs = _rows(table)
    res = e.aggregate(packed=(atts, arena), sig_points192=sigs)
    assert res["n_groups"] == 64 and res["sig192"].shape == (64, 192)
    for g in range(64):
        members = np.nonzero(res["group_of"] == g)[0]
        k = (len(members) * a + int(members.sum()) * b) % g1.R_ORDER
        assert res["sig192"][g].tobytes() == g2.to_bytes192(g2.mul(k, g2.G2)), g


def test_g2_decompress_vs_oracle(engine_factory):
    """96-byte compressed signatures -> affine on the GPU (Fp2 square root + sign) against oracle/g2.py."""
    e = engine_factory()
    pts = g2.synthetic_points(1500, 0x1234567, 0x89ABCDE)
    comp = np.frombuffer(b"".join(g2.compress(p) for p in pts), dtype=np.uint8).reshape(-1, 96).copy()
    want = _rows(pts)
    assert np.array_equal(e.g2_compress(want), comp)
    out, status = e.g2_decompress(comp)
    assert not status.any() and np.array_equal(out, want)
    gen = g2.compress(g2.G2)
    assert gen.hex().startswith("93e02b6052719f60")
    # an x whose right-hand side is not a square in Fp2
    x0 = 1
    while g2.f2_sqrt(g2.f2_add(g2.f2_mul(g2.f2_sqr((x0, 0)), (x0, 0)), g2.B2)) is not None:
        x0 += 1
    off = bytearray(bytes(48) + x0.to_bytes(48, "big"))
    off[0] |= 0x80
    special = [gen, bytes([gen[0] ^ 0x20]) + gen[1:], g2.compress(None), bytes(96), bytes([0xE0]) + bytes(95),
               bytes([0x9F]) + b"\xff" * 95, bytes(off)]
    out, status = e.g2_decompress(np.frombuffer(b"".join(special), dtype=np.uint8).reshape(-1, 96))
    assert list(status) == [0, 0, 0, 1, 1, 1, 2]
    assert out[0].tobytes() == g2.to_bytes192(g2.G2) and out[1].tobytes() == g2.to_bytes192(g2.neg(g2.G2))
    assert out[2][0] == 0x40 and not out[2][1:].any() and not out[3:].any()
    # wire-to-wire: compressed signatures in, compressed aggregate out
    agg = e.g2_compress(e.g2_sum(e.g2_decompress(comp[:64])[0], [0, 64]))[0].tobytes()
    assert agg == g2.compress(g2.sum_points(pts[:64]))
