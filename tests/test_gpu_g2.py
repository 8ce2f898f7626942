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


# ---------------------------------------------------------------- the signature leg of pe_aggregate (pe_aggregate_signed)
R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def _off_subgroup_point():
    """A point of E'(Fp2) outside G2: decode small x values until one lies on the curve (no cofactor clearing)."""
    x0 = 1
    while True:
        b = bytearray((0).to_bytes(48, "big") + x0.to_bytes(48, "big"))
        b[0] |= 0x80
        try:
            p = g2.decompress(bytes(b))
        except ValueError:
            x0 += 1
            continue
        if g2.mul(R_ORDER, p) is not None:
            return p
        x0 += 1


def _off_curve_x():
    x0 = 1
    while True:
        b = bytearray((0).to_bytes(48, "big") + x0.to_bytes(48, "big"))
        b[0] |= 0x80
        try:
            g2.decompress(bytes(b))
        except ValueError:
            return bytes(b)
        x0 += 1


@pytest.mark.parametrize("rows_mode", ["host", "device"])
@pytest.mark.parametrize("compressed", [True, False])
def test_aggregate_signed_vs_oracle(engine_factory, rows_mode, compressed):
    """bls.Aggregate as a leg of pe_aggregate through the C ABI: per group the sum of its members' BLSSignatures in the
    compressed wire form, against oracle/g2.py's closed form; undecodable / off-curve / off-subgroup members are
    reported per row, left out of the sum, and cost the group its PE_ATT_FLAG_SIGNATURE_VALID."""
    import pos_evolution_amd as pea
    from pos_evolution_amd import _abi
    from tests.test_gpu_pipeline import _world
    from tests.test_gpu_resident_rows import _dev_arena, _dev_rows

    w = _world(engine_factory, 6000, 64, seed=3, density=0.8, parts=3)
    e, atts, arena = w["e"], w["atts"], w["arena"]
    n = len(atts)
    a, b = 0xABCDEF12345, 0x1357
    pts = g2.synthetic_points(n, a, b)          # signature of row i = (a + i * b) * G2
    off_sub = _off_subgroup_point()
    bad = {5: "malformed", 9: "off_curve", 11: "off_subgroup", 40: "infinity"} if compressed else {11: "off_subgroup", 40: "infinity"}
    wire = []
    for i, p in enumerate(pts):
        kind = bad.get(i)
        if kind == "off_subgroup":
            p = off_sub
        elif kind == "infinity":
            p = None
        enc = bytearray(g2.compress(p) if compressed else g2.to_bytes192(p))
        if kind == "malformed":
            enc[0] &= 0x7F                       # the compression bit cleared
        elif kind == "off_curve":
            enc = bytearray(_off_curve_x())
        wire.append(bytes(enc))
    sigs = np.frombuffer(b"".join(wire), dtype=np.uint8).reshape(n, -1)
    packed = (_dev_rows(atts), _dev_arena(arena)) if rows_mode == "device" else (atts, arena)
    ref = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    res = e.aggregate_signed(sigs, packed=packed, compressed=compressed, check_subgroup=True, want_aggregate_pubkeys=True)
    g = res["n_groups"]
    assert g == ref["n_groups"] and np.array_equal(res["group_of"][:n], ref["group_of"])
    assert np.array_equal(res["aggpk96"], ref["aggpk96"]) and np.array_equal(res["count"], ref["count"])
    want_st = np.zeros(n, dtype=np.int32)
    for i, kind in bad.items():
        want_st[i] = {"malformed": 1, "off_curve": 2, "off_subgroup": 3, "infinity": 0}[kind]
    assert np.array_equal(res["sig_status"], want_st)
    gof = np.asarray(ref["group_of"])
    for k in range(g):
        members = np.nonzero(gof == k)[0]
        good = [int(i) for i in members if want_st[i] == 0 and bad.get(int(i)) != "infinity"]
        exp = g2.mul((a * len(good) + b * sum(good)) % R_ORDER, g2.G2) if good else None
        assert res["sig96c"][k].tobytes() == g2.compress(exp), f"group {k}"
        valid = bool(res["atts"][k]["flags"] & _abi.PE_ATT_FLAG_SIGNATURE_VALID)
        was_valid = bool(ref["atts"][k]["flags"] & _abi.PE_ATT_FLAG_SIGNATURE_VALID)
        assert valid == (was_valid and all(want_st[i] == 0 for i in members)), f"group {k}: signature-valid flag"
    # the same inside a pipeline: outputs complete at its end
    with e.pipeline():
        res2 = e.aggregate_signed(sigs, packed=packed, compressed=compressed, check_subgroup=True)
    assert np.array_equal(res2["sig96c"], res["sig96c"]) and np.array_equal(res2["sig_status"], want_st)
    # without the subgroup check the off-subgroup member is summed like any curve point
    res3 = e.aggregate_signed(sigs, packed=packed, compressed=compressed, check_subgroup=False)
    k = int(gof[11])
    members = np.nonzero(gof == k)[0]
    good = [int(i) for i in members if want_st[i] in (0, 3) and bad.get(int(i)) not in ("infinity", "off_subgroup")]
    exp = g2.add(g2.mul((a * len(good) + b * sum(good)) % R_ORDER, g2.G2), off_sub)
    assert res3["sig_status"][11] == 0 and res3["sig96c"][k].tobytes() == g2.compress(exp)


def test_g2_subgroup_check(engine_factory):
    e = engine_factory()
    good = g2.synthetic_points(9, 77, 5)
    off = _off_subgroup_point()
    pts = _rows(good[:4] + [off, None] + good[4:] + [g2.double(off)])
    st = e.g2_subgroup_check(pts)
    want = [0, 0, 0, 0, 3, 0, 0, 0, 0, 0, 0, 3]
    assert st.tolist() == want


# ---------------------------------------------------------------- pe_aggregate_signatures: an epoch's unaggregated signatures
@pytest.mark.parametrize("where", ["host", "device"])
def test_aggregate_signatures_vs_oracle(engine_factory, where):
    """bls.Aggregate per committee over compressed signatures (pe:717 one BLSSignature per attester; pe:659 / pe:1536 summed
    per committee): ragged groups over an index list, an empty group, members that do not decode (malformed, off the curve,
    off the subgroup with the check on) left out of their sums and counted, an infinity member -- against oracle/g2.py."""
    import torch

    import pos_evolution_amd as pea

    e = engine_factory()
    n = 700
    pts = g2.synthetic_points(n, 0x5151, 0x77)
    comp = [g2.compress(p) for p in pts]
    off_sub = _off_subgroup_point()
    comp[5] = bytes(96)                         # malformed (no compression flag)
    comp[6] = _off_curve_x()                    # not on the curve
    comp[7] = g2.compress(off_sub)              # on the curve, outside G2
    comp[8] = g2.compress(None)                 # the identity: a valid encoding, adds nothing
    sig = np.frombuffer(b"".join(comp), dtype=np.uint8).reshape(-1, 96).copy()
    rng = np.random.default_rng(3)
    index = rng.permutation(n).astype(np.uint32)
    cuts = np.sort(rng.choice(np.arange(1, n), size=9, replace=False))
    offsets = np.concatenate([[0], cuts[:4], [cuts[3]], cuts[4:], [n]]).astype(np.uint32)   # one empty group
    if where == "device":
        st = torch.from_numpy(sig.reshape(-1)).cuda()
        it = torch.from_numpy(index).cuda()
        sig_in = pea.DeviceArena(st.data_ptr(), st.numel(), keep=st)
        idx_in = pea.DeviceArena(it.data_ptr(), it.numel() * 4, keep=it)
    else:
        sig_in, idx_in = sig, index
    for check in (False, True):
        agg, status, bad = e.aggregate_signatures(sig_in, offsets, index=idx_in, check_subgroup=check)
        want_status = np.zeros(n, dtype=np.int32)
        want_status[5], want_status[6] = 1, 2
        if check:
            want_status[7] = 3
        assert np.array_equal(status, want_status)
        for g in range(offsets.size - 1):
            members = index[offsets[g]:offsets[g + 1]]
            good = [int(i) for i in members if want_status[i] == 0]
            total = g2.sum_points([off_sub if i == 7 else None if i == 8 else pts[i] for i in good])
            assert bytes(agg[g]) == g2.compress(total), (check, g)
            assert bad[g] == len(members) - len(good)
    # contiguous groups (index NULL) and the degenerate calls
    agg, status, bad = e.aggregate_signatures(sig[20:84], [0, 64])
    assert bytes(agg[0]) == g2.compress(g2.sum_points(pts[20:84])) and not status.any() and not bad.any()
    agg, _, _ = e.aggregate_signatures(np.zeros((0, 96), dtype=np.uint8), [0, 0])
    assert bytes(agg[0]) == g2.compress(None)
    with pytest.raises(AssertionError):
        e.aggregate_signatures(sig, [0, 10], index=np.array([n] * 10, dtype=np.uint32))
    if where == "device":   # the host cannot read a device-resident index: it is range-checked on the device
        bad_idx = index.copy()
        bad_idx[17] = n + 5
        bt = torch.from_numpy(bad_idx).cuda()
        with pytest.raises(AssertionError):
            e.aggregate_signatures(sig_in, offsets, index=pea.DeviceArena(bt.data_ptr(), bt.numel() * 4, keep=bt))
        agg2, status2, bad2 = e.aggregate_signatures(sig_in, offsets, index=idx_in)   # the handle is fine afterwards
        assert np.array_equal(agg2, e.aggregate_signatures(sig, offsets, index=index)[0]) and int(bad2.sum()) == 2


def test_aggregate_signatures_of_a_whole_epoch(engine_factory):
    """BASELINE configs[3]'s epoch at full size: 1 048 576 compressed signatures resident in HBM, 2048 committees of 512 by
    a random partition, every committee's aggregate against the closed form (validator v signs with (a + (v mod 16384) b) G2:
    sum = (|S| a + b sum(v mod 16384)) G2, computed by oracle/g2.py's own double-and-add) -- all 2048 of them."""
    import torch

    import pos_evolution_amd as pea
    import pos_evolution_amd.synth as synth

    e = engine_factory()
    V, C, period = 1 << 20, 2048, 16384
    a, b = 0xABCDEF12345, 0x1357
    base = synth.signature_points(e, period, a, b)
    sig = np.ascontiguousarray(np.tile(base, (V // period, 1)))
    st = torch.from_numpy(sig.reshape(-1)).cuda()
    rng = np.random.default_rng(20)
    perm = rng.permutation(V).astype(np.uint32)
    attests = rng.random(V) < 0.99
    lists = [perm[c * 512:(c + 1) * 512][attests[c * 512:(c + 1) * 512]] for c in range(C)]
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint32)
    index = np.concatenate(lists).astype(np.uint32)
    it = torch.from_numpy(index).cuda()
    agg, status, bad = e.aggregate_signatures(pea.DeviceArena(st.data_ptr(), st.numel(), keep=st), offsets,
                                              index=pea.DeviceArena(it.data_ptr(), it.numel() * 4, keep=it))
    assert not status.any() and not bad.any()
    for c in range(C):
        m = lists[c].astype(np.int64) % period
        want = g2.compress(g2.mul((len(m) * a + b * int(m.sum())) % R_ORDER, g2.G2))
        assert bytes(agg[c]) == want, c
