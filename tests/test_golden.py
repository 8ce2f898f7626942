"""Committed fixtures (tests/golden/, produced by tests/golden/generate.py from the L0 oracle).
CPU: the oracles still reproduce them.  -m gpu: the engine reproduces them without running L0 at test time."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    return json.load(open(os.path.join(HERE, "golden", name)))


def test_g1_vectors_oracles():
    from oracle import cport, g1
    v = _load("g1_vectors.json")
    assert g1.compress(g1.G).hex() == v["generator_compressed"]
    G96 = g1.to_bytes96(g1.G)
    for row in v["scalar_mul"]:
        k = int(row["k"], 16)
        assert g1.compress(g1.mul(k, g1.G)).hex() == row["compressed"]
        assert cport.g1_scalar_mul(k, G96).hex() == row["uncompressed"]
    for row in v["subset_sums"]:
        a, b = int(row["a"], 16), int(row["b"], 16)
        n = max(row["indices"], default=0) + 1
        pts = cport.g1_arith_progression(g1.to_bytes96(g1.mul(a, g1.G)), g1.to_bytes96(g1.mul(b, g1.G)), n)
        got = cport.g1_sum_groups(pts, np.array(row["indices"], dtype=np.uint32),
                                  np.array([0, len(row["indices"])], dtype=np.uint32))
        assert got[0].tobytes().hex() == row["sum_uncompressed"]


def test_shuffle_vectors_oracle():
    from oracle import spec
    for row in _load("shuffle_vectors.json"):
        spec.use_preset(row["preset"])
        assert spec.SHUFFLE_ROUND_COUNT == row["rounds"]
        seed = bytes.fromhex(row["seed"])
        assert [spec.compute_shuffled_index(i, row["index_count"], seed) for i in range(row["index_count"])] == row["shuffled"]
        assert sorted(row["shuffled"]) == list(range(row["index_count"]))
    spec.use_preset("mainnet")


@pytest.mark.gpu
def test_g1_vectors_engine(engine_factory):
    import pos_evolution_amd.synth as synth
    e = engine_factory()
    v = _load("g1_vectors.json")
    for row in v["subset_sums"]:
        a, b = int(row["a"], 16), int(row["b"], 16)
        n = max(row["indices"], default=0) + 1
        pts = synth.registry_points(e, n, a, b)
        got = e.g1_sum([0, len(row["indices"])], index=np.array(row["indices"], dtype=np.uint32), points96=pts)
        assert got[0].tobytes().hex() == row["sum_uncompressed"]


@pytest.mark.gpu
def test_forkchoice_trace_engine(engine_factory):
    """Replays the committed event stream through the raw engine API and checks head / latest-message digest /
    boost root after every event against the values the L0 oracle produced when the fixture was generated."""
    import hashlib
    import pos_evolution_amd as pea
    tr = _load("forkchoice_trace.json")
    assert tr["preset"] == "minimal"
    e = engine_factory(slots_per_epoch=8, seconds_per_slot=6, intervals_per_slot=3, safe_slots_to_update_justified=2,
                       max_committee_tables=16)
    anchor = bytes.fromhex(tr["anchor_root"])
    e.store_init(0, 0, anchor)
    n = tr["n_validators"]
    e.set_validators(np.full(n, 32 * 10**9, dtype=np.uint64), np.full(n, 1, dtype=np.uint8))
    for ep, comms in tr["committees_epoch"].items():
        sizes = [len(c) for c in comms]
        e.set_committees(int(ep), np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32),
                         np.concatenate([np.asarray(c, dtype=np.uint32) for c in comms]) if sum(sizes) else [])

    def digest():
        ep, blk = e.latest_messages()
        h = hashlib.sha256()
        cnt = 0
        for v in range(n):
            if blk[v] != 0xFFFFFFFF:
                h.update(v.to_bytes(8, "little") + int(ep[v]).to_bytes(8, "little") + e.block_root_at(int(blk[v])))
                cnt += 1
        return h.hexdigest(), cnt

    for ev in tr["events"]:
        if ev["op"] == "tick":
            e.on_tick(ev["time"])
        elif ev["op"] == "block":
            e.on_block(bytes.fromhex(ev["root"]), bytes.fromhex(ev["parent"]), ev["slot"], (0, bytes(32)), (0, bytes(32)))
        else:
            row = pea.AttRow(ev["slot"], ev["index"], bytes.fromhex(ev["beacon_block_root"]), 0, bytes(32),
                             ev["target_epoch"], bytes.fromhex(ev["target_root"]), np.array(ev["bits"], dtype=np.uint8))
            status, _, _ = e.on_attestation_batch([row])
            assert (status[0] == 0) == ev["accepted"]
        x = ev["expect"]
        assert e.get_head().hex() == x["head"]
        d, cnt = digest()
        assert cnt == x["n_messages"] and d == x["lm_digest"]
        assert e.store_scalars()["proposer_boost_root"].hex() == x["boost"]
