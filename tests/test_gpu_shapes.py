"""-m gpu: the whole step (aggregate -> on_attestation -> get_head -> process_attestation) at the BASELINE config
shapes round 1 left without a G1 parity check, and the workload variations SURVEY.md 8(d) asks for (bit density
50 % / 100 %, proposer boost on a leaf), against the C oracle -- through the same helpers bench.py's own
cross-check uses, so the bench's "checked_against_oracle" and these tests cannot drift apart."""
import hashlib

import numpy as np
import pytest

import bench
import pos_evolution_amd as pea
import pos_evolution_amd.synth as synth
from oracle import cport

pytestmark = pytest.mark.gpu


def _workload(e, n_val, n_comm, n_blocks, seed, density, parts, mixed=False, kind="bushy", shuffle=True):
    spe = 32
    tree = synth.random_tree(n_blocks, seed, kind)
    e.store_init(0, 0, tree.roots[0].tobytes())
    for i in range(1, n_blocks):
        e.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
    bal = synth.balances(n_val, seed, mixed=mixed)
    flags = synth.validator_flags(n_val, seed, inactive_frac=0.005)
    pts = synth.registry_points(e, n_val)
    e.set_validators(bal, flags, pts)
    ep = int(tree.slot.max()) // spe + 1
    if shuffle:   # the reference's swap-or-not shuffle on the GPU (pe:495-534), as the bench builds its tables
        s = hashlib.sha256(b"shape" + seed.to_bytes(4, "little")).digest()
        off, mem = e.compute_committees(ep, s, np.arange(n_val, dtype=np.uint32), n_comm, 90)
        comm = synth.Committees(off, mem)
    else:
        comm = synth.random_committees(n_val, n_comm, seed)
        e.set_committees(ep, comm.offsets, comm.members)
    atts, arena, _ = synth.epoch_attestations(comm, tree, ep, spe, seed=seed, density=density, parts=parts,
                                              source=(0, tree.roots[0].tobytes()), vote_recent=64)
    w = dict(tree=tree, bal=bal, flags=flags, pts=pts, spe=spe)
    st = dict(epoch=ep, comm=comm, atts=atts, arena=arena)
    st["ctx"] = bench.state_ctx(w, ep)
    w["steps"] = [st]
    return w, st


def _check_step(e, w, st, boost_root=None, pipelined=True):
    inp = bench.cpu_step_inputs(w, st)
    V = w["bal"].size
    vote_epoch, vote_block = np.zeros(V, dtype=np.uint64), np.full(V, 0xFFFFFFFF, dtype=np.uint32)
    want = bench.cpu_step(w, st, inp, False, vote_epoch, vote_block)
    if boost_root is not None:   # the oracle's head / weights with the boost on that block
        bi = next(i for i in range(w["tree"].roots.shape[0]) if w["tree"].roots[i].tobytes() == boost_root)
        h, wts = cport.get_head(w["tree"].parent, np.ones(w["tree"].parent.size, np.uint8), w["tree"].roots,
                                want["vote_block"], w["bal"], w["flags"], 0, boost_idx=bi)
        want["head"], want["weights"] = w["tree"].roots[h].tobytes(), wts
    r = bench.run_step_single(e, w, st, pipelined=pipelined, lagged=False)
    rows = r["rows"]
    C = st["comm"].offsets.size - 1
    pos = ((rows["slot"] % 32) * (C // 32) + rows["index"]).astype(np.int64)
    inv = np.argsort(pos)
    assert np.array_equal(pos[inv], np.arange(C))
    agg = r["agg"]
    assert np.array_equal(agg["count"][inv], want["count"])
    assert np.array_equal(np.concatenate([np.packbits(agg["bits"][g], bitorder="little") for g in inv]), want["union"])
    assert np.array_equal(agg["aggpk96"][inv], want["aggpk"]), "aggregate pubkeys"
    assert (r["status"] == 0).all() and (r["pstatus"] == 0).all()
    assert np.array_equal(e.latest_messages()[1], want["vote_block"])
    assert r["head"] == want["head"]
    assert np.array_equal(e.get_weights(), want["weights"])
    assert np.array_equal(r["numerators"][inv], want["numerators"])
    assert np.array_equal(e.participation_get(0), want["part_cur"])
    assert np.array_equal(e.participation_get(1), want["part_prev"])
    return r, want, inv


def test_config2_shape_65536_validators_64_committees_per_slot(engine_factory):
    """BASELINE configs[1]: 65 536 validators, 64 committees per slot (2048 committees of 32), 2048-block tree."""
    e = engine_factory()
    w, st = _workload(e, 65536, 2048, 2048, seed=2, density=0.99, parts=4, kind="branchy")
    _check_step(e, w, st)


@pytest.mark.parametrize("density,parts", [(0.5, 4), (1.0, 1), (1.0, 4)])
def test_bit_density_variations(engine_factory, density, parts):
    """SURVEY.md 8(d) "also vary bit density {50 %, 99 %, 100 %}": 262 144 validators, 2048 committees of 128."""
    e = engine_factory()
    w, st = _workload(e, 262144, 2048, 512, seed=3, density=density, parts=parts)
    r, want, _ = _check_step(e, w, st)
    if density == 1.0:
        assert int(want["count"].sum()) == 262144


def test_proposer_boost_on_a_leaf(engine_factory):
    """SURVEY.md 8(d) "boost {unset, set on a leaf}": the boost lands on a leaf (every ancestor gains the proposer
    score, A.1); head and all weights against the oracle with the same boost."""
    e = engine_factory()
    w, st = _workload(e, 65536, 2048, 300, seed=5, density=0.9, parts=2)
    tree = w["tree"]
    is_parent = np.zeros(tree.parent.size, dtype=bool)
    is_parent[tree.parent[1:]] = True
    leaf = int(np.nonzero(~is_parent)[0][-3])
    e.on_tick((st["epoch"] + 1) * 32 * 12)
    e.set_proposer_boost(tree.roots[leaf].tobytes())
    # run_step_single ticks to the same time again: no new slot, the boost stays (pe:943-944)
    r, want, _ = _check_step(e, w, st, boost_root=tree.roots[leaf].tobytes())
    plain, _ = cport.get_head(tree.parent, np.ones(tree.parent.size, np.uint8), tree.roots, want["vote_block"],
                              w["bal"], w["flags"], 0)
    assert np.array_equal(e.get_weights(), want["weights"])
    assert want["weights"][leaf] > 0 and want["weights"][0] > 0


def test_config5_shape_aggregate_2048_committees_of_2048_over_4m_registry(engine_factory):
    """BASELINE configs[4]: 4 194 304 validators, EIP-7251-style mixed balances, 2048 committees of 2048 (the
    MAX_VALIDATORS_PER_COMMITTEE of pe:715), 8192-block tree.  pe_aggregate's 2048 sums of ~2027 points each: EVERY
    group against the closed form, the whole step (union, LMD, head, weights, flags, numerators and all 2048 aggregate
    pubkeys) against the C oracle."""
    e = engine_factory()
    w, st = _workload(e, 1 << 22, 2048, 8192, seed=4, density=0.99, parts=4, mixed=True)
    r, want, inv = _check_step(e, w, st)
    agg, comm = r["agg"], st["comm"]
    rows = agg["atts"]
    pos = ((rows["slot"] % 32) * 64 + rows["index"]).astype(np.int64)
    for g in range(agg["n_groups"]):
        mem = comm.members[comm.offsets[pos[g]]:comm.offsets[pos[g] + 1]]
        assert agg["aggpk96"][g].tobytes() == synth.registry_closed_form(mem[agg["bits"][g]]), g


@pytest.mark.parametrize("shape,lag,n_epochs,boost,tree_kind", [
    ("configs1", 2, 3, False, None), ("configs1", 4, 5, False, None), ("configs1", 4, 5, True, None),
    ("configs2", 2, 3, False, None), ("configs2", 4, 3, True, None),
    ("configs2", 4, 3, False, "chain"), ("configs2", 2, 3, True, "chain"),   # SURVEY 8(d) c3's worst case: ONE chain, 4096 deep
    ("configs3", 2, 3, False, None), ("configs3", 4, 3, False, None), ("configs3", 4, 3, True, None),
    ("configs4", 4, 2, False, None)])
def test_the_timed_path_against_the_oracle(shape, lag, n_epochs, boost, tree_kind):
    """The path bench.py TIMES -- attestation rows and bits resident in HBM (PE_ROWS_RESIDENT: grouped, resolved and validated
    on the device), streaming pipelines whose outputs complete `lag` steps later, pe_get_head_async -- held against the C
    oracle DIRECTLY, on bench.build_workload's own workload at the BASELINE configs[1..4] shapes (configs[2] with its
    4096-block tree, configs[4] with mixed balances and 8192 blocks): consecutive epochs, every step's union bits, counts,
    aggregate pubkeys, statuses, head and reward numerators, and the store behind the last one (latest messages, all
    weights, both participation arrays).  (until now only bench.py's own step-0 check compared this path with
    the oracle; the -m gpu tests held it against the host-row path.)  Round 5: configs[1] on SURVEY 8(d)'s 2048-block
    branchy chain, configs[4] with its 1 % equivocating validators (pe:1438 and A.1's mask inside the timed path), and
    `boost`: proposer_boost_root set on a leaf behind every step's on_tick (pe:1020-1024; the oracle's head and weights
    carry the same boost).  Round 6: configs[2] also on the 4096-deep single chain SURVEY 8(d) names as c3's worst case (the
    paired lean tree k_pair_union_tree<512, 8, true> walks it in the streaming steps), and the boost at configs[3]."""
    import types

    cfg = bench.SHAPES[shape]
    args = types.SimpleNamespace(validators_local=cfg["validators"], blocks=cfg["blocks"], committees=cfg["committees"], parts=4,
                                 mixed_balances=cfg["mixed_balances"], host_arena=False, host_rows=False, with_shuffle=False,
                                 by_committee=False, world=1, shuffle_variant_from=n_epochs,
                                 tree_kind=tree_kind or cfg["tree_kind"], equivocating_frac=cfg["equivocating_frac"], boost=boost)
    e = pea.Engine(max_committee_tables=n_epochs + 3)
    w = bench.build_workload(e, args, 0, n_epochs)
    if tree_kind == "chain":
        assert int((w["tree"].parent[1:] == np.arange(cfg["blocks"] - 1)).sum()) == cfg["blocks"] - 1, "one chain, every block on it"
    assert all("rows_in" in st and "arena_in" in st for st in w["steps"])
    assert (w["equivocating"] is not None and len(w["equivocating"]) == cfg["validators"] // 100) == (shape == "configs4")
    assert all(("boost_idx" in st) == boost for st in w["steps"])
    e.set_pipeline_lag(lag)
    e.reuse_outputs(max(n_epochs, lag) + 2)   # the output ring must be deeper than the lag
    got = [bench.run_step_single(e, w, st, pipelined=True, lagged=True, sync_head=False) for st in w["steps"]]
    e.drain()
    V = w["bal"].size
    C = cfg["committees"]
    vote_epoch, vote_block = np.zeros(V, dtype=np.uint64), np.full(V, 0xFFFFFFFF, dtype=np.uint32)
    for k, (st, r) in enumerate(zip(w["steps"], got)):
        want = bench.cpu_step(w, st, bench.cpu_step_inputs(w, st), True, vote_epoch, vote_block)
        agg = r["agg"]
        assert int(agg["n_groups"]) == C
        rows = agg["atts"]
        pos = ((rows["slot"] % 32) * (C // 32) + rows["index"]).astype(np.int64)
        inv = np.argsort(pos)
        assert np.array_equal(pos[inv], np.arange(C)), k
        assert np.array_equal(np.asarray(agg["count"])[inv], want["count"]), k
        assert np.array_equal(np.asarray(r["count"])[:C][inv], want["count"]), k
        assert np.array_equal(np.concatenate([np.packbits(agg["bits"][g], bitorder="little") for g in inv]), want["union"]), k
        assert np.array_equal(np.asarray(agg["aggpk96"])[inv], want["aggpk"]), (k, "aggregate pubkeys")
        assert (np.asarray(r["status"])[:C] == 0).all() and (np.asarray(r["pstatus"])[:C] == 0).all(), k
        assert bytes(r["head"]) == want["head"], k
        assert np.array_equal(np.asarray(r["numerators"])[:C][inv], want["numerators"]), k
        last = want
    assert np.array_equal(e.latest_messages()[1], last["vote_block"])
    assert np.array_equal(e.get_weights(), last["weights"])
    # the state's slot is the first of the NEXT epoch (bench.state_ctx): the step's attestations are previous-epoch ones, their
    # flags land in the previous-epoch array on top of what the rotation left there (nothing: the current-epoch array stays empty)
    assert np.array_equal(e.participation_get(0), last["part_cur"]) and not last["part_cur"].any()
    assert np.array_equal(e.participation_get(1), last["part_prev"]) and last["part_prev"].any()
    e.close()


def test_the_signed_steps_of_the_bench():
    """bench.py's `with_signatures` leg at the configs[1] shape: streaming steps whose aggregate is pe_aggregate_signed over
    signatures resident in HBM -- its own checks (every step == its synchronous host-row replay, signatures and statuses
    included; sampled aggregate signatures == the oracle's closed form) are assertions inside."""
    import types

    cfg = bench.SHAPES["configs1"]
    args = types.SimpleNamespace(validators_local=cfg["validators"], blocks=cfg["blocks"], committees=cfg["committees"], parts=4,
                                 mixed_balances=False, host_arena=False, host_rows=False, with_shuffle=False,
                                 by_committee=False, world=1, shuffle_variant_from=6)
    e = pea.Engine(max_committee_tables=8)
    w = bench.build_workload(e, args, 0, 6)
    e.close()
    out = bench.signed_steps(pea, w, 0, 2, 4, 4)
    assert out["steps_verified"] == 4 and out["signatures_per_step"] == 8192
    assert out["aggregate_signatures_checked_against_oracle"] == 32 and out["ms_per_step_with_signatures"] > 0
