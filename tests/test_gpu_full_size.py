"""BASELINE.json's full sizes through size-independent properties (the oracle is too slow or too memory-hungry to be
the checker everywhere at 1 M validators; where it is fast enough -- get_head -- it still is):
linearity of the G1 sums, closed-form committee sums, idempotence of the union / LMD / flag updates, weight
conservation in the block tree, wire-format round trips."""
import numpy as np
import pytest

import pos_evolution_amd.synth as synth
from oracle import cport
from tests import helpers as H

pytestmark = pytest.mark.gpu
NONE32 = 0xFFFFFFFF
V, C_COMM, SPE, B = 1 << 20, 2048, 32, 4096


@pytest.fixture(scope="module")
def world(engine_factory):
    e = engine_factory(max_committee_tables=4)
    tree = synth.random_tree(B, 4, "bushy")
    H.load_tree(e, tree)
    bal = synth.balances(V, 4, mixed=True)
    flags = synth.validator_flags(V, 4, inactive_frac=0.005)
    pts = synth.registry_points(e, V)
    e.set_validators(bal, flags, pts)
    E = int(tree.slot.max()) // SPE + 1
    comm = synth.random_committees(V, C_COMM, 4)
    e.set_committees(E, comm.offsets, comm.members)
    e.on_tick((E + 1) * SPE * 12)
    return dict(e=e, tree=tree, bal=bal, flags=flags, pts=pts, E=E, comm=comm)


def _epoch_rows(w, parts, seed, density=0.9):
    atts, arena, bit_rows = synth.epoch_attestations(w["comm"], w["tree"], w["E"], SPE, seed=seed, density=density,
                                                     parts=parts, source=(0, w["tree"].roots[0].tobytes()), vote_recent=64)
    return atts, arena, bit_rows


def test_aggregate_linearity_and_closed_form_at_1m(world):
    """sum over (A | B) = sum over A + sum over B for disjoint partial aggregates, at 1 M validators / 2048 committees;
    three committees against the closed form of the synthetic registry."""
    e = world["e"]
    atts, arena, bit_rows = _epoch_rows(world, parts=2, seed=11)
    whole = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    assert whole["n_groups"] == C_COMM and int(whole["count"].sum()) == sum(int(np.sum(b)) for b in bit_rows)
    # each partial aggregate on its own (one group per row), then P_A + P_B through the engine's own adder
    halves = []
    for k in range(2):
        sel = np.arange(k, len(atts), 2)
        sub_arena, offs, nb = synth.pack_bit_rows([bit_rows[i] for i in sel])
        sub = atts[sel].copy()
        sub["bits_offset"], sub["n_bits"] = offs, nb
        r = e.aggregate(packed=(sub, sub_arena), want_aggregate_pubkeys=True)
        assert r["n_groups"] == C_COMM
        halves.append(r)
    ra, rb = halves
    # groups come out in first-appearance order = committee order, in both half calls and in the whole call
    pair = np.concatenate([ra["aggpk96"], rb["aggpk96"]])
    idx = np.stack([np.arange(C_COMM), np.arange(C_COMM) + C_COMM], axis=1).reshape(-1).astype(np.uint32)
    summed = e.g1_sum(np.arange(0, 2 * C_COMM + 1, 2), index=idx, points96=pair)
    assert np.array_equal(summed, whole["aggpk96"])
    cps = C_COMM // SPE
    for g in (0, 777, C_COMM - 1):
        row = whole["atts"][g]
        c = int((row["slot"] % SPE) * cps + row["index"])
        mem = world["comm"].members[world["comm"].offsets[c]:world["comm"].offsets[c + 1]]
        assert whole["aggpk96"][g].tobytes() == synth.registry_closed_form(mem[whole["bits"][g]])


def test_union_and_lmd_and_flags_are_idempotent_at_1m(world):
    from pos_evolution_amd._abi import pe_state_ctx
    e, tree, E = world["e"], world["tree"], world["E"]
    atts, arena, _ = _epoch_rows(world, parts=4, seed=12)
    once = e.aggregate(packed=(atts, arena))
    twice = e.aggregate(packed=(np.concatenate([atts, atts]), np.concatenate([arena, arena])))    # same bits offsets: reused
    assert twice["n_groups"] == once["n_groups"]
    assert np.array_equal(twice["count"], once["count"])
    assert all(np.array_equal(a, b) for a, b in zip(once["bits"], twice["bits"]))
    rows, out_arena = once["atts"], once["out_arena"]
    st1, _, cnt1 = e.on_attestation_batch(packed=(rows, out_arena))
    assert not st1.any()
    ep1, blk1 = e.latest_messages()
    st2, _, cnt2 = e.on_attestation_batch(packed=(rows, out_arena))       # same epoch again: strictly-later rule
    ep2, blk2 = e.latest_messages()
    assert not st2.any() and np.array_equal(ep1, ep2) and np.array_equal(blk1, blk2) and np.array_equal(cnt1, cnt2)
    voted = blk1 != NONE32
    assert voted.sum() == int(cnt1.sum())                                  # committees partition the registry
    ctx = pe_state_ctx()
    ctx.slot = (E + 1) * SPE
    ctx.chain_tip_root[:] = tree.roots[B - 1].tobytes()
    ctx.current_justified_root[:] = tree.roots[0].tobytes()
    ctx.previous_justified_root[:] = tree.roots[0].tobytes()
    ctx.base_reward_per_increment = 2264
    e.participation_rotate()
    e.participation_rotate()
    s1, num1 = e.process_attestation_batch(ctx, packed=(rows, out_arena))
    p1 = e.participation_get(1).copy()
    s2, num2 = e.process_attestation_batch(ctx, packed=(rows, out_arena))
    assert not s1.any() and not s2.any()
    assert np.array_equal(e.participation_get(1), p1) and not num2.any() and num1.any()   # flags already set: no reward


def test_weight_conservation_and_oracle_head_at_1m(world):
    """After the batch above: the root's weight = balance of every counted voter; a parent weighs at least as much as
    each child; the head and all 4096 weights equal the C oracle's (10 ms on the CPU: affordable even here)."""
    e, tree, bal, flags = world["e"], world["tree"], world["bal"], world["flags"]
    _, blk = e.latest_messages()
    w = e.get_weights()
    counted = (blk != NONE32) & ((flags & 1) != 0) & ((flags & 4) == 0)
    assert int(w[0]) == int(bal[counted].astype(object).sum())
    par = tree.parent
    assert all(int(w[par[i]]) >= int(w[i]) for i in range(1, B))
    head_o, w_o = cport.get_head(par.copy(), np.ones(B, np.uint8), tree.roots, blk, bal, flags, 0, NONE32)
    assert np.array_equal(w, w_o) and e.get_head() == tree.roots[head_o].tobytes()


def test_pubkey_wire_format_round_trip_at_1m(world):
    """compress -> GPU decompress is the identity on the whole 1 M registry (one square root per key)."""
    e, pts = world["e"], world["pts"]
    comp = e.g1_compress(pts)
    out, status = e.g1_decompress(comp)
    assert not status.any() and np.array_equal(out, pts)
