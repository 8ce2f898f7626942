"""-m gpu: edge cases of the domain -- empty and ragged inputs, maximum sizes, infinity points, equivocators,
capacity limits, and the two largest BASELINE.json shapes (config 3 deep chain is in test_gpu_engine; here config 5:
4 194 304 validators, EIP-7251-style mixed balances up to 2048 ETH, 8192-block tree)."""
import numpy as np
import pytest

import pos_evolution_amd as pea
import pos_evolution_amd.synth as synth
from oracle import cport, g1
from tests import helpers as H

pytestmark = pytest.mark.gpu
NONE32 = 0xFFFFFFFF


def test_empty_store_and_empty_batches(engine_factory):
    e = engine_factory()
    root = bytes(range(32))
    e.store_init(100, 0, root)
    assert e.get_head() == root                                   # no validators, one block (K10)
    assert list(e.get_weights()) == [0]
    e.set_validators(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint8))
    assert e.get_head() == root
    status, agg, count = e.on_attestation_batch([])
    assert len(status) == 0
    res = e.aggregate([])
    assert res["n_groups"] == 0
    assert e.g1_sum([0], points96=np.zeros((0, 96), dtype=np.uint8)).shape == (0, 96)
    assert e.ffg_balances() == (10**9, 10**9, 10**9)              # get_total_balance's max(increment, 0)


def test_handlers_require_store_and_reject_bad_arguments(engine_factory):
    e = engine_factory()
    with pytest.raises(pea.EngineError):
        e.get_head()                                              # store not initialised
    e.store_init(0, 0, bytes(32))
    with pytest.raises(pea.EngineError):
        e.add_block(bytes([1]) * 32, bytes([9]) * 32, 1)          # unknown parent
    e.add_block(bytes([1]) * 32, bytes(32), 5)
    with pytest.raises(pea.EngineError):
        e.add_block(bytes([2]) * 32, bytes([1]) * 32, 5)          # slot must exceed the parent's
    with pytest.raises(pea.EngineError):
        e.set_proposer_boost(bytes([7]) * 32)                     # unknown root
    e.set_validators(np.full(4, 32 * 10**9, dtype=np.uint64), np.ones(4, dtype=np.uint8))
    with pytest.raises(pea.EngineError):
        e.set_committees(0, [0, 2, 4], [0, 1, 2, 3])              # 2 committees: not a multiple of SLOTS_PER_EPOCH
    with pytest.raises(pea.EngineError):
        e.mark_equivocating([4])                                  # index out of range
    with pytest.raises(pea.EngineError):
        e.set_validators(np.full(4, 70000 * 10**9, dtype=np.uint64), np.ones(4, dtype=np.uint8))  # > 65535 increments


def test_block_table_capacity(engine_factory):
    e = engine_factory()
    tree = synth.random_tree(8192, 1, "bushy")
    H.load_tree(e, tree)
    assert e.num_blocks == 8192
    with pytest.raises(pea.EngineError) as err:
        e.add_block(bytes([0xEE]) * 32, tree.roots[5].tobytes(), int(tree.slot.max()) + 5)
    assert err.value.status == -10                                # PE_ERR_CAPACITY, store untouched
    assert e.num_blocks == 8192
    e.set_validators(np.full(1000, 32 * 10**9, dtype=np.uint64), np.ones(1000, dtype=np.uint8))
    head = e.get_head()
    h_o, _ = cport.get_head(tree.parent, np.ones(8192, np.uint8), tree.roots, np.full(1000, NONE32, np.uint32),
                            np.full(1000, 32 * 10**9, np.uint64), np.ones(1000, np.uint8), 0)
    assert head == tree.roots[h_o].tobytes()                      # pure tie-break descent over 8192 blocks


def test_ragged_committees_and_bit_lengths(engine_factory):
    """Committees of size 0, 1 and MAX_VALIDATORS_PER_COMMITTEE (2048); bit lists longer / shorter than the committee."""
    e = engine_factory()
    tree = synth.random_tree(8, 2, "chain")
    H.load_tree(e, tree)
    n_val = 2048 + 1 + 40
    pts, (a, b) = H.oracle_points(n_val)
    e.set_validators(synth.balances(n_val, 2), np.ones(n_val, dtype=np.uint8), pts)
    sizes = [2048, 0, 1] + [0] * 28 + [40]                        # 32 committees (1 per slot)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    members = np.random.default_rng(2).permutation(n_val).astype(np.uint32)
    epoch = 1
    e.set_committees(epoch, offsets, members)
    e.on_tick(3 * 32 * 12)
    blk = tree.roots[7].tobytes()
    tgt = tree.roots[synth.ancestor_at(tree, 7, epoch * 32)].tobytes()

    def row(slot_in_epoch, bits, **kw):
        return pea.AttRow(epoch * 32 + slot_in_epoch, 0, blk, 0, tree.roots[0].tobytes(), epoch, tgt,
                          np.asarray(bits, dtype=np.uint8), is_from_block=True, **kw)

    rows = [
        row(0, np.ones(2048)),                       # maximum committee, all bits
        row(1, np.zeros(0)),                         # empty committee: no indices -> invalid (A.7)
        row(2, [1]),                                 # singleton
        row(2, [1, 1, 1]),                           # longer than the committee: extra bits are never read (A.6)
        row(31, np.ones(39)),                        # shorter than the committee: bits[i] would raise -> rejected
        row(31, np.r_[np.zeros(39), 1]),             # only the last member
        row(5, np.zeros(0)),                         # another empty committee
    ]
    status, aggpk, count = e.on_attestation_batch(rows, want_aggregate_pubkeys=True)
    assert list(status) == [0, 11, 0, 0, 10, 0, 11]
    assert list(count) == [2048, 0, 1, 1, 0, 1, 0]
    m0 = members[:2048]
    assert aggpk[0].tobytes() == H.closed_form_sum(m0, a, b)
    assert aggpk[2].tobytes() == H.closed_form_sum([members[2048]], a, b) == aggpk[3].tobytes()
    assert aggpk[5].tobytes() == H.closed_form_sum([members[-1]], a, b)
    assert aggpk[1][0] == 0x40 and aggpk[4][0] == 0x40            # rejected rows: infinity
    ep, bi = e.latest_messages()
    voted = np.zeros(n_val, dtype=bool)
    voted[m0] = True
    voted[members[2048]] = True
    voted[members[-1]] = True
    assert np.array_equal(bi != NONE32, voted)


def test_aggregate_mixed_groups_and_overlaps(engine_factory):
    """Groups of different bit lengths, overlapping partial aggregates, an all-zero member, a lone attestation."""
    e = engine_factory()
    tree = synth.random_tree(4, 3, "chain")
    H.load_tree(e, tree)
    n_val = 96
    pts, (a, b) = H.oracle_points(n_val)
    e.set_validators(synth.balances(n_val, 3), np.ones(n_val, dtype=np.uint8), pts)
    comm = synth.random_committees(n_val, 32, 3)                  # 3 members each
    e.set_committees(0, comm.offsets, comm.members)
    r0 = tree.roots[0].tobytes()

    def row(slot, bits, blk=r0):
        return pea.AttRow(slot, 0, blk, 0, r0, 0, r0, np.asarray(bits, dtype=np.uint8))

    rows = [row(1, [1, 0, 0]), row(2, [0, 0, 0]), row(1, [1, 1, 0]), row(3, [1, 1, 1]), row(1, [0, 1, 0]),
            row(1, [0, 0, 1], blk=tree.roots[1].tobytes())]       # same committee, different data: own group
    res = e.aggregate(rows, want_aggregate_pubkeys=True)
    assert res["n_groups"] == 4 and list(res["group_of"]) == [0, 1, 0, 2, 0, 3]
    assert [list(b.astype(int)) for b in res["bits"]] == [[1, 1, 0], [0, 0, 0], [1, 1, 1], [0, 0, 1]]
    assert list(res["count"]) == [2, 0, 3, 1]
    c1 = comm.members[comm.offsets[1]:comm.offsets[2]]
    assert res["aggpk96"][0].tobytes() == H.closed_form_sum(c1[:2], a, b)
    assert res["aggpk96"][1][0] == 0x40                           # no attesters: infinity
    assert res["aggpk96"][3].tobytes() == H.closed_form_sum(c1[2:], a, b)


def test_g1_infinity_and_cancellation_inside_committees(engine_factory):
    e = engine_factory()
    A = g1.mul(11, g1.G)
    pts = np.stack([np.frombuffer(g1.to_bytes96(p), dtype=np.uint8)
                    for p in (A, g1.neg(A), None, A, A, g1.mul(22, g1.G), None, None)])
    out = e.g1_sum([0, 2, 3, 5, 6, 8, 8], points96=pts)
    assert out[0][0] == 0x40                                      # A + (-A)
    assert out[1][0] == 0x40                                      # lone infinity
    assert out[2].tobytes() == g1.to_bytes96(g1.mul(22, g1.G))    # A + A: doubling on the very first add
    assert out[3].tobytes() == g1.to_bytes96(g1.mul(22, g1.G))
    assert out[4][0] == 0x40 and out[5][0] == 0x40                # only infinities / empty
    # cancellation deep inside a tree: 512 copies of A and 512 of -A, interleaved, one group
    big = np.stack([pts[0], pts[1]] * 512)
    assert e.g1_sum([0, 1024], points96=big)[0][0] == 0x40
    # 1023 copies of A then -A...: many equal partial sums meet in the tree (doubling inside g1x_add_pair's slow path)
    same = np.stack([pts[0]] * 1024)
    assert e.g1_sum([0, 1024], points96=same)[0].tobytes() == g1.to_bytes96(g1.mul(11 * 1024, g1.G))


def test_config5_shape_mixed_balances(engine_factory):
    """BASELINE configs[4] shape on one GPU: 4 194 304 validators, 32..2048 ETH, 1 % equivocating, 0.5 % inactive,
    8192-block tree: u64 weights (8.6e18 Gwei total fits), head and every block weight vs the C oracle."""
    e = engine_factory()
    V, B = 1 << 22, 8192
    tree = synth.random_tree(B, 5, "bushy")
    H.load_tree(e, tree)
    bal = synth.balances(V, 5, mixed=True)
    flags = synth.validator_flags(V, 5, inactive_frac=0.005)
    e.set_validators(bal, flags)
    comm = synth.random_committees(V, 2048, 5)                    # 2048 committees of 2048
    epoch = int(tree.slot.max()) // 32 + 1
    e.set_committees(epoch, comm.offsets, comm.members)
    rng = np.random.default_rng(5)
    equiv = rng.choice(V, size=V // 100, replace=False)
    e.mark_equivocating(equiv)
    flags_o = flags.copy()
    flags_o[equiv] |= 4
    e.on_tick((epoch + 2) * 32 * 12)
    atts, arena, _ = synth.epoch_attestations(comm, tree, epoch, 32, seed=5, density=0.99, parts=1, from_block=True,
                                              vote_recent=64)
    status, _, count = e.on_attestation_batch(packed=(atts, arena))
    assert (status == 0).all() and count.sum() > 0.98 * V
    vote_epoch = np.zeros(V, dtype=np.uint64)
    vote_block = np.full(V, NONE32, dtype=np.uint32)
    mo, nb, bo = H.att_device_rows(atts, comm, 32)
    blk = np.array([e.block_index_of(r["beacon_block_root"].tobytes()) for r in atts], dtype=np.uint32)
    cport.update_latest_messages(mo, nb, bo, atts["target_epoch"], blk, arena, comm.members, flags_o, vote_epoch,
                                 vote_block)
    _, bi = e.latest_messages()
    assert np.array_equal(bi, vote_block)
    e.set_proposer_boost(tree.roots[B - 1].tobytes())
    head_o, w_o = cport.get_head(tree.parent, np.ones(B, np.uint8), tree.roots, vote_block, bal, flags_o, 0, B - 1)
    assert np.array_equal(e.get_weights(), w_o)
    assert e.get_head() == tree.roots[head_o].tobytes()
    assert int(w_o[0]) > 2**60
    # every validator at the EIP-7251 cap: 4 194 304 x 2048 ETH = 8.59e18 Gwei > 2^62, still < 2^64
    bal2 = np.full(V, 2048 * 10**9, dtype=np.uint64)
    e.set_balances(bal2, flags)
    head_o, w_o = cport.get_head(tree.parent, np.ones(B, np.uint8), tree.roots, vote_block, bal2, flags_o, 0, B - 1)
    assert np.array_equal(e.get_weights(), w_o)
    assert e.get_head() == tree.roots[head_o].tobytes()
    assert int(w_o[0]) > 2**62


def test_results_are_deterministic_run_to_run(engine_factory):
    """SURVEY.md 5 "race detection": the batch kernels use atomics (LDS histograms, 64-bit atomicMax tie-break, global
    weight adds) -- all of them on integers, so two runs of the same inputs on fresh engines must agree bit for bit in
    everything they return: latest messages, participation flags, weights, head, aggregates."""
    from pos_evolution_amd._abi import pe_state_ctx

    def run():
        e = engine_factory()
        n_val, spe = 60000, 32
        tree = synth.random_tree(500, 33, "branchy")
        H.load_tree(e, tree)
        pts, _ = H.oracle_points(n_val)
        e.set_validators(synth.balances(n_val, 33, True), synth.validator_flags(n_val, 33), pts)
        E = int(tree.slot.max()) // spe + 1
        comm = synth.random_committees(n_val, 128, 33)
        e.set_committees(E, comm.offsets, comm.members)
        e.on_tick((E + 1) * spe * 12)
        e.set_proposer_boost(tree.roots[499].tobytes())
        atts, arena, _ = synth.epoch_attestations(comm, tree, E, spe, seed=33, density=0.8, parts=3,
                                                  source=(0, tree.roots[0].tobytes()), vote_recent=40)
        agg = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
        # the un-aggregated rows too: overlapping committees in one batch exercise the atomicMax tie-break path
        st_raw, _, _ = e.on_attestation_batch(packed=(atts, arena))
        st_agg, pk, cnt = e.on_attestation_batch(packed=(agg["atts"], agg["out_arena"]), want_aggregate_pubkeys=True)
        ctx = pe_state_ctx()
        ctx.slot = (E + 1) * spe
        ctx.chain_tip_root[:] = tree.roots[499].tobytes()
        ctx.current_justified_root[:] = tree.roots[0].tobytes()
        ctx.previous_justified_root[:] = tree.roots[0].tobytes()
        ctx.base_reward_per_increment = 9000
        st_p, num = e.process_attestation_batch(ctx, packed=(agg["atts"], agg["out_arena"]))
        ep, blk = e.latest_messages()
        return [agg["aggpk96"].tobytes(), agg["out_arena"].tobytes(), agg["count"].tobytes(), st_raw.tobytes(),
                st_agg.tobytes(), pk.tobytes(), cnt.tobytes(), st_p.tobytes(), num.tobytes(), ep.tobytes(), blk.tobytes(),
                e.participation_get(0).tobytes(), e.participation_get(1).tobytes(), e.get_weights().tobytes(), e.get_head()]

    a, b = run(), run()
    assert [i for i, (x, y) in enumerate(zip(a, b)) if x != y] == []
    assert any(a[13])  # the weights are not trivially zero
