"""-m gpu: the reference-shaped interface (pos_evolution_amd.forkchoice) on the MI355X engine, run in lockstep
with the L0 literal oracle: after EVERY handler call the head, the checkpoints, the boost root and the whole
latest-message table must agree, and accept/reject decisions must match (tests/scenario.py)."""
import numpy as np
import pytest

from oracle import spec
from tests import fc_scenarios
from tests.scenario import l0_ffg_balances, l0_proposer_reward_numerator, new_world, slot_committee_members

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scenario", fc_scenarios.ALL, ids=lambda f: f.__name__)
def test_scenario_engine_vs_literal_oracle(scenario, engine_factory):
    scenario(lambda n, **kw: new_world(n, "minimal", engine_factory=engine_factory, **kw))


@pytest.mark.parametrize("seed,eta", [(1, 0), (2, 0), (3, 0), (4, 1), (5, 3), (6, 9)])
def test_random_event_stream(engine_factory, seed, eta):
    """Random ticks / forks / attestations (wire + from-block, valid + stale) / slashings on the minimal preset;
    eta > 0 runs the same stream under the vote-expiry variant (RLMD-GHOST, pe:1585-1596)."""
    rng = np.random.default_rng(seed)
    w = new_world(96, "minimal", engine_factory=engine_factory, VOTE_EXPIRY_SLOTS=eta)
    roots = [w.store.justified_checkpoint.root]
    slot = 0
    for step in range(60):
        slot += int(rng.integers(1, 3))
        w.tick_to_slot(slot, offset=int(rng.integers(0, spec.SECONDS_PER_SLOT)))
        parent = roots[int(rng.integers(max(0, len(roots) - 4), len(roots)))]
        if w.store.blocks[parent].slot < slot:
            roots.append(w.block(parent, slot, graffiti=bytes([step])))
        # attest for an earlier slot (in the past), to a random recent block not newer than that slot
        a_slot = slot - 1
        cand = [r for r in roots[-6:] if w.store.blocks[r].slot <= a_slot]
        if cand and a_slot >= 0:
            target_block = cand[int(rng.integers(0, len(cand)))]
            voters = slot_committee_members(w.store, a_slot)
            rng.shuffle(voters)
            for att in w.attestation_for(voters[: max(1, len(voters) // 2)], target_block, a_slot):
                w.attest(att, is_from_block=bool(rng.integers(0, 2)))
        if step % 17 == 16:
            members = slot_committee_members(w.store, a_slot)
            eq = sorted(members[:3])
            tgt = spec.Checkpoint(spec.compute_epoch_at_slot(a_slot), roots[0])
            d1 = spec.AttestationData(slot=a_slot, index=0, beacon_block_root=roots[-1], target=tgt)
            d2 = spec.AttestationData(slot=a_slot, index=0, beacon_block_root=roots[0], target=tgt)
            w.slash(spec.IndexedAttestation(eq, d1), spec.IndexedAttestation(eq, d2))
    assert len(w.store.latest_messages) > 0
    spec.use_preset("minimal")


def test_on_attestation_aggregate_pubkey_matches_bls_oracle(engine_factory):
    """The G1 sum FastAggregateVerify consumes (A.7): engine vs exact Python ints, on the (i+1)*G key set whose
    first additions hit the doubling branch."""
    from oracle import g1
    from pos_evolution_amd.forkchoice import _att_row
    w = new_world(64, "minimal", engine_factory=engine_factory, with_pubkeys=True)
    anchor = w.store.justified_checkpoint.root
    w.tick_to_slot(1)
    b1 = w.block(anchor, 1)
    w.tick_to_slot(3)
    state = w.store.block_states[anchor]
    for s in (1, 2):
        voters = slot_committee_members(w.store, s)
        for att in w.attestation_for(voters, b1, s):
            status, aggpk, count = w.mirror.engine.on_attestation_batch([_att_row(att)], want_aggregate_pubkeys=True)
            idx = spec.get_indexed_attestation(state, att).attesting_indices
            assert status[0] == 0 and count[0] == len(idx)
            got_idx, got_data, _ = w.fc.get_indexed_attestation(w.mirror, att)   # A.6 through the mirror
            assert got_idx == list(idx) and got_data == att.data
            assert aggpk[0].tobytes() == g1.to_bytes96(spec.aggregate_pubkeys(state, idx))


def test_store_from_compressed_pubkeys(engine_factory):
    """A state whose validators carry 48-byte BLSPubkeys (the pyspec's own type, pe:37) binds through the mirror:
    decompression on the GPU, same aggregate pubkeys as the affine registry."""
    from oracle import g1
    from pos_evolution_amd.forkchoice import _att_row
    w = new_world(32, "minimal", engine_factory=engine_factory, with_pubkeys=True)
    state = w.store.block_states[w.store.justified_checkpoint.root]
    affine = [v.pubkey for v in state.validators]
    for v in state.validators:
        v.pubkey = g1.compress(v.pubkey)
    w.mirror.set_justified_state(state)
    for v, p in zip(state.validators, affine):
        v.pubkey = p
    anchor = w.store.justified_checkpoint.root
    w.tick_to_slot(1)
    b1 = w.block(anchor, 1)
    w.tick_to_slot(2)
    voters = slot_committee_members(w.store, 1)
    for att in w.attestation_for(voters, b1, 1):
        status, aggpk, count = w.mirror.engine.on_attestation_batch([_att_row(att)], want_aggregate_pubkeys=True)
        idx = spec.get_indexed_attestation(state, att).attesting_indices
        assert status[0] == 0 and aggpk[0].tobytes() == g1.to_bytes96(spec.aggregate_pubkeys(state, idx))


def test_process_attestation_vs_literal_oracle(engine_factory):
    """process_attestation (pe:722-754) through the mirror: participation flags, proposer reward, asserts."""
    import copy
    import pos_evolution_amd.forkchoice as fc
    w = new_world(128, "minimal", engine_factory=engine_factory)
    anchor = w.store.justified_checkpoint.root
    roots = [anchor]
    for s in range(1, 7):
        w.tick_to_slot(s)
        roots.append(w.block(roots[-1], s))
    w.tick_to_slot(8)
    tip = roots[-1]
    # the state process_attestation runs on: post-state of the tip advanced to slot 7 (a block is being built)
    state = w.store.block_states[tip].copy()
    spec.process_slots(state, 7)
    state.proposer_index_override = 5
    mstate = state.copy()
    eng = w.mirror.engine
    w._ensure_committees(0, state)
    fc.bind_state(eng, mstate, tip, spec.get_base_reward_per_increment(state))
    atts = []
    for s in (3, 5, 6, 6):   # slot 6 twice: the second inclusion must earn nothing new
        voters = slot_committee_members(w.store, s)
        head_at_s = spec.get_block_root_at_slot(state, s)
        atts += w.attestation_for(voters, head_at_s, s)
    wrong_head = w.attestation_for(slot_committee_members(w.store, 4), roots[1], 4)
    atts += wrong_head
    import dataclasses
    bad_source = dataclasses.replace(atts[0], data=dataclasses.replace(atts[0].data, source=spec.Checkpoint(3, tip)))
    too_new = w.attestation_for(slot_committee_members(w.store, 7), tip, 7)   # slot + 1 > state.slot
    for att in atts + [bad_source] + too_new:
        ok = True
        pre_cur, pre_prev = list(state.current_epoch_participation), list(state.previous_epoch_participation)
        try:
            spec.process_attestation(state, att)
        except (AssertionError, KeyError):
            ok = False
        m_ok = True
        try:
            fc.process_attestation(mstate, att, get_beacon_proposer_index=spec.get_beacon_proposer_index)
            with pytest.raises(AssertionError):   # the proposer (pe:754) is never defaulted
                fc.process_attestation(mstate, att)
        except AssertionError:
            m_ok = False
        assert ok == m_ok
        if ok:
            assert mstate._last_proposer_reward_numerator == l0_proposer_reward_numerator(state, pre_cur, pre_prev)
        assert mstate.current_epoch_participation == state.current_epoch_participation
        assert mstate.previous_epoch_participation == state.previous_epoch_participation
        assert mstate.balances == state.balances
    assert sum(state.current_epoch_participation) > 0


def test_justification_and_finalization_vs_literal_oracle(engine_factory):
    """process_justification_and_finalization (pe:791-802) + weigh_justification_and_finalization (pe:815-853):
    the three Gwei sums come from the GPU, the checkpoint/bit logic is the reference's, both must match L0 through
    several epochs with uneven balances, slashed and exiting validators, and partial participation."""
    import pos_evolution_amd.forkchoice as fc
    rng = np.random.default_rng(9)
    w = new_world(160, "minimal", engine_factory=engine_factory)
    anchor = w.store.justified_checkpoint.root
    spe = spec.SLOTS_PER_EPOCH
    roots = [anchor]
    for s in range(1, 5 * spe):
        w.tick_to_slot(s)
        roots.append(w.block(roots[-1], s))
    tip = roots[-1]
    state = w.store.block_states[tip].copy()          # slot 5*spe - 1: last slot of epoch 4, where process_epoch runs
    for i, v in enumerate(state.validators):
        v.effective_balance = int(rng.integers(16, 33)) * 10**9
        if i % 11 == 0:
            v.slashed = True
        if i % 13 == 0:
            v.exit_epoch = 4                            # active in the previous epoch (3) only
        if i % 17 == 0:
            v.activation_epoch = 4                      # active in the current epoch (4) only
    state.justification_bits = [False, True, True, False]
    state.previous_justified_checkpoint = spec.Checkpoint(1, spec.get_block_root(state, 1))
    state.current_justified_checkpoint = spec.Checkpoint(2, spec.get_block_root(state, 2))
    for target_frac_prev, target_frac_cur in ((0.9, 0.1), (0.5, 0.8), (0.7, 0.69)):
        st = state.copy()
        n = len(st.validators)
        st.previous_epoch_participation = [int(rng.integers(0, 8)) | (2 if rng.random() < target_frac_prev else 0) & 7 for _ in range(n)]
        st.current_epoch_participation = [int(rng.integers(0, 2)) | (2 if rng.random() < target_frac_cur else 0) for _ in range(n)]
        mst = st.copy()
        fc.bind_state(w.mirror.engine, mst, tip, 1)
        want_balances = l0_ffg_balances(st)
        spec.process_justification_and_finalization(st)
        fc.process_justification_and_finalization(mst, get_block_root=spec.get_block_root)
        assert mst._last_ffg_balances == want_balances
        assert mst.justification_bits == st.justification_bits
        assert mst.current_justified_checkpoint == st.current_justified_checkpoint
        assert mst.previous_justified_checkpoint == st.previous_justified_checkpoint
        assert mst.finalized_checkpoint == st.finalized_checkpoint


def test_get_head_refuses_a_stale_justified_state(engine_factory):
    """on_block can move store.justified_checkpoint; the reference then weighs votes with THAT checkpoint's
    state (A.1).  The mirror tracks which checkpoint the engine's balances belong to: get_head raises until the new
    state is handed over -- directly or through store.checkpoint_state_provider."""
    import pos_evolution_amd.forkchoice as fc
    from pos_evolution_amd import EngineError
    w = new_world(64, "minimal", engine_factory=engine_factory)
    anchor = w.store.justified_checkpoint.root
    spe = spec.SLOTS_PER_EPOCH
    w.tick_to_slot(1)
    b1 = w.block(anchor, 1)
    for i, v in enumerate(w.store.block_states[b1].validators):
        v.effective_balance = (8 + i % 3) * 10**9
    spec.on_tick(w.store, w.store.genesis_time + spe * spec.SECONDS_PER_SLOT)
    fc.on_tick(w.mirror, w.store.genesis_time + spe * spec.SECONDS_PER_SLOT)
    just = spec.Checkpoint(1, b1)
    blk = spec.BeaconBlock(slot=spe, parent_root=b1, body=spec.BeaconBlockBody(graffiti=b"j"))
    signed = spec.SignedBeaconBlock(message=blk, scripted_checkpoints=(just, spec.Checkpoint(0, anchor)))
    spec.on_block(w.store, signed)
    fc.on_block(w.mirror, signed, w.store.block_states[spec.hash_tree_root(blk)])
    assert w.mirror.justified_checkpoint.epoch == 1
    with pytest.raises(EngineError):
        fc.get_head(w.mirror)
    spec.store_target_checkpoint_state(w.store, just)
    w.mirror.checkpoint_state_provider = lambda cp: w.store.checkpoint_states[spec.Checkpoint(cp.epoch, spec.Root(cp.root))]
    assert fc.get_head(w.mirror) == bytes(spec.get_head(w.store))
    assert w.mirror.balances_checkpoint == w.mirror.justified_checkpoint
    w.check()
