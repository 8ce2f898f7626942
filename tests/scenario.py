"""Scenario toolkit: drives the L0 literal oracle (oracle/spec.py) and, optionally, the engine's
spec-shaped mirror (pos_evolution_amd.forkchoice) with the SAME sequence of handler calls, so the
tests read like pyspec fork-choice tests and every event is a differential check."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from oracle import g1, spec


def synthetic_pubkey(i: int):
    """(i+1)*G: the edge-case key set of SURVEY.md 8(d) config 1 (forces the doubling branch)."""
    return g1.mul(i + 1, g1.G)


def genesis(n_validators: int, balance: int = 32 * 10**9, with_pubkeys: bool = False, genesis_time: int = 0):
    """Genesis BeaconState + anchor block for the currently bound preset."""
    pks = None
    if with_pubkeys:
        pks, cur = [], None
        for _ in range(n_validators):
            cur = g1.add(cur, g1.G)
            pks.append(cur)
    state = spec.BeaconState(
        genesis_time=genesis_time, slot=0,
        validators=[spec.Validator(pubkey=(pks[i] if pks else None), effective_balance=balance)
                    for i in range(n_validators)],
        balances=[balance] * n_validators,
        randao_mixes=[spec.sha256(b"mix" + i.to_bytes(4, "little")) for i in range(spec.EPOCHS_PER_HISTORICAL_VECTOR)],
        previous_epoch_participation=[0] * n_validators,
        current_epoch_participation=[0] * n_validators,
    )
    block = spec.BeaconBlock(slot=0, state_root=spec.hash_tree_root(state))
    # the anchor is the state's latest block: get_block_root_at_slot(state, 0) resolves to it
    state.block_roots[0] = spec.hash_tree_root(block)
    state.latest_block_root = state.block_roots[0]
    return state, block


@dataclass
class World:
    """One L0 store plus (optionally) the engine mirror, kept in lockstep."""
    store: spec.Store
    mirror: object = None          # pos_evolution_amd.forkchoice.Store or None
    fc: object = None              # the forkchoice module
    committees_loaded: set = field(default_factory=set)

    # ---- helpers ---------------------------------------------------------
    def _ensure_committees(self, epoch: int, state: spec.BeaconState):
        if self.mirror is None or epoch in self.committees_loaded:
            return
        cps = spec.get_committee_count_per_slot(state, epoch)
        comms = []
        for s in range(spec.SLOTS_PER_EPOCH):
            for i in range(cps):
                comms.append(spec.get_beacon_committee(state, epoch * spec.SLOTS_PER_EPOCH + s, i))
        self.mirror.set_committees(epoch, comms)
        self.committees_loaded.add(epoch)

    def _sync_justified(self):
        """get_latest_attesting_balance reads checkpoint_states[justified_checkpoint] (A.1); a client has that
        state materialised by the time the checkpoint becomes justified.  Do the same for L0, and hand the
        engine the balances of that state (pe_set_balances)."""
        jc = self.store.justified_checkpoint
        if jc not in self.store.checkpoint_states:
            spec.store_target_checkpoint_state(self.store, jc)
        # the engine holds one registry view: hand it the justified state's whenever the checkpoint it belongs to is
        # no longer the store's (also when on_attestation had materialised that checkpoint state earlier)
        if self.mirror is not None:
            have = self.mirror.balances_checkpoint
            if have is None or (have.epoch, bytes(have.root)) != (jc.epoch, bytes(jc.root)):
                self.mirror.set_justified_state(self.store.checkpoint_states[jc])

    def check(self):
        """After-event differential: head, scalars, latest messages."""
        self._sync_justified()
        if self.mirror is None:
            return
        s, m = self.store, self.mirror
        assert m.time == s.time
        assert tuple(m.justified_checkpoint.__dict__.values()) == (s.justified_checkpoint.epoch, bytes(s.justified_checkpoint.root))
        assert tuple(m.finalized_checkpoint.__dict__.values()) == (s.finalized_checkpoint.epoch, bytes(s.finalized_checkpoint.root))
        assert tuple(m.best_justified_checkpoint.__dict__.values()) == (s.best_justified_checkpoint.epoch, bytes(s.best_justified_checkpoint.root))
        assert m.proposer_boost_root == bytes(s.proposer_boost_root)
        lm = m.latest_messages
        assert {k: (v.epoch, bytes(v.root)) for k, v in s.latest_messages.items()} == \
               {k: (v.epoch, v.root) for k, v in lm.items()}
        assert self.fc.get_head(m) == bytes(spec.get_head(s))
        # per-block weights (get_latest_attesting_balance incl. proposer boost, A.1) -- after get_head, same view
        eng = m.engine
        weights = eng.get_weights()
        for i in range(eng.num_blocks):
            root = eng.block_root_at(i)
            assert int(weights[i]) == spec.get_latest_attesting_balance(s, spec.Root(root)), (i, root.hex())

    # ---- handlers, applied to both -----------------------------------------
    def tick(self, time: int):
        spec.on_tick(self.store, time)
        if self.mirror is not None:
            self.fc.on_tick(self.mirror, time)
        self.check()

    def tick_to_slot(self, slot: int, offset: int = 0):
        self.tick(self.store.genesis_time + slot * spec.SECONDS_PER_SLOT + offset)

    def block(self, parent_root, slot: int, attestations: Sequence = (), scripted=None, graffiti: bytes = b"",
              expect_fail: bool = False):
        """Build + on_block.  Returns the block root."""
        blk = spec.BeaconBlock(slot=slot, parent_root=parent_root,
                               body=spec.BeaconBlockBody(attestations=list(attestations), graffiti=graffiti))
        signed = spec.SignedBeaconBlock(message=blk, scripted_checkpoints=scripted)
        root = spec.hash_tree_root(blk)
        ok = True
        try:
            spec.on_block(self.store, signed)
        except (AssertionError, KeyError):
            ok = False
        if self.mirror is not None:
            # the mirror needs the post-state: from the oracle when it accepted, else a stand-in
            post = self.store.block_states.get(root)
            m_ok = True
            try:
                if post is None:
                    post = spec.BeaconState()
                self.fc.on_block(self.mirror, signed, post)
            except AssertionError:
                m_ok = False
            assert m_ok == ok, f"on_block acceptance differs: oracle {ok}, engine {m_ok}"
        assert ok != expect_fail, "unexpected on_block outcome"
        self.check()
        return root

    def attestation_for(self, validators: Sequence[int], block_root, slot: int, index: Optional[int] = None,
                        target_epoch: Optional[int] = None, signature_valid: bool = True):
        """Attestations (one per committee touched) in which exactly `validators` attest to block_root at `slot`."""
        epoch = spec.compute_epoch_at_slot(slot) if target_epoch is None else target_epoch
        target_root = spec.get_ancestor(self.store, block_root, spec.compute_start_slot_at_epoch(epoch))
        target = spec.Checkpoint(epoch, target_root)
        spec.store_target_checkpoint_state(self.store, target)
        tstate = self.store.checkpoint_states[target]
        self._ensure_committees(epoch, tstate)
        cps = spec.get_committee_count_per_slot(tstate, epoch)
        out = []
        want = set(validators)
        for i in ([index] if index is not None else range(cps)):
            committee = spec.get_beacon_committee(tstate, slot, i)
            bits = [v in want for v in committee]
            if any(bits):
                data = spec.AttestationData(slot=slot, index=i, beacon_block_root=block_root,
                                            source=tstate.current_justified_checkpoint, target=target)
                out.append(spec.Attestation(aggregation_bits=bits, data=data, signature_valid=signature_valid))
        return out

    def attest(self, attestation, is_from_block: bool = False, expect_fail: Optional[bool] = None):
        ok = True
        before = dict(self.store.latest_messages)
        try:
            spec.on_attestation(self.store, attestation, is_from_block)
        except (AssertionError, KeyError, IndexError):
            ok = False
            assert self.store.latest_messages == before  # pe:1041
        if self.mirror is not None:
            tgt = attestation.data.target
            if tgt in self.store.checkpoint_states:
                self._ensure_committees(tgt.epoch, self.store.checkpoint_states[tgt])
            m_ok = True
            try:
                self.fc.on_attestation(self.mirror, attestation, is_from_block)
            except AssertionError:
                m_ok = False
            assert m_ok == ok, f"on_attestation acceptance differs: oracle {ok}, engine {m_ok}"
        if expect_fail is not None:
            assert ok != expect_fail
        self.check()
        return ok

    def vote(self, validators: Sequence[int], block_root, slot: int):
        """Every listed validator that sits in a committee of `slot` attests to block_root."""
        for a in self.attestation_for(validators, block_root, slot):
            self.attest(a)

    def slash(self, att1: spec.IndexedAttestation, att2: spec.IndexedAttestation, expect_fail: bool = False):
        sl = spec.AttesterSlashing(att1, att2)
        ok = True
        try:
            spec.on_attester_slashing(self.store, sl)
        except AssertionError:
            ok = False
        if self.mirror is not None:
            m_ok = True
            try:
                self.fc.on_attester_slashing(self.mirror, sl)
            except AssertionError:
                m_ok = False
            assert m_ok == ok
        assert ok != expect_fail
        self.check()

    def head(self):
        h = spec.get_head(self.store)
        if self.mirror is not None:
            assert self.fc.get_head(self.mirror) == bytes(h)
        return h


def l0_proposer_reward_numerator(state, pre_current: Sequence[int], pre_previous: Sequence[int]) -> int:
    """proposer_reward_numerator of the process_attestation call that turned (pre_current, pre_previous) into the
    state's participation arrays: sum of get_base_reward(i) * weight over the flags it newly set (pe:746-749).  The
    reference keeps the numerator in a local, so the harness recovers it from the before/after flags."""
    total = 0
    for pre, post in ((pre_current, state.current_epoch_participation), (pre_previous, state.previous_epoch_participation)):
        for i, (a, b) in enumerate(zip(pre, post)):
            new = b & ~a
            if new:
                for flag_index, weight in enumerate(spec.PARTICIPATION_FLAG_WEIGHTS):
                    if spec.has_flag(new, flag_index):
                        total += spec.get_base_reward(state, i) * weight
    return total


def l0_ffg_balances(state):
    """The three Gwei sums process_justification_and_finalization (pe:797-801) hands to
    weigh_justification_and_finalization, computed with the same calls."""
    previous_indices = spec.get_unslashed_participating_indices(state, spec.TIMELY_TARGET_FLAG_INDEX, spec.get_previous_epoch(state))
    current_indices = spec.get_unslashed_participating_indices(state, spec.TIMELY_TARGET_FLAG_INDEX, spec.get_current_epoch(state))
    return (spec.get_total_active_balance(state), spec.get_total_balance(state, previous_indices),
            spec.get_total_balance(state, current_indices))


def engine_config_for_preset() -> dict:
    """pe_config fields matching the constants currently bound in oracle.spec."""
    return dict(
        slots_per_epoch=spec.SLOTS_PER_EPOCH, seconds_per_slot=spec.SECONDS_PER_SLOT,
        intervals_per_slot=spec.INTERVALS_PER_SLOT,
        safe_slots_to_update_justified=spec.SAFE_SLOTS_TO_UPDATE_JUSTIFIED,
        proposer_score_boost=spec.PROPOSER_SCORE_BOOST,
        effective_balance_increment=spec.EFFECTIVE_BALANCE_INCREMENT,
        min_attestation_inclusion_delay=spec.MIN_ATTESTATION_INCLUSION_DELAY,
        max_validators_per_committee=spec.MAX_VALIDATORS_PER_COMMITTEE,
        max_committee_tables=16, vote_expiry_slots=spec.VOTE_EXPIRY_SLOTS,
    )


def new_world(n_validators: int, preset: str = "minimal", engine_factory=None, with_pubkeys: bool = False,
              **overrides) -> World:
    spec.use_preset(preset, **overrides)
    state, block = genesis(n_validators, with_pubkeys=with_pubkeys)
    store = spec.get_forkchoice_store(state, block)
    mirror = fc = None
    if engine_factory is not None:
        import pos_evolution_amd.forkchoice as fc
        engine = engine_factory(**engine_config_for_preset())
        mirror = fc.get_forkchoice_store(state, block, hash_tree_root=spec.hash_tree_root, engine=engine)
    return World(store=store, mirror=mirror, fc=fc)


def slot_committee_members(store: spec.Store, slot: int) -> List[int]:
    """All validators attesting in `slot` (union of its committees), per the anchor state's shuffling."""
    state = store.block_states[store.justified_checkpoint.root]
    epoch = spec.compute_epoch_at_slot(slot)
    cps = spec.get_committee_count_per_slot(state, epoch)
    out = []
    for i in range(cps):
        out += spec.get_beacon_committee(state, slot, i)
    return out
