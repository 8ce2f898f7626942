"""Host-side mirror of the reference's fork-choice / attestation interface, backed by the
MI355X engine through the C ABI.

Same names, argument meaning and error behaviour as the pyspec excerpts in
pos-evolution.md (``pe:N``):

    get_forkchoice_store(anchor_state, anchor_block)        pe:1077-1095
    on_tick(store, time)                                    pe:934-955
    on_block(store, signed_block, post_state)               pe:986-1036   (state_transition is the caller's)
    on_attestation(store, attestation, is_from_block=False) pe:963-979, pe:1423-1428
    on_attester_slashing(store, attester_slashing)          pe:1447-1461
    get_head(store) -> Root                                 pe:1102-1116
    process_attestation(state, attestation)                 pe:722-754    (state bound with bind_state)

Objects are duck-typed: anything exposing the pyspec's field names works
(``attestation.data.beacon_block_root``, ``attestation.aggregation_bits``,
``state.validators[i].effective_balance`` ...).  A failed handler raises
``AssertionError`` (``EngineError``) and leaves the store unmodified (pe:1041).

Out of scope and therefore supplied by the caller: SSZ
``hash_tree_root`` (pass the function), ``state_transition`` (pass the post-state),
committee shuffling (pass the committees), the pairing check (its boolean rides on
``attestation.signature_valid``, default True).

This module performs no arithmetic of the hot path itself and never imports the
oracle: without libposevo.so + a HIP device it raises.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import _abi
from .engine import AttRow, Engine, EngineError, ZERO_ROOT
from ._abi import pe_state_ctx

GENESIS_EPOCH = 0
PARTICIPATION_FLAG_WEIGHTS = [14, 26, 14]
PROPOSER_WEIGHT = 8
WEIGHT_DENOMINATOR = 64


@dataclass(eq=True, frozen=True)
class Checkpoint:
    """pe:219-221"""
    epoch: int = 0
    root: bytes = ZERO_ROOT


@dataclass(eq=True, frozen=True)
class LatestMessage:
    """pe:286-289"""
    epoch: int
    root: bytes


def _att_row(attestation, is_from_block: bool = False) -> AttRow:
    d = attestation.data
    return AttRow(
        slot=int(d.slot), index=int(d.index), beacon_block_root=bytes(d.beacon_block_root),
        source_epoch=int(d.source.epoch), source_root=bytes(d.source.root),
        target_epoch=int(d.target.epoch), target_root=bytes(d.target.root),
        bits=np.asarray(attestation.aggregation_bits, dtype=np.uint8),
        signature_valid=bool(getattr(attestation, "signature_valid", True)),
        is_from_block=is_from_block,
    )


class Store:
    """Engine-backed ``Store`` (pe:889-901).  Scalars live in the engine; this object exposes them
    under the pyspec's attribute names."""

    def __init__(self, engine: Engine, hash_tree_root: Optional[Callable] = None):
        self.engine = engine
        self.hash_tree_root = hash_tree_root
        self.equivocating_indices = set()   # host mirror of pe:897 (the engine holds the per-validator bit)
        self.blocks: Dict[bytes, object] = {}  # root -> the caller's block object (pe:898)
        # get_latest_attesting_balance reads checkpoint_states[store.justified_checkpoint] on EVERY call (Appendix A.1);
        # the engine holds ONE registry view.  Which checkpoint it belongs to is tracked here, and get_head refuses
        # to weigh votes with another checkpoint's balances (on_block / on_tick / set_checkpoints can move
        # justified_checkpoint).  checkpoint_state_provider(checkpoint) -> state, when set, is asked for the missing
        # state instead (a client has it materialised by the time the checkpoint is justified).
        self.balances_checkpoint: Optional[Checkpoint] = None
        self.checkpoint_state_provider: Optional[Callable] = None

    # -- the spec's fields ---------------------------------------------------
    @property
    def time(self) -> int:
        return self.engine.store_scalars()["time"]

    @property
    def genesis_time(self) -> int:
        return self.engine.store_scalars()["genesis_time"]

    @property
    def justified_checkpoint(self) -> Checkpoint:
        return Checkpoint(*self.engine.store_scalars()["justified"])

    @property
    def finalized_checkpoint(self) -> Checkpoint:
        return Checkpoint(*self.engine.store_scalars()["finalized"])

    @property
    def best_justified_checkpoint(self) -> Checkpoint:
        return Checkpoint(*self.engine.store_scalars()["best_justified"])

    @property
    def proposer_boost_root(self) -> bytes:
        return self.engine.store_scalars()["proposer_boost_root"]

    @proposer_boost_root.setter
    def proposer_boost_root(self, root: bytes):
        self.engine.set_proposer_boost(bytes(root))

    @property
    def latest_messages(self) -> Dict[int, LatestMessage]:
        epoch, block = self.engine.latest_messages()
        roots = [self.engine.block_root_at(i) for i in range(self.engine.num_blocks)]
        return {int(i): LatestMessage(int(epoch[i]), roots[int(block[i])])
                for i in np.nonzero(block != _abi.NONE32)[0]}

    # -- inputs the pyspec derives from states the engine does not hold -----------
    def ensure_justified_state(self):
        """The engine's balances must be those of checkpoint_states[justified_checkpoint] (A.1)."""
        jc = self.justified_checkpoint
        if self.balances_checkpoint == jc:
            return
        if self.checkpoint_state_provider is not None:
            self.set_justified_state(self.checkpoint_state_provider(jc))
            return
        raise EngineError(_abi.PE_ERR_STATE,
                          f"justified checkpoint moved to epoch {jc.epoch} but the engine still holds the balances of "
                          f"{self.balances_checkpoint}: call store.set_justified_state(checkpoint_states[justified_checkpoint]) "
                          "or set store.checkpoint_state_provider")

    def set_justified_state(self, state, slots_per_epoch: Optional[int] = None):
        """checkpoint_states[justified_checkpoint] (Appendix A.1): balances/activity feeding the weights,
        and the pubkeys feeding the G1 sums.  Records that the engine's view now belongs to the store's current
        justified checkpoint."""
        spe = slots_per_epoch or int(self.engine.cfg.slots_per_epoch)
        epoch = int(state.slot) // spe
        n = len(state.validators)
        bal = np.fromiter((int(v.effective_balance) for v in state.validators), dtype=np.uint64, count=n)
        flags = np.fromiter(
            ((_abi.PE_VAL_ACTIVE if int(v.activation_epoch) <= epoch < int(v.exit_epoch) else 0)
             | (_abi.PE_VAL_SLASHED if v.slashed else 0) for v in state.validators), dtype=np.uint8, count=n)
        pk = pk48 = None
        if n and all(getattr(v, "pubkey", None) is not None for v in state.validators):
            if all(isinstance(v.pubkey, (bytes, bytearray)) and len(v.pubkey) == 48 for v in state.validators):
                # the pyspec's own type: BLSPubkey = 48-byte compressed (pe:37); decompressed on the GPU
                pk48 = np.frombuffer(b"".join(bytes(v.pubkey) for v in state.validators), dtype=np.uint8)
            else:
                pk = np.frombuffer(b"".join(_point96(v.pubkey) for v in state.validators), dtype=np.uint8)
        if self.engine.num_validators == n and pk is None and pk48 is None:
            self.engine.set_balances(bal, flags)
        else:
            self.engine.set_validators(bal, flags, pk)
            if pk48 is not None:
                self.engine.set_pubkeys_compressed(pk48)
        self.balances_checkpoint = self.justified_checkpoint

    def set_committees(self, epoch: int, committees: Sequence[Sequence[int]]):
        """get_beacon_committee(state, slot, index) for every (slot, index) of ``epoch``, in committee-id
        order (slot-major): the L2 feeder's output (compute_committee, pe:495-504)."""
        sizes = np.fromiter((len(c) for c in committees), dtype=np.uint32, count=len(committees))
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
        members = (np.concatenate([np.asarray(c, dtype=np.uint32) for c in committees])
                   if len(committees) else np.zeros(0, dtype=np.uint32))
        self.engine.set_committees(epoch, offsets, members)


    def compute_committees(self, epoch: int, seed: bytes, active_indices: Sequence[int], committees_per_slot: int,
                           shuffle_round_count: int = 90):
        """The same table computed on the GPU from get_seed(state, epoch, DOMAIN_BEACON_ATTESTER) (pe:481-486)
        and get_active_validator_indices(state, epoch): compute_committee over compute_shuffled_index
        (pe:495-534).  Store method so that the client no longer runs the shuffle itself."""
        n = int(self.engine.cfg.slots_per_epoch) * committees_per_slot
        return self.engine.compute_committees(epoch, bytes(seed), active_indices, n, shuffle_round_count)


def _point96(pt) -> bytes:
    """Affine (x, y) int tuple / None / 96 raw bytes -> 96-byte uncompressed encoding."""
    if isinstance(pt, (bytes, bytearray)):
        assert len(pt) == 96
        return bytes(pt)
    if pt is None:
        return bytes([0x40]) + bytes(95)
    x, y = pt
    return int(x).to_bytes(48, "big") + int(y).to_bytes(48, "big")


# ----------------------------------------------------------------------------- handlers
def get_forkchoice_store(anchor_state, anchor_block, *, hash_tree_root: Callable, engine: Optional[Engine] = None,
                         **engine_config) -> Store:
    """pe:1077-1095"""
    assert bytes(anchor_block.state_root) == bytes(hash_tree_root(anchor_state))
    anchor_root = bytes(hash_tree_root(anchor_block))
    eng = engine or Engine(**engine_config)
    eng.store_init(int(anchor_state.genesis_time), int(anchor_state.slot), anchor_root)
    store = Store(eng, hash_tree_root)
    store.blocks[anchor_root] = anchor_block
    store.set_justified_state(anchor_state)  # checkpoint_states = {justified_checkpoint: anchor_state}
    return store


def on_tick(store: Store, time: int) -> None:
    """pe:934-955"""
    store.engine.on_tick(int(time))


def on_block(store: Store, signed_block, post_state) -> None:
    """pe:986-1036.  ``post_state`` = the block's post-state as computed by the caller's
    ``state_transition`` (pe:1009); only its two checkpoints are read."""
    block = signed_block.message
    root = bytes(store.hash_tree_root(block))
    j, f = post_state.current_justified_checkpoint, post_state.finalized_checkpoint
    store.engine.on_block(root, bytes(block.parent_root), int(block.slot), (int(j.epoch), bytes(j.root)),
                          (int(f.epoch), bytes(f.root)))
    store.blocks[root] = block


def on_attestation(store: Store, attestation, is_from_block: bool = False) -> None:
    """pe:963-979 with the ``is_from_block`` form of pe:1423-1428."""
    status, _, _ = store.engine.on_attestation_batch([_att_row(attestation, is_from_block)])
    if status[0] != 0:
        raise EngineError(int(status[0]), "on_attestation: " + _abi.ATT_STATUS_NAMES.get(int(status[0]), "?"))


def on_attestations(store: Store, attestations: Iterable, is_from_block: bool = False) -> np.ndarray:
    """Batch form: ``on_attestation`` x n applied as if sequentially; returns the per-attestation status
    (0 = applied) instead of raising on the first invalid one."""
    rows = [_att_row(a, is_from_block) for a in attestations]
    status, _, _ = store.engine.on_attestation_batch(rows)
    return status


def on_attester_slashing(store: Store, attester_slashing) -> None:
    """pe:1447-1461"""
    a1, a2 = attester_slashing.attestation_1, attester_slashing.attestation_2

    def row(ia):
        d = ia.data
        return AttRow(int(d.slot), int(d.index), bytes(d.beacon_block_root), int(d.source.epoch), bytes(d.source.root),
                      int(d.target.epoch), bytes(d.target.root), np.zeros(0, dtype=np.uint8),
                      bool(getattr(ia, "signature_valid", True)))

    store.engine.on_attester_slashing(row(a1), list(a1.attesting_indices), row(a2), list(a2.attesting_indices))
    store.equivocating_indices |= set(a1.attesting_indices).intersection(a2.attesting_indices)


def get_indexed_attestation(store: Store, attestation):
    """get_indexed_attestation(state, attestation) (Appendix A.6): the sorted attesting indices, resolved against the
    committee table of the attestation's target epoch on the GPU.  Returns (attesting_indices, data, signature)."""
    status, offsets, indices = store.engine.get_indexed_attestations([_att_row(attestation)])
    if status[0] != 0:
        raise EngineError(int(status[0]), "get_indexed_attestation: " + _abi.ATT_STATUS_NAMES.get(int(status[0]), "?"))
    return [int(i) for i in indices], attestation.data, getattr(attestation, "signature", None)


def get_head(store: Store) -> bytes:
    """pe:1102-1116.  Raises (AssertionError) when the justified checkpoint has moved and the balances of its state
    have not been handed over (Store.ensure_justified_state)."""
    store.ensure_justified_state()
    return store.engine.get_head()


# ----------------------------------------------------------------------------- process_attestation
class StateBinding:
    """Binds a BeaconState-like object to the engine's working participation arrays so that
    ``process_attestation(state, attestation)`` can mutate it as the pyspec does."""

    def __init__(self, engine: Engine, state, chain_tip_root: bytes, base_reward_per_increment: int):
        self.engine = engine
        self.state = state
        self.chain_tip_root = bytes(chain_tip_root)
        self.base_reward_per_increment = int(base_reward_per_increment)
        engine.participation_set(0, np.asarray(state.current_epoch_participation, dtype=np.uint8))
        engine.participation_set(1, np.asarray(state.previous_epoch_participation, dtype=np.uint8))
        # the working state's own registry view (effective balances, activity in current/previous epoch, slashed)
        spe = int(engine.cfg.slots_per_epoch)
        cur = int(state.slot) // spe
        prev = max(cur - 1, GENESIS_EPOCH)
        n = len(state.validators)
        bal = np.fromiter((int(v.effective_balance) for v in state.validators), dtype=np.uint64, count=n)
        fl = np.fromiter(
            ((_abi.PE_VAL_ACTIVE if int(v.activation_epoch) <= cur < int(v.exit_epoch) else 0)
             | (_abi.PE_VAL_SLASHED if v.slashed else 0)
             | (_abi.PE_VAL_ACTIVE_PREV if int(v.activation_epoch) <= prev < int(v.exit_epoch) else 0)
             for v in state.validators), dtype=np.uint8, count=n)
        engine.state_set_validators(bal, fl)

    def ctx(self) -> pe_state_ctx:
        s = self.state
        c = pe_state_ctx()
        c.slot = int(s.slot)
        c.chain_tip_root[:] = self.chain_tip_root
        c.current_justified_epoch = int(s.current_justified_checkpoint.epoch)
        c.current_justified_root[:] = bytes(s.current_justified_checkpoint.root)
        c.previous_justified_epoch = int(s.previous_justified_checkpoint.epoch)
        c.previous_justified_root[:] = bytes(s.previous_justified_checkpoint.root)
        c.base_reward_per_increment = self.base_reward_per_increment
        return c

    def sync_back(self):
        self.state.current_epoch_participation[:] = [int(x) for x in self.engine.participation_get(0)]
        self.state.previous_epoch_participation[:] = [int(x) for x in self.engine.participation_get(1)]


_BINDING_ATTR = "_posevo_state_binding"


def bind_state(engine: Engine, state, chain_tip_root: bytes, base_reward_per_increment: int) -> StateBinding:
    """The binding lives ON the state object (not in a table keyed by id(state), which would leak and could hand a
    recycled id a stale binding); ``unbind_state`` drops it."""
    b = StateBinding(engine, state, chain_tip_root, base_reward_per_increment)
    object.__setattr__(state, _BINDING_ATTR, b)
    return b


def unbind_state(state) -> None:
    if getattr(state, _BINDING_ATTR, None) is not None:
        object.__setattr__(state, _BINDING_ATTR, None)


def _binding(state) -> Optional[StateBinding]:
    b = getattr(state, _BINDING_ATTR, None)
    return b if b is not None and b.state is state else None   # a copy of a bound state is not bound


def process_attestation(state, attestation, *, get_beacon_proposer_index: Optional[Callable] = None,
                        proposer_index: Optional[int] = None) -> None:
    """pe:722-754.  ``state`` must have been bound with ``bind_state``.  The proposer that earns the reward
    (pe:754) is ``get_beacon_proposer_index(state)`` or the explicit ``proposer_index``: one of them is required --
    crediting a default validator would silently corrupt ``state.balances``."""
    b = _binding(state)
    assert b is not None, "process_attestation: bind_state(engine, state, ...) first"
    assert get_beacon_proposer_index is not None or proposer_index is not None, \
        "process_attestation: pass get_beacon_proposer_index or proposer_index (pe:754)"
    status, numerators = b.engine.process_attestation_batch(b.ctx(), [_att_row(attestation)])
    if status[0] != 0:
        raise EngineError(int(status[0]), "process_attestation: " + _abi.ATT_STATUS_NAMES.get(int(status[0]), "?"))
    b.sync_back()
    # Reward proposer (pe:752-754)
    denominator = WEIGHT_DENOMINATOR * (WEIGHT_DENOMINATOR - PROPOSER_WEIGHT) // PROPOSER_WEIGHT
    proposer_reward = int(numerators[0]) // denominator
    proposer = int(proposer_index) if proposer_index is not None else int(get_beacon_proposer_index(state))
    state.balances[proposer] += proposer_reward
    state._last_proposer_reward_numerator = int(numerators[0])


# ----------------------------------------------------------------------------- FFG
JUSTIFICATION_BITS_LENGTH = 4


# The finalization rules of pe:841-852 as data: (justified bits that must all be set, which of the two checkpoints held BEFORE
# this epoch's update is the source, how many epochs back that source must lie).  Rules apply in this order; a later
# match overrides an earlier one, as the reference's four consecutive `if`s do.
_FINALITY_RULES = (
    (range(1, 4), "previous", 3),   # epochs 2-4 back justified, the 2nd finalizes its source, the 4th
    (range(1, 3), "previous", 2),   # epochs 2-3 back justified, source = the 3rd
    (range(0, 3), "current", 2),    # epochs 1-3 back justified, source = the 3rd
    (range(0, 2), "current", 1),    # epochs 1-2 back justified, source = the 2nd
)


def weigh_justification_and_finalization(state, total_active_balance: int, previous_epoch_target_balance: int,
                                         current_epoch_target_balance: int, *, get_block_root: Callable,
                                         slots_per_epoch: int) -> None:
    """The FFG verdict of pe:815-853 on the three Gwei sums `pe_ffg_balances` delivers.  Same state transition as the
    reference's function (tests/test_gpu_forkchoice.py runs both on the same states), written as two tables: which epoch
    a 2/3 supermajority justifies (bit position = epochs back), and `_FINALITY_RULES`."""
    epoch_now = int(state.slot) // slots_per_epoch
    epoch_before = max(epoch_now - 1, GENESIS_EPOCH)
    held = {"previous": state.previous_justified_checkpoint, "current": state.current_justified_checkpoint}
    make_checkpoint = type(held["current"])

    # justification: the bit vector ages by one epoch, then a supermajority of the target balance sets its epoch's bit
    aged = [False] + [bool(x) for x in state.justification_bits[:JUSTIFICATION_BITS_LENGTH - 1]]
    state.previous_justified_checkpoint = held["current"]
    for bit, epoch, target_balance in ((1, epoch_before, previous_epoch_target_balance),
                                       (0, epoch_now, current_epoch_target_balance)):
        if 3 * target_balance >= 2 * total_active_balance:
            aged[bit] = True
            state.current_justified_checkpoint = make_checkpoint(epoch=epoch, root=get_block_root(state, epoch))
    for i, v in enumerate(aged):
        state.justification_bits[i] = v

    # finalization
    for needed, source, distance in _FINALITY_RULES:
        if all(aged[i] for i in needed) and held[source].epoch + distance == epoch_now:
            state.finalized_checkpoint = held[source]


def process_justification_and_finalization(state, *, get_block_root: Callable) -> None:
    """pe:791-802.  ``state`` must have been bound with ``bind_state`` (its participation arrays live on the GPU);
    the three balance sums come from one streaming kernel over the working-state registry view."""
    b = _binding(state)
    assert b is not None, "process_justification_and_finalization: bind_state(engine, state, ...) first"
    spe = int(b.engine.cfg.slots_per_epoch)
    if int(state.slot) // spe <= GENESIS_EPOCH + 1:
        return
    total_active_balance, previous_target_balance, current_target_balance = b.engine.ffg_balances()
    state._last_ffg_balances = (total_active_balance, previous_target_balance, current_target_balance)
    weigh_justification_and_finalization(state, total_active_balance, previous_target_balance,
                                         current_target_balance, get_block_root=get_block_root, slots_per_epoch=spe)
