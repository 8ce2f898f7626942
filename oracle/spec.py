"""L0 oracle: literal, executable restatement of the pos-evolution pyspec excerpts.

ORACLE / TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this; the product path
(``pos_evolution_amd/``) never does.

Citation convention: ``pe:N`` = ``/root/reference/pos-evolution.md`` line N.

Two kinds of function live here and every one is tagged:

* ``[REF pe:a-b]``  -- transcribed from the code box at those lines.  The body is
  the reference's, with only the stand-in types below substituted.
* ``[UPSTREAM-MEMORY A.n]`` -- the reference CALLS it but never defines it
  (SURVEY.md Appendix A).  Restated from memory of ethereum/consensus-specs
  (~v1.2.0, Bellatrix era); no copy is on disk and there is no network, so for
  these **parity is unpinned by the reference**.  What pins them are the prose
  known answers K1-K10 (tests/test_oracle_forkchoice.py).
* ``[STAND-IN]`` -- replaces machinery that is out of scope for the hot path
  (SSZ merkleisation, full state transition, pairing check).

ORACLE OF RECORD.  At import the reference's OWN text of every ``[REF]`` function (``oracle/_ref/``, generated
by ``oracle/ref_extract.py`` from ``/root/reference/pos-evolution.md``; a git-ignored build output) is executed
inside this module's namespace and REPLACES the transcriptions below: ``spec.get_head``, ``spec.on_block``,
``spec.compute_shuffled_index`` ... are then the reference's code objects calling this file's
``[UPSTREAM-MEMORY]`` / ``[STAND-IN]`` callees.  ``ORACLE_OF_RECORD`` says which is in force ("reference" or
"transcription").  The transcriptions stay as the fallback and are held equal to the reference text, AST for AST
modulo the integer-cast stand-ins, by ``tests/test_oracle_ref_pin.py``.

Stand-in types: SSZ containers are plain dataclasses, ``Root`` is 32 raw bytes,
``hash`` is SHA-256, ``hash_tree_root`` is a SHA-256 over a deterministic field
dump (NOT SSZ merkleisation -- the hot path only compares and looks up roots,
pe:1110, pe:1116), uintN are Python ints.
"""
from __future__ import annotations

import builtins as _builtins
import hashlib
from copy import copy as _shallow_copy
from copy import deepcopy
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Set, Tuple

from . import g1 as _g1

_builtin_hash = _builtins.hash

# --------------------------------------------------------------------------
# Constants (names cited in the reference, values
# [UPSTREAM-MEMORY] except where the prose confirms them: 32 slots/epoch and 64
# committees/slot pe:472-474, 32 ETH pe:110, 12 s = 3 x 4 s pe:1536).
# --------------------------------------------------------------------------
PRESETS = {
    "mainnet": dict(
        SLOTS_PER_EPOCH=32, MAX_COMMITTEES_PER_SLOT=64, TARGET_COMMITTEE_SIZE=128,
        MAX_VALIDATORS_PER_COMMITTEE=2048, SHUFFLE_ROUND_COUNT=90,
        MAX_EFFECTIVE_BALANCE=32 * 10**9, EFFECTIVE_BALANCE_INCREMENT=10**9,
        SECONDS_PER_SLOT=12, INTERVALS_PER_SLOT=3, SAFE_SLOTS_TO_UPDATE_JUSTIFIED=8,
        PROPOSER_SCORE_BOOST=40, MIN_ATTESTATION_INCLUSION_DELAY=1, MIN_SEED_LOOKAHEAD=1,
        EPOCHS_PER_HISTORICAL_VECTOR=65536, SLOTS_PER_HISTORICAL_ROOT=8192,
        BASE_REWARD_FACTOR=64, MAX_ATTESTATIONS=128, VOTE_EXPIRY_SLOTS=0,
    ),
    "minimal": dict(
        SLOTS_PER_EPOCH=8, MAX_COMMITTEES_PER_SLOT=4, TARGET_COMMITTEE_SIZE=4,
        MAX_VALIDATORS_PER_COMMITTEE=2048, SHUFFLE_ROUND_COUNT=10,
        MAX_EFFECTIVE_BALANCE=32 * 10**9, EFFECTIVE_BALANCE_INCREMENT=10**9,
        SECONDS_PER_SLOT=6, INTERVALS_PER_SLOT=3, SAFE_SLOTS_TO_UPDATE_JUSTIFIED=2,
        PROPOSER_SCORE_BOOST=40, MIN_ATTESTATION_INCLUSION_DELAY=1, MIN_SEED_LOOKAHEAD=1,
        EPOCHS_PER_HISTORICAL_VECTOR=64, SLOTS_PER_HISTORICAL_ROOT=64,
        BASE_REWARD_FACTOR=64, MAX_ATTESTATIONS=128, VOTE_EXPIRY_SLOTS=0,
    ),
}
GENESIS_EPOCH = 0
GENESIS_SLOT = 0
FAR_FUTURE_EPOCH = 2**64 - 1
DOMAIN_BEACON_ATTESTER = bytes.fromhex("01000000")
TIMELY_SOURCE_FLAG_INDEX = 0
TIMELY_TARGET_FLAG_INDEX = 1
TIMELY_HEAD_FLAG_INDEX = 2
PARTICIPATION_FLAG_WEIGHTS = [14, 26, 14]
PROPOSER_WEIGHT = 8
WEIGHT_DENOMINATOR = 64
# Era switch: this era's get_latest_attesting_balance does not
# filter slashed validators; later upstream versions (get_weight) do.
FILTER_SLASHED = False
# [VARIANT pe:1585-1596] RLMD-GHOST's vote expiry period eta, in slots: only latest messages from the most recent
# eta slots count in get_latest_attesting_balance (eta = 1: Goldfish's GHOST-Eph, pe:1549).  0 = no expiry = the
# reference's executable LMD-GHOST.  The reference gives this as prose only; the predicate below is this repo's
# restatement of it (parity unpinned).
VOTE_EXPIRY_SLOTS = 0


def use_preset(name: str, **overrides) -> None:
    """Bind the preset's constants as module globals (the pyspec's own style)."""
    vals = dict(PRESETS[name])
    vals.update(overrides)
    globals().update(vals)
    globals()["PRESET_NAME"] = name


use_preset("mainnet")



class Root(bytes):
    """32 opaque bytes; ``Root()`` is the all-zero root as in the pyspec (pe:943, pe:1083)."""

    def __new__(cls, value: bytes = b"\x00" * 32):
        assert len(value) == 32
        return super().__new__(cls, value)


ZERO_ROOT = Root()


def sha256(data: bytes) -> Root:
    """The pyspec's ``hash`` (pe:486, 522, 525).  Not named ``hash`` here: a module-level ``hash`` would
    shadow the builtin that the frozen dataclasses' generated ``__hash__`` calls."""
    return Root(hashlib.sha256(data).digest())


class uint64(int):
    """[STAND-IN] SSZ uintN: a Python int that remembers its serialised width (the pyspec's ``uint_to_bytes`` takes
    the width from the type: pe:522 hashes ONE byte of ``uint8(current_round)``, pe:525 FOUR of ``uint32(...)``).
    Arithmetic returns plain ints (unbounded; the reference's "Avoid underflow" forms stay harmless)."""
    BYTE_LEN = 8


class uint32(int):
    BYTE_LEN = 4


class uint8(int):
    BYTE_LEN = 1


Epoch = Slot = Gwei = ValidatorIndex = CommitteeIndex = uint64  # [STAND-IN] SSZ aliases used as casts (pe:485, pe:752)


def uint_to_bytes(n: int, length: Optional[int] = None) -> bytes:
    """[UPSTREAM-MEMORY A.10] little-endian, width of the SSZ type (plain ints are uint64: every untyped call site
    in the reference passes an Epoch, pe:486)."""
    if length is None:
        length = getattr(type(n), "BYTE_LEN", 8)
    return int(n).to_bytes(length, "little")


def hash(data):  # noqa: A001 -- the pyspec's name (pe:486, 522, 525)
    """The pyspec's ``hash`` = SHA-256 for the reference's own code (oracle/_ref).  Anything that is not a byte
    string goes to the builtin: the frozen dataclasses' generated ``__hash__`` calls ``hash((fields...))`` through
    this module's globals."""
    if isinstance(data, (bytes, bytearray)):
        return sha256(bytes(data))
    return _builtin_hash(data)


def bytes_to_uint64(data: bytes) -> int:
    """[UPSTREAM-MEMORY A.10]"""
    return int.from_bytes(data, "little")


def integer_squareroot(n: int) -> int:
    """[UPSTREAM-MEMORY] largest x with x*x <= n (Newton iteration as upstream)."""
    x = n
    y = (x + 1) // 2
    while y < x:
        x = y
        y = (x + n // x) // 2
    return x


# --------------------------------------------------------------------------
# L0 data model
# --------------------------------------------------------------------------
@dataclass
class Validator:
    """[REF pe:36-45]  pubkey here is an affine G1 point (x, y) or None."""
    pubkey: object = None
    withdrawal_credentials: bytes = ZERO_ROOT
    effective_balance: int = 0
    slashed: bool = False
    activation_eligibility_epoch: int = 0
    activation_epoch: int = 0
    exit_epoch: int = FAR_FUTURE_EPOCH
    withdrawable_epoch: int = FAR_FUTURE_EPOCH


@dataclass(eq=True, frozen=True)
class Checkpoint:
    """[REF pe:219-221]"""
    epoch: int = 0
    root: Root = ZERO_ROOT


@dataclass(eq=True, frozen=True)
class LatestMessage(object):
    """[REF pe:286-289]"""
    epoch: int
    root: Root


@dataclass(eq=True, frozen=True)
class AttestationData:
    """[REF pe:689-697]"""
    slot: int = 0
    index: int = 0
    beacon_block_root: Root = ZERO_ROOT
    source: Checkpoint = Checkpoint()
    target: Checkpoint = Checkpoint()


@dataclass
class Attestation:
    """[REF pe:714-717]  signature stand-in: ``signature_valid`` is the injected
    result of the (out-of-scope) pairing check; ``signature`` may carry a G1 point."""
    aggregation_bits: List[bool] = field(default_factory=list)
    data: AttestationData = AttestationData()
    signature: object = None
    signature_valid: bool = True


@dataclass
class IndexedAttestation:
    """[UPSTREAM-MEMORY A.6] container shape."""
    attesting_indices: List[int] = field(default_factory=list)
    data: AttestationData = AttestationData()
    signature: object = None
    signature_valid: bool = True


@dataclass
class AttesterSlashing:
    """[REF pe:1159-1162]"""
    attestation_1: IndexedAttestation = None
    attestation_2: IndexedAttestation = None


@dataclass
class BeaconBlockBody:
    """[REF pe:632-645] only the field the hot path reads."""
    attestations: List[Attestation] = field(default_factory=list)
    graffiti: bytes = b""


@dataclass
class BeaconBlock:
    """[REF pe:671-676]"""
    slot: int = 0
    proposer_index: int = 0
    parent_root: Root = ZERO_ROOT
    state_root: Root = ZERO_ROOT
    body: BeaconBlockBody = field(default_factory=BeaconBlockBody)


@dataclass
class SignedBeaconBlock:
    message: BeaconBlock = None
    signature: object = None
    # [STAND-IN] scripted post-state checkpoints (justified, finalized) applied by
    # the stand-in state_transition: the real FFG epoch processing (pe:793-853)
    # is out of scope and scenarios need control over them.
    scripted_checkpoints: Optional[Tuple[Checkpoint, Checkpoint]] = None


@dataclass
class BeaconState:
    """[REF pe:338-375] restricted to the fields the hot path reads."""
    genesis_time: int = 0
    slot: int = 0
    validators: List[Validator] = field(default_factory=list)
    balances: List[int] = field(default_factory=list)
    randao_mixes: List[bytes] = field(default_factory=list)
    block_roots: Dict[int, Root] = field(default_factory=dict)  # slot -> root (sparse stand-in)
    previous_epoch_participation: List[int] = field(default_factory=list)
    current_epoch_participation: List[int] = field(default_factory=list)
    previous_justified_checkpoint: Checkpoint = Checkpoint()
    current_justified_checkpoint: Checkpoint = Checkpoint()
    finalized_checkpoint: Checkpoint = Checkpoint()
    latest_block_root: Root = ZERO_ROOT
    proposer_index_override: Optional[int] = None
    justification_bits: List[bool] = field(default_factory=lambda: [False] * 4)  # JUSTIFICATION_BITS_LENGTH = 4 (pe:403)

    def copy(self) -> "BeaconState":
        return deepcopy(self)


def copy(obj):
    """The pyspec's ``copy`` (pe:992, pe:1092): value copy of an SSZ object."""
    return deepcopy(obj)


def hash_tree_root(obj) -> Root:
    """[STAND-IN] deterministic SHA-256 over a field dump, NOT SSZ merkleisation."""
    if isinstance(obj, BeaconBlock):
        body = hashlib.sha256(
            repr([(a.aggregation_bits, a.data) for a in obj.body.attestations]).encode() + obj.body.graffiti
        ).digest()
        return sha256(
            b"blk" + uint_to_bytes(obj.slot) + uint_to_bytes(obj.proposer_index)
            + obj.parent_root + obj.state_root + body
        )
    if isinstance(obj, BeaconState):
        return sha256(
            b"st" + uint_to_bytes(obj.genesis_time) + uint_to_bytes(obj.slot)
            + uint_to_bytes(len(obj.validators))
        )
    raise TypeError(type(obj))


# --------------------------------------------------------------------------
# Small helpers [UPSTREAM-MEMORY A.10]
# --------------------------------------------------------------------------
def compute_epoch_at_slot(slot: int) -> int:
    return slot // SLOTS_PER_EPOCH


def compute_start_slot_at_epoch(epoch: int) -> int:
    return epoch * SLOTS_PER_EPOCH


def compute_slots_since_epoch_start(slot: int) -> int:
    return slot - compute_start_slot_at_epoch(compute_epoch_at_slot(slot))


def get_current_epoch(state: BeaconState) -> int:
    return compute_epoch_at_slot(state.slot)


def get_previous_epoch(state: BeaconState) -> int:
    current_epoch = get_current_epoch(state)
    return GENESIS_EPOCH if current_epoch == GENESIS_EPOCH else current_epoch - 1


def get_current_slot(store: "Store") -> int:
    return GENESIS_SLOT + (store.time - store.genesis_time) // SECONDS_PER_SLOT


def get_randao_mix(state: BeaconState, epoch: int) -> bytes:
    return state.randao_mixes[epoch % EPOCHS_PER_HISTORICAL_VECTOR]


def is_active_validator(validator: Validator, epoch: int) -> bool:
    return validator.activation_epoch <= epoch < validator.exit_epoch


def get_active_validator_indices(state: BeaconState, epoch: int) -> List[int]:
    """[UPSTREAM-MEMORY A.6]"""
    return [i for i, v in enumerate(state.validators) if is_active_validator(v, epoch)]


def get_total_balance(state: BeaconState, indices) -> int:
    return max(EFFECTIVE_BALANCE_INCREMENT, sum(state.validators[i].effective_balance for i in indices))


def get_total_active_balance(state: BeaconState) -> int:
    """[UPSTREAM-MEMORY A.1]"""
    return get_total_balance(state, set(get_active_validator_indices(state, get_current_epoch(state))))


# --------------------------------------------------------------------------
# Committees  [REF pe:461-468, 481-486, 495-504, 513-534]
# --------------------------------------------------------------------------
def get_committee_count_per_slot(state: BeaconState, epoch: int) -> int:
    """[REF pe:461-468]"""
    return max(1, min(
        MAX_COMMITTEES_PER_SLOT,
        len(get_active_validator_indices(state, epoch)) // SLOTS_PER_EPOCH // TARGET_COMMITTEE_SIZE,
    ))


def get_seed(state: BeaconState, epoch: int, domain_type: bytes) -> bytes:
    """[REF pe:481-486]"""
    mix = get_randao_mix(state, epoch + EPOCHS_PER_HISTORICAL_VECTOR - MIN_SEED_LOOKAHEAD - 1)  # Avoid underflow
    return sha256(domain_type + uint_to_bytes(epoch) + mix)


def compute_shuffled_index(index: int, index_count: int, seed: bytes) -> int:
    """[REF pe:513-534] swap-or-not, SHUFFLE_ROUND_COUNT rounds."""
    assert index < index_count

    for current_round in range(SHUFFLE_ROUND_COUNT):
        pivot = bytes_to_uint64(sha256(seed + uint_to_bytes(current_round, 1))[0:8]) % index_count
        flip = (pivot + index_count - index) % index_count
        position = max(index, flip)
        source = sha256(
            seed
            + uint_to_bytes(current_round, 1)
            + uint_to_bytes(position // 256, 4)
        )
        byte = source[(position % 256) // 8]
        bit = (byte >> (position % 8)) % 2
        index = flip if bit else index

    return index


def compute_committee(indices: Sequence[int], seed: bytes, index: int, count: int) -> List[int]:
    """[REF pe:495-504]"""
    start = (len(indices) * index) // count
    end = (len(indices) * (index + 1)) // count
    return [indices[compute_shuffled_index(i, len(indices), seed)] for i in range(start, end)]


_committee_cache: Dict[tuple, List[int]] = {}


def get_beacon_committee(state: BeaconState, slot: int, index: int) -> List[int]:
    """[UPSTREAM-MEMORY A.6]  (memoised on (seed, active set, position): the literal
    form recomputes the whole shuffle per call, pe:504.)"""
    epoch = compute_epoch_at_slot(slot)
    committees_per_slot = get_committee_count_per_slot(state, epoch)
    indices = get_active_validator_indices(state, epoch)
    seed = get_seed(state, epoch, DOMAIN_BEACON_ATTESTER)
    pos = (slot % SLOTS_PER_EPOCH) * committees_per_slot + index
    count = committees_per_slot * SLOTS_PER_EPOCH
    key = (seed, sha256(repr(indices).encode()), pos, count, SHUFFLE_ROUND_COUNT)
    if key not in _committee_cache:
        _committee_cache[key] = compute_committee(indices=indices, seed=seed, index=pos, count=count)
    return list(_committee_cache[key])


def get_attesting_indices(state: BeaconState, data: AttestationData, bits: Sequence[bool]) -> Set[int]:
    """[UPSTREAM-MEMORY A.6]"""
    committee = get_beacon_committee(state, data.slot, data.index)
    return set(index for i, index in enumerate(committee) if bits[i])


def get_indexed_attestation(state: BeaconState, attestation: Attestation) -> IndexedAttestation:
    """[UPSTREAM-MEMORY A.6]"""
    attesting_indices = get_attesting_indices(state, attestation.data, attestation.aggregation_bits)
    return IndexedAttestation(
        attesting_indices=sorted(attesting_indices),
        data=attestation.data,
        signature=attestation.signature,
        signature_valid=attestation.signature_valid,
    )


def aggregate_pubkeys(state: BeaconState, indices: Sequence[int]):
    """The G1 sum inside FastAggregateVerify [UPSTREAM-MEMORY A.7]: sum of the
    attesters' pubkeys.  This is the quantity the engine's G1 kernels are pinned to."""
    return _g1.sum_points(state.validators[i].pubkey for i in indices)


def is_valid_indexed_attestation(state: BeaconState, indexed_attestation: IndexedAttestation) -> bool:
    """[UPSTREAM-MEMORY A.7]  structural checks are literal; the pairing equality
    e(sum pk, H(m)) == e(G, sig) is [STAND-IN]: its boolean is injected
    (``signature_valid``); the G1 sum it consumes is ``aggregate_pubkeys``."""
    indices = indexed_attestation.attesting_indices
    if len(indices) == 0 or not indices == sorted(set(indices)):
        return False
    return bool(indexed_attestation.signature_valid)


# --------------------------------------------------------------------------
# Aggregation (validator guide)  [UPSTREAM-MEMORY A.8; reference: prose only,
# pe:474, pe:659, pe:715, pe:1536; shape check pe:730]
# --------------------------------------------------------------------------
def aggregate_attestations(attestations: Sequence[Attestation]) -> Attestation:
    """OR the bitlists, add the signature points (bls.Aggregate = point addition).
    All inputs must carry identical ``data`` and equal-length bitlists."""
    assert len(attestations) > 0
    data = attestations[0].data
    n = len(attestations[0].aggregation_bits)
    bits = [False] * n
    sig = None
    valid = True
    for a in attestations:
        assert a.data == data and len(a.aggregation_bits) == n
        bits = [x or y for x, y in zip(bits, a.aggregation_bits)]
        if a.signature is not None or sig is not None:
            sig = _g1.add(sig, a.signature)
        valid = valid and a.signature_valid
    return Attestation(aggregation_bits=bits, data=data, signature=sig, signature_valid=valid)


# --------------------------------------------------------------------------
# Participation / rewards (Altair)  [UPSTREAM-MEMORY A.9]
# --------------------------------------------------------------------------
def has_flag(flags: int, flag_index: int) -> bool:
    flag = 2**flag_index
    return flags & flag == flag


def add_flag(flags: int, flag_index: int) -> int:
    flag = 2**flag_index
    return flags | flag


def get_block_root_at_slot(state: BeaconState, slot: int) -> Root:
    assert slot < state.slot <= slot + SLOTS_PER_HISTORICAL_ROOT
    # sparse stand-in for the block_roots ring: root of the latest block at or before slot
    s = slot
    while s not in state.block_roots:
        assert s > 0
        s -= 1
    return state.block_roots[s]


def get_block_root(state: BeaconState, epoch: int) -> Root:
    return get_block_root_at_slot(state, compute_start_slot_at_epoch(epoch))


def get_attestation_participation_flag_indices(state: BeaconState, data: AttestationData,
                                               inclusion_delay: int) -> List[int]:
    """[UPSTREAM-MEMORY A.9]"""
    if data.target.epoch == get_current_epoch(state):
        justified_checkpoint = state.current_justified_checkpoint
    else:
        justified_checkpoint = state.previous_justified_checkpoint

    is_matching_source = data.source == justified_checkpoint
    is_matching_target = is_matching_source and data.target.root == get_block_root(state, data.target.epoch)
    is_matching_head = is_matching_target and data.beacon_block_root == get_block_root_at_slot(state, data.slot)
    assert is_matching_source

    participation_flag_indices = []
    if is_matching_source and inclusion_delay <= integer_squareroot(SLOTS_PER_EPOCH):
        participation_flag_indices.append(TIMELY_SOURCE_FLAG_INDEX)
    if is_matching_target and inclusion_delay <= SLOTS_PER_EPOCH:
        participation_flag_indices.append(TIMELY_TARGET_FLAG_INDEX)
    if is_matching_head and inclusion_delay == MIN_ATTESTATION_INCLUSION_DELAY:
        participation_flag_indices.append(TIMELY_HEAD_FLAG_INDEX)
    return participation_flag_indices


def get_base_reward_per_increment(state: BeaconState) -> int:
    return EFFECTIVE_BALANCE_INCREMENT * BASE_REWARD_FACTOR // integer_squareroot(get_total_active_balance(state))


def get_base_reward(state: BeaconState, index: int) -> int:
    increments = state.validators[index].effective_balance // EFFECTIVE_BALANCE_INCREMENT
    return increments * get_base_reward_per_increment(state)


def get_beacon_proposer_index(state: BeaconState) -> int:
    """[STAND-IN] compute_proposer_index (pe:604-618) is out of scope (once per
    slot, scalar); scenarios pin the proposer explicitly."""
    if state.proposer_index_override is not None:
        return state.proposer_index_override
    return 0


def increase_balance(state: BeaconState, index: int, delta: int) -> None:
    state.balances[index] += delta


def process_attestation(state: BeaconState, attestation: Attestation) -> None:
    """[REF pe:722-754]"""
    data = attestation.data
    assert data.target.epoch in (get_previous_epoch(state), get_current_epoch(state))
    assert data.target.epoch == compute_epoch_at_slot(data.slot)
    assert data.slot + MIN_ATTESTATION_INCLUSION_DELAY <= state.slot <= data.slot + SLOTS_PER_EPOCH
    assert data.index < get_committee_count_per_slot(state, data.target.epoch)

    committee = get_beacon_committee(state, data.slot, data.index)
    assert len(attestation.aggregation_bits) == len(committee)

    # Participation flag indices
    participation_flag_indices = get_attestation_participation_flag_indices(state, data, state.slot - data.slot)

    # Verify signature
    assert is_valid_indexed_attestation(state, get_indexed_attestation(state, attestation))

    # Update epoch participation flags
    if data.target.epoch == get_current_epoch(state):
        epoch_participation = state.current_epoch_participation
    else:
        epoch_participation = state.previous_epoch_participation

    proposer_reward_numerator = 0
    for index in get_attesting_indices(state, data, attestation.aggregation_bits):
        for flag_index, weight in enumerate(PARTICIPATION_FLAG_WEIGHTS):
            if flag_index in participation_flag_indices and not has_flag(epoch_participation[index], flag_index):
                epoch_participation[index] = add_flag(epoch_participation[index], flag_index)
                proposer_reward_numerator += get_base_reward(state, index) * weight

    # Reward proposer
    proposer_reward_denominator = (WEIGHT_DENOMINATOR - PROPOSER_WEIGHT) * WEIGHT_DENOMINATOR // PROPOSER_WEIGHT
    proposer_reward = proposer_reward_numerator // proposer_reward_denominator
    increase_balance(state, get_beacon_proposer_index(state), proposer_reward)


# --------------------------------------------------------------------------
# FFG: justification and finalization  [REF pe:791-802, pe:815-853]
# --------------------------------------------------------------------------
JUSTIFICATION_BITS_LENGTH = 4


def get_unslashed_participating_indices(state: BeaconState, flag_index: int, epoch: int) -> Set[int]:
    """[UPSTREAM-MEMORY, Altair]  (described in prose at pe:805)"""
    assert epoch in (get_previous_epoch(state), get_current_epoch(state))
    if epoch == get_current_epoch(state):
        epoch_participation = state.current_epoch_participation
    else:
        epoch_participation = state.previous_epoch_participation
    active_validator_indices = get_active_validator_indices(state, epoch)
    participating_indices = [i for i in active_validator_indices if has_flag(epoch_participation[i], flag_index)]
    return set(filter(lambda index: not state.validators[index].slashed, participating_indices))


def weigh_justification_and_finalization(state: BeaconState, total_active_balance: int,
                                         previous_epoch_target_balance: int, current_epoch_target_balance: int) -> None:
    """[REF pe:815-853]"""
    previous_epoch = get_previous_epoch(state)
    current_epoch = get_current_epoch(state)
    old_previous_justified_checkpoint = state.previous_justified_checkpoint
    old_current_justified_checkpoint = state.current_justified_checkpoint

    # Process justifications
    state.previous_justified_checkpoint = state.current_justified_checkpoint
    state.justification_bits[1:] = state.justification_bits[:JUSTIFICATION_BITS_LENGTH - 1]
    state.justification_bits[0] = 0b0
    if previous_epoch_target_balance * 3 >= total_active_balance * 2:
        state.current_justified_checkpoint = Checkpoint(epoch=previous_epoch,
                                                        root=get_block_root(state, previous_epoch))
        state.justification_bits[1] = 0b1
    if current_epoch_target_balance * 3 >= total_active_balance * 2:
        state.current_justified_checkpoint = Checkpoint(epoch=current_epoch,
                                                        root=get_block_root(state, current_epoch))
        state.justification_bits[0] = 0b1

    # Process finalizations
    bits = state.justification_bits
    # The 2nd/3rd/4th most recent epochs are justified, the 2nd using the 4th as source
    if all(bits[1:4]) and old_previous_justified_checkpoint.epoch + 3 == current_epoch:
        state.finalized_checkpoint = old_previous_justified_checkpoint
    # The 2nd/3rd most recent epochs are justified, the 2nd using the 3rd as source
    if all(bits[1:3]) and old_previous_justified_checkpoint.epoch + 2 == current_epoch:
        state.finalized_checkpoint = old_previous_justified_checkpoint
    # The 1st/2nd/3rd most recent epochs are justified, the 1st using the 3rd as source
    if all(bits[0:3]) and old_current_justified_checkpoint.epoch + 2 == current_epoch:
        state.finalized_checkpoint = old_current_justified_checkpoint
    # The 1st/2nd most recent epochs are justified, the 1st using the 2nd as source
    if all(bits[0:2]) and old_current_justified_checkpoint.epoch + 1 == current_epoch:
        state.finalized_checkpoint = old_current_justified_checkpoint


def process_justification_and_finalization(state: BeaconState) -> None:
    """[REF pe:791-802]"""
    # Initial FFG checkpoint values have a `0x00` stub for `root`.
    # Skip FFG updates in the first two epochs to avoid corner cases that might result in modifying this stub.
    if get_current_epoch(state) <= GENESIS_EPOCH + 1:
        return
    previous_indices = get_unslashed_participating_indices(state, TIMELY_TARGET_FLAG_INDEX, get_previous_epoch(state))
    current_indices = get_unslashed_participating_indices(state, TIMELY_TARGET_FLAG_INDEX, get_current_epoch(state))
    total_active_balance = get_total_active_balance(state)
    previous_target_balance = get_total_balance(state, previous_indices)
    current_target_balance = get_total_balance(state, current_indices)
    weigh_justification_and_finalization(state, total_active_balance, previous_target_balance, current_target_balance)


# --------------------------------------------------------------------------
# [STAND-IN] state transition: the full transition (pe:412-424 -> process_slots,
# process_block, FFG pe:793-853) is out of scope.  This keeps exactly what the
# fork-choice handlers observe: slot advance, participation rotation at epoch
# boundaries, block-root bookkeeping, process_attestation over the body, and
# scripted post-state checkpoints.
# --------------------------------------------------------------------------
def process_slots(state: BeaconState, slot: int) -> None:
    assert state.slot < slot
    while state.slot < slot:
        if (state.slot + 1) % SLOTS_PER_EPOCH == 0:
            # process_participation_flag_updates
            state.previous_epoch_participation = state.current_epoch_participation
            state.current_epoch_participation = [0] * len(state.validators)
        state.slot += 1


def state_transition(state: BeaconState, signed_block: SignedBeaconBlock, validate_result: bool = True) -> None:
    block = signed_block.message
    process_slots(state, block.slot)
    for attestation in block.body.attestations:
        process_attestation(state, attestation)
    if signed_block.scripted_checkpoints is not None:
        justified, finalized = signed_block.scripted_checkpoints
        if justified != state.current_justified_checkpoint:
            state.previous_justified_checkpoint = state.current_justified_checkpoint
            state.current_justified_checkpoint = justified
        state.finalized_checkpoint = finalized
    root = hash_tree_root(block)
    state.block_roots[block.slot] = root
    state.latest_block_root = root


# --------------------------------------------------------------------------
# Fork choice
# --------------------------------------------------------------------------
@dataclass
class Store(object):
    """[REF pe:889-901]"""
    time: int
    genesis_time: int
    justified_checkpoint: Checkpoint
    finalized_checkpoint: Checkpoint
    best_justified_checkpoint: Checkpoint
    proposer_boost_root: Root
    equivocating_indices: Set[int]
    blocks: Dict[Root, BeaconBlock] = field(default_factory=dict)
    block_states: Dict[Root, BeaconState] = field(default_factory=dict)
    checkpoint_states: Dict[Checkpoint, BeaconState] = field(default_factory=dict)
    latest_messages: Dict[int, LatestMessage] = field(default_factory=dict)


def get_forkchoice_store(anchor_state: BeaconState, anchor_block: BeaconBlock) -> Store:
    """[REF pe:1077-1095]"""
    assert anchor_block.state_root == hash_tree_root(anchor_state)
    anchor_root = hash_tree_root(anchor_block)
    anchor_epoch = get_current_epoch(anchor_state)
    justified_checkpoint = Checkpoint(epoch=anchor_epoch, root=anchor_root)
    finalized_checkpoint = Checkpoint(epoch=anchor_epoch, root=anchor_root)
    proposer_boost_root = Root()
    return Store(
        time=anchor_state.genesis_time + SECONDS_PER_SLOT * anchor_state.slot,
        genesis_time=anchor_state.genesis_time,
        justified_checkpoint=justified_checkpoint,
        finalized_checkpoint=finalized_checkpoint,
        best_justified_checkpoint=justified_checkpoint,
        proposer_boost_root=proposer_boost_root,
        equivocating_indices=set(),
        blocks={anchor_root: copy(anchor_block)},
        block_states={anchor_root: copy(anchor_state)},
        checkpoint_states={justified_checkpoint: copy(anchor_state)},
    )


def get_ancestor(store: Store, root: Root, slot: int) -> Root:
    """[UPSTREAM-MEMORY A.2] (iterative form of the recursive upstream text)."""
    block = store.blocks[root]
    while block.slot > slot:
        root = block.parent_root
        block = store.blocks[root]
    # block.slot == slot -> itself; block.slot < slot -> skip-slot case: most recent root prior to slot
    return root


def is_vote_expired(store: Store, index: int) -> bool:
    """[VARIANT pe:1585-1596] "only messages from the most recent eta slots are utilized".  The slot of the
    attestation behind a latest message is not part of the reference's ``LatestMessage`` (pe:286-289): the variant
    keeps it in a side table, ``store.latest_message_slots`` (see ``_record_message_slots``)."""
    if VOTE_EXPIRY_SLOTS == 0:
        return False
    return store.__dict__.get("latest_message_slots", {}).get(index, 0) + VOTE_EXPIRY_SLOTS < get_current_slot(store)


def get_latest_attesting_balance(store: Store, root: Root) -> int:
    """[UPSTREAM-MEMORY A.1]"""
    state = store.checkpoint_states[store.justified_checkpoint]
    active_indices = get_active_validator_indices(state, get_current_epoch(state))
    attestation_score = sum(
        state.validators[i].effective_balance for i in active_indices
        if (i in store.latest_messages
            and i not in store.equivocating_indices
            and not (FILTER_SLASHED and state.validators[i].slashed)
            and not is_vote_expired(store, i)
            and get_ancestor(store, store.latest_messages[i].root, store.blocks[root].slot) == root)
    )
    if store.proposer_boost_root == Root():
        # Return only attestation score if ``proposer_boost_root`` is not set
        return attestation_score

    # Calculate proposer score if ``proposer_boost_root`` is set
    proposer_score = 0
    # Boost is applied if ``root`` is an ancestor of ``proposer_boost_root``
    if get_ancestor(store, store.proposer_boost_root, store.blocks[root].slot) == root:
        num_validators = len(get_active_validator_indices(state, get_current_epoch(state)))
        avg_balance = get_total_active_balance(state) // num_validators
        committee_size = num_validators // SLOTS_PER_EPOCH
        committee_weight = committee_size * avg_balance
        proposer_score = (committee_weight * PROPOSER_SCORE_BOOST) // 100
    return attestation_score + proposer_score


def filter_block_tree(store: Store, block_root: Root, blocks: Dict[Root, BeaconBlock]) -> bool:
    """[UPSTREAM-MEMORY A.3]"""
    block = store.blocks[block_root]
    children = [
        root for root in store.blocks.keys()
        if store.blocks[root].parent_root == block_root
    ]

    # If any children branches contain expected finalized/justified checkpoints,
    # add to filtered block-tree and signal viability to parent.
    if any(children):
        filter_block_tree_result = [filter_block_tree(store, child, blocks) for child in children]
        if any(filter_block_tree_result):
            blocks[block_root] = block
            return True
        return False

    # If leaf block, check finalized/justified checkpoints as matching latest.
    head_state = store.block_states[block_root]

    correct_justified = (
        store.justified_checkpoint.epoch == GENESIS_EPOCH
        or head_state.current_justified_checkpoint == store.justified_checkpoint
    )
    correct_finalized = (
        store.finalized_checkpoint.epoch == GENESIS_EPOCH
        or head_state.finalized_checkpoint == store.finalized_checkpoint
    )
    # If expected finalized/justified, add to viable block-tree and signal viability to parent.
    if correct_justified and correct_finalized:
        blocks[block_root] = block
        return True

    # Otherwise, branch not viable
    return False


def get_filtered_block_tree(store: Store) -> Dict[Root, BeaconBlock]:
    """[UPSTREAM-MEMORY A.3]"""
    base = store.justified_checkpoint.root
    blocks: Dict[Root, BeaconBlock] = {}
    filter_block_tree(store, base, blocks)
    return blocks


def get_head(store: Store) -> Root:
    """[REF pe:1102-1116]"""
    # Get filtered block-tree that only includes viable branches
    blocks = get_filtered_block_tree(store)
    # Execute the LMD-GHOST fork-choice
    head = store.justified_checkpoint.root
    while True:
        children = [
            root for root in blocks.keys()
            if blocks[root].parent_root == head
        ]
        if len(children) == 0:
            return head
        # Sort by latest attesting balance with ties broken lexicographically
        # Ties broken by favoring block with lexicographically higher root
        head = max(children, key=lambda root: (get_latest_attesting_balance(store, root), root))


def should_update_justified_checkpoint(store: Store, new_justified_checkpoint: Checkpoint) -> bool:
    """[REF pe:1046-1061]"""
    if compute_slots_since_epoch_start(get_current_slot(store)) < SAFE_SLOTS_TO_UPDATE_JUSTIFIED:
        return True

    justified_slot = compute_start_slot_at_epoch(store.justified_checkpoint.epoch)
    if not get_ancestor(store, new_justified_checkpoint.root, justified_slot) == store.justified_checkpoint.root:
        return False

    return True


def validate_target_epoch_against_current_time(store: Store, attestation: Attestation) -> None:
    """[UPSTREAM-MEMORY A.4]"""
    target = attestation.data.target

    # Attestations must be from the current or previous epoch
    current_epoch = compute_epoch_at_slot(get_current_slot(store))
    # Use GENESIS_EPOCH for previous when genesis to avoid underflow
    previous_epoch = current_epoch - 1 if current_epoch > GENESIS_EPOCH else GENESIS_EPOCH
    # If attestation target is from a future epoch, delay consideration until the epoch arrives
    assert target.epoch in [current_epoch, previous_epoch]


def validate_on_attestation(store: Store, attestation: Attestation, is_from_block: bool) -> None:
    """[UPSTREAM-MEMORY A.4]"""
    target = attestation.data.target

    # If the given attestation is not from a beacon block message, we have to check the target epoch scope.
    if not is_from_block:
        validate_target_epoch_against_current_time(store, attestation)

    # Check that the epoch number and slot number are matching
    assert target.epoch == compute_epoch_at_slot(attestation.data.slot)

    # Attestations target be for a known block. If target block is unknown, delay consideration until the block is found
    assert target.root in store.blocks

    # Attestations must be for a known block. If block is unknown, delay consideration until the block is found
    assert attestation.data.beacon_block_root in store.blocks
    # Attestations must not be for blocks in the future. If not, the attestation should not be considered
    assert store.blocks[attestation.data.beacon_block_root].slot <= attestation.data.slot

    # LMD vote must be consistent with FFG vote target
    target_slot = compute_start_slot_at_epoch(target.epoch)
    assert target.root == get_ancestor(store, attestation.data.beacon_block_root, target_slot)

    # Attestations can only affect the fork choice of subsequent slots.
    # Delay consideration in the fork choice until their slot is in the past.
    assert get_current_slot(store) >= attestation.data.slot + 1


def store_target_checkpoint_state(store: Store, target: Checkpoint) -> None:
    """[UPSTREAM-MEMORY A.5]"""
    # Store target checkpoint state if not yet seen
    if target not in store.checkpoint_states:
        base_state = copy(store.block_states[target.root])
        if base_state.slot < compute_start_slot_at_epoch(target.epoch):
            process_slots(base_state, compute_start_slot_at_epoch(target.epoch))
        store.checkpoint_states[target] = base_state


def update_latest_messages(store: Store, attesting_indices: Sequence[int], attestation: Attestation) -> None:
    """[REF pe:1435-1441]"""
    target = attestation.data.target
    beacon_block_root = attestation.data.beacon_block_root
    non_equivocating_attesting_indices = [i for i in attesting_indices if i not in store.equivocating_indices]
    for i in non_equivocating_attesting_indices:
        if i not in store.latest_messages or target.epoch > store.latest_messages[i].epoch:
            store.latest_messages[i] = LatestMessage(epoch=target.epoch, root=beacon_block_root)


def on_tick(store: Store, time: int) -> None:
    """[REF pe:934-955]"""
    previous_slot = get_current_slot(store)

    # update store time
    store.time = time

    current_slot = get_current_slot(store)

    # Reset store.proposer_boost_root if this is a new slot
    if current_slot > previous_slot:
        store.proposer_boost_root = Root()

    # Not a new epoch, return
    if not (current_slot > previous_slot and compute_slots_since_epoch_start(current_slot) == 0):
        return

    # Update store.justified_checkpoint if a better checkpoint on the store.finalized_checkpoint chain
    if store.best_justified_checkpoint.epoch > store.justified_checkpoint.epoch:
        finalized_slot = compute_start_slot_at_epoch(store.finalized_checkpoint.epoch)
        ancestor_at_finalized_slot = get_ancestor(store, store.best_justified_checkpoint.root, finalized_slot)
        if ancestor_at_finalized_slot == store.finalized_checkpoint.root:
            store.justified_checkpoint = store.best_justified_checkpoint


def on_attestation(store: Store, attestation: Attestation, is_from_block: bool = False) -> None:
    """[REF pe:963-979 + pe:1423-1428 (is_from_block form)]

    Invalid calls must not modify ``store`` (pe:1041): ``validate_on_attestation``
    only reads; ``store_target_checkpoint_state`` adds a derived cache entry
    (as upstream does) which does not change any fork-choice result."""
    validate_on_attestation(store, attestation, is_from_block)
    store_target_checkpoint_state(store, attestation.data.target)

    # Get state at the `target` to fully validate attestation
    target_state = store.checkpoint_states[attestation.data.target]
    indexed_attestation = get_indexed_attestation(target_state, attestation)
    assert is_valid_indexed_attestation(target_state, indexed_attestation)

    # Update latest messages for attesting indices
    update_latest_messages(store, indexed_attestation.attesting_indices, attestation)


def on_block(store: Store, signed_block: SignedBeaconBlock) -> None:
    """[REF pe:986-1036]  (the Bellatrix merge-transition check pe:1011-1013 calls two [STAND-IN]s:
    execution-layer validation is out of scope)."""
    block = signed_block.message
    # Parent block must be known
    assert block.parent_root in store.block_states
    # Make a copy of the state to avoid mutability issues
    pre_state = copy(store.block_states[block.parent_root])
    # Blocks cannot be in the future. If they are, their consideration must be delayed until they are in the past.
    assert get_current_slot(store) >= block.slot

    # Check that block is later than the finalized epoch slot (optimization to reduce calls to get_ancestor)
    finalized_slot = compute_start_slot_at_epoch(store.finalized_checkpoint.epoch)
    assert block.slot > finalized_slot
    # Check block is a descendant of the finalized block at the checkpoint finalized slot
    assert get_ancestor(store, block.parent_root, finalized_slot) == store.finalized_checkpoint.root

    # Check the block is valid and compute the post-state
    state = pre_state.copy()
    state_transition(state, signed_block, True)

    # [New in Bellatrix]
    if is_merge_transition_block(pre_state, block.body):
        validate_merge_block(block)

    # Add new block to the store
    store.blocks[hash_tree_root(block)] = block
    # Add new state for this block to the store
    store.block_states[hash_tree_root(block)] = state

    # Add proposer score boost if the block is timely
    time_into_slot = (store.time - store.genesis_time) % SECONDS_PER_SLOT
    is_before_attesting_interval = time_into_slot < SECONDS_PER_SLOT // INTERVALS_PER_SLOT
    if get_current_slot(store) == block.slot and is_before_attesting_interval:
        store.proposer_boost_root = hash_tree_root(block)

    # Update justified checkpoint
    if state.current_justified_checkpoint.epoch > store.justified_checkpoint.epoch:
        if state.current_justified_checkpoint.epoch > store.best_justified_checkpoint.epoch:
            store.best_justified_checkpoint = state.current_justified_checkpoint
        if should_update_justified_checkpoint(store, state.current_justified_checkpoint):
            store.justified_checkpoint = state.current_justified_checkpoint

    # Update finalized checkpoint
    if state.finalized_checkpoint.epoch > store.finalized_checkpoint.epoch:
        store.finalized_checkpoint = state.finalized_checkpoint
        store.justified_checkpoint = state.current_justified_checkpoint


def is_merge_transition_block(state: BeaconState, body: BeaconBlockBody) -> bool:
    """[STAND-IN] pe:1012: execution-layer (PoW -> PoS transition) check; this chain is post-merge from genesis."""
    return False


def validate_merge_block(block: BeaconBlock) -> None:
    """[STAND-IN] pe:1013: never reached (is_merge_transition_block is False)."""
    raise AssertionError("validate_merge_block is out of scope")


def is_slashable_attestation_data(data_1: AttestationData, data_2: AttestationData) -> bool:
    """[REF pe:1134-1143]"""
    return (
        # Double vote
        (data_1 != data_2 and data_1.target.epoch == data_2.target.epoch) or
        # Surround vote
        (data_1.source.epoch < data_2.source.epoch and data_2.target.epoch < data_1.target.epoch)
    )


def on_attester_slashing(store: Store, attester_slashing: AttesterSlashing) -> None:
    """[REF pe:1447-1461]"""
    attestation_1 = attester_slashing.attestation_1
    attestation_2 = attester_slashing.attestation_2
    assert is_slashable_attestation_data(attestation_1.data, attestation_2.data)
    state = store.block_states[store.justified_checkpoint.root]
    assert is_valid_indexed_attestation(state, attestation_1)
    assert is_valid_indexed_attestation(state, attestation_2)

    indices = set(attestation_1.attesting_indices).intersection(attestation_2.attesting_indices)
    for index in indices:
        store.equivocating_indices.add(index)


# --------------------------------------------------------------------------
# The reference's own text takes over (see the module docstring).  Everything ABOVE this line that is tagged
# [REF] is replaced by the code object compiled from the fence it cites.
# --------------------------------------------------------------------------
from . import ref_extract as _ref_extract  # noqa: E402

REF_MANIFEST = _ref_extract.overlay(globals())
ORACLE_OF_RECORD = "reference" if REF_MANIFEST else "transcription"


def _record_message_slots(literal):
    """[VARIANT pe:1585-1596] wraps the (reference's, literal) ``update_latest_messages`` so that the RLMD-GHOST /
    Goldfish variant knows the slot of every latest message.  With VOTE_EXPIRY_SLOTS == 0 -- the reference's
    executable protocol -- the wrapper is a straight call."""
    def update_latest_messages(store, attesting_indices, attestation):
        if VOTE_EXPIRY_SLOTS == 0:
            return literal(store, attesting_indices, attestation)
        before = {i: store.latest_messages.get(i) for i in attesting_indices}
        literal(store, attesting_indices, attestation)
        slots = store.__dict__.setdefault("latest_message_slots", {})
        for i in attesting_indices:
            if store.latest_messages.get(i) is not before[i]:
                slots[i] = attestation.data.slot
    update_latest_messages.literal = literal
    update_latest_messages.__doc__ = literal.__doc__
    return update_latest_messages


update_latest_messages = _record_message_slots(update_latest_messages)
