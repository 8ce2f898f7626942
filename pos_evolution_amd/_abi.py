"""ctypes declarations of include/posevo.h -- the only way Python reaches the engine.

The library is built in-tree (``pos_evolution_amd/libposevo.so``, see
``__graft_entry__.build``).  There is no fallback: a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# POSEVO_LIB_PATH: another build of the same sources (kernel-variant experiments, tools/); the default is the in-tree library
LIB_PATH = os.environ.get("POSEVO_LIB_PATH") or os.path.join(_HERE, "libposevo.so")

PE_OK = 0
PE_ERR_NO_DEVICE = -2
PE_ERR_CAPACITY = -10
PE_ERR_STATE = -14
PE_ERR_TIMEOUT = -15
PE_DIST_SINGLE_COMM = 1
PE_SIG_G2_COMPRESSED, PE_SIG_G2_UNCOMPRESSED, PE_SIG_CHECK_SUBGROUP = 1, 2, 0x100
NONE32 = 0xFFFFFFFF

PE_VAL_ACTIVE, PE_VAL_SLASHED, PE_VAL_EQUIVOCATING, PE_VAL_ACTIVE_PREV = 0x01, 0x02, 0x04, 0x08
PE_ATT_FLAG_SIGNATURE_VALID, PE_ATT_FLAG_FROM_BLOCK = 0x1, 0x2
PE_G1_PARTIAL_BYTES = 192
PE_EXCHANGE_EXTRA = 512
PE_DIST_ID_BYTES = 256
(PE_KERNEL_G1_ACCUMULATE, PE_KERNEL_G1_NORMALISE, PE_KERNEL_VOTES, PE_KERNEL_TREE, PE_KERNEL_LMD,
 PE_KERNEL_PARTICIPATION, PE_KERNEL_BITS_UNION, PE_KERNEL_G2_ACCUMULATE, PE_KERNEL_G2_NORMALISE,
 PE_KERNEL_G1_TREE, PE_KERNEL_ATT_GROUP, PE_KERNEL_ATT_VALIDATE, PE_KERNEL_PAIR_INGEST_VALIDATE,
 PE_KERNEL_PAIR_PLAN_LMD, PE_KERNEL_PAIR_MEMBERS_VOTES, PE_KERNEL_PAIR_UNION_TREE, PE_KERNEL_G2_DECOMPRESS,
 PE_KERNEL_COUNT) = range(18)
KERNEL_NAMES = ["g1_accumulate", "g1_normalise", "votes", "tree", "lmd", "participation", "bits_union",
                "g2_accumulate", "g2_normalise", "g1_tree", "att_group", "att_validate", "pair_ingest_validate",
                "pair_plan_lmd", "pair_members_votes", "pair_union_tree", "g2_decompress"]

ATT_STATUS_NAMES = {
    0: "ok", 1: "target epoch not current or previous", 2: "target epoch != epoch(slot)",
    3: "unknown target root", 4: "unknown beacon block root", 5: "block after attestation slot",
    6: "target not ancestor of beacon block", 7: "attestation slot not in the past", 8: "no committee table",
    9: "committee index out of range", 10: "bits length mismatch", 11: "empty or invalid indices",
    12: "bad signature", 13: "outside inclusion window", 14: "source mismatch",
}


class pe_config(C.Structure):
    _fields_ = [
        ("slots_per_epoch", C.c_uint64), ("seconds_per_slot", C.c_uint64), ("intervals_per_slot", C.c_uint64),
        ("safe_slots_to_update_justified", C.c_uint64), ("proposer_score_boost", C.c_uint64),
        ("effective_balance_increment", C.c_uint64), ("min_attestation_inclusion_delay", C.c_uint64),
        ("max_validators_per_committee", C.c_uint64), ("filter_slashed", C.c_uint32), ("device", C.c_int32),
        ("reserve_validators", C.c_uint64), ("reserve_blocks", C.c_uint32), ("max_committee_tables", C.c_uint32),
        ("vote_expiry_slots", C.c_uint64),
    ]


class pe_attestation(C.Structure):
    _fields_ = [
        ("slot", C.c_uint64), ("index", C.c_uint64), ("beacon_block_root", C.c_uint8 * 32),
        ("source_epoch", C.c_uint64), ("source_root", C.c_uint8 * 32),
        ("target_epoch", C.c_uint64), ("target_root", C.c_uint8 * 32),
        ("bits_offset", C.c_uint32), ("n_bits", C.c_uint32), ("flags", C.c_uint32), ("reserved0", C.c_uint32),
    ]


assert C.sizeof(pe_attestation) == 144


class pe_state_ctx(C.Structure):
    _fields_ = [
        ("slot", C.c_uint64), ("chain_tip_root", C.c_uint8 * 32),
        ("current_justified_epoch", C.c_uint64), ("current_justified_root", C.c_uint8 * 32),
        ("previous_justified_epoch", C.c_uint64), ("previous_justified_root", C.c_uint8 * 32),
        ("base_reward_per_increment", C.c_uint64),
    ]


_P = C.POINTER
# Buffer arguments are declared void*: the wrappers pass raw addresses (engine._ptr), which costs 0.5 us per argument
# against 2.8 us for ndarray.ctypes.data_as -- some twenty pointers cross the boundary per step.
_u8p = _u32p = _u64p = _i32p = _attp = C.c_void_p
_H = C.c_void_p

# name -> (restype, argtypes); every symbol include/posevo.h declares
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class pe_collectives(C.Structure):
    """struct pe_collectives (include/posevo.h): the caller's collectives for pe_dist_init_custom."""
    _fields_ = [("user", C.c_void_p), ("all_reduce_u64", ALL_REDUCE_FN), ("all_gather", ALL_GATHER_FN)]


SIGNATURES = {
    "pe_abi_version": (C.c_uint32, []),
    "pe_config_default": (None, [_P(pe_config)]),
    "pe_engine_create": (C.c_int, [_P(pe_config), _P(_H)]),
    "pe_engine_destroy": (None, [_H]),
    "pe_strerror": (C.c_char_p, [C.c_int]),
    "pe_last_error": (C.c_char_p, [_H]),
    "pe_set_stream": (C.c_int, [_H, C.c_void_p]),
    "pe_store_init": (C.c_int, [_H, C.c_uint64, C.c_uint64, _u8p]),
    "pe_set_validators": (C.c_int, [_H, C.c_uint64, _u8p, _u64p, _u8p]),
    "pe_set_balances": (C.c_int, [_H, C.c_uint64, _u64p, _u8p]),
    "pe_on_tick": (C.c_int, [_H, C.c_uint64]),
    "pe_on_block": (C.c_int, [_H, _u8p, _u8p, C.c_uint64, C.c_uint64, _u8p, C.c_uint64, _u8p]),
    "pe_add_block": (C.c_int, [_H, _u8p, _u8p, C.c_uint64, C.c_uint64, _u8p, C.c_uint64, _u8p]),
    "pe_set_checkpoints": (C.c_int, [_H, C.c_uint64, _u8p, C.c_uint64, _u8p]),
    "pe_set_proposer_boost": (C.c_int, [_H, _u8p]),
    "pe_mark_equivocating": (C.c_int, [_H, _u64p, C.c_uint64]),
    "pe_on_attester_slashing": (C.c_int, [_H, _attp, _u64p, C.c_uint64, _attp, _u64p,
                                          C.c_uint64]),
    "pe_set_committees": (C.c_int, [_H, C.c_uint64, C.c_uint32, _u32p, _u32p]),
    "pe_compute_committees": (C.c_int, [_H, C.c_uint64, _u8p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p]),
    "pe_compute_committees_async": (C.c_int, [_H, C.c_uint64, _u8p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pe_get_head": (C.c_int, [_H, _u8p]),
    "pe_get_head_async": (C.c_int, [_H, _u8p]),
    "pe_aggregate_signed": (C.c_int, [_H, _attp, C.c_uint32, _u8p, C.c_uint64, _u8p, C.c_uint32, _attp, _u32p, _u32p, _u8p,
                                      C.c_uint64, _u8p, _i32p, _u8p, _u32p]),
    "pe_g2_subgroup_check": (C.c_int, [_H, _u8p, C.c_uint64, _i32p]),
    "pe_pipeline_set_lag": (C.c_int, [_H, C.c_uint32]),
    "pe_pipeline_get_lag": (C.c_uint32, [_H]),
    "pe_pipeline_generation": (C.c_uint64, [_H]),
    "pe_pipeline_completed": (C.c_uint64, [_H]),
    "pe_get_weights": (C.c_int, [_H, _u64p, C.c_uint32]),
    "pe_on_attestation_batch": (C.c_int, [_H, _attp, C.c_uint32, _u8p, C.c_uint64, _i32p, _u8p, _u32p]),
    "pe_get_indexed_attestations": (C.c_int, [_H, _attp, C.c_uint32, _u8p, C.c_uint64, _i32p, _u32p, _u32p,
                                              C.c_uint64]),
    "pe_aggregate": (C.c_int, [_H, _attp, C.c_uint32, _u8p, C.c_uint64, _u8p, _attp,
                               _u32p, _u32p, _u8p, C.c_uint64, _u8p, _u8p, _u32p]),
    "pe_process_attestation_batch": (C.c_int, [_H, _P(pe_state_ctx), _attp, C.c_uint32, _u8p,
                                               C.c_uint64, _i32p, _u64p]),
    "pe_participation_set": (C.c_int, [_H, C.c_int, _u8p, C.c_uint64]),
    "pe_participation_get": (C.c_int, [_H, C.c_int, _u8p, C.c_uint64]),
    "pe_participation_rotate": (C.c_int, [_H]),
    "pe_state_set_validators": (C.c_int, [_H, C.c_uint64, _u64p, _u8p]),
    "pe_state_get_validators": (C.c_int, [_H, C.c_uint64, _u64p, _u8p, C.c_void_p]),
    "pe_get_committee_epochs": (C.c_int, [_H, _u64p, C.c_uint32, _u32p]),
    "pe_get_committees": (C.c_int, [_H, C.c_uint64, _u32p, _u32p, C.c_uint32, _u32p, C.c_uint64]),
    "pe_ffg_balances": (C.c_int, [_H, _u64p]),
    "pe_g1_sum": (C.c_int, [_H, _u8p, C.c_uint64, _u32p, _u32p, C.c_uint32, _u8p]),
    "pe_get_block": (C.c_int, [_H, C.c_uint32, _u8p, _u32p, _u64p, _u64p, _u8p, _u64p, _u8p]),
    "pe_get_validator_flags": (C.c_int, [_H, _u8p, C.c_uint64]),
    "pe_get_latest_message_slots": (C.c_int, [_H, _u32p, C.c_uint64]),
    "pe_set_latest_messages": (C.c_int, [_H, C.c_uint64, _u64p, _u32p, _u32p]),
    "pe_set_best_justified": (C.c_int, [_H, C.c_uint64, _u8p]),
    "pe_g1_key_validate": (C.c_int, [_H, _u8p, C.c_uint64, _i32p]),
    "pe_g1_decompress": (C.c_int, [_H, _u8p, C.c_uint64, _u8p, _i32p]),
    "pe_set_pubkeys_compressed": (C.c_int, [_H, C.c_uint64, _u8p, _i32p]),
    "pe_g1_compress": (C.c_int, [_u8p, C.c_uint64, _u8p]),
    "pe_g2_decompress": (C.c_int, [_H, _u8p, C.c_uint64, _u8p, _i32p]),
    "pe_g2_compress": (C.c_int, [_u8p, C.c_uint64, _u8p]),
    "pe_g2_sum": (C.c_int, [_H, _u8p, C.c_uint64, _u32p, _u32p, C.c_uint32, _u8p]),
    "pe_aggregate_signatures": (C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, _u32p, C.c_uint32, C.c_uint32, _u8p, _i32p,
                                         _u32p]),
    "pe_num_blocks": (C.c_uint32, [_H]),
    "pe_num_validators": (C.c_uint64, [_H]),
    "pe_block_root_at": (C.c_int, [_H, C.c_uint32, _u8p]),
    "pe_block_index_of": (C.c_int, [_H, _u8p, _u32p]),
    "pe_get_latest_messages": (C.c_int, [_H, _u64p, _u32p, C.c_uint64]),
    "pe_get_store_scalars": (C.c_int, [_H, _u64p, _u64p, _u64p, _u8p, _u64p, _u8p, _u64p, _u8p, _u8p]),
    "pe_votes_partial": (C.c_int, [_H, C.c_void_p, C.c_uint32]),
    "pe_head_from_weights": (C.c_int, [_H, C.c_void_p, C.c_uint32, _u8p]),
    "pe_aggregate_partial": (C.c_int, [_H, _attp, C.c_uint32, _u8p, C.c_uint64, _attp,
                                       _u32p, _u32p, _u8p, C.c_uint64, _u32p, C.c_void_p, C.c_uint32]),
    "pe_g1_partial": (C.c_int, [_H, _u32p, _u32p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "pe_get_last_weights": (C.c_int, [_H, _u64p, C.c_uint32]),
    "pe_g1_finish": (C.c_int, [_H, C.c_void_p, C.c_uint32, C.c_uint32, _u8p]),
    "pe_dist_unique_id": (C.c_int, [_u8p]),
    "pe_dist_init": (C.c_int, [_H, _u8p, C.c_int, C.c_int]),
    "pe_dist_destroy": (C.c_int, [_H]),
    "pe_dist_init_ex": (C.c_int, [_H, _u8p, C.c_int, C.c_int, C.c_uint32]),
    "pe_dist_init_custom": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "pe_dist_set_timeout_ms": (C.c_int, [_H, C.c_uint32]),
    "pe_dist_set_max_groups": (C.c_int, [_H, C.c_uint32]),
    "pe_aggregate_exchange": (C.c_int, [_H, _attp, _u32p, _u8p, C.c_uint64, _u32p, C.c_uint32]),
    "pe_get_head_sharded": (C.c_int, [_H, _u8p]),
    "pe_get_head_sharded_async": (C.c_int, [_H, _u8p]),
    "pe_aggregate_sharded": (C.c_int, [_H, _attp, C.c_uint32, _u8p, C.c_uint64, _attp, _u32p, _u32p, _u8p,
                                       C.c_uint64, _u8p, _u32p]),
    "pe_pipeline_begin": (C.c_int, [_H]),
    "pe_pipeline_end": (C.c_int, [_H]),
    "pe_pipeline_end_lagged": (C.c_int, [_H]),
    "pe_pipeline_begin_streaming": (C.c_int, [_H]),
    "pe_profile_enable": (C.c_int, [_H, C.c_int]),
    "pe_profile_reset": (C.c_int, [_H]),
    "pe_profile_get": (C.c_int, [_H, C.c_int, _u64p, _P(C.c_double)]),
    "pe_profile_timeline": (C.c_int, [_H, _i32p, C.c_void_p, C.c_void_p, C.c_uint32, _P(C.c_uint32)]),
    "pe_profile_queue_classes": (C.c_int, [_H, _i32p]),
    "pe_profile_arena_growths": (C.c_int, [_H, _P(C.c_uint64)]),
    "pe_profile_accumulate_mhz": (C.c_int, [_H, C.c_void_p, C.c_uint32, _P(C.c_uint32)]),
}

PE_ROWS_RESIDENT = 1  # include/posevo.h: "every group of the last pe_aggregate over rows in device memory"
PE_BITS_RESIDENT = 1  # the address include/posevo.h defines as "bits are where the last pe_aggregate left them"
PE_ATT_FLAG_OVERLAPPING_BITS = 0x4

_lib = None


def load():
    """dlopen libposevo.so and attach the signatures.  Raises if the library is absent:
    the product path has no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
