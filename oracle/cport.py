"""ctypes binding of the L1 C oracle (``posevo_oracle.c``) -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this.  numpy arrays in, numpy arrays out.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# POSEVO_ORACLE_LIB selects another build of the same source (tests/test_oracle_properties.py: the ASan/UBSan build)
_LIB_PATH = os.environ.get("POSEVO_ORACLE_LIB") or os.path.join(_HERE, "libposevo_oracle.so")
NONE = 0xFFFFFFFF


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "posevo_oracle.c")
    if os.environ.get("POSEVO_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libposevo_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _p(a, ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def max_threads() -> int:
    return int(lib().po_max_threads())


def set_threads(n: int) -> None:
    lib().po_set_threads(C.c_int(int(n)))


def _fn(name: str, mt: bool):
    """mt=True: the OpenMP all-cores form of the same loop (bench.py's cpu_baseline leg; equality with the single-thread
    function is a CPU test)."""
    return getattr(lib(), name + ("_mt" if mt else ""))


def get_head(parent, leaf_ok, roots, vote_block, eff_balance, flags, justified_idx, boost_idx=NONE,
             filter_slashed=False, slots_per_epoch=32, boost_percent=40, balance_increment=10**9, mt=False):
    """Returns (head_index, weights[u64 n_blocks])."""
    parent, leaf_ok, roots = _u32(parent), _u8(leaf_ok), _u8(roots)
    vote_block, eff_balance, flags = _u32(vote_block), _u64(eff_balance), _u8(flags)
    n_blocks = parent.shape[0]
    assert roots.size == 32 * n_blocks
    weights = np.zeros(n_blocks, dtype=np.uint64)
    head = C.c_uint32(0)
    rc = _fn("po_get_head", mt)(
        C.c_uint32(n_blocks), _p(parent, C.c_uint32), _p(leaf_ok, C.c_uint8), _p(roots, C.c_uint8),
        C.c_uint64(vote_block.shape[0]), _p(vote_block, C.c_uint32), _p(eff_balance, C.c_uint64),
        _p(flags, C.c_uint8), C.c_int(int(filter_slashed)), C.c_uint32(justified_idx), C.c_uint32(boost_idx),
        C.c_uint64(slots_per_epoch), C.c_uint64(boost_percent), C.c_uint64(balance_increment),
        _p(weights, C.c_uint64), C.byref(head))
    assert rc == 0, rc
    return int(head.value), weights


def update_latest_messages(member_off, n_bits, bits_off, target_epoch, block_idx, arena, members, val_flags,
                           vote_epoch, vote_block, mt=False):
    """In place on vote_epoch (u64) / vote_block (u32).  mt=True needs pairwise disjoint committees in the batch."""
    member_off, n_bits, bits_off = _u32(member_off), _u32(n_bits), _u32(bits_off)
    target_epoch, block_idx = _u64(target_epoch), _u32(block_idx)
    arena, members, val_flags = _u8(arena), _u32(members), _u8(val_flags)
    assert vote_epoch.dtype == np.uint64 and vote_block.dtype == np.uint32
    _fn("po_update_latest_messages", mt)(
        C.c_uint32(member_off.shape[0]), _p(member_off, C.c_uint32), _p(n_bits, C.c_uint32),
        _p(bits_off, C.c_uint32), _p(target_epoch, C.c_uint64), _p(block_idx, C.c_uint32),
        _p(arena, C.c_uint8), _p(members, C.c_uint32), _p(val_flags, C.c_uint8),
        _p(vote_epoch, C.c_uint64), _p(vote_block, C.c_uint32))


def process_attestation_flags(member_off, n_bits, bits_off, flag_mask, which, arena, members, eff_balance,
                              increment, base_reward_per_increment, part_current, part_previous, mt=False):
    """In place on the two participation arrays (u8); returns numerators (u64 per attestation).  mt=True needs pairwise
    disjoint committees in the batch."""
    member_off, n_bits, bits_off = _u32(member_off), _u32(n_bits), _u32(bits_off)
    flag_mask, which, arena = _u8(flag_mask), _u8(which), _u8(arena)
    members, eff_balance = _u32(members), _u64(eff_balance)
    assert part_current.dtype == np.uint8 and part_previous.dtype == np.uint8
    out = np.zeros(member_off.shape[0], dtype=np.uint64)
    _fn("po_process_attestation_flags", mt)(
        C.c_uint32(member_off.shape[0]), _p(member_off, C.c_uint32), _p(n_bits, C.c_uint32),
        _p(bits_off, C.c_uint32), _p(flag_mask, C.c_uint8), _p(which, C.c_uint8), _p(arena, C.c_uint8),
        _p(members, C.c_uint32), _p(eff_balance, C.c_uint64), C.c_uint64(increment),
        C.c_uint64(base_reward_per_increment), _p(part_current, C.c_uint8), _p(part_previous, C.c_uint8),
        _p(out, C.c_uint64))
    return out


def bits_union(group_start, att_list, att_bits_off, arena, group_n_bits, out_bits_off, out_arena_len, mt=False):
    group_start, att_list, att_bits_off = _u32(group_start), _u32(att_list), _u32(att_bits_off)
    arena, group_n_bits, out_bits_off = _u8(arena), _u32(group_n_bits), _u32(out_bits_off)
    n_groups = group_n_bits.shape[0]
    out = np.zeros(out_arena_len, dtype=np.uint8)
    count = np.zeros(n_groups, dtype=np.uint32)
    _fn("po_bits_union", mt)(
        C.c_uint32(n_groups), _p(group_start, C.c_uint32), _p(att_list, C.c_uint32),
        _p(att_bits_off, C.c_uint32), _p(arena, C.c_uint8), _p(group_n_bits, C.c_uint32),
        _p(out_bits_off, C.c_uint32), _p(out, C.c_uint8), _p(count, C.c_uint32))
    return out, count


def g1_sum_groups(points96, index, offsets, mt=False):
    """points96: (n, 96) u8; index: u32 or None; offsets: u32 (n_groups+1).  -> (n_groups, 96) u8."""
    points96 = _u8(points96).reshape(-1, 96)
    offsets = _u32(offsets)
    index = None if index is None else _u32(index)
    n_groups = offsets.shape[0] - 1
    out = np.zeros((n_groups, 96), dtype=np.uint8)
    rc = _fn("po_g1_sum_groups", mt)(_p(points96, C.c_uint8), C.c_uint64(points96.shape[0]), _p(index, C.c_uint32),
                                _p(offsets, C.c_uint32), C.c_uint32(n_groups), _p(out, C.c_uint8))
    assert rc == 0, rc
    return out


def g1_sum_attesters(member_off, n_bits, bits_off, arena, members, points96, mt=False):
    """Aggregate pubkey per attestation straight from its bits (no index list): (n_att, 96) u8."""
    member_off, n_bits, bits_off = _u32(member_off), _u32(n_bits), _u32(bits_off)
    arena, members = _u8(arena), _u32(members)
    points96 = _u8(points96).reshape(-1, 96)
    out = np.zeros((member_off.shape[0], 96), dtype=np.uint8)
    rc = _fn("po_g1_sum_attesters", mt)(
        C.c_uint32(member_off.shape[0]), _p(member_off, C.c_uint32), _p(n_bits, C.c_uint32), _p(bits_off, C.c_uint32),
        _p(arena, C.c_uint8), _p(members, C.c_uint32), _p(points96, C.c_uint8), C.c_uint64(points96.shape[0]),
        _p(out, C.c_uint8))
    assert rc == 0, rc
    return out


def g1_partial_groups(points96, index, offsets):
    points96 = _u8(points96).reshape(-1, 96)
    offsets = _u32(offsets)
    index = None if index is None else _u32(index)
    n_groups = offsets.shape[0] - 1
    out = np.zeros((n_groups, 144), dtype=np.uint8)
    rc = lib().po_g1_partial_groups(_p(points96, C.c_uint8), C.c_uint64(points96.shape[0]),
                                    _p(index, C.c_uint32), _p(offsets, C.c_uint32), C.c_uint32(n_groups),
                                    _p(out, C.c_uint8))
    assert rc == 0, rc
    return out


def g1_finish_partials(gathered144, n_ranks, n_groups):
    gathered144 = _u8(gathered144)
    assert gathered144.size == 144 * n_ranks * n_groups
    out = np.zeros((n_groups, 96), dtype=np.uint8)
    rc = lib().po_g1_finish_partials(_p(gathered144, C.c_uint8), C.c_uint32(n_ranks), C.c_uint32(n_groups),
                                     _p(out, C.c_uint8))
    assert rc == 0, rc
    return out


def g1_arith_progression(a96: bytes, b96: bytes, n: int):
    """(n, 96) u8 with row i = A + i*B."""
    a = _u8(np.frombuffer(a96, dtype=np.uint8))
    b = _u8(np.frombuffer(b96, dtype=np.uint8))
    out = np.zeros((n, 96), dtype=np.uint8)
    rc = lib().po_g1_arith_progression(_p(a, C.c_uint8), _p(b, C.c_uint8), C.c_uint64(n), _p(out, C.c_uint8))
    assert rc == 0, rc
    return out


def g1_scalar_mul(k: int, p96: bytes) -> bytes:
    kb = _u8(np.frombuffer(int(k).to_bytes(32, "big"), dtype=np.uint8))
    p = _u8(np.frombuffer(p96, dtype=np.uint8))
    out = np.zeros(96, dtype=np.uint8)
    lib().po_g1_scalar_mul(_p(kb, C.c_uint8), _p(p, C.c_uint8), _p(out, C.c_uint8))
    return out.tobytes()


def g1_is_on_curve(p96: bytes) -> bool:
    p = _u8(np.frombuffer(p96, dtype=np.uint8))
    return bool(lib().po_g1_is_on_curve(_p(p, C.c_uint8)))
