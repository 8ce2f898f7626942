"""numpy-level wrapper over the C ABI (include/posevo.h).  Thin: no arithmetic here.

Every method maps to exactly one ``pe_*`` entry point and raises ``EngineError``
(an ``AssertionError`` subclass, so spec-style ``assert``-driven tests behave as
with the pyspec) on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import sys
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from ._abi import pe_attestation, pe_config, pe_state_ctx

ZERO_ROOT = bytes(32)


class EngineError(AssertionError):
    def __init__(self, status: int, message: str):
        super().__init__(f"posevo status {status}: {message}")
        self.status = status


_c_char = C.c_char


class DeviceArena:
    """A bit arena that already lies in device memory (or pinned host memory): ``ptr`` = its address, ``size`` = bytes.
    pe_aggregate copies it with the copy engine instead of passing over it on the host; keep it unchanged (and ``keep``,
    the owning object, alive) until the call's outputs are complete."""
    __slots__ = ("ptr", "size", "keep")

    def __init__(self, ptr: int, size: int, keep=None):
        self.ptr, self.size, self.keep = int(ptr), int(size), keep


class DeviceRows:
    """Attestation rows (``n`` x struct pe_attestation, 144 bytes each, 16-byte aligned) that lie in DEVICE memory:
    ``Engine.aggregate(packed=(DeviceRows(...), arena))`` groups, resolves and validates them on the device
    (include/posevo.h, PE_ROWS_RESIDENT) -- the host reads nothing of them.  Keep them unchanged (and ``keep`` alive)
    until the call's outputs are complete."""
    __slots__ = ("ptr", "n", "keep")

    def __init__(self, ptr: int, n: int, keep=None):
        self.ptr, self.n, self.keep = int(ptr), int(n), keep

    def __len__(self):
        return self.n


def _ptr(a, ctype=None):
    """Address of a C-contiguous numpy buffer (None stays NULL).  c_char.from_buffer + addressof is the cheapest route
    ctypes offers; read-only or empty arrays take the slower ndarray.ctypes path."""
    if a is None:
        return None
    if a.__class__ is DeviceArena:
        return a.ptr
    try:
        return C.addressof(_c_char.from_buffer(a))
    except (TypeError, ValueError, BufferError):
        return a.ctypes.data


def _att_ptr(arr):
    """ctypes array of pe_attestation or numpy structured array (synth.ATT_DTYPE) -> address."""
    if isinstance(arr, np.ndarray):
        assert arr.dtype.itemsize == 144 and arr.flags["C_CONTIGUOUS"]
        return _ptr(arr)
    return arr


def _root(b: bytes):
    assert len(b) == 32, "roots are 32 bytes"
    return (C.c_uint8 * 32).from_buffer_copy(bytes(b))


@dataclass
class AttRow:
    """AttestationData (pe:689-697) + aggregation bits + the injected signature verdict."""
    slot: int
    index: int
    beacon_block_root: bytes
    source_epoch: int
    source_root: bytes
    target_epoch: int
    target_root: bytes
    bits: np.ndarray  # bool or 0/1, one entry per committee position
    signature_valid: bool = True
    is_from_block: bool = False


def pack_attestations(rows: Sequence[AttRow]):
    """-> (ctypes array of pe_attestation, uint8 arena).  Bits are packed LSB-first (SSZ order)."""
    n = len(rows)
    arr = (pe_attestation * max(n, 1))()
    chunks = []
    off = 0
    for i, r in enumerate(rows):
        a = arr[i]
        a.slot, a.index = int(r.slot), int(r.index)
        C.memmove(a.beacon_block_root, bytes(r.beacon_block_root), 32)
        a.source_epoch = int(r.source_epoch)
        C.memmove(a.source_root, bytes(r.source_root), 32)
        a.target_epoch = int(r.target_epoch)
        C.memmove(a.target_root, bytes(r.target_root), 32)
        bits = np.asarray(r.bits).astype(np.uint8)
        packed = np.packbits(bits, bitorder="little") if bits.size else np.zeros(0, dtype=np.uint8)
        a.bits_offset, a.n_bits = off, int(bits.size)
        a.flags = (_abi.PE_ATT_FLAG_SIGNATURE_VALID if r.signature_valid else 0) | (
            _abi.PE_ATT_FLAG_FROM_BLOCK if r.is_from_block else 0)
        chunks.append(packed)
        off += packed.size
    arena = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    if arena.size == 0:
        arena = np.zeros(1, dtype=np.uint8)
    return arr, np.ascontiguousarray(arena)


_ATT_DTYPE = np.dtype([
    ("slot", "<u8"), ("index", "<u8"), ("beacon_block_root", "u1", (32,)),
    ("source_epoch", "<u8"), ("source_root", "u1", (32,)),
    ("target_epoch", "<u8"), ("target_root", "u1", (32,)),
    ("bits_offset", "<u4"), ("n_bits", "<u4"), ("flags", "<u4"), ("reserved0", "<u4"),
])  # == synth.ATT_DTYPE == struct pe_attestation (144 bytes)


_I32, _U32, _U64, _U8 = np.dtype(np.int32), np.dtype(np.uint32), np.dtype(np.uint64), np.dtype(np.uint8)


class _Resident:
    """Sentinel for ``packed=(rows, RESIDENT)``: the rows are rows of the last ``aggregate`` result and their OR-ed
    bits are used where pe_aggregate left them on the device (PE_BITS_RESIDENT, include/posevo.h)."""
    size = 0

    def __repr__(self):
        return "RESIDENT"


RESIDENT = _Resident()


class _RowsResident:
    """Sentinel for ``packed=(ROWS_RESIDENT, RESIDENT)``: every group of the last ``aggregate`` over DeviceRows, in
    group order, validated on the device (PE_ROWS_RESIDENT)."""

    def __repr__(self):
        return "ROWS_RESIDENT"


ROWS_RESIDENT = _RowsResident()


class AggregateResult(dict):
    """Result of Engine.aggregate; ``res["bits"]`` decodes the OR-ed bitfields on demand."""

    def __getitem__(self, key):
        if key == "bits" and "bits" not in self:
            out, arena = [], dict.__getitem__(self, "out_arena")
            for a in dict.__getitem__(self, "atts"):
                off, nbits = int(a["bits_offset"]), int(a["n_bits"])
                nb = (nbits + 7) // 8
                out.append(np.unpackbits(arena[off:off + nb], bitorder="little")[:nbits].astype(bool))
            self["bits"] = out
        return dict.__getitem__(self, key)


class ResidentAggregateResult(dict):
    """Result of Engine.aggregate over DeviceRows: nothing is host-derived, so every field -- the number of groups
    included -- is valid once the call's outputs are complete (at return for a synchronous call, at the pipeline's
    completion otherwise).  Fields are sliced to the groups formed when they are read."""

    def __getitem__(self, key):
        raw = dict.__getitem__(self, "_raw")
        g = int(raw["n_groups"][0])
        if key == "n_groups":
            return g
        if key == "group_of":
            return raw["group_of"]
        if key == "out_arena":
            if g == 0:
                return raw["out_arena"][:0]
            last = raw["atts"][g - 1]
            return raw["out_arena"][: int(last["bits_offset"]) + (int(last["n_bits"]) + 7) // 8]
        if key == "bits":
            arena, out = raw["out_arena"], []
            for a in raw["atts"][:g]:
                off, nbits = int(a["bits_offset"]), int(a["n_bits"])
                out.append(np.unpackbits(arena[off:off + (nbits + 7) // 8], bitorder="little")[:nbits].astype(bool))
            return out
        if key in ("atts", "aggpk96", "count", "sig96c"):
            v = raw.get(key)
            return None if v is None else v[:g]
        if key == "sig_status":
            return raw.get(key)
        return dict.__getitem__(self, key)


class _Pipeline:
    def __init__(self, engine, lagged=False):
        self.e = engine
        self.lagged = lagged

    def __enter__(self):
        e = self.e
        # a lagged pipeline still in flight stays in flight
        if self.lagged and e._ring is not None and len(e._ring) < e._lag + 2:
            # outputs of lagged pipeline k are written at the exit of k + lag: a set handed out again before k + lag + 1
            # would never be stable to read
            raise ValueError(f"reuse_outputs(depth) must be >= {e._lag + 2} with lagged pipelines of lag depth {e._lag}")
        e._check((e._lib.pe_pipeline_begin_streaming if self.lagged else e._lib.pe_pipeline_begin)(e._h))
        e._pipe_keep = []
        e._ring_ord = {}
        return e

    def __exit__(self, exc_type, exc, tb):
        e = self.e
        if self.lagged and exc_type is None:
            rc = e._lib.pe_pipeline_end_lagged(e._h)
            # lag depth L: the generation L back is complete now, keep this one and the L - 1 before it alive
            e._lagged_keep = [e._pipe_keep] + (e._lagged_keep or [])[:e._lag - 1]
        else:
            rc = e._lib.pe_pipeline_end(e._h)
            e._lagged_keep = None
        e._pipe_keep = None
        if e._ring is not None:
            e._ring_i = (e._ring_i + 1) % len(e._ring)
        if exc_type is None:
            e._check(rc)
        return False


class Engine:
    """One engine handle = one fork-choice store + validator registry on one GPU."""

    def __init__(self, device: int = -1, **config):
        self._lib = _abi.load()
        cfg = pe_config()
        self._lib.pe_config_default(C.byref(cfg))
        cfg.device = device
        for k, v in config.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown config field {k}")
            setattr(cfg, k, v)
        self.cfg = cfg
        h = C.c_void_p()
        rc = self._lib.pe_engine_create(C.byref(cfg), C.byref(h))
        if rc != _abi.PE_OK:
            raise EngineError(rc, self._lib.pe_strerror(rc).decode() +
                              " (the engine needs a HIP device; there is no CPU fallback)")
        self._h = h
        self._pipe_keep = None
        self._lagged_keep = None
        self._ring = None
        self._ring_i = 0
        self._ring_ord = {}
        self._lag = int(self._lib.pe_pipeline_get_lag(self._h)) or 2

    def set_pipeline_lag(self, depth: int):
        """Lag depth of ``pipeline(lagged=True)`` blocks (pe_pipeline_set_lag): a block's outputs are complete when the
        depth-th next lagged block exits.  Completes everything in flight first."""
        self._check(self._lib.pe_pipeline_set_lag(self._h, int(depth)))
        self._lag = int(depth)
        self._lagged_keep = None

    # -- plumbing ---------------------------------------------------------
    def reuse_outputs(self, depth: int = 4):
        """Opt in to output-buffer reuse: the arrays the batch calls return come from a ring of `depth` buffer sets that
        advances at every pipeline exit (and after every batch call made outside a pipeline), instead of fresh
        ``np.empty`` allocations (whose first touch page-faults: 50+ us per step for the 1 MB of rows and bits an epoch
        returns).  An array stays valid for depth - 1 further pipelines / synchronous calls; copy what must live longer.
        Two calls of the same kind inside ONE pipeline get distinct sets (keyed by their ordinal in the pipeline).
        depth >= lag + 2 with lagged pipelines (4 at the default lag depth 2: entering one with a shallower ring raises)."""
        self._ring = [dict() for _ in range(max(depth, 1))]
        self._ring_i = 0
        self._ring_ord = {}

    def fill_ring(self):
        """Allocate (and touch) in every ring slot the output sets seen so far in any slot: a caller that wants to keep
        the outputs of a whole run -- ring depth = number of pipelines -- pays the allocations before its timed region."""
        if self._ring is None:
            return
        seen = {}
        for d in self._ring:
            for k, (arrs, _) in d.items():
                seen.setdefault(k, tuple(None if a is None else (a.shape, a.dtype) for a in arrs))
        for d in self._ring:
            for k, specs in seen.items():
                if k not in d:
                    arrs = tuple(None if sp is None else np.zeros(sp[0], dtype=sp[1]) for sp in specs)
                    for a in arrs:
                        if a is not None:
                            a.fill(0)  # np.zeros maps pages lazily: touch them now
                    d[k] = (arrs, tuple(_ptr(a) for a in arrs))

    def _outs(self, key, specs):
        """The output arrays of one call and their addresses: ``specs`` = ((shape, dtype) or None, ...).  With
        reuse_outputs() the set comes from the ring slot of the current pipeline -- one dict lookup instead of one
        allocation and one address computation per array (a step makes ~20 pointer arguments)."""
        if self._ring is None:
            arrs = tuple(None if sp is None else np.empty(sp[0], dtype=sp[1]) for sp in specs)
            return arrs, tuple(_ptr(a) for a in arrs)
        d = self._ring[self._ring_i]
        if self._pipe_keep is None:
            # a synchronous call: its results are complete at return, the next call gets the next set
            k = (key, specs, 0)
            self._ring_i = (self._ring_i + 1) % len(self._ring)
        else:
            # inside a pipeline every call's buffers are written at the pipeline's end: the n-th call of a kind gets
            # the n-th set of this ring slot (two aggregates of one pipeline must not share their outputs)
            o = self._ring_ord.get((key, specs), 0)
            self._ring_ord[(key, specs)] = o + 1
            k = (key, specs, o)
        hit = d.get(k)
        if hit is None:
            arrs = tuple(None if sp is None else np.zeros(sp[0], dtype=sp[1]) for sp in specs)  # zeros: touch the pages here
            hit = d[k] = (arrs, tuple(_ptr(a) for a in arrs))
        return hit

    def _keep(self, *buffers):
        """Inside a pipeline the engine writes into these buffers at pe_pipeline_end: keep them alive until then."""
        if self._pipe_keep is not None:
            self._pipe_keep.extend(b for b in buffers if b is not None)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pe_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != _abi.PE_OK:
            detail = self._lib.pe_last_error(self._h).decode()
            raise EngineError(rc, f"{self._lib.pe_strerror(rc).decode()}: {detail}")

    def pipeline(self, lagged: bool = False):
        """``with engine.pipeline(): ...`` -- the batch calls inside return once their device work is enqueued
        (pe_pipeline_begin); their output arrays are complete when the block exits (pe_pipeline_end): one wait per
        step instead of one per call.  get_head() inside the block is still synchronous.
        lagged=True (pe_pipeline_end_lagged): the block's outputs are complete when the L-th next lagged block exits
        (L = lag depth, default 2; or at drain() / any other synchronous call) -- step N's G1 sums run while the steps
        behind it are being prepared."""
        return _Pipeline(self, lagged)

    def drain(self):
        """Complete every pipelined call still in flight (pe_pipeline_end outside a pipeline does exactly that)."""
        self._check(self._lib.pe_pipeline_end(self._h))
        self._lagged_keep = None

    def set_stream(self, hip_stream: int):
        self._check(self._lib.pe_set_stream(self._h, C.c_void_p(hip_stream)))

    # -- store ------------------------------------------------------------
    def store_init(self, genesis_time: int, anchor_slot: int, anchor_root: bytes):
        self._check(self._lib.pe_store_init(self._h, genesis_time, anchor_slot, _root(anchor_root)))

    def set_validators(self, effective_balance, flags, pubkeys96=None):
        bal = np.ascontiguousarray(effective_balance, dtype=np.uint64)
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        assert bal.shape == fl.shape
        pk = None
        if pubkeys96 is not None:
            pk = np.ascontiguousarray(pubkeys96, dtype=np.uint8)
            assert pk.size == 96 * bal.size
        self._check(self._lib.pe_set_validators(self._h, bal.size, _ptr(pk, C.c_uint8), _ptr(bal, C.c_uint64),
                                                _ptr(fl, C.c_uint8)))

    def set_balances(self, effective_balance, flags):
        bal = np.ascontiguousarray(effective_balance, dtype=np.uint64)
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        self._check(self._lib.pe_set_balances(self._h, bal.size, _ptr(bal, C.c_uint64), _ptr(fl, C.c_uint8)))

    def on_tick(self, time: int):
        self._check(self._lib.pe_on_tick(self._h, time))

    def on_block(self, root, parent_root, slot, post_justified=(0, ZERO_ROOT), post_finalized=(0, ZERO_ROOT)):
        self._check(self._lib.pe_on_block(self._h, _root(root), _root(parent_root), slot, post_justified[0],
                                          _root(post_justified[1]), post_finalized[0], _root(post_finalized[1])))

    def add_block(self, root, parent_root, slot, post_justified=(0, ZERO_ROOT), post_finalized=(0, ZERO_ROOT)):
        self._check(self._lib.pe_add_block(self._h, _root(root), _root(parent_root), slot, post_justified[0],
                                           _root(post_justified[1]), post_finalized[0], _root(post_finalized[1])))

    def set_checkpoints(self, justified: Tuple[int, bytes], finalized: Tuple[int, bytes]):
        self._check(self._lib.pe_set_checkpoints(self._h, justified[0], _root(justified[1]), finalized[0],
                                                 _root(finalized[1])))

    def set_proposer_boost(self, root: bytes):
        self._check(self._lib.pe_set_proposer_boost(self._h, _root(root)))

    def mark_equivocating(self, indices):
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        self._check(self._lib.pe_mark_equivocating(self._h, _ptr(idx, C.c_uint64), idx.size))

    def on_attester_slashing(self, row1: AttRow, indices1, row2: AttRow, indices2):
        a1, _ = pack_attestations([row1])
        a2, _ = pack_attestations([row2])
        i1 = np.ascontiguousarray(indices1, dtype=np.uint64)
        i2 = np.ascontiguousarray(indices2, dtype=np.uint64)
        self._check(self._lib.pe_on_attester_slashing(self._h, a1, _ptr(i1, C.c_uint64), i1.size, a2,
                                                      _ptr(i2, C.c_uint64), i2.size))

    def set_committees(self, epoch: int, offsets, members):
        off = np.ascontiguousarray(offsets, dtype=np.uint32)
        mem = np.ascontiguousarray(members, dtype=np.uint32)
        assert off[-1] == mem.size
        if mem.size == 0:
            mem = np.zeros(1, dtype=np.uint32)
        self._check(self._lib.pe_set_committees(self._h, epoch, off.size - 1, _ptr(off, C.c_uint32),
                                                _ptr(mem, C.c_uint32)))

    def compute_committees(self, epoch: int, seed: bytes, active_indices, n_committees: int,
                           shuffle_round_count: int = 90, want_result: bool = True):
        """compute_committee / compute_shuffled_index (pe:495-534) for the whole epoch on the GPU; registers the
        table for `epoch` (it stays on the device).  active_indices: the index array, or an int n meaning validators
        0 .. n - 1.  -> (offsets uint32[C+1], members uint32[n_active]) when want_result, else nothing is read back."""
        if isinstance(active_indices, (int, np.integer)):   # "every validator 0 .. n - 1 is active": nothing to upload
            act, n_act = None, int(active_indices)
        else:
            act = np.ascontiguousarray(active_indices, dtype=np.uint32)
            n_act = act.size
        off = np.empty(n_committees + 1, dtype=np.uint32) if want_result else None
        mem = np.empty(max(n_act, 1), dtype=np.uint32) if want_result else None
        self._check(self._lib.pe_compute_committees(self._h, epoch, _root(seed), _ptr(act, C.c_uint32), n_act,
                                                    n_committees, shuffle_round_count, _ptr(off, C.c_uint32),
                                                    _ptr(mem, C.c_uint32)))
        return (off, mem[:n_act]) if want_result else None

    def compute_committees_async(self, epoch: int, seed: bytes, n_active: int, n_committees: int,
                                 shuffle_round_count: int = 90):
        """pe_compute_committees_async over validators 0 .. n_active - 1: enqueued on the state-transition stream, nothing
        waited for or read back; the table is usable by the calls that follow."""
        rc = self._lib.pe_compute_committees_async(self._h, epoch, _root(seed), None, int(n_active), n_committees,
                                                   shuffle_round_count)
        if rc:
            self._check(rc)

    # -- hot path ---------------------------------------------------------
    def get_head(self) -> bytes:
        out = (C.c_uint8 * 32)()
        self._check(self._lib.pe_get_head(self._h, out))
        return bytes(out)

    def get_head_async(self) -> np.ndarray:
        """pe_get_head_async: -> a 32-byte array that holds the root once the pipeline's outputs are complete (inside a
        pipeline nothing waits; outside one the call is synchronous)."""
        (out,), (p_out,) = self._outs("head", ((32, _U8),))
        if self._pipe_keep is not None:
            self._pipe_keep.append(out)
        rc = self._lib.pe_get_head_async(self._h, p_out)
        if rc:
            self._check(rc)
        return out

    def get_weights(self) -> np.ndarray:
        n = self.num_blocks
        out = np.zeros(n, dtype=np.uint64)
        self._check(self._lib.pe_get_weights(self._h, _ptr(out, C.c_uint64), n))
        return out

    def last_weights(self) -> np.ndarray:
        """Per-block weights left behind by the last head computation (get_head / head_from_weights), not recomputed."""
        n = self.num_blocks
        out = np.zeros(n, dtype=np.uint64)
        self._check(self._lib.pe_get_last_weights(self._h, _ptr(out, C.c_uint64), n))
        return out

    def on_attestation_batch(self, rows=None, packed=None, want_aggregate_pubkeys=False, cap: int = 0):
        """-> (status int32[n], aggpk (n,96) u8 or None, count uint32[n]).
        ``packed=(ROWS_RESIDENT, RESIDENT), cap=c``: every group of the last aggregate over DeviceRows; the arrays hold c
        entries (c >= the groups formed; entries past them read 0)."""
        arr, arena = packed if packed is not None else pack_attestations(rows)
        if arr is ROWS_RESIDENT:
            assert arena is RESIDENT and cap > 0 and not want_aggregate_pubkeys
            (status, count), (p_status, p_count) = self._outs("ratt", ((cap, _I32), (cap, _U32)))
            if self._pipe_keep is not None:
                self._pipe_keep.append((status, count))
            rc = self._lib.pe_on_attestation_batch(self._h, _abi.PE_ROWS_RESIDENT, cap, _abi.PE_BITS_RESIDENT, 0, p_status,
                                                   None, p_count)
            if rc:
                self._check(rc)
            return status, None, count
        n = len(rows) if rows is not None else len(arr)
        m = max(n, 1)
        (status, count, agg), (p_status, p_count, p_agg) = self._outs(
            "att", ((m, _I32), (m, _U32), ((m, 96), _U8) if want_aggregate_pubkeys else None))
        arena_p = _abi.PE_BITS_RESIDENT if arena is RESIDENT else _ptr(arena, C.c_uint8)
        if self._pipe_keep is not None:
            self._pipe_keep.append((arr, status, count, agg))
        rc = self._lib.pe_on_attestation_batch(self._h, _att_ptr(arr), n, arena_p, arena.size, p_status, p_agg, p_count)
        if rc:
            self._check(rc)
        return status[:n], (agg[:n] if agg is not None else None), count[:n]

    def get_indexed_attestations(self, rows=None, packed=None):
        """get_indexed_attestation x n (A.6): -> (status int32[n], offsets uint32[n+1], sorted attesting indices)."""
        arr, arena = packed if packed is not None else pack_attestations(rows)
        n = len(rows) if rows is not None else len(arr)
        status = np.empty(max(n, 1), dtype=np.int32)
        offsets = np.empty(n + 1, dtype=np.uint32)
        cap = int(sum(int(r.bits.size) for r in rows)) if rows is not None else int(arr["n_bits"].sum())
        indices = np.empty(max(cap, 1), dtype=np.uint32)
        self._check(self._lib.pe_get_indexed_attestations(self._h, _att_ptr(arr), n, _ptr(arena, C.c_uint8), arena.size,
                                                          _ptr(status, C.c_int32), _ptr(offsets, C.c_uint32),
                                                          _ptr(indices, C.c_uint32), indices.size))
        return status[:n], offsets, indices[:int(offsets[-1])]

    def aggregate(self, rows=None, packed=None, sig_points96=None, want_aggregate_pubkeys=False, sig_points192=None):
        """-> AggregateResult(n_groups, atts (ATT_DTYPE rows of the groups), group_of, out_arena, sig96, aggpk96,
        count; ``["bits"]`` decodes the OR-ed bitfields on demand).  sig_points192: the members' signatures as G2
        points (192-byte uncompressed BLSSignature, pe:717); their per-group sums come back as ``sig192``
        (pe_g2_sum over pe_aggregate's grouping)."""
        arr, arena = packed if packed is not None else pack_attestations(rows)
        n = len(rows) if rows is not None else len(arr)
        m = max(n, 1)
        if arr.__class__ is DeviceRows:  # rows in device memory: grouped / resolved / validated there
            assert sig_points96 is None and sig_points192 is None, "signature points take host rows"
            (out_atts, group_of, out_arena, out_pk, count, ng), (p_atts, p_gof, p_arena, p_pk, p_count, p_ng) = self._outs(
                "ragg", ((m, _ATT_DTYPE), (m, _U32), (max(arena.size, 1), _U8),
                         ((m, 96), _U8) if want_aggregate_pubkeys else None, (m, _U32), (1, _U32)))
            ng[0] = 0
            if self._pipe_keep is not None:
                self._pipe_keep.append((arr, arena, out_atts, group_of, out_arena, out_pk, count, ng))
            rc = self._lib.pe_aggregate(self._h, arr.ptr, n, _ptr(arena, C.c_uint8), arena.size, None, p_atts, p_ng, p_gof,
                                        p_arena, out_arena.size, None, p_pk, p_count)
            if rc:
                self._check(rc)
            return ResidentAggregateResult(_raw=dict(n_groups=ng, atts=out_atts, group_of=group_of, out_arena=out_arena,
                                                     aggpk96=out_pk, count=count), sig96=None, sig192=None)
        sig = None
        if sig_points96 is not None:
            sig = np.ascontiguousarray(sig_points96, dtype=np.uint8)
            assert sig.size == 96 * n
        # output buffers: written by the engine, never read before that.  "tail" receives (offset, n_bits) of the last
        # group's bits through two of the row fields the C side fills in, read back as plain u32s below
        (out_atts, group_of, out_arena, out_sig, out_pk, count), (p_atts, p_gof, p_arena, p_sig, p_pk, p_count) = self._outs(
            "agg", ((m, _ATT_DTYPE), (m, _U32), (max(arena.size, 1), _U8), ((m, 96), _U8) if sig is not None else None,
                    ((m, 96), _U8) if want_aggregate_pubkeys else None, (m, _U32)))
        n_groups = C.c_uint32(0)
        if self._pipe_keep is not None:
            self._pipe_keep.append((arr, arena, sig, out_atts, group_of, out_arena, out_sig, out_pk, count))
        rc = self._lib.pe_aggregate(self._h, _att_ptr(arr), n, _ptr(arena, C.c_uint8), arena.size, _ptr(sig, C.c_uint8),
                                    p_atts, C.byref(n_groups), p_gof, p_arena, out_arena.size, p_sig, p_pk, p_count)
        if rc:
            self._check(rc)
        g = n_groups.value
        if g:  # the groups' unions are laid out back to back: hand back only the written part of the arena
            tail = out_atts.view(np.uint32).reshape(-1, 36)[g - 1]   # u32 words 32 / 33 of a row: bits_offset, n_bits
            out_arena = out_arena[: int(tail[32]) + (int(tail[33]) + 7) // 8]
        sig192 = None
        if sig_points192 is not None:
            s2 = np.ascontiguousarray(sig_points192, dtype=np.uint8)
            assert s2.size == 192 * n
            order = np.argsort(group_of[:n], kind="stable").astype(np.uint32)   # members of a group in input order
            offs = np.concatenate([[0], np.cumsum(np.bincount(group_of[:n], minlength=g))]).astype(np.uint32)
            sig192 = self.g2_sum(s2, offs, index=order)
        return AggregateResult(n_groups=g, atts=out_atts[:g], group_of=group_of[:n], out_arena=out_arena,
                               sig96=None if out_sig is None else out_sig[:g], sig192=sig192,
                               aggpk96=None if out_pk is None else out_pk[:g], count=count[:g])

    def aggregate_signed(self, signatures, rows=None, packed=None, compressed: bool = True, check_subgroup: bool = False,
                         want_aggregate_pubkeys: bool = False):
        """pe_aggregate_signed: pe_aggregate plus bls.Aggregate over the members' BLSSignatures (pe:659, pe:714-717).
        signatures: (n, 96) compressed or (n, 192) uncompressed uint8 (or a DeviceArena holding them).  -> the aggregate
        result with ``sig96c`` ((groups, 96) uint8: the compressed aggregate signature of every group) and ``sig_status``
        (int32[n]: PE_SIG_* per input row).  Host rows or DeviceRows, synchronous or inside a pipeline, as aggregate()."""
        arr, arena = packed if packed is not None else pack_attestations(rows)
        n = len(rows) if rows is not None else len(arr)
        m = max(n, 1)
        fmt = (_abi.PE_SIG_G2_COMPRESSED if compressed else _abi.PE_SIG_G2_UNCOMPRESSED) | (
            _abi.PE_SIG_CHECK_SUBGROUP if check_subgroup else 0)
        sig = signatures if signatures.__class__ is DeviceArena else np.ascontiguousarray(signatures, dtype=np.uint8)
        assert sig.size == (96 if compressed else 192) * n, "one signature per input row"
        dev = arr.__class__ is DeviceRows
        (out_atts, group_of, out_arena, out_pk, count, ng, osig, sst), (p_atts, p_gof, p_arena, p_pk, p_count, p_ng, p_osig,
                                                                       p_sst) = self._outs(
            "saggd" if dev else "sagg", ((m, _ATT_DTYPE), (m, _U32), (max(arena.size, 1), _U8),
                                         ((m, 96), _U8) if want_aggregate_pubkeys else None, (m, _U32), (1, _U32),
                                         ((m, 96), _U8), (m, _I32)))
        ng[0] = 0
        if self._pipe_keep is not None:
            self._pipe_keep.append((arr, arena, sig, out_atts, group_of, out_arena, out_pk, count, ng, osig, sst))
        rc = self._lib.pe_aggregate_signed(self._h, arr.ptr if dev else _att_ptr(arr), n, _ptr(arena, C.c_uint8), arena.size,
                                           _ptr(sig, C.c_uint8), fmt, p_atts, p_ng, p_gof, p_arena, out_arena.size, p_osig,
                                           p_sst, p_pk, p_count)
        if rc:
            self._check(rc)
        return ResidentAggregateResult(_raw=dict(n_groups=ng, atts=out_atts, group_of=group_of, out_arena=out_arena,
                                                 aggpk96=out_pk, count=count, sig96c=osig, sig_status=sst[:n]),
                                       sig96=None, sig192=None)

    def g2_subgroup_check(self, points192) -> np.ndarray:
        """pe_g2_subgroup_check: int32[n], 0 = in G2 (r * P = infinity), 3 = not."""
        pts = np.ascontiguousarray(points192, dtype=np.uint8).reshape(-1, 192)
        st = np.zeros(max(len(pts), 1), dtype=np.int32)
        self._check(self._lib.pe_g2_subgroup_check(self._h, _ptr(pts, C.c_uint8), len(pts), _ptr(st, C.c_int32)))
        return st[:len(pts)]

    def process_attestation_batch(self, state_ctx: pe_state_ctx, rows=None, packed=None, cap: int = 0):
        """-> (status int32[n], proposer_reward_numerator uint64[n]).  ``packed=(ROWS_RESIDENT, RESIDENT), cap=c`` as for
        on_attestation_batch."""
        arr, arena = packed if packed is not None else pack_attestations(rows)
        if arr is ROWS_RESIDENT:
            assert arena is RESIDENT and cap > 0
            (status, num), (p_status, p_num) = self._outs("rproc", ((cap, _I32), (cap, _U64)))
            if self._pipe_keep is not None:
                self._pipe_keep.append((status, num))
            rc = self._lib.pe_process_attestation_batch(self._h, C.byref(state_ctx), _abi.PE_ROWS_RESIDENT, cap,
                                                        _abi.PE_BITS_RESIDENT, 0, p_status, p_num)
            if rc:
                self._check(rc)
            return status, num
        n = len(rows) if rows is not None else len(arr)
        m = max(n, 1)
        (status, num), (p_status, p_num) = self._outs("proc", ((m, _I32), (m, _U64)))
        arena_p = _abi.PE_BITS_RESIDENT if arena is RESIDENT else _ptr(arena, C.c_uint8)
        if self._pipe_keep is not None:
            self._pipe_keep.append((arr, status, num))
        rc = self._lib.pe_process_attestation_batch(self._h, C.byref(state_ctx), _att_ptr(arr), n, arena_p, arena.size,
                                                    p_status, p_num)
        if rc:
            self._check(rc)
        return status[:n], num[:n]

    def participation_set(self, which: int, flags):
        f = np.ascontiguousarray(flags, dtype=np.uint8)
        self._check(self._lib.pe_participation_set(self._h, which, _ptr(f, C.c_uint8), f.size))

    def participation_get(self, which: int) -> np.ndarray:
        out = np.zeros(self.num_validators, dtype=np.uint8)
        self._check(self._lib.pe_participation_get(self._h, which, _ptr(out, C.c_uint8), out.size))
        return out

    def participation_rotate(self):
        self._check(self._lib.pe_participation_rotate(self._h))

    def state_set_validators(self, effective_balance, flags):
        bal = np.ascontiguousarray(effective_balance, dtype=np.uint64)
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        self._check(self._lib.pe_state_set_validators(self._h, bal.size, _ptr(bal, C.c_uint64), _ptr(fl, C.c_uint8)))

    def ffg_balances(self):
        """-> (total_active_balance, previous_target_balance, current_target_balance) of pe:791-802."""
        out = np.zeros(3, dtype=np.uint64)
        self._check(self._lib.pe_ffg_balances(self._h, _ptr(out, C.c_uint64)))
        return int(out[0]), int(out[1]), int(out[2])

    def g1_sum(self, offsets, index=None, points96=None) -> np.ndarray:
        off = np.ascontiguousarray(offsets, dtype=np.uint32)
        idx = None if index is None else np.ascontiguousarray(index, dtype=np.uint32)
        pts = None if points96 is None else np.ascontiguousarray(points96, dtype=np.uint8)
        n_points = 0 if pts is None else pts.size // 96
        n_groups = off.size - 1
        out = np.zeros((max(n_groups, 1), 96), dtype=np.uint8)
        self._check(self._lib.pe_g1_sum(self._h, _ptr(pts, C.c_uint8), n_points, _ptr(idx, C.c_uint32),
                                        _ptr(off, C.c_uint32), n_groups, _ptr(out, C.c_uint8)))
        return out[:n_groups]

    def g1_key_validate(self, points96=None) -> np.ndarray:
        """KeyValidate (A.7): status per key -- 0 valid, 3 not in the prime-order subgroup, 4 identity.  points96 None =
        the registry's pubkeys as loaded."""
        if points96 is None:
            n, p = self.num_validators, None
        else:
            p = np.ascontiguousarray(points96, dtype=np.uint8).reshape(-1, 96)
            n = p.shape[0]
        status = np.empty(max(n, 1), dtype=np.int32)
        self._check(self._lib.pe_g1_key_validate(self._h, _ptr(p, C.c_uint8), n, _ptr(status, C.c_int32)))
        return status[:n]

    def g1_decompress(self, keys48):
        """48-byte compressed BLSPubkeys (pe:37) -> (96-byte uncompressed (n, 96), status int32[n])."""
        k = np.ascontiguousarray(keys48, dtype=np.uint8).reshape(-1, 48)
        n = k.shape[0]
        out = np.empty((max(n, 1), 96), dtype=np.uint8)
        status = np.empty(max(n, 1), dtype=np.int32)
        self._check(self._lib.pe_g1_decompress(self._h, _ptr(k, C.c_uint8), n, _ptr(out, C.c_uint8), _ptr(status, C.c_int32)))
        return out[:n], status[:n]

    def set_pubkeys_compressed(self, keys48):
        """Load the registry's pubkeys from their 48-byte wire form; raises if any key does not decode."""
        k = np.ascontiguousarray(keys48, dtype=np.uint8).reshape(-1, 48)
        status = np.empty(max(k.shape[0], 1), dtype=np.int32)
        self._check(self._lib.pe_set_pubkeys_compressed(self._h, k.shape[0], _ptr(k, C.c_uint8), _ptr(status, C.c_int32)))

    def g1_compress(self, points96) -> np.ndarray:
        p = np.ascontiguousarray(points96, dtype=np.uint8).reshape(-1, 96)
        out = np.empty((max(p.shape[0], 1), 48), dtype=np.uint8)
        self._check(self._lib.pe_g1_compress(_ptr(p, C.c_uint8), p.shape[0], _ptr(out, C.c_uint8)))
        return out[:p.shape[0]]

    def g2_decompress(self, sigs96):
        """96-byte compressed BLSSignatures (pe:37) -> (192-byte uncompressed (n, 192), status int32[n])."""
        k = np.ascontiguousarray(sigs96, dtype=np.uint8).reshape(-1, 96)
        n = k.shape[0]
        out = np.empty((max(n, 1), 192), dtype=np.uint8)
        status = np.empty(max(n, 1), dtype=np.int32)
        self._check(self._lib.pe_g2_decompress(self._h, _ptr(k, C.c_uint8), n, _ptr(out, C.c_uint8), _ptr(status, C.c_int32)))
        return out[:n], status[:n]

    def g2_compress(self, points192) -> np.ndarray:
        p = np.ascontiguousarray(points192, dtype=np.uint8).reshape(-1, 192)
        out = np.empty((max(p.shape[0], 1), 96), dtype=np.uint8)
        self._check(self._lib.pe_g2_compress(_ptr(p, C.c_uint8), p.shape[0], _ptr(out, C.c_uint8)))
        return out[:p.shape[0]]

    def g2_sum(self, points192, offsets, index=None) -> np.ndarray:
        """bls.Aggregate over G2 signature points (pe:659, pe:1536): 192-byte uncompressed in, 192-byte affine out."""
        off = np.ascontiguousarray(offsets, dtype=np.uint32)
        idx = None if index is None else np.ascontiguousarray(index, dtype=np.uint32)
        pts = np.ascontiguousarray(points192, dtype=np.uint8)
        n_groups = off.size - 1
        out = np.empty((max(n_groups, 1), 192), dtype=np.uint8)
        self._check(self._lib.pe_g2_sum(self._h, _ptr(pts, C.c_uint8), pts.size // 192, _ptr(idx, C.c_uint32),
                                        _ptr(off, C.c_uint32), n_groups, _ptr(out, C.c_uint8)))
        return out[:n_groups]

    def aggregate_signatures(self, signatures, offsets, index=None, check_subgroup: bool = False):
        """pe_aggregate_signatures: bls.Aggregate per committee over an epoch's unaggregated compressed signatures
        (pe:717, pe:659, pe:1536).  signatures: (n, 96) uint8 array or a DeviceArena of n * 96 bytes; index: uint32 array
        or a DeviceArena of 4-byte entries (None = contiguous groups).
        -> (aggregates (n_groups, 96) uint8, per-signature status int32 (n,), undecodable members per group uint32)."""
        off = np.ascontiguousarray(offsets, dtype=np.uint32)
        n_groups = off.size - 1
        if isinstance(signatures, DeviceArena):
            sig_ptr, n = signatures.ptr, signatures.size // 96
        else:
            sig = np.ascontiguousarray(signatures, dtype=np.uint8)
            sig_ptr, n = sig.ctypes.data, sig.size // 96
        if index is None:
            idx_ptr = None
        elif isinstance(index, DeviceArena):
            idx_ptr = index.ptr
        else:
            idx = np.ascontiguousarray(index, dtype=np.uint32)
            idx_ptr = idx.ctypes.data
        out = np.empty((max(n_groups, 1), 96), dtype=np.uint8)
        status = np.zeros(max(n, 1), dtype=np.int32)
        bad = np.zeros(max(n_groups, 1), dtype=np.uint32)
        flags = _abi.PE_SIG_CHECK_SUBGROUP if check_subgroup else 0
        self._check(self._lib.pe_aggregate_signatures(self._h, C.c_void_p(sig_ptr), n, C.c_void_p(idx_ptr) if idx_ptr else None,
                                                      _ptr(off, C.c_uint32), n_groups, flags, _ptr(out, C.c_uint8),
                                                      _ptr(status, C.c_int32), _ptr(bad, C.c_uint32)))
        return out[:n_groups], status[:n], bad[:n_groups]

    # -- multi-GPU exchange inside the C ABI (RCCL owned by the engine) -------
    def dist_unique_id(self) -> bytes:
        """rank 0: the PE_DIST_ID_BYTES (two RCCL unique ids) to ship to the other ranks (pe_dist_unique_id)."""
        out = (C.c_uint8 * _abi.PE_DIST_ID_BYTES)()
        rc = self._lib.pe_dist_unique_id(out)
        if rc != _abi.PE_OK:
            raise EngineError(rc, "pe_dist_unique_id: librccl not available")
        return bytes(out)

    def dist_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == _abi.PE_DIST_ID_BYTES
        buf = (C.c_uint8 * _abi.PE_DIST_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._lib.pe_dist_init(self._h, buf, rank, world))

    def dist_init_ex(self, unique_id: bytes, rank: int, world: int, single_comm: bool = False):
        assert len(unique_id) == _abi.PE_DIST_ID_BYTES
        buf = (C.c_uint8 * _abi.PE_DIST_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._lib.pe_dist_init_ex(self._h, buf, rank, world, _abi.PE_DIST_SINGLE_COMM if single_comm else 0))

    def dist_init_custom(self, rank: int, world: int, all_reduce_u64, all_gather):
        """pe_dist_init_custom: the two exchange steps through Python callables
        ``all_reduce_u64(dev_ptr, count, hip_stream) -> int`` (in-place sum over ranks) and
        ``all_gather(dev_send, dev_recv, bytes_per_rank, hip_stream) -> int``, both on device memory, ordered on the stream
        (pos_evolution_amd.sharded.HostStagedCollectives is one such pair over any torch.distributed backend)."""
        def _ar(_user, buf, count, stream):
            try:
                return int(all_reduce_u64(buf or 0, int(count), stream or 0) or 0)
            except Exception as err:  # never unwind through the C frames
                print(f"[posevo] all_reduce_u64 callback failed: {err!r}", file=sys.stderr)
                return 1

        def _ag(_user, send, recv, nbytes, stream):
            try:
                return int(all_gather(send or 0, recv or 0, int(nbytes), stream or 0) or 0)
            except Exception as err:
                print(f"[posevo] all_gather callback failed: {err!r}", file=sys.stderr)
                return 1

        fn = _abi.pe_collectives(None, _abi.ALL_REDUCE_FN(_ar), _abi.ALL_GATHER_FN(_ag))
        self._coll_keep = fn  # the engine calls through these pointers until dist_destroy
        self._check(self._lib.pe_dist_init_custom(self._h, rank, world, C.addressof(fn)))

    def dist_set_timeout_ms(self, ms: int):
        self._check(self._lib.pe_dist_set_timeout_ms(self._h, int(ms)))

    def dist_set_max_groups(self, n: int):
        self._check(self._lib.pe_dist_set_max_groups(self._h, int(n)))

    def aggregate_exchange(self, cap_groups: int, max_bits: int = 2048):
        """pe_aggregate_exchange (committee-sharded steps): the aggregates of the last aggregate() over DeviceRows are
        all-gathered and become the resident aggregate for the handlers.  -> the gathered aggregates (atts, out_arena,
        count; fields valid when the call's outputs are complete).  cap_groups >= world x the local bound."""
        m = max(int(cap_groups), 1)
        (out_atts, out_arena, count, ng), (p_atts, p_arena, p_count, p_ng) = self._outs(
            "xagg", ((m, _ATT_DTYPE), (m * ((max_bits + 31) // 32) * 4, _U8), (m, _U32), (1, _U32)))
        ng[0] = 0
        if self._pipe_keep is not None:
            self._pipe_keep.append((out_atts, out_arena, count, ng))
        rc = self._lib.pe_aggregate_exchange(self._h, p_atts, p_ng, p_arena, out_arena.size, p_count, m)
        if rc:
            self._check(rc)
        return ResidentAggregateResult(_raw=dict(n_groups=ng, atts=out_atts, group_of=None, out_arena=out_arena, aggpk96=None,
                                                 count=count), sig96=None, sig192=None)

    def dist_destroy(self):
        self._check(self._lib.pe_dist_destroy(self._h))
        self._coll_keep = None

    def get_head_sharded(self) -> bytes:
        """get_head over all shards through the engine's own RCCL communicator (one all-reduce on its stream)."""
        out = (C.c_uint8 * 32)()
        self._check(self._lib.pe_get_head_sharded(self._h, out))
        return bytes(out)

    def get_head_sharded_async(self) -> np.ndarray:
        """pe_get_head_sharded_async: -> a 32-byte array that holds the root once the pipeline's outputs are complete."""
        (out,), (p_out,) = self._outs("headsh", ((32, _U8),))
        if self._pipe_keep is not None:
            self._pipe_keep.append(out)
        rc = self._lib.pe_get_head_sharded_async(self._h, p_out)
        if rc:
            self._check(rc)
        return out

    def aggregate_sharded(self, rows=None, packed=None):
        """pe_aggregate over all shards (one all-gather of the XYZZ partials inside): rank-local unions, global
        aggregate pubkeys.  Inside a pipeline() block nothing waits; the unions can be handed on as RESIDENT."""
        arr, arena = packed if packed is not None else pack_attestations(rows)
        n = len(rows) if rows is not None else len(arr)
        m = max(n, 1)
        if arr.__class__ is DeviceRows:  # rows in device memory: grouped / resolved there; the handlers take ROWS_RESIDENT
            (out_atts, group_of, out_arena, out_pk, count, ng), (p_atts, p_gof, p_arena, p_pk, p_count, p_ng) = self._outs(
                "raggsh", ((m, _ATT_DTYPE), (m, _U32), (max(arena.size, 1), _U8), ((m, 96), _U8), (m, _U32), (1, _U32)))
            ng[0] = 0
            if self._pipe_keep is not None:
                self._pipe_keep.append((arr, arena, out_atts, group_of, out_arena, out_pk, count, ng))
            rc = self._lib.pe_aggregate_sharded(self._h, arr.ptr, n, _ptr(arena, C.c_uint8), arena.size, p_atts, p_ng,
                                                p_gof, p_arena, out_arena.size, p_pk, p_count)
            if rc:
                self._check(rc)
            return ResidentAggregateResult(_raw=dict(n_groups=ng, atts=out_atts, group_of=group_of, out_arena=out_arena,
                                                     aggpk96=out_pk, count=count), sig96=None, sig192=None)
        (out_atts, group_of, out_arena, out_pk, count), (p_atts, p_gof, p_arena, p_pk, p_count) = self._outs(
            "aggsh", ((m, _ATT_DTYPE), (m, _U32), (max(arena.size, 1), _U8), ((m, 96), _U8), (m, _U32)))
        n_groups = C.c_uint32(0)
        if self._pipe_keep is not None:
            self._pipe_keep.append((arr, arena, out_atts, group_of, out_arena, out_pk, count))
        rc = self._lib.pe_aggregate_sharded(self._h, _att_ptr(arr), n, _ptr(arena, C.c_uint8), arena.size, p_atts,
                                            C.byref(n_groups), p_gof, p_arena, out_arena.size, p_pk, p_count)
        if rc:
            self._check(rc)
        g = n_groups.value
        if g:
            tail = out_atts.view(np.uint32).reshape(-1, 36)[g - 1]   # u32 words 32 / 33 of a row: bits_offset, n_bits
            out_arena = out_arena[: int(tail[32]) + (int(tail[33]) + 7) // 8]
        return AggregateResult(n_groups=g, atts=out_atts[:g], group_of=group_of[:n], out_arena=out_arena, sig96=None,
                               sig192=None, aggpk96=out_pk[:g], count=count[:g])

    # -- multi-GPU exchange -------------------------------------------------
    def votes_partial(self, dev_ptr: int):
        """dev_ptr: device buffer of num_blocks + PE_EXCHANGE_EXTRA u64 (weights | per-workgroup active totals)."""
        self._check(self._lib.pe_votes_partial(self._h, C.c_void_p(dev_ptr), self.num_blocks))

    def head_from_weights(self, dev_ptr: int) -> bytes:
        out = (C.c_uint8 * 32)()
        self._check(self._lib.pe_head_from_weights(self._h, C.c_void_p(dev_ptr), self.num_blocks, out))
        return bytes(out)

    def aggregate_partial(self, dev_ptr: int, rows=None, packed=None, capacity_groups: int = 0):
        """pe_aggregate with the aggregate pubkeys left as this shard's XYZZ partials (192 B each) at dev_ptr, a device
        buffer of capacity_groups partials (a batch forming more groups fails with PE_ERR_CAPACITY, nothing written)."""
        arr, arena = packed if packed is not None else pack_attestations(rows)
        n = len(rows) if rows is not None else len(arr)
        out_atts = np.empty(max(n, 1), dtype=_ATT_DTYPE)
        n_groups = C.c_uint32(0)
        group_of = np.empty(max(n, 1), dtype=np.uint32)
        out_arena = np.empty(max(arena.size, 1), dtype=np.uint8)
        count = np.empty(max(n, 1), dtype=np.uint32)
        self._check(self._lib.pe_aggregate_partial(self._h, _att_ptr(arr), n, _ptr(arena, C.c_uint8), arena.size,
                                                   _att_ptr(out_atts), C.byref(n_groups), _ptr(group_of, C.c_uint32),
                                                   _ptr(out_arena, C.c_uint8), out_arena.size,
                                                   _ptr(count, C.c_uint32), C.c_void_p(dev_ptr), capacity_groups))
        g = n_groups.value
        if g:
            last = out_atts[g - 1]
            out_arena = out_arena[: int(last["bits_offset"]) + (int(last["n_bits"]) + 7) // 8]
        return dict(n_groups=g, atts=out_atts[:g], group_of=group_of[:n], out_arena=out_arena, count=count[:g])

    def g1_partial(self, offsets, index, dev_ptr: int, capacity_groups: int):
        off = np.ascontiguousarray(offsets, dtype=np.uint32)
        idx = None if index is None else np.ascontiguousarray(index, dtype=np.uint32)
        self._check(self._lib.pe_g1_partial(self._h, _ptr(idx, C.c_uint32), _ptr(off, C.c_uint32), off.size - 1,
                                            C.c_void_p(dev_ptr), capacity_groups))

    def g1_finish(self, dev_ptr: int, n_ranks: int, n_groups: int) -> np.ndarray:
        out = np.zeros((max(n_groups, 1), 96), dtype=np.uint8)
        self._check(self._lib.pe_g1_finish(self._h, C.c_void_p(dev_ptr), n_ranks, n_groups, _ptr(out, C.c_uint8)))
        return out[:n_groups]

    # -- inspection -------------------------------------------------------
    @property
    def num_blocks(self) -> int:
        return int(self._lib.pe_num_blocks(self._h))

    @property
    def num_validators(self) -> int:
        return int(self._lib.pe_num_validators(self._h))

    def block_root_at(self, i: int) -> bytes:
        out = (C.c_uint8 * 32)()
        self._check(self._lib.pe_block_root_at(self._h, i, out))
        return bytes(out)

    def block_index_of(self, root: bytes) -> int:
        out = C.c_uint32(0)
        self._check(self._lib.pe_block_index_of(self._h, _root(root), C.byref(out)))
        return out.value

    def latest_messages(self):
        """-> (epoch uint64[V], block_index uint32[V]); block_index 0xFFFFFFFF = no message."""
        n = self.num_validators
        ep = np.zeros(max(n, 1), dtype=np.uint64)
        bi = np.zeros(max(n, 1), dtype=np.uint32)
        self._check(self._lib.pe_get_latest_messages(self._h, _ptr(ep, C.c_uint64), _ptr(bi, C.c_uint32), n))
        return ep[:n], bi[:n]

    # -- checkpoint / resume ------------------------------------------------
    def state_validators(self):
        """-> (effective_balance u64[n], flags u8[n], is_set): the working-state view process_attestation and the FFG
        sums read (pe_state_set_validators); is_set False while it still mirrors the registry."""
        n = self.num_validators
        bal, flags, is_set = np.zeros(max(n, 1), dtype=np.uint64), np.zeros(max(n, 1), dtype=np.uint8), C.c_int(0)
        self._check(self._lib.pe_state_get_validators(self._h, n, _ptr(bal), _ptr(flags), C.byref(is_set)))
        return bal[:n], flags[:n], bool(is_set.value)

    def committee_epochs(self):
        n = C.c_uint32(0)
        self._check(self._lib.pe_get_committee_epochs(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.uint64)
        self._check(self._lib.pe_get_committee_epochs(self._h, _ptr(out), n.value, C.byref(n)))
        return [int(x) for x in out[: n.value]]

    def committees(self, epoch: int):
        """-> (offsets u32[C + 1], members u32[total]) of the table held for `epoch` (set or computed on the GPU)."""
        nc = C.c_uint32(0)
        self._check(self._lib.pe_get_committees(self._h, epoch, C.byref(nc), None, 0, None, 0))
        offsets = np.zeros(nc.value + 1, dtype=np.uint32)
        self._check(self._lib.pe_get_committees(self._h, epoch, C.byref(nc), _ptr(offsets), offsets.size, None, 0))
        members = np.zeros(max(int(offsets[-1]), 1), dtype=np.uint32)
        self._check(self._lib.pe_get_committees(self._h, epoch, C.byref(nc), None, 0, _ptr(members), int(offsets[-1])))
        return offsets, members[: int(offsets[-1])]

    def export_state(self) -> dict:
        """The store's dynamic state as flat arrays (SURVEY.md 5 "checkpoint / resume"): scalars, the block table in
        insertion order, validator flags (incl. the equivocating bit), latest messages, participation flags, the
        committee tables and the working-state view.  The registry itself (balances, pubkeys) is the caller's input and
        is passed again to import_state."""
        nb, nv = self.num_blocks, self.num_validators
        roots = np.zeros((nb, 32), dtype=np.uint8)
        parent = np.zeros(nb, dtype=np.uint32)
        slot = np.zeros(nb, dtype=np.uint64)
        pj_e, pf_e = np.zeros(nb, dtype=np.uint64), np.zeros(nb, dtype=np.uint64)
        pj_r, pf_r = np.zeros((nb, 32), dtype=np.uint8), np.zeros((nb, 32), dtype=np.uint8)
        r, jr, fr = ((C.c_uint8 * 32)() for _ in range(3))
        pi, sl, je, fe = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        for i in range(nb):
            self._check(self._lib.pe_get_block(self._h, i, r, C.byref(pi), C.byref(sl), C.byref(je), jr, C.byref(fe), fr))
            roots[i], parent[i], slot[i] = np.frombuffer(r, dtype=np.uint8), pi.value, sl.value
            pj_e[i], pf_e[i] = je.value, fe.value
            pj_r[i], pf_r[i] = np.frombuffer(jr, dtype=np.uint8), np.frombuffer(fr, dtype=np.uint8)
        flags = np.zeros(max(nv, 1), dtype=np.uint8)
        lm_slot = np.zeros(max(nv, 1), dtype=np.uint32)
        if nv:
            self._check(self._lib.pe_get_validator_flags(self._h, _ptr(flags), nv))
            self._check(self._lib.pe_get_latest_message_slots(self._h, _ptr(lm_slot), nv))
        ep, bi = self.latest_messages()
        sbal, sflags, s_set = self.state_validators()
        return dict(committees={e: self.committees(e) for e in self.committee_epochs()},
                    state_view=(sbal.copy(), sflags.copy()) if s_set else None,
                    scalars=self.store_scalars(), roots=roots, parent=parent, slot=slot, post_justified_epoch=pj_e,
                    post_justified_root=pj_r, post_finalized_epoch=pf_e, post_finalized_root=pf_r, flags=flags[:nv],
                    lm_epoch=ep.copy(), lm_block=bi.copy(), lm_slot=lm_slot[:nv],
                    participation=(self.participation_get(0), self.participation_get(1)))

    def import_state(self, st: dict, effective_balance, pubkeys96=None):
        """Rebuild the store exported by export_state on this (fresh) handle."""
        sc = st["scalars"]
        roots = st["roots"]
        self.store_init(sc["genesis_time"], int(st["slot"][0]), roots[0].tobytes())
        for i in range(1, roots.shape[0]):
            self.add_block(roots[i].tobytes(), roots[int(st["parent"][i])].tobytes(), int(st["slot"][i]),
                           (int(st["post_justified_epoch"][i]), st["post_justified_root"][i].tobytes()),
                           (int(st["post_finalized_epoch"][i]), st["post_finalized_root"][i].tobytes()))
        flags = np.ascontiguousarray(st["flags"], dtype=np.uint8)
        self.set_validators(effective_balance, flags & np.uint8(0xFF ^ _abi.PE_VAL_EQUIVOCATING), pubkeys96)
        equiv = np.nonzero(flags & _abi.PE_VAL_EQUIVOCATING)[0]
        if equiv.size:
            self.mark_equivocating(equiv)
        ep = np.ascontiguousarray(st["lm_epoch"], dtype=np.uint64)
        bi = np.ascontiguousarray(st["lm_block"], dtype=np.uint32)
        sl = np.ascontiguousarray(st["lm_slot"], dtype=np.uint32)
        self._check(self._lib.pe_set_latest_messages(self._h, ep.size, _ptr(ep), _ptr(bi), _ptr(sl)))
        self.on_tick(sc["time"])
        self.set_checkpoints(sc["justified"], sc["finalized"])
        self._check(self._lib.pe_set_best_justified(self._h, sc["best_justified"][0], _root(sc["best_justified"][1])))
        self.set_proposer_boost(sc["proposer_boost_root"])
        self.participation_set(0, st["participation"][0])
        self.participation_set(1, st["participation"][1])
        for epoch, (offsets, members) in sorted(st.get("committees", {}).items()):
            self.set_committees(epoch, offsets, members)
        if st.get("state_view") is not None:
            self.state_set_validators(*st["state_view"])

    def store_scalars(self) -> dict:
        t, g, je, fe, be = (C.c_uint64(0) for _ in range(5))
        jr, fr, br, boost = ((C.c_uint8 * 32)() for _ in range(4))
        self._check(self._lib.pe_get_store_scalars(self._h, C.byref(t), C.byref(g), C.byref(je), jr, C.byref(fe), fr,
                                                   C.byref(be), br, boost))
        return dict(time=t.value, genesis_time=g.value, justified=(je.value, bytes(jr)),
                    finalized=(fe.value, bytes(fr)), best_justified=(be.value, bytes(br)),
                    proposer_boost_root=bytes(boost))

    # -- profiling --------------------------------------------------------
    def profile_enable(self, on=True):
        """0 / False off, 1 / True per-kernel totals, 3 totals of the accumulation only, 2 totals + the timeline of
        profile_timeline()."""
        self._check(self._lib.pe_profile_enable(self._h, int(on)))

    def profile_timeline(self):
        """-> list of (kernel name, start_ms, duration_ms) of the launches bracketed since profile_reset(), by start time
        (pe_profile_timeline; profile_enable(2))."""
        n = C.c_uint32(0)
        self._check(self._lib.pe_profile_timeline(self._h, None, None, None, 0, C.byref(n)))
        k = np.zeros(max(n.value, 1), dtype=np.int32)
        t0 = np.zeros(max(n.value, 1), dtype=np.float64)
        dt = np.zeros(max(n.value, 1), dtype=np.float64)
        self._check(self._lib.pe_profile_timeline(self._h, _ptr(k, C.c_int32), _ptr(t0, C.c_double), _ptr(dt, C.c_double),
                                                  n.value, C.byref(n)))
        rows = [(_abi.KERNEL_NAMES[int(k[i])], float(t0[i]), float(dt[i])) for i in range(min(n.value, k.size))]
        return sorted(rows, key=lambda r: r[1])

    def profile_queue_classes(self):
        """pe_profile_queue_classes: for the engine / accumulation / tree / finish streams, the lowest of them that shares
        its hardware queue -- [0, 1, 2, 3] when each has a queue of its own."""
        out = np.zeros(4, dtype=np.int32)
        self._check(self._lib.pe_profile_queue_classes(self._h, _ptr(out, C.c_int32)))
        return [int(x) for x in out]

    def profile_arena_growths(self) -> int:
        """pe_profile_arena_growths: (re)allocations of arena buffers since the handle was created."""
        n = C.c_uint64(0)
        self._check(self._lib.pe_profile_arena_growths(self._h, C.byref(n)))
        return int(n.value)

    def profile_accumulate_mhz(self):
        """pe_profile_accumulate_mhz: the shader clock (MHz) of each k_g1_accumulate launch since profile_reset(), in launch
        order (launches made while profiling was on; the last 4096)."""
        out = np.zeros(4096, dtype=np.float64)
        n = C.c_uint32(0)
        self._check(self._lib.pe_profile_accumulate_mhz(self._h, out.ctypes.data_as(C.c_void_p), 4096, C.byref(n)))
        return out[:min(int(n.value), 4096)].copy()

    def profile_reset(self):
        self._check(self._lib.pe_profile_reset(self._h))

    def profile(self) -> dict:
        out = {}
        for k, name in enumerate(_abi.KERNEL_NAMES):
            n, ms = C.c_uint64(0), C.c_double(0)
            self._check(self._lib.pe_profile_get(self._h, k, C.byref(n), C.byref(ms)))
            out[name] = dict(launches=n.value, total_ms=ms.value)
        return out
