"""Run by tests/test_host_wrapper.py in a subprocess with POSEVO_LIB_PATH pointing at a STUB of libposevo.so (every
entry point of include/posevo.h present, none of them computing anything): what is checked here is the host logic of the
Python package above the C ABI -- argument marshalling, the output-set ring, pipeline book-keeping, the pyspec-level
guards of forkchoice.py -- which needs no GPU."""
import ctypes as C
import sys
import types

import numpy as np

import pos_evolution_amd as pea
import pos_evolution_amd.forkchoice as fc
import pos_evolution_amd.synth as synth
from pos_evolution_amd import _abi

lib = _abi.load()
log = (C.c_char * 65536).in_dll(lib, "stub_log")


def calls():
    names = bytes(log.value).decode().split()
    lib.stub_reset()
    return names


# ---- pack_attestations: rows -> 144-byte records + one bit arena ----
rows = [pea.AttRow(slot=5, index=1, beacon_block_root=b"\x11" * 32, source_epoch=0, source_root=b"\x00" * 32,
                   target_epoch=0, target_root=b"\x22" * 32, bits=[True, False, True] + [False] * 9 + [True],
                   signature_valid=True, is_from_block=False),
        pea.AttRow(slot=6, index=0, beacon_block_root=b"\x33" * 32, source_epoch=0, source_root=b"\x00" * 32,
                   target_epoch=0, target_root=b"\x22" * 32, bits=[False] * 7 + [True], signature_valid=False,
                   is_from_block=True)]
arr, arena = pea.pack_attestations(rows)
view = np.frombuffer(arr, dtype=synth.ATT_DTYPE, count=2)
assert list(view["n_bits"]) == [13, 8] and list(view["bits_offset"]) == [0, 2]
assert bytes(arena[:3]) == bytes([0b00000101, 0b00010000, 0b10000000])
assert list(view["flags"]) == [_abi.PE_ATT_FLAG_SIGNATURE_VALID, _abi.PE_ATT_FLAG_FROM_BLOCK]
assert view["beacon_block_root"][1].tobytes() == b"\x33" * 32 and int(view["slot"][0]) == 5

# ---- Engine: create / call order / output-set ring / pipelines ----
e = pea.Engine()
assert calls()[:1] == ["pe_engine_create"]
atts = np.zeros(8, dtype=synth.ATT_DTYPE)
atts["n_bits"], atts["bits_offset"] = 64, np.arange(8) * 8
bits = np.zeros(64, dtype=np.uint8)
ctx = _abi.pe_state_ctx()


def step(lagged):
    with e.pipeline(lagged=lagged):
        agg = e.aggregate(packed=(atts, bits), want_aggregate_pubkeys=True)
        st, _, cnt = e.on_attestation_batch(packed=(agg["atts"], pea.RESIDENT))
        head = e.get_head()
        pst, num = e.process_attestation_batch(ctx, packed=(agg["atts"], pea.RESIDENT))
    return agg, st, cnt, pst, num, head


r = step(False)
assert calls() == ["pe_pipeline_begin", "pe_aggregate", "pe_on_attestation_batch", "pe_get_head",
                   "pe_process_attestation_batch", "pe_pipeline_end"]
assert r[0]["n_groups"] == 2 and len(r[0]["atts"]) == 2 and r[0]["aggpk96"].shape == (2, 96)   # the stub forms n / 4 groups
assert r[0]["out_arena"].size == 8 + 8            # trimmed to the last group's bits (stub: offsets 0 and 8, 64 bits each)
r = step(True)
assert calls() == ["pe_pipeline_begin_streaming", "pe_aggregate", "pe_on_attestation_batch", "pe_get_head",
                   "pe_process_attestation_batch", "pe_pipeline_end_lagged"]
# without reuse_outputs every call gets fresh arrays; with a ring of depth d the arrays of pipeline k return at k + d
a1, a2 = step(True), step(True)
assert a1[1].ctypes.data != a2[1].ctypes.data
e.reuse_outputs(3)
try:                                    # lag depth 2: a set handed out again before k + 3 is never stable to read
    step(True)
    raise SystemExit("a lagged pipeline must refuse a ring shallower than 4")
except ValueError:
    pass
e.reuse_outputs()                       # default depth 4
ring = [step(True) for _ in range(9)]
addr = [x[1].ctypes.data for x in ring]
assert len(set(addr[:4])) == 4 and addr[4:8] == addr[:4] and addr[8] == addr[0]
assert all(ring[k][0]["aggpk96"].ctypes.data == ring[k + 4][0]["aggpk96"].ctypes.data for k in range(5))
e.drain()
# two calls of one kind inside ONE pipeline get distinct output sets (they aliased), and the pair returns
# with the ring slot
def two_aggregates():
    with e.pipeline():
        x = e.aggregate(packed=(atts, bits), want_aggregate_pubkeys=True)
        y = e.aggregate(packed=(atts, bits), want_aggregate_pubkeys=True)
    return x, y
pairs = [two_aggregates() for _ in range(5)]
for x, y in pairs:
    assert x["aggpk96"].ctypes.data != y["aggpk96"].ctypes.data and x["atts"].ctypes.data != y["atts"].ctypes.data
assert pairs[0][0]["aggpk96"].ctypes.data == pairs[4][0]["aggpk96"].ctypes.data
assert pairs[0][1]["aggpk96"].ctypes.data == pairs[4][1]["aggpk96"].ctypes.data
# synchronous calls outside a pipeline advance the ring too: consecutive results do not alias
s1 = e.aggregate(packed=(atts, bits), want_aggregate_pubkeys=True)
s2 = e.aggregate(packed=(atts, bits), want_aggregate_pubkeys=True)
assert s1["aggpk96"].ctypes.data != s2["aggpk96"].ctypes.data
assert "pe_pipeline_end" in calls()
# a DeviceArena travels as its address; the host-repacking calls never see one from the wrapper's own paths
dev = pea.DeviceArena(0xDEAD0000, 64)
with e.pipeline():
    e.aggregate(packed=(atts, dev))
lib.stub_last_arena.restype = C.c_size_t
assert lib.stub_last_arena() == 0xDEAD0000

# ---- error mapping: a failing status raises EngineError carrying it ----
lib.stub_fail_next(-10)
try:
    e.get_head()
    raise SystemExit("expected EngineError")
except pea.EngineError as err:
    assert err.status == -10 and isinstance(err, AssertionError)

# ---- forkchoice.py guards ----
store = fc.Store(e)
store.balances_checkpoint = fc.Checkpoint(3, b"\x01" * 32)      # the engine's balances belong to another checkpoint
try:
    fc.get_head(store)
    raise SystemExit("get_head must refuse balances of another justified state")
except pea.EngineError as err:
    assert err.status == _abi.PE_ERR_STATE
seen = []
state = types.SimpleNamespace(slot=64, validators=[types.SimpleNamespace(effective_balance=32 * 10**9, activation_epoch=0,
                                                                        exit_epoch=2**64 - 1, slashed=False, pubkey=None)] * 4)
store.checkpoint_state_provider = lambda cp: (seen.append(cp), state)[1]
lib.stub_reset()
assert len(fc.get_head(store)) == 32 and len(seen) == 1
assert "pe_set_validators" in calls() and store.balances_checkpoint == store.justified_checkpoint
fc.get_head(store)
assert len(seen) == 1                                            # provider asked once per justified checkpoint
# process_attestation: unbound state / missing proposer are refused before anything reaches the engine
st2 = types.SimpleNamespace()
for kwargs in ({}, {"proposer_index": 0}):
    try:
        fc.process_attestation(st2, object(), **kwargs)
        raise SystemExit("process_attestation must refuse an unbound state")
    except AssertionError:
        pass
# ---- sharded.py over the engine's own collectives: id from rank 0, init, then the two sharded calls ----
from pos_evolution_amd.sharded import ShardedForkChoice
lib.stub_reset()
sh = ShardedForkChoice(e, n_groups_max=4, use_engine_rccl=True)
assert calls() == ["pe_dist_unique_id", "pe_dist_init_ex"]
assert len(sh.get_head()) == 32
with e.pipeline(lagged=True):
    res = sh.aggregate(packed=(atts, bits))
    e.on_attestation_batch(packed=(res["atts"], pea.RESIDENT))
    sh.get_head()
assert calls() == ["pe_get_head_sharded", "pe_pipeline_begin_streaming", "pe_aggregate_sharded", "pe_on_attestation_batch",
                   "pe_get_head_sharded", "pe_pipeline_end_lagged"]
# ---- round 3: lag depth, rows in device memory, the signature leg, the committee-sharded exchange, custom collectives ----
lib.stub_reset()
e.drain()
e.reuse_outputs(4)
e.set_pipeline_lag(4)                    # pe_pipeline_set_lag; the ring must then hold lag + 2 sets
assert "pe_pipeline_set_lag" in calls()
try:
    step(True)
    raise SystemExit("a ring of 4 is too shallow for lag depth 4")
except ValueError:
    pass
e.reuse_outputs(6)
ring = [step(True) for _ in range(13)]
addr = [x[1].ctypes.data for x in ring]
assert len(set(addr[:6])) == 6 and addr[6:12] == addr[:6]
assert len(e._lagged_keep) == 4          # this block's buffers + the three before it stay alive
e.drain()
e.set_pipeline_lag(2)
e.reuse_outputs(4)
lib.stub_reset()
rows = pea.DeviceRows(atts.ctypes.data, 8)   # rows "in device memory" travel as their address (the stub reads them: host memory here)
with e.pipeline(lagged=True):
    ragg = e.aggregate(packed=(rows, pea.DeviceArena(0xDEAD0000, 64)), want_aggregate_pubkeys=True)
    st_r, _, cnt_r = e.on_attestation_batch(packed=(pea.ROWS_RESIDENT, pea.RESIDENT), cap=8)
    hd = e.get_head_async()
    pst_r, num_r = e.process_attestation_batch(ctx, packed=(pea.ROWS_RESIDENT, pea.RESIDENT), cap=8)
    e.compute_committees_async(7, b"\x05" * 32, 64, 32)
assert calls() == ["pe_pipeline_begin_streaming", "pe_aggregate", "pe_on_attestation_batch", "pe_get_head_async",
                   "pe_process_attestation_batch", "pe_compute_committees_async", "pe_pipeline_end_lagged"]
assert st_r.shape == (8,) and num_r.shape == (8,) and hd.shape == (32,) and ragg["n_groups"] == 2   # the stub's n / 4
sigs = np.zeros((8, 96), dtype=np.uint8)
sres = e.aggregate_signed(sigs, packed=(atts, bits), compressed=True, check_subgroup=True)
assert "pe_aggregate_signed" in calls() and sres["sig96c"].shape[1] == 96 and sres["sig_status"].shape == (8,)
try:
    e.aggregate_signed(sigs[:3], packed=(atts, bits))
    raise SystemExit("one signature per input row")
except AssertionError:
    pass
lib.stub_reset()
seen_cb = []
e.dist_init_custom(0, 2, lambda buf, count, stream: seen_cb.append("ar") or 0, lambda s_, r_, nb, st_: seen_cb.append("ag") or 0)
e.dist_set_max_groups(5)
gx = e.aggregate_exchange(cap_groups=10)
assert calls() == ["pe_dist_init_custom", "pe_dist_set_max_groups", "pe_aggregate_exchange"]
assert gx["n_groups"] == 0 and e._coll_keep is not None        # the callback table outlives the call
e.dist_destroy()
assert e._coll_keep is None
# ---- profiling: per-kernel totals, and the timeline mode (the stub brackets nothing) ----
lib.stub_reset()
e.profile_enable(2)
e.profile_reset()
assert e.profile_timeline() == [] and set(e.profile()) >= {"g1_accumulate", "att_group", "att_validate"}
assert calls()[:3] == ["pe_profile_enable", "pe_profile_reset", "pe_profile_timeline"]
print("host wrapper ok")
