"""-m gpu: the paired launches of a streaming caller (engine_pair.cpp / pair_kernels.hip, round 5).  In a streaming pipeline over
rows in device memory pe_on_attestation_batch and pe_get_head_async hold their kernels back; they go out with the NEXT step's
pe_aggregate, each as a block range of the same grid as one of its row kernels -- or alone and in order the moment anything
else needs the stream.  Every way into and out of that state is driven here against a twin engine that makes synchronous
calls over host rows (the path the other -m gpu tests hold against the oracle): same outputs step by step, same store."""
import types

import numpy as np
import pytest

import bench
import pos_evolution_amd as pea
from pos_evolution_amd import RESIDENT, ROWS_RESIDENT

pytestmark = pytest.mark.gpu
PAIRS = ("pair_ingest_validate", "pair_plan_lmd", "pair_members_votes", "pair_union_tree")


def _args(V, C, B, n_epochs, **kw):
    return types.SimpleNamespace(validators_local=V, blocks=B, committees=C, parts=kw.get("parts", 3), mixed_balances=True,
                                 host_arena=False, host_rows=False, with_shuffle=False, by_committee=False, world=1,
                                 shuffle_variant_from=n_epochs, tree_kind=kw.get("tree_kind", "bushy"),
                                 equivocating_frac=kw.get("equivocating_frac", 0.0), boost=kw.get("boost", False))


def _twin(w, n_epochs):
    """A fresh engine with the workload's store and registry, for synchronous host-row calls."""
    e2 = pea.Engine(max_committee_tables=n_epochs + 3)
    tree = w["tree"]
    e2.store_init(0, 0, tree.roots[0].tobytes())
    for i in range(1, tree.roots.shape[0]):
        e2.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
    bench.load_registry(e2, w)
    for st in w["steps"]:
        e2.set_committees(st["epoch"], st["comm"].offsets, st["comm"].members)
    return e2


def _stream_step(e, w, st, head=True, att=True):
    """One streaming step over device rows; head / att: leave the call out (a client need not make all four)."""
    e.on_tick((st["epoch"] + 1) * w["spe"] * 12)
    if "boost_idx" in st:
        e.set_proposer_boost(w["tree"].roots[st["boost_idx"]].tobytes())
    e.participation_rotate()
    cap = st["comm"].offsets.size - 1
    with e.pipeline(lagged=True):
        agg = e.aggregate(packed=(st["rows_in"], st["arena_in"]), want_aggregate_pubkeys=True)
        r = dict(agg=agg)
        if att:
            r["status"], _, r["count"] = e.on_attestation_batch(packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
        if head:
            r["head"] = e.get_head_async()
        r["pstatus"], r["numerators"] = e.process_attestation_batch(st["ctx"], packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
    return r


def _sync_step(e, w, st, head=True, att=True):
    e.on_tick((st["epoch"] + 1) * w["spe"] * 12)
    if "boost_idx" in st:
        e.set_proposer_boost(w["tree"].roots[st["boost_idx"]].tobytes())
    e.participation_rotate()
    agg = e.aggregate(packed=(st["atts"], st["arena"]), want_aggregate_pubkeys=True)
    r = dict(agg=agg)
    rows = agg["atts"]
    if att:
        r["status"], _, r["count"] = e.on_attestation_batch(packed=(rows, agg["out_arena"]))
    if head:
        r["head"] = e.get_head()
    r["pstatus"], r["numerators"] = e.process_attestation_batch(st["ctx"], packed=(rows, agg["out_arena"]))
    return r


def _same_step(a, b, k):
    g = int(a["agg"]["n_groups"])
    assert g == int(b["agg"]["n_groups"]), k
    for key in ("atts", "count", "aggpk96", "group_of"):
        assert np.array_equal(np.asarray(a["agg"][key])[:g] if key != "group_of" else a["agg"][key],
                              np.asarray(b["agg"][key])[:g] if key != "group_of" else b["agg"][key]), (k, key)
    assert np.array_equal(a["agg"]["out_arena"], b["agg"]["out_arena"]), k
    for key in ("status", "count", "pstatus", "numerators"):
        if key in a or key in b:
            assert np.array_equal(np.asarray(a[key])[:g], np.asarray(b[key])[:g]), (k, key)
    if "head" in a or "head" in b:
        assert bytes(a["head"]) == bytes(b["head"]), (k, "head")


def _same_store(ea, eb):
    la, lb = ea.latest_messages(), eb.latest_messages()
    assert np.array_equal(la[0], lb[0]) and np.array_equal(la[1], lb[1]), "latest messages"
    assert np.array_equal(ea.get_weights(), eb.get_weights())
    assert np.array_equal(ea.participation_get(0), eb.participation_get(0))
    assert np.array_equal(ea.participation_get(1), eb.participation_get(1))


def _launches(e):
    return {k: v["launches"] for k, v in e.profile().items()}


@pytest.mark.parametrize("lag", [1, 2, 4])
def test_every_step_but_the_last_goes_out_paired(lag):
    """N streaming steps: N - 1 launches of each pair (step k's fork choice beside step k + 1's rows), the last step's
    fork-choice kernels alone at the drain, nothing else of those kernels -- and every output equals the synchronous twin's."""
    n = 6
    e = pea.Engine(max_committee_tables=n + 3)
    w = bench.build_workload(e, _args(32768, 256, 300, n, boost=True, equivocating_frac=0.02), 0, n)
    e.set_pipeline_lag(lag)
    e.reuse_outputs(n + lag + 2)
    e.profile_enable(True)
    e.profile_reset()
    got = [_stream_step(e, w, st) for st in w["steps"]]
    e.drain()
    ln = _launches(e)
    e.profile_enable(False)
    assert all(ln[p] == n - 1 for p in PAIRS), ln
    assert ln["votes"] == 1 and ln["tree"] == 1 and ln["lmd"] == 1 and ln["bits_union"] == 1, ln
    assert 1 <= ln["g1_accumulate"] <= n          # totals mode brackets one accumulation in four (engine_g1.cpp)
    e2 = _twin(w, n)
    for k, st in enumerate(w["steps"]):
        _same_step(got[k], _sync_step(e2, w, st), k)
    _same_store(e, e2)
    e.close()
    e2.close()


def test_whatever_comes_between_two_steps(monkeypatch):
    """Between streaming steps: a synchronous read (pe_get_weights), a new block (the tree tables are rewritten), new
    equivocation marks, a step without pe_get_head_async, a step without pe_on_attestation_batch, a polled pe_get_head in
    front of the next step.  Held-back launches go out before any of it can be seen; twin: the same calls, synchronous."""
    n = 8
    e = pea.Engine(max_committee_tables=n + 3)
    w = bench.build_workload(e, _args(16384, 128, 200, n), 0, n)
    e2 = _twin(w, n)
    e.set_pipeline_lag(2)
    e.reuse_outputs(n + 6)
    e.profile_enable(True)
    e.profile_reset()
    tree = w["tree"]
    tip = tree.roots[tree.roots.shape[0] - 1].tobytes()
    new_root = bytes([7]) * 32
    got, want = [], []

    def both(fn):
        return fn(e), fn(e2)

    for k, st in enumerate(w["steps"]):
        head, att = k != 3, k != 4
        got.append(_stream_step(e, w, st, head=head, att=att))
        want.append(_sync_step(e2, w, st, head=head, att=att))
        if k == 0:
            a, b = both(lambda x: x.get_weights())                   # synchronous: completes everything first
            assert np.array_equal(a, b)
        if k == 1:
            both(lambda x: x.add_block(new_root, tip, int(tree.slot.max()) + 1))
        if k == 2:
            both(lambda x: x.mark_equivocating(np.arange(5, 16384, 41)))
        if k == 5:
            a, b = both(lambda x: x.get_head())                      # polled head between two streaming steps
            assert a == b
    e.drain()
    ln = _launches(e)
    e.profile_enable(False)
    for k in range(n):
        _same_step(got[k], want[k], k)
    _same_store(e, e2)
    # steps 3 -> 4 and 4 -> 5 went out half paired (no head held / no handlers held)
    assert 0 < ln["pair_members_votes"] < n - 1 and 0 < ln["pair_plan_lmd"] < n - 1, ln
    e.close()
    e2.close()


def test_a_tree_too_large_for_the_pair_launches_it_alone():
    """More than 4096 blocks: k_tree needs its 147 KB workgroup, beside which no union block fits -- that pair goes out as two
    launches, the other three stay paired."""
    n = 3
    e = pea.Engine(max_committee_tables=n + 3)
    w = bench.build_workload(e, _args(16384, 64, 5000, n), 0, n)
    e.set_pipeline_lag(2)
    e.reuse_outputs(n + 4)
    e.profile_enable(True)
    e.profile_reset()
    got = [_stream_step(e, w, st) for st in w["steps"]]
    e.drain()
    ln = _launches(e)
    e.profile_enable(False)
    assert ln["pair_union_tree"] == n - 1 and ln["pair_plan_lmd"] == n - 1, ln   # the bracket is the pair's, one or two launches
    e2 = _twin(w, n)
    for k, st in enumerate(w["steps"]):
        _same_step(got[k], _sync_step(e2, w, st), k)
    _same_store(e, e2)
    e.close()
    e2.close()


def test_pairing_switched_off_is_the_old_chain(monkeypatch):
    """POSEVO_PAIR=0 (the A/B switch): nothing is held back, every kernel runs alone, same results."""
    monkeypatch.setenv("POSEVO_PAIR", "0")
    n = 4
    e = pea.Engine(max_committee_tables=n + 3)
    w = bench.build_workload(e, _args(16384, 128, 200, n), 0, n)
    e.set_pipeline_lag(2)
    e.reuse_outputs(n + 4)
    e.profile_enable(True)
    e.profile_reset()
    got = [_stream_step(e, w, st) for st in w["steps"]]
    e.drain()
    ln = _launches(e)
    e.profile_enable(False)
    assert all(ln[p] == 0 for p in PAIRS) and ln["votes"] == n and ln["tree"] == n, ln
    monkeypatch.delenv("POSEVO_PAIR")
    e2 = _twin(w, n)
    for k, st in enumerate(w["steps"]):
        _same_step(got[k], _sync_step(e2, w, st), k)
    _same_store(e, e2)
    e.close()
    e2.close()


KNOB_SETS = [
    {"POSEVO_ACC_EXCLUSIVE": "0", "POSEVO_STATE_ON": "0"},                       # round 4's signatures and streams
    {"POSEVO_ACC_EXCLUSIVE": "1", "POSEVO_STATE_ON": "2"},                       # the defaults, spelled out
    {"POSEVO_ACC_EXCLUSIVE": "1", "POSEVO_STATE_ON": "1"},                       # the flag passes always on the tree's stream
    {"POSEVO_ACC_EXCLUSIVE": "1", "POSEVO_STATE_ON": "0"},
]


@pytest.mark.parametrize("knobs", KNOB_SETS, ids=lambda k: ",".join(f"{a[7:].lower()}={b}" for a, b in k.items()))
def test_the_scheduling_knobs_change_no_result(knobs, monkeypatch):
    """The scheduling knobs of the streaming G1 chain (engine_internal.h Tune) decide WHERE and WHEN a step's
    kernels run -- which stream carries the state-transition work, whether two accumulation workgroups may share a CU --
    never what they compute: every
    setting against the synchronous twin, step by step, over enough steps for every arena to be reused twice."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)   # read once per handle, when it is created
    n, lag = 9, 3
    e = pea.Engine(max_committee_tables=n + 3)
    w = bench.build_workload(e, _args(65536, 256, 700, n, equivocating_frac=0.01), 0, n)
    e.set_pipeline_lag(lag)
    e.reuse_outputs(n + lag + 2)
    got = [_stream_step(e, w, st) for st in w["steps"]]
    e.drain()
    for k in list(knobs):
        monkeypatch.delenv(k)
    e2 = _twin(w, n)
    for k, st in enumerate(w["steps"]):
        _same_step(got[k], _sync_step(e2, w, st), k)
    _same_store(e, e2)
    e.close()
    e2.close()


def test_steps_of_changing_size_move_the_flag_passes_between_streams_and_change_no_result():
    """Tune::state_on = 2 (the default): the flag passes of process_attestation ride the tree's stream in steps over more than
    1024 rows and the accumulation's in smaller ones (engine_core.cpp: state_stream_begin).  The numerators of a step depend on
    the flags its predecessors set (pe:744-750), so a change of stream must keep their order: steps of 1536 and 700 rows in
    every succession -- large, small, small, large, small, large, large -- against the synchronous twin, step by step."""
    from pos_evolution_amd import DeviceRows

    n, lag = 7, 3
    small = (False, True, True, False, True, False, False)
    e = pea.Engine(max_committee_tables=n + 3)
    w = bench.build_workload(e, _args(65536, 512, 600, n), 0, n)
    assert w["steps"][0]["atts"].shape[0] == 1536
    steps = []
    for st, sm in zip(w["steps"], small):
        if sm:  # the first 700 rows of the step, on the device and on the host
            st = dict(st)
            st["rows_in"] = DeviceRows(st["rows_in"].ptr, 700, keep=st["rows_in"].keep)
            st["atts"] = np.ascontiguousarray(st["atts"][:700])
        steps.append(st)
    e.set_pipeline_lag(lag)
    e.reuse_outputs(n + lag + 2)
    got = [_stream_step(e, w, st) for st in steps]
    e.drain()
    e2 = _twin(w, n)
    for k, st in enumerate(steps):
        _same_step(got[k], _sync_step(e2, w, st), k)
    _same_store(e, e2)
    e.close()
    e2.close()


# ---------------------------------------------------------------- collected signature legs (round 6)
def _signed_stream_step(e, w, st, sigs):
    e.on_tick((st["epoch"] + 1) * w["spe"] * 12)
    e.participation_rotate()
    cap = st["comm"].offsets.size - 1
    with e.pipeline(lagged=True):
        agg = e.aggregate_signed(sigs, packed=(st["rows_in"], st["arena_in"]), want_aggregate_pubkeys=True)
        status, _, count = e.on_attestation_batch(packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
        head = e.get_head_async()
        pst, num = e.process_attestation_batch(st["ctx"], packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
    return dict(agg=agg, status=status, count=count, head=head, pstatus=pst, numerators=num)


def _signed_sync_step(e, w, st, sigs):
    e.on_tick((st["epoch"] + 1) * w["spe"] * 12)
    e.participation_rotate()
    agg = e.aggregate_signed(sigs, packed=(st["atts"], st["arena"]), want_aggregate_pubkeys=True)
    rows = agg["atts"]
    status, _, count = e.on_attestation_batch(packed=(rows, agg["out_arena"]))
    head = e.get_head()
    pst, num = e.process_attestation_batch(st["ctx"], packed=(rows, agg["out_arena"]))
    return dict(agg=agg, status=status, count=count, head=head, pstatus=pst, numerators=num)


@pytest.mark.parametrize("batch,n,lag,where", [(4, 6, 5, "device"),    # 4 + 2: the drain flushes a half-filled batch
                                                (3, 7, 2, "device"),    # lag < batch: a lagged end flushes what its arena still waits for
                                                (4, 5, 5, "host"),      # a HOST signature buffer refilled for every step
                                                (1, 4, 3, "device")])   # a launch per step: round 5's shape
def test_collected_signature_legs_equal_the_synchronous_calls(batch, n, lag, where, monkeypatch):
    """pe_aggregate_signed in streaming steps: the legs of POSEVO_SIG_BATCH steps share ONE decompression launch
    (engine_g1.cpp sig_batch_flush); a pipeline that completes before its batch is full -- the drain, a lag shorter than the
    batch -- launches what has been collected.  Every step's aggregate signatures, per-row statuses and everything else of
    the step against a twin's synchronous host-row calls; signatures differ from step to step."""
    import torch
    import pos_evolution_amd.synth as synth
    from pos_evolution_amd import DeviceArena

    monkeypatch.setenv("POSEVO_SIG_BATCH", str(batch))
    e = pea.Engine(max_committee_tables=n + 3)
    monkeypatch.delenv("POSEVO_SIG_BATCH")
    w = bench.build_workload(e, _args(32768, 128, 300, n), 0, n)
    n_rows = len(w["steps"][0]["atts"])
    base = synth.signature_points(e, n_rows + n)          # step k signs row i with (a + (i + k) b) G2
    per_step = [np.ascontiguousarray(base[k:k + n_rows]) for k in range(n)]
    e.set_pipeline_lag(lag)
    e.reuse_outputs(n + lag + 2)
    got, keep = [], []
    host_buf = np.empty_like(per_step[0])
    for k, st in enumerate(w["steps"]):
        if where == "host":
            host_buf[:] = per_step[k]                     # the same buffer, new contents: read before the call returns
            sigs = host_buf
        else:
            t = torch.from_numpy(per_step[k].reshape(-1).copy()).cuda()
            keep.append(t)
            sigs = DeviceArena(t.data_ptr(), t.numel(), keep=t)
        got.append(_signed_stream_step(e, w, st, sigs))
    e.drain()
    e2 = _twin(w, n)
    for k, st in enumerate(w["steps"]):
        want = _signed_sync_step(e2, w, st, per_step[k])
        _same_step(got[k], want, k)
        g = int(want["agg"]["n_groups"])
        assert np.array_equal(np.asarray(got[k]["agg"]["sig96c"])[:g], np.asarray(want["agg"]["sig96c"])[:g]), (k, "signatures")
        assert np.array_equal(got[k]["agg"]["sig_status"], want["agg"]["sig_status"]) and not want["agg"]["sig_status"].any()
    _same_store(e, e2)
    e.close()
    e2.close()


@pytest.mark.parametrize("lag", [2, 6, 12])
def test_a_stream_of_like_steps_allocates_nothing_after_its_first_step(lag):
    """Every arena of the rotation is sized before the stream runs: an arena that sized itself at ITS first pipeline cost 7 ms of
    pinned allocations inside the stream (round 6: the driver's command read 1.5 ms per step at lag 8, the signed step 1.5 ms at
    lag 15) -- also the arenas that JOIN the rotation when the caller sets its lag after the registry load has grown the first
    three.  pe_profile_arena_growths counts every (re)allocation of an arena buffer; signed steps included (their scratch too)."""
    import torch
    import pos_evolution_amd.synth as synth
    from pos_evolution_amd import DeviceArena

    n = lag + 5
    e = pea.Engine(max_committee_tables=n + 3)
    w = bench.build_workload(e, _args(32768, 128, 300, n), 0, n)      # loads the registry: the default rotation's blocks grow
    sg = synth.signature_points(e, len(w["steps"][0]["atts"]))
    t = torch.from_numpy(sg.reshape(-1).copy()).cuda()
    sigs = DeviceArena(t.data_ptr(), t.numel(), keep=t)
    e.set_pipeline_lag(lag)
    e.reuse_outputs(n + lag + 2)
    got = [_signed_stream_step(e, w, w["steps"][0], sigs)]
    e.drain()
    before = e.profile_arena_growths()
    got += [_signed_stream_step(e, w, st, sigs) for st in w["steps"][1:]]
    e.drain()
    assert e.profile_arena_growths() == before, "an arena buffer was (re)allocated inside the stream"
    assert all(int(r["agg"]["n_groups"]) == 128 and not np.asarray(r["status"])[:128].any() for r in got)
    e.close()
