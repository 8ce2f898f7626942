"""-m gpu: the awkward corners of pipelined / streaming calls: a failing call inside a pipeline, staging blocks that
have to grow while work is enqueued, store mutations between streaming pipelines, a handle destroyed or re-initialised
with pipelines in flight.  The reference point is always a twin engine driven with synchronous calls."""
import hashlib

import numpy as np
import pytest

import pos_evolution_amd as pea
import pos_evolution_amd.synth as synth
from pos_evolution_amd._abi import pe_state_ctx
from tests.test_gpu_pipeline import _same_results, _step_sync, _world

pytestmark = pytest.mark.gpu


def _step_streaming(w, lagged=True):
    e = w["e"]
    with e.pipeline(lagged=lagged):
        agg = e.aggregate(packed=(w["atts"], w["arena"]), want_aggregate_pubkeys=True)
        status, _, count = e.on_attestation_batch(packed=(agg["atts"], pea.RESIDENT))
        head = e.get_head()
        pst, num = e.process_attestation_batch(w["ctx"], packed=(agg["atts"], pea.RESIDENT))
    return agg, status, count, pst, num, head


def test_failing_call_inside_a_pipeline_leaves_the_rest_intact(engine_factory):
    """A call that is refused inside a pipeline (rows that are not rows of the resident aggregate) changes nothing; the
    calls around it complete as usual."""
    wa = _world(engine_factory, 12000, 64, seed=31)
    wb = _world(engine_factory, 12000, 64, seed=31)
    ra = _step_sync(wa)
    e = wb["e"]
    with e.pipeline():
        agg = e.aggregate(packed=(wb["atts"], wb["arena"]), want_aggregate_pubkeys=True)
        foreign = wb["atts"][:5].copy()
        foreign["bits_offset"] += 3           # no group of the aggregate starts there
        with pytest.raises(AssertionError) as ei:
            e.on_attestation_batch(packed=(foreign, pea.RESIDENT))
        assert ei.value.status == -1   # PE_ERR_INVALID_ARG
        status, _, count = e.on_attestation_batch(packed=(agg["atts"], pea.RESIDENT))
        pst, num = e.process_attestation_batch(wb["ctx"], packed=(agg["atts"], pea.RESIDENT))
        head = e.get_head()
    _same_results(ra, (agg, status, count, pst, num, head))
    # the same inside a streaming pipeline, where the G1 launch is deferred behind get_head
    for w in (wa, wb):
        w["e"].participation_rotate()
    ra = _step_sync(wa)
    with e.pipeline(lagged=True):
        agg = e.aggregate(packed=(wb["atts"], wb["arena"]), want_aggregate_pubkeys=True)
        with pytest.raises(AssertionError):
            e.on_attestation_batch(packed=(foreign, pea.RESIDENT))
        status, _, count = e.on_attestation_batch(packed=(agg["atts"], pea.RESIDENT))
        head = e.get_head()
        pst, num = e.process_attestation_batch(wb["ctx"], packed=(agg["atts"], pea.RESIDENT))
    e.drain()
    _same_results(ra, (agg, status, count, pst, num, head))
    assert np.array_equal(wa["e"].get_weights(), e.get_weights())
    assert np.array_equal(wa["e"].participation_get(0), e.participation_get(0))


def test_staging_blocks_grow_while_streaming(engine_factory):
    """Streaming pipelines whose batches grow from one step to the next: the pinned staging / output blocks and the
    resident buffers are re-allocated with earlier steps still in flight.  Every step equals the synchronous twin."""
    n_val, spe = 60000, 32
    worlds = [_world(engine_factory, n_val, 64, seed=41), _world(engine_factory, n_val, 64, seed=41)]
    tree = worlds[0]["tree"]
    out = [[], []]
    for k, (n_comm, parts) in enumerate([(32, 1), (64, 2), (256, 3), (2048, 4), (64, 1), (1024, 6)]):
        ep = worlds[0]["epoch"] + k
        seed = hashlib.sha256(b"grow%d" % k).digest()
        for w in worlds:   # an engine keeps a handful of tables: each epoch's is computed right before its step
            off, mem = w["e"].compute_committees(ep, seed, n_val, n_comm, 8)
        comm = synth.Committees(off, mem)
        atts, arena, _ = synth.epoch_attestations(comm, tree, ep, spe, seed=41 + k, density=0.9, parts=parts,
                                                  source=(0, tree.roots[0].tobytes()))
        ctx = pe_state_ctx()
        ctx.slot = (ep + 1) * spe
        ctx.chain_tip_root[:] = tree.roots[-1].tobytes()
        ctx.current_justified_root[:] = tree.roots[0].tobytes()
        ctx.previous_justified_root[:] = tree.roots[0].tobytes()
        ctx.base_reward_per_increment = 300 + k
        for which, w in enumerate(worlds):
            e = w["e"]
            e.on_tick((ep + 1) * spe * 12)
            e.participation_rotate()
            wk = dict(e=e, atts=atts, arena=arena, ctx=ctx)
            out[which].append(_step_sync(wk) if which == 0 else _step_streaming(wk))
    worlds[1]["e"].drain()
    for a, b in zip(*out):
        _same_results(a, b)
        assert (b[1] == 0).all() and (b[3] == 0).all()
    assert np.array_equal(worlds[0]["e"].get_weights(), worlds[1]["e"].get_weights())


def test_store_mutations_between_streaming_pipelines(engine_factory):
    """on_block / on_tick / proposer boost / equivocation marks / a balance refresh between streaming pipelines: the
    ones that touch device state first complete what is in flight (they are synchronous entry points), the host-only
    ones (on_tick, the boost root) do not need to, and the next step sees the new store -- exactly as with synchronous
    calls."""
    n_val, n_comm, spe = 20000, 64, 32
    worlds = [_world(engine_factory, n_val, n_comm, seed=51), _world(engine_factory, n_val, n_comm, seed=51)]
    tree = worlds[0]["tree"]
    tip = tree.roots[-1].tobytes()
    heads = [[], []]
    for which, w in enumerate(worlds):
        e = w["e"]
        step = _step_sync if which == 0 else _step_streaming
        res = [step(w)]
        new1 = b"\x51" * 32
        e.on_block(new1, tip, int(tree.slot.max()) + 1, (0, tree.roots[0].tobytes()), (0, tree.roots[0].tobytes()))
        e.set_proposer_boost(new1)
        e.participation_rotate()
        res.append(step(w))
        e.mark_equivocating(np.arange(0, n_val, 97))
        bal2 = w["bal"].copy()
        bal2[::3] = 17_000_000_000
        e.set_balances(bal2, w["flags"])
        e.participation_rotate()
        res.append(step(w))
        e.on_tick((w["epoch"] + 1) * spe * 12 + 12)   # a new slot: the boost is cleared (pe:943-944)
        e.participation_rotate()
        res.append(step(w))
        if which == 1:
            e.drain()
        heads[which] = res
    for a, b in zip(*heads):
        _same_results(a, b)
    assert np.array_equal(worlds[0]["e"].get_weights(), worlds[1]["e"].get_weights())


def test_destroy_and_reinit_with_pipelines_in_flight(engine_factory):
    """A handle closed, or its store re-initialised, while streaming pipelines are still in flight: no hang, no crash,
    and the outputs of the steps already issued are complete (closing a handle drains it first)."""
    w = _world(engine_factory, 30000, 128, seed=61)
    twin = _world(engine_factory, 30000, 128, seed=61)
    ref = _step_sync(twin)
    e = w["e"]
    got = [_step_streaming(w) for _ in range(3)]   # same votes each time; flags are set by the first step only
    e.store_init(0, int(w["tree"].slot[0]), w["tree"].roots[0].tobytes())   # synchronous: completes what is in flight
    _same_results(ref, got[0])
    with pytest.raises(AssertionError):
        e.on_attestation_batch(packed=(got[0][0]["atts"], pea.RESIDENT))   # the resident aggregate went with the store
    # ... and a handle that is simply closed with work in flight
    w2 = _world(engine_factory, 30000, 128, seed=61)
    got2 = [_step_streaming(w2) for _ in range(2)]
    w2["e"].close()
    _same_results(ref, got2[0])


@pytest.mark.parametrize("lagged", [False, True])
def test_two_aggregates_in_one_pipeline(engine_factory, lagged):
    """Two pe_aggregate calls inside one pipeline (the slots of an epoch in two halves): the second replaces the first as
    the resident aggregate, both share the arena's G1 scratch on the side stream, and each one's rows are handed to the
    handlers right after it.  Outputs equal the synchronous calls on a twin engine."""
    wa = _world(engine_factory, 30000, 128, seed=71, density=0.85, parts=2)
    wb = _world(engine_factory, 30000, 128, seed=71, density=0.85, parts=2)
    first = wa["atts"]["slot"] % 32 < 16
    halves = []
    for mask in (first, ~first):
        halves.append(np.ascontiguousarray(wa["atts"][mask]))
    ref = []
    ea = wa["e"]
    for atts in halves:
        agg = ea.aggregate(packed=(atts, wa["arena"]), want_aggregate_pubkeys=True)
        st, _, cnt = ea.on_attestation_batch(packed=(agg["atts"], agg["out_arena"]))
        ref.append((agg, st, cnt))
    ref_head = ea.get_head()
    e = wb["e"]
    got = []
    # five rounds, every one checked: from the third on the arenas are pre-sized and nothing flushes a deferred G1
    # launch by accident -- the steady state in which a second aggregate used to overwrite the unions the first one's
    # deferred pubkey sum still had to read
    for rep in range(5):
        got = []
        with e.pipeline(lagged=lagged):
            for atts in halves:
                agg = e.aggregate(packed=(atts, wb["arena"]), want_aggregate_pubkeys=True)
                st, _, cnt = e.on_attestation_batch(packed=(agg["atts"], pea.RESIDENT))
                got.append((agg, st, cnt))
            head = e.get_head()
        e.drain()
        for (ra, rs, rc), (ga, gs, gc) in zip(ref, got):
            assert ga["n_groups"] == ra["n_groups"]
            for k in ("aggpk96", "count", "out_arena", "group_of"):
                assert np.array_equal(ga[k], ra[k]), (rep, k)
            assert np.array_equal(gs, rs) and np.array_equal(gc, rc) and (gs == 0).all()
    assert head == ref_head
    assert np.array_equal(e.get_weights(), ea.get_weights())
    # the first aggregate is no longer resident once the second one was made
    with pytest.raises(AssertionError):
        with e.pipeline():
            a1 = e.aggregate(packed=(halves[0], wb["arena"]), want_aggregate_pubkeys=True)
            e.aggregate(packed=(halves[1], wb["arena"]), want_aggregate_pubkeys=True)
            e.on_attestation_batch(packed=(a1["atts"], pea.RESIDENT))


@pytest.mark.parametrize("dev_rows", [False, True])
def test_two_aggregates_with_process_attestation_between(engine_factory, dev_rows):
    """aggregate -> on_attestation -> process_attestation -> aggregate -> ... inside ONE pipeline: the flag pass of the
    first half runs on the state-transition stream and reads the arena's group descriptors, member lists and resident
    union words, which the second aggregate rewrites on the engine's stream -- the second aggregate must be ordered behind
    it (before, only the G1 side stream was joined).  Rows in host memory and in HBM, ten rounds each (a
    missing join shows as a reward numerator or a participation byte of the first half computed over the second half's
    unions), against synchronous calls on a twin."""
    import torch

    wa = _world(engine_factory, 30000, 128, seed=73, density=0.85, parts=2)
    wb = _world(engine_factory, 30000, 128, seed=73, density=0.85, parts=2)
    first = wa["atts"]["slot"] % 32 < 16
    halves = [np.ascontiguousarray(wa["atts"][m]) for m in (first, ~first)]
    ea, e = wa["e"], wb["e"]
    part0 = (ea.participation_get(0).copy(), ea.participation_get(1).copy())
    ref = []
    for atts in halves:
        agg = ea.aggregate(packed=(atts, wa["arena"]), want_aggregate_pubkeys=True)
        st, _, cnt = ea.on_attestation_batch(packed=(agg["atts"], agg["out_arena"]))
        pst, num = ea.process_attestation_batch(wa["ctx"], packed=(agg["atts"], agg["out_arena"]))
        ref.append((agg, st.copy(), cnt.copy(), pst.copy(), num.copy()))
    ref_part = (ea.participation_get(0).copy(), ea.participation_get(1).copy())
    assert sum(int(r[4].sum()) for r in ref) > 0
    keep = []
    for rep in range(10):
        e.participation_set(0, part0[0])
        e.participation_set(1, part0[1])
        got = []
        with e.pipeline():
            for atts in halves:
                if dev_rows:
                    t = torch.from_numpy(atts.view(np.uint8).reshape(-1).copy()).cuda()
                    rows = pea.DeviceRows(t.data_ptr(), len(atts), keep=t)
                    keep.append(rows)
                    agg = e.aggregate(packed=(rows, wb["arena"]), want_aggregate_pubkeys=True)
                    st, _, cnt = e.on_attestation_batch(packed=(pea.ROWS_RESIDENT, pea.RESIDENT), cap=len(atts))
                    pst, num = e.process_attestation_batch(wb["ctx"], packed=(pea.ROWS_RESIDENT, pea.RESIDENT), cap=len(atts))
                else:
                    agg = e.aggregate(packed=(atts, wb["arena"]), want_aggregate_pubkeys=True)
                    st, _, cnt = e.on_attestation_batch(packed=(agg["atts"], pea.RESIDENT))
                    pst, num = e.process_attestation_batch(wb["ctx"], packed=(agg["atts"], pea.RESIDENT))
                got.append((agg, st, cnt, pst, num))
        for (ra, rs, rc, rp, rn), (ga, gs, gc, gp, gn) in zip(ref, got):
            ng = ra["n_groups"]
            assert ga["n_groups"] == ng
            assert np.array_equal(ga["aggpk96"][:ng], ra["aggpk96"][:ng]), rep
            assert np.array_equal(gs[:ng], rs[:ng]) and np.array_equal(gc[:ng], rc[:ng]), rep
            assert np.array_equal(gp[:ng], rp[:ng]) and np.array_equal(gn[:ng], rn[:ng]), rep
        assert np.array_equal(e.participation_get(0), ref_part[0]), rep
        assert np.array_equal(e.participation_get(1), ref_part[1]), rep


def test_a_handles_hot_streams_get_a_hardware_queue_each_whatever_was_created_before():
    """The runtime maps streams onto four hardware queues in creation order and a queue runs its packets in order: a handle
    created after other streams -- torch's, another handle's -- used to share queues with ITSELF (round 5: its finish kernel in
    its row chain's queue, 308 us per slot-step instead of 120).  pe_engine_create now asks the device which of its streams
    share a queue and keeps four that do not (engine_core.cpp: probe_queue_classes); pe_profile_queue_classes asks again."""
    import torch

    torch.zeros(8, device="cuda").add_(1)             # work on torch's null stream
    side = [torch.cuda.Stream() for _ in range(3)]    # ... and three more streams of the host's own
    for s in side:
        with torch.cuda.stream(s):
            torch.zeros(8, device="cuda").add_(1)
    torch.cuda.synchronize()
    first = pea.Engine()
    second = pea.Engine()                             # beside another handle AND the host's streams
    for e in (first, second):
        assert e.profile_queue_classes() == [0, 1, 2, 3], "two of the handle's hot streams share a hardware queue"
    first.close()
    second.close()
    del side


def test_every_profiled_accumulation_reports_the_clock_it_ran_at(engine_factory):
    """pe_profile_accumulate_mhz: workgroup 0 of each k_g1_accumulate launch made while profiling is on counts shader cycles
    against the fixed 100 MHz counter; one reading per launch since profile_reset, in launch order, in the range of the
    device's clock (power management moves it between ~2.0 and ~2.45 GHz: profiles/r06_clockramp.txt)."""
    w = _world(engine_factory, n_val=1 << 14, n_comm=32, seed=11)
    e = w["e"]
    assert e.profile_accumulate_mhz().size == 0
    e.profile_enable(True)
    e.profile_reset()
    for _ in range(3):
        _step_streaming(w)
    e.drain()
    mhz = e.profile_accumulate_mhz()
    e.profile_enable(False)
    assert mhz.size == 3, mhz
    assert ((mhz > 800.0) & (mhz < 3200.0)).all(), mhz
    e.profile_reset()
    assert e.profile_accumulate_mhz().size == 0
