"""-m gpu: attestation rows resident in device memory (PE_ROWS_RESIDENT, include/posevo.h): grouping, committee
resolution, validate_on_attestation (A.4) and the asserts of process_attestation (pe:724-730) on the device give exactly
the outputs, statuses and store state of the host-row path -- which the other -m gpu tests hold against the oracle."""
import hashlib

import numpy as np
import pytest

import pos_evolution_amd as pea
import pos_evolution_amd.synth as synth
from pos_evolution_amd._abi import PE_ATT_FLAG_OVERLAPPING_BITS, pe_state_ctx
from tests import helpers as H
from tests.test_gpu_pipeline import _world

pytestmark = pytest.mark.gpu
NONE32 = 0xFFFFFFFF
RR, RES = pea.ROWS_RESIDENT, pea.RESIDENT


def _dev_rows(atts):
    import torch

    t = torch.from_numpy(np.ascontiguousarray(atts).view(np.uint8).reshape(-1).copy()).cuda()
    return pea.DeviceRows(t.data_ptr(), len(atts), keep=t)


def _dev_arena(arena):
    import torch

    t = torch.from_numpy(arena.copy()).cuda()
    return pea.DeviceArena(t.data_ptr(), t.numel(), keep=t)


def _host_step(e, atts, arena, ctx, want_pk=True):
    """The host-row path in the form the device path mirrors: rows + OR-ed bits handed on with PE_BITS_RESIDENT."""
    with e.pipeline():
        agg = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=want_pk)
        status, _, count = e.on_attestation_batch(packed=(agg["atts"], RES))
        head = e.get_head()
        pst, num = e.process_attestation_batch(ctx, packed=(agg["atts"], RES))
    return agg, status, count, pst, num, head


def _resident_step(e, atts, arena, ctx, mode="sync", want_pk=True, cap=None, dev_arena=False):
    rows = _dev_rows(atts)
    bits = _dev_arena(arena) if dev_arena else arena
    cap = len(atts) if cap is None else cap

    def calls():
        agg = e.aggregate(packed=(rows, bits), want_aggregate_pubkeys=want_pk)
        status, _, count = e.on_attestation_batch(packed=(RR, RES), cap=cap)
        head = e.get_head()
        pst, num = e.process_attestation_batch(ctx, packed=(RR, RES), cap=cap)
        return agg, status, count, pst, num, head

    if mode == "sync":
        return calls()
    with e.pipeline(lagged=(mode == "lagged")):
        out = calls()
    if mode == "lagged":
        e.drain()
    return out


def _assert_same(host, res, want_pk=True):
    agg_h, st_h, cnt_h, pst_h, num_h, head_h = host
    agg_r, st_r, cnt_r, pst_r, num_r, head_r = res
    g = agg_h["n_groups"]
    assert agg_r["n_groups"] == g
    assert np.array_equal(agg_r["atts"], agg_h["atts"]), "output rows (data, bits_offset, flags)"
    assert np.array_equal(agg_r["group_of"], agg_h["group_of"])
    assert np.array_equal(agg_r["out_arena"], agg_h["out_arena"])
    assert np.array_equal(agg_r["count"], agg_h["count"])
    if want_pk:
        assert np.array_equal(agg_r["aggpk96"], agg_h["aggpk96"])
    assert np.array_equal(st_r[:g], st_h) and (st_r[g:] == 0).all()
    assert np.array_equal(cnt_r[:g], cnt_h) and (cnt_r[g:] == 0).all()
    assert np.array_equal(pst_r[:g], pst_h) and (pst_r[g:] == 0).all()
    assert np.array_equal(num_r[:g], num_h) and (num_r[g:] == 0).all()
    assert head_r == head_h


def _assert_same_state(ea, eb):
    la, lb = ea.latest_messages(), eb.latest_messages()
    assert np.array_equal(la[0], lb[0]) and np.array_equal(la[1], lb[1]), "latest messages"
    assert np.array_equal(ea.participation_get(0), eb.participation_get(0))
    assert np.array_equal(ea.participation_get(1), eb.participation_get(1))
    assert np.array_equal(ea.get_weights(), eb.get_weights())


@pytest.mark.parametrize("mode", ["sync", "pipelined", "lagged"])
@pytest.mark.parametrize("n_val,n_comm,parts,density", [(20000, 64, 3, 0.9), (70000, 2048, 4, 0.99), (3000, 32, 1, 0.5),
                                                        (65536 + 77, 64, 2, 0.8)])
def test_resident_rows_step_equals_host_rows_step(engine_factory, n_val, n_comm, parts, density, mode):
    wa = _world(engine_factory, n_val, n_comm, seed=11, density=density, parts=parts)
    wb = _world(engine_factory, n_val, n_comm, seed=11, density=density, parts=parts)
    host = _host_step(wa["e"], wa["atts"], wa["arena"], wa["ctx"])
    res = _resident_step(wb["e"], wb["atts"], wb["arena"], wb["ctx"], mode=mode, dev_arena=(mode != "sync"))
    _assert_same(host, res)
    assert (host[1] == 0).all() and (host[3] == 0).all() and host[0]["n_groups"] == n_comm
    _assert_same_state(wa["e"], wb["e"])
    # and against the closed form of the synthetic registry: the sums themselves, not only their agreement
    a, b = wb["ab"]
    comm, agg = wb["comm"], res[0]
    for g in range(0, agg["n_groups"], max(1, agg["n_groups"] // 16)):
        r = agg["atts"][g]
        c = int((r["slot"] % 32) * (n_comm // 32) + r["index"])
        mem = comm.members[comm.offsets[c]:comm.offsets[c + 1]]
        assert agg["aggpk96"][g].tobytes() == H.closed_form_sum(mem[agg["bits"][g]], a, b)


def test_rows_in_shuffled_order_keep_first_appearance_order(engine_factory):
    """Groups are numbered by first appearance in the input -- with the input shuffled, members of a group far apart."""
    wa = _world(engine_factory, 30000, 128, seed=5, parts=4)
    wb = _world(engine_factory, 30000, 128, seed=5, parts=4)
    perm = np.random.Generator(np.random.PCG64(99)).permutation(len(wa["atts"]))
    atts = np.ascontiguousarray(wa["atts"][perm])
    host = _host_step(wa["e"], atts, wa["arena"], wa["ctx"])
    res = _resident_step(wb["e"], atts, wb["arena"], wb["ctx"], mode="pipelined")
    _assert_same(host, res)
    _assert_same_state(wa["e"], wb["e"])


@pytest.mark.parametrize("n_val,n_comm,parts,shuffle", [(200000, 2048, 40, True),     # 81 920 rows = 320 workgroups: five look-back windows
                                                         (150000, 36864, 1, False),    # more groups than round 5's plan could number (32 768)
                                                         (9000, 64, 37, True)])        # 2368 rows: a ragged last workgroup
def test_the_plan_over_many_workgroups_equals_the_host_grouping(engine_factory, n_val, n_comm, parts, shuffle):
    """k_att_plan runs one lane per row over ceil(n / 256) workgroups that hand their running sums to each other through
    look-back records (att_kernels.hip): group ids, union offsets, member lists, committee row lists and the G1 plan of a
    batch that spans many workgroups -- rows of a group far apart -- must be what the host path derives; twice on the same
    engine (the records have to come back clean)."""
    wa = _world(engine_factory, n_val, n_comm, seed=21, density=0.7, parts=parts)
    wb = _world(engine_factory, n_val, n_comm, seed=21, density=0.7, parts=parts)
    atts = wa["atts"]
    if shuffle:
        atts = np.ascontiguousarray(atts[np.random.Generator(np.random.PCG64(7)).permutation(len(atts))])
    host = _host_step(wa["e"], atts, wa["arena"], wa["ctx"])
    res = _resident_step(wb["e"], atts, wb["arena"], wb["ctx"], mode="pipelined", dev_arena=True)
    _assert_same(host, res)
    assert host[0]["n_groups"] == n_comm
    assert set(np.unique(host[1]).tolist()) <= {0, 11}, "only empty unions (committees of a few members) may be refused"
    _assert_same_state(wa["e"], wb["e"])
    again = _resident_step(wb["e"], atts, wb["arena"], wb["ctx"], mode="sync", want_pk=False)
    assert again[0]["n_groups"] == n_comm and np.array_equal(again[0]["atts"], host[0]["atts"])
    assert np.array_equal(again[0]["out_arena"], host[0]["out_arena"]) and np.array_equal(again[0]["group_of"], host[0]["group_of"])


def _violations(w, e_list):
    """One row per assert of validate_on_attestation (A.4) / process_attestation (pe:724-730) / A.7, each on a committee
    of its own, plus untouched rows.  Needs a block inside the attested epoch and a store clock 10 slots into the next
    epoch (applied to every engine of e_list)."""
    spe, epoch, tree, comm = 32, w["epoch"], w["tree"], w["comm"]
    tip = tree.roots[tree.roots.shape[0] - 1].tobytes()
    late_root = hashlib.sha256(b"late block").digest()
    for e in e_list:
        e.add_block(late_root, tip, epoch * spe + 20)
        e.on_tick(((epoch + 1) * spe + 10) * 12)
    base, arena = w["atts"], w["arena"]           # parts = 1: row c is committee c's attestation
    n_comm = comm.offsets.size - 1
    cps = n_comm // spe
    rows, expect = [], []
    extra = bytearray()

    def take(c):
        return base[c:c + 1].copy()

    def put(r, code_fc, code_st):
        rows.append(r)
        expect.append((code_fc, code_st))

    cc = 5 * cps                                  # a committee of slot epoch * 32 + 5, used below
    c = iter(x for x in range(3, n_comm) if x != cc)
    old = epoch - 2 if epoch >= 2 else epoch + 5  # neither the current (epoch + 1) nor the previous (epoch) epoch
    put(take(0), 0, 0)
    r = take(next(c)); r["target_epoch"] = old; r["slot"] = old * spe + 1; put(r, 1, 1)
    r = take(next(c)); r["slot"] = (epoch + 1) * spe + 1; put(r, 2, 2)                       # target epoch != epoch(slot)
    r = take(next(c)); r["target_root"] = np.frombuffer(hashlib.sha256(b"x").digest(), np.uint8); put(r, 3, None)
    r = take(next(c)); r["beacon_block_root"] = np.frombuffer(hashlib.sha256(b"y").digest(), np.uint8); put(r, 4, None)
    # vote for the block of slot epoch*32 + 20 from an attestation of slot epoch*32 + 5
    r = take(cc); r["beacon_block_root"] = np.frombuffer(late_root, np.uint8); put(r, 5, None)
    # FFG target that is not the LMD vote's ancestor at the epoch start: a block off the vote's chain
    r = take(next(c))
    blk = next(i for i in range(tree.roots.shape[0]) if tree.roots[i].tobytes() == r["beacon_block_root"][0].tobytes())
    anc = set()
    b = blk
    while True:
        anc.add(b)
        if int(tree.parent[b]) == NONE32 or b == 0:
            break
        b = int(tree.parent[b])
    off_chain = next(i for i in range(tree.roots.shape[0] - 1, 0, -1) if i not in anc)
    r["target_root"] = tree.roots[off_chain]; put(r, 6, None)
    # an attestation of the current slot (not in the past yet); for the state it is outside the inclusion window
    r = take(next(c)); r["slot"] = (epoch + 1) * spe + 10; r["target_epoch"] = epoch + 1
    r["beacon_block_root"] = np.frombuffer(late_root, np.uint8); r["target_root"] = np.frombuffer(late_root, np.uint8)
    put(r, 7, 13)
    # a valid attestation of the CURRENT epoch, for which no committee table is loaded
    r = take(next(c)); r["slot"] = (epoch + 1) * spe + 3; r["target_epoch"] = epoch + 1
    r["beacon_block_root"] = np.frombuffer(late_root, np.uint8); r["target_root"] = np.frombuffer(late_root, np.uint8)
    put(r, 8, None)
    # data.index >= committees per slot: process_attestation asserts it (pe:727); on_attestation's get_beacon_committee
    # (A.6) only needs the flat committee id (slot % 32) * cps + index to exist, so an early slot's row lands on a later
    # slot's committee there -- and a flat id beyond the table is refused by both
    r = take(2); r["index"] = cps + 3; put(r, 0, 9)
    r = take(next(c)); r["index"] = n_comm; put(r, 9, 9)
    # len(aggregation_bits) != len(committee): bits of a longer list appended to the arena
    r = take(next(c))
    nb = int(r["n_bits"][0]) + 8
    r["bits_offset"] = len(arena) + len(extra); r["n_bits"] = nb
    extra += bytes([0xFF] * ((nb + 7) // 8))
    put(r, 10, 10)
    # no bit set
    r = take(next(c))
    r["bits_offset"] = len(arena) + len(extra)
    extra += bytes((int(r["n_bits"][0]) + 7) // 8)
    put(r, 11, 11)
    r = take(next(c)); r["flags"] = 0; put(r, 12, 12)                                        # signature verdict false
    # two members of one group sharing bits (A.8): the same row twice
    dup = take(next(c)); put(dup, 12, 12); rows.append(dup.copy())
    # source checkpoint that is not the state's justified one
    r = take(next(c)); r["source_epoch"] = 7; put(r, 0, 14)
    for _ in range(6):
        put(take(next(c)), 0, 0)
    atts = np.ascontiguousarray(np.concatenate(rows))
    arena2 = np.concatenate([arena, np.frombuffer(bytes(extra), dtype=np.uint8)])
    return atts, arena2, expect


def test_statuses_with_every_assert_violated_once(engine_factory):
    wa = _world(engine_factory, 40000, 64, seed=21, density=0.9, parts=1)
    wb = _world(engine_factory, 40000, 64, seed=21, density=0.9, parts=1)
    atts, arena, expect = _violations(wa, [wa["e"], wb["e"]])
    ctx = wa["ctx"]   # state.slot = (epoch + 1) * 32
    host = _host_step(wa["e"], atts, arena, ctx, want_pk=False)
    res = _resident_step(wb["e"], atts, arena, ctx, mode="pipelined", want_pk=False)
    _assert_same(host, res, want_pk=False)
    _assert_same_state(wa["e"], wb["e"])
    g = host[0]["n_groups"]
    assert g == len(expect)
    for k, (fc, st) in enumerate(expect):
        assert res[1][k] == fc, (k, "on_attestation", res[1][k], fc)
        if st is not None:
            assert res[3][k] == st, (k, "process_attestation", res[3][k], st)
    assert set(int(x) for x in res[1][:g]) == set(range(13)), "every pe_att_status of the fork-choice side occurs"
    assert {1, 2, 9, 10, 11, 12, 13, 14} <= set(int(x) for x in res[3][:g])
    assert bool(res[0]["atts"]["flags"][[k for k, e in enumerate(expect) if e == (12, 12)][1]] & PE_ATT_FLAG_OVERLAPPING_BITS)


def test_two_target_epochs_and_several_votes_per_committee(engine_factory):
    """Rows of the previous and of the current epoch in one batch (both candidate tables), and two different votes per
    committee: the batch-order rule of update_latest_messages (pe:1435-1441: first in batch order among equal target
    epochs) and the order of the flag loop (pe:745-749: the first attestation to set a flag earns it)."""
    n_val, n_comm, spe = 50000, 128, 32
    worlds = []
    for _ in range(2):
        w = _world(engine_factory, n_val, n_comm, seed=33, density=0.7, parts=2)
        comm2 = synth.random_committees(n_val, n_comm, 77)
        w["e"].set_committees(w["epoch"] + 1, comm2.offsets, comm2.members)
        w["e"].on_tick(((w["epoch"] + 1) * spe + 20) * 12)
        w["comm2"] = comm2
        worlds.append(w)
    wa, wb = worlds
    tree, epoch = wa["tree"], wa["epoch"]
    src = (0, tree.roots[0].tobytes())
    a1, ar1, _ = synth.epoch_attestations(wa["comm"], tree, epoch, spe, seed=1, density=0.7, parts=2, source=src)
    a2, ar2, _ = synth.epoch_attestations(wa["comm"], tree, epoch, spe, seed=2, density=0.6, parts=1, source=src, vote_seed=9)
    a3, ar3, _ = synth.epoch_attestations(wa["comm2"], tree, epoch + 1, spe, seed=3, density=0.8, parts=2, source=src)
    a3 = a3[a3["slot"] < (epoch + 1) * spe + 20]
    a2 = a2.copy(); a2["bits_offset"] += len(ar1)
    a3 = a3.copy(); a3["bits_offset"] += len(ar1) + len(ar2)
    rng = np.random.Generator(np.random.PCG64(4))
    atts = np.concatenate([a1, a2, a3])
    atts = np.ascontiguousarray(atts[rng.permutation(len(atts))])
    arena = np.concatenate([ar1, ar2, ar3])
    ctx = pe_state_ctx()
    ctx.slot = (epoch + 1) * spe + 20
    ctx.chain_tip_root[:] = tree.roots[tree.roots.shape[0] - 1].tobytes()
    ctx.current_justified_root[:] = tree.roots[0].tobytes()
    ctx.previous_justified_root[:] = tree.roots[0].tobytes()
    ctx.base_reward_per_increment = 555
    host = _host_step(wa["e"], atts, arena, ctx)
    res = _resident_step(wb["e"], atts, arena, ctx, mode="pipelined", dev_arena=True)
    _assert_same(host, res)
    _assert_same_state(wa["e"], wb["e"])
    st, pst = res[1][:host[0]["n_groups"]], res[3][:host[0]["n_groups"]]
    assert (st == 0).sum() > n_comm and (pst == 0).sum() > n_comm // 2   # both epochs' rows were applied
    assert len({int(x) for x in host[0]["atts"]["target_epoch"]}) == 2


def test_errors_are_deferred_and_form_no_groups(engine_factory):
    w = _world(engine_factory, 20000, 64, seed=41, parts=2)
    e, atts, arena, ctx = w["e"], w["atts"], w["arena"], w["ctx"]
    n = len(atts)
    before = e.latest_messages()[1].copy()
    # handlers without a resident aggregate
    with pytest.raises(pea.EngineError) as err:
        e.on_attestation_batch(packed=(RR, RES), cap=n)
    assert err.value.status == pea._abi.PE_ERR_STATE
    # a row whose bits lie outside the arena: the aggregate fails where its outputs complete, the handlers apply nothing
    bad = atts.copy()
    bad["bits_offset"][7] = len(arena)
    reached, st, cnt = [], None, None
    with pytest.raises(pea.EngineError) as err:
        with e.pipeline():
            agg = e.aggregate(packed=(_dev_rows(bad), arena), want_aggregate_pubkeys=True)
            reached.append("aggregate")   # the aggregate itself only enqueues: the refusal is found on the device
            st, _, cnt = e.on_attestation_batch(packed=(RR, RES), cap=n)
            e.get_head()
    # raised where the aggregate's outputs complete: at the pipeline's end, or earlier if a later call of the pipeline has
    # to wait for what is enqueued (a block that grows) -- never inside the aggregate, and nothing is applied either way
    assert reached == ["aggregate"], (reached, str(err.value))
    assert err.value.status == -1   # PE_ERR_INVALID_ARG
    assert np.array_equal(e.latest_messages()[1], before)
    assert st is None or ((st == 0).all() and (cnt == 0).all())
    # a length that wraps 32-bit byte arithmetic ((n_bits + 7) / 8 == 0) is refused like any other row outside the arena,
    # rows in host memory or in HBM
    for n_bits in (0xFFFFFFFF, 0xFFFFFFF9, 0x80000000):
        wrap = atts.copy()
        wrap["n_bits"][5] = n_bits
        for rows in (wrap, _dev_rows(wrap)):
            with pytest.raises(pea.EngineError) as err:
                e.aggregate(packed=(rows, arena), want_aggregate_pubkeys=True)
            assert err.value.status == -1
    assert np.array_equal(e.latest_messages()[1], before)
    # aggregate pubkeys asked for a target epoch without a table
    other = atts.copy()
    other["target_epoch"][3] += 1
    with pytest.raises(pea.EngineError) as err:
        e.aggregate(packed=(_dev_rows(other), arena), want_aggregate_pubkeys=True)
    assert err.value.status == -11   # PE_ERR_NO_COMMITTEES
    # status arrays shorter than the groups formed
    agg = e.aggregate(packed=(_dev_rows(atts), arena), want_aggregate_pubkeys=True)
    assert agg["n_groups"] == 64
    with pytest.raises(pea.EngineError) as err:
        e.on_attestation_batch(packed=(RR, RES), cap=10)
    assert err.value.status == pea._abi.PE_ERR_CAPACITY
    assert np.array_equal(e.latest_messages()[1], before)
    # the same for process_attestation (the flag kernel used to run anyway and write numerators[g] for every
    # g < groups into a block sized by cap -- over the error word, so the call returned PE_OK with all-zero statuses):
    # synchronous, and as the middle call of a pipeline whose neighbours' outputs lie behind its block
    part_before = (e.participation_get(0).copy(), e.participation_get(1).copy())
    with pytest.raises(pea.EngineError) as err:
        e.process_attestation_batch(ctx, packed=(RR, RES), cap=10)
    assert err.value.status == pea._abi.PE_ERR_CAPACITY
    with pytest.raises(pea.EngineError) as err:
        with e.pipeline():
            e.process_attestation_batch(ctx, packed=(RR, RES), cap=10)
            e.get_head()
    assert err.value.status == pea._abi.PE_ERR_CAPACITY
    assert np.array_equal(e.participation_get(0), part_before[0])
    assert np.array_equal(e.participation_get(1), part_before[1])
    # the clock crossing an epoch between the aggregate and its handler
    e.on_tick((w["epoch"] + 2) * 32 * 12)
    with pytest.raises(pea.EngineError) as err:
        e.on_attestation_batch(packed=(RR, RES), cap=n)
    assert err.value.status == pea._abi.PE_ERR_STATE
    # and the path still works afterwards
    e2w = _world(engine_factory, 20000, 64, seed=41, parts=2)
    host = _host_step(e2w["e"], atts, arena, ctx)
    wb = _world(engine_factory, 20000, 64, seed=41, parts=2)
    # exactly enough entries
    res = _resident_step(wb["e"], atts, arena, ctx, mode="sync", cap=64)
    _assert_same(host, res)


def test_streaming_epochs_resident_rows_equal_synchronous_host_rows(engine_factory):
    """Six epochs through streaming pipelines with rows + bits resident in HBM (the bench's call pattern) against a twin
    engine driven with synchronous host-row calls: outputs of every step and the final store state."""
    n_val, n_comm, spe, steps = 60000, 256, 32, 6
    ea, eb = engine_factory(max_committee_tables=steps + 1), engine_factory(max_committee_tables=steps + 1)
    tree = synth.random_tree(128, 4, "bushy")
    pts, _ = H.oracle_points(n_val)
    bal = synth.balances(n_val, 4, mixed=True)
    flags = synth.validator_flags(n_val, 4, inactive_frac=0.01)
    for e in (ea, eb):
        H.load_tree(e, tree)
        e.set_validators(bal, flags, pts)
    ep0 = int(tree.slot.max()) // spe + 1
    work = []
    for s in range(steps):
        ep = ep0 + s
        seed = hashlib.sha256(b"rr" + ep.to_bytes(8, "little")).digest()
        for e in (ea, eb):
            off, mem = e.compute_committees(ep, seed, n_val, n_comm, 10)
        comm = synth.Committees(off, mem)
        atts, arena, _ = synth.epoch_attestations(comm, tree, ep, spe, seed=4, density=0.95, parts=3,
                                                  source=(0, tree.roots[0].tobytes()), vote_recent=16)
        ctx = pe_state_ctx()
        ctx.slot = (ep + 1) * spe
        ctx.chain_tip_root[:] = tree.roots[tree.roots.shape[0] - 1].tobytes()
        ctx.current_justified_root[:] = tree.roots[0].tobytes()
        ctx.previous_justified_root[:] = tree.roots[0].tobytes()
        ctx.base_reward_per_increment = 999
        work.append(dict(ep=ep, atts=atts, arena=arena, ctx=ctx, rows=_dev_rows(atts), bits=_dev_arena(arena)))
    ref = []
    for wk in work:
        ea.on_tick((wk["ep"] + 1) * spe * 12)
        ea.participation_rotate()
        agg = ea.aggregate(packed=(wk["atts"], wk["arena"]), want_aggregate_pubkeys=True)
        st, _, cnt = ea.on_attestation_batch(packed=(agg["atts"], agg["out_arena"]))
        head = ea.get_head()
        pst, num = ea.process_attestation_batch(wk["ctx"], packed=(agg["atts"], agg["out_arena"]))
        ref.append((agg, st, cnt, pst, num, head))
    got = []
    for wk in work:
        eb.on_tick((wk["ep"] + 1) * spe * 12)
        eb.participation_rotate()
        with eb.pipeline(lagged=True):
            agg = eb.aggregate(packed=(wk["rows"], wk["bits"]), want_aggregate_pubkeys=True)
            st, _, cnt = eb.on_attestation_batch(packed=(RR, RES), cap=n_comm)
            head = eb.get_head()
            pst, num = eb.process_attestation_batch(wk["ctx"], packed=(RR, RES), cap=n_comm)
        got.append((agg, st, cnt, pst, num, head))
    eb.drain()
    for r, g in zip(ref, got):
        _assert_same(r, g)
        assert (r[1] == 0).all() and (r[3] == 0).all()
    _assert_same_state(ea, eb)
