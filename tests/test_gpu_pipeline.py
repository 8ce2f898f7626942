"""-m gpu: pipelined calls (pe_pipeline_begin/_end), the device-resident hand-over of pe_aggregate's outputs
(PE_BITS_RESIDENT) and the overlap verdict (validator guide A.8) give exactly the results of the synchronous,
host-buffer calls -- and of the oracle."""
import numpy as np
import pytest

import pos_evolution_amd as pea
import pos_evolution_amd.synth as synth
from oracle import cport
from pos_evolution_amd._abi import PE_ATT_FLAG_OVERLAPPING_BITS, pe_state_ctx
from tests import helpers as H

pytestmark = pytest.mark.gpu
NONE32 = 0xFFFFFFFF


def _world(engine_factory, n_val, n_comm, seed, density=0.9, parts=3):
    e = engine_factory()
    spe = 32
    tree = synth.random_tree(96, seed, "branchy")
    H.load_tree(e, tree)
    pts, ab = H.oracle_points(n_val)
    bal = synth.balances(n_val, seed, mixed=True)
    flags = synth.validator_flags(n_val, seed, inactive_frac=0.01)
    e.set_validators(bal, flags, pts)
    epoch = int(tree.slot.max()) // spe + 1
    comm = synth.random_committees(n_val, n_comm, seed)
    e.set_committees(epoch, comm.offsets, comm.members)
    e.on_tick((epoch + 1) * spe * 12)
    atts, arena, bit_rows = synth.epoch_attestations(comm, tree, epoch, spe, seed=seed, density=density, parts=parts,
                                                     source=(0, tree.roots[0].tobytes()))
    ctx = pe_state_ctx()
    ctx.slot = (epoch + 1) * spe
    ctx.chain_tip_root[:] = tree.roots[tree.roots.shape[0] - 1].tobytes()
    ctx.current_justified_root[:] = tree.roots[0].tobytes()
    ctx.previous_justified_root[:] = tree.roots[0].tobytes()
    ctx.base_reward_per_increment = 777
    return dict(e=e, tree=tree, comm=comm, atts=atts, arena=arena, bit_rows=bit_rows, ctx=ctx, ab=ab, epoch=epoch,
                bal=bal, flags=flags, n_val=n_val)


def _step_sync(w):
    e = w["e"]
    agg = e.aggregate(packed=(w["atts"], w["arena"]), want_aggregate_pubkeys=True)
    status, _, count = e.on_attestation_batch(packed=(agg["atts"], agg["out_arena"]))
    pst, num = e.process_attestation_batch(w["ctx"], packed=(agg["atts"], agg["out_arena"]))
    head = e.get_head()
    return agg, status, count, pst, num, head


def _step_pipelined(w):
    e = w["e"]
    with e.pipeline():
        agg = e.aggregate(packed=(w["atts"], w["arena"]), want_aggregate_pubkeys=True)
        status, _, count = e.on_attestation_batch(packed=(agg["atts"], pea.RESIDENT))
        pst, num = e.process_attestation_batch(w["ctx"], packed=(agg["atts"], pea.RESIDENT))
        head = e.get_head()
    return agg, status, count, pst, num, head


def _same_results(a, b):
    agg_a, st_a, cnt_a, pst_a, num_a, head_a = a
    agg_b, st_b, cnt_b, pst_b, num_b, head_b = b
    assert agg_a["n_groups"] == agg_b["n_groups"]
    assert np.array_equal(agg_a["atts"], agg_b["atts"])
    assert np.array_equal(agg_a["out_arena"], agg_b["out_arena"])
    assert np.array_equal(agg_a["count"], agg_b["count"])
    assert np.array_equal(agg_a["aggpk96"], agg_b["aggpk96"])
    assert np.array_equal(agg_a["group_of"], agg_b["group_of"])
    assert np.array_equal(st_a, st_b) and np.array_equal(cnt_a, cnt_b)
    assert np.array_equal(pst_a, pst_b) and np.array_equal(num_a, num_b)
    assert head_a == head_b


@pytest.mark.parametrize("n_val,n_comm,parts,density", [(20000, 64, 3, 0.9), (70000, 2048, 4, 0.99), (3000, 32, 1, 0.5)])
def test_pipelined_resident_step_equals_synchronous_step(engine_factory, n_val, n_comm, parts, density):
    """The same epoch through two engines: synchronous calls over host buffers vs one pipeline with the aggregate's
    outputs handed over resident.  Every output and the whole device state must agree, and both must equal the
    oracle's union / closed-form G1 sums / LMD table / flags / head."""
    wa = _world(engine_factory, n_val, n_comm, seed=n_comm, density=density, parts=parts)
    wb = _world(engine_factory, n_val, n_comm, seed=n_comm, density=density, parts=parts)
    ra = _step_sync(wa)
    rb = _step_pipelined(wb)
    _same_results(ra, rb)
    ea, eb = wa["e"], wb["e"]
    for x, y in zip(ea.latest_messages(), eb.latest_messages()):
        assert np.array_equal(x, y)
    assert np.array_equal(ea.participation_get(0), eb.participation_get(0))
    assert np.array_equal(ea.participation_get(1), eb.participation_get(1))
    assert np.array_equal(ea.get_weights(), eb.get_weights())
    # ... and the oracle
    agg, status, count, pst, num, head = rb
    comm, tree, spe = wb["comm"], wb["tree"], 32
    cps = n_comm // spe
    a, b = wb["ab"]
    assert (status == 0).all() and (pst == 0).all()
    rows = agg["atts"]
    pos = ((rows["slot"] % spe) * cps + rows["index"]).astype(np.int64)
    for g in range(0, agg["n_groups"], max(1, agg["n_groups"] // 64)):
        mem = comm.members[comm.offsets[pos[g]]:comm.offsets[pos[g] + 1]]
        assert agg["aggpk96"][g].tobytes() == H.closed_form_sum(mem[agg["bits"][g]], a, b)
        assert count[g] == agg["bits"][g].sum() == agg["count"][g]
    vote_epoch = np.zeros(n_val, dtype=np.uint64)
    vote_block = np.full(n_val, NONE32, dtype=np.uint32)
    blk = np.array([eb.block_index_of(r["beacon_block_root"].tobytes()) for r in rows], dtype=np.uint32)
    cport.update_latest_messages(comm.offsets[pos], rows["n_bits"], rows["bits_offset"], rows["target_epoch"], blk,
                                 agg["out_arena"], comm.members, wb["flags"], vote_epoch, vote_block)
    assert np.array_equal(eb.latest_messages()[1], vote_block)
    head_o, w_o = cport.get_head(tree.parent, np.ones(tree.parent.size, dtype=np.uint8), tree.roots, vote_block,
                                 wb["bal"], wb["flags"], 0)
    assert head == tree.roots[head_o].tobytes()
    assert np.array_equal(eb.get_weights(), w_o)


def test_two_steps_in_a_row_and_sync_call_inside_a_pipeline(engine_factory):
    """Pipelines back to back reuse the staging blocks; a synchronous entry point called inside a pipeline first
    completes what is enqueued (here: latest_messages() between on_attestation and get_head)."""
    wa = _world(engine_factory, 12000, 64, seed=5)
    wb = _world(engine_factory, 12000, 64, seed=5)
    for rep in range(3):
        ra = _step_sync(wa)
        e = wb["e"]
        with e.pipeline():
            agg = e.aggregate(packed=(wb["atts"], wb["arena"]), want_aggregate_pubkeys=True)
            status, _, count = e.on_attestation_batch(packed=(agg["atts"], pea.RESIDENT))
            mid = e.latest_messages()          # forces completion of the two calls above
            assert (status == 0).all() and agg["count"].sum() == count.sum()
            pst, num = e.process_attestation_batch(wb["ctx"], packed=(agg["atts"], pea.RESIDENT))
            head = e.get_head()
        _same_results(ra, (agg, status, count, pst, num, head))
        assert np.array_equal(mid[1], wa["e"].latest_messages()[1])
        for w in (wa, wb):   # next repetition: a later epoch would need new tables; rotate participation instead
            w["e"].participation_rotate()


def test_lagged_pipelines_equal_synchronous_steps(engine_factory):
    """pe_pipeline_end_lagged: step N's outputs are complete when step N+1's block exits (the last at drain()).  Four
    epochs in a row, each with its own committee table and attestations: every output and the final device state equal
    the synchronous run's."""
    import hashlib
    n_val, n_comm, spe = 30000, 128, 32
    worlds = [_world(engine_factory, n_val, n_comm, seed=21), _world(engine_factory, n_val, n_comm, seed=21)]
    steps = []
    for k in range(4):
        ep = worlds[0]["epoch"] + k
        seed = hashlib.sha256(b"lag%d" % k).digest()
        comm = None
        for w in worlds:
            off, mem = w["e"].compute_committees(ep, seed, np.arange(n_val, dtype=np.uint32), n_comm, 10)
            comm = synth.Committees(off, mem)
        atts, arena, _ = synth.epoch_attestations(comm, worlds[0]["tree"], ep, spe, seed=21 + k, density=0.8 + 0.05 * k,
                                                  parts=2 + k % 2, source=(0, worlds[0]["tree"].roots[0].tobytes()))
        ctx = pe_state_ctx()
        ctx.slot = (ep + 1) * spe
        ctx.chain_tip_root[:] = worlds[0]["tree"].roots[-1].tobytes()
        ctx.current_justified_root[:] = worlds[0]["tree"].roots[0].tobytes()
        ctx.previous_justified_root[:] = worlds[0]["tree"].roots[0].tobytes()
        ctx.base_reward_per_increment = 500 + k
        steps.append(dict(epoch=ep, atts=atts, arena=arena, ctx=ctx))
    out = [[], []]
    for which, w in enumerate(worlds):
        e = w["e"]
        for st in steps:
            e.on_tick((st["epoch"] + 1) * spe * 12)
            e.participation_rotate()
            wk = dict(e=e, atts=st["atts"], arena=st["arena"], ctx=st["ctx"])
            if which == 0:
                out[0].append(_step_sync(wk))
            else:
                with e.pipeline(lagged=True):
                    agg = e.aggregate(packed=(st["atts"], st["arena"]), want_aggregate_pubkeys=True)
                    status, _, count = e.on_attestation_batch(packed=(agg["atts"], pea.RESIDENT))
                    pst, num = e.process_attestation_batch(st["ctx"], packed=(agg["atts"], pea.RESIDENT))
                    head = e.get_head()
                out[1].append((agg, status, count, pst, num, head))
        e.drain()
    for a, b in zip(*out):
        _same_results(a, b)
        assert (b[1] == 0).all() and (b[3] == 0).all()
    ea, eb = worlds[0]["e"], worlds[1]["e"]
    for x, y in zip(ea.latest_messages(), eb.latest_messages()):
        assert np.array_equal(x, y)
    assert np.array_equal(ea.participation_get(0), eb.participation_get(0))
    assert np.array_equal(ea.participation_get(1), eb.participation_get(1))
    assert np.array_equal(ea.get_weights(), eb.get_weights())


def test_overlapping_members_are_flagged_and_rejected(engine_factory):
    """ADVICE r1 / validator guide A.8: members of a group that share a bit (a gossip duplicate is enough) make the
    summed signature count that validator twice.  pe_aggregate says so (PE_ATT_FLAG_OVERLAPPING_BITS, signature
    verdict cleared) and the fork-choice / state handlers reject the row -- on the host path and on the resident
    path -- leaving the store untouched (pe:1041)."""
    w = _world(engine_factory, 5000, 32, seed=3, parts=2)
    e, atts, arena = w["e"], w["atts"], w["arena"]
    # group 0: duplicate its first member; group 1: untouched; group 2: make its two parts overlap in one bit
    g_of = e.aggregate(packed=(atts, arena))["group_of"]
    m0 = np.nonzero(g_of == 0)[0]
    m2 = np.nonzero(g_of == 2)[0]
    arena2 = arena.copy()
    b_first = int(atts[m2[0]]["bits_offset"]), int(atts[m2[1]]["bits_offset"])
    nb = int(atts[m2[0]]["n_bits"])
    bits_a = np.unpackbits(arena2[b_first[0]:b_first[0] + (nb + 7) // 8], bitorder="little")[:nb]
    share = int(np.nonzero(bits_a)[0][0])
    arena2[b_first[1] + share // 8] |= np.uint8(1 << (share % 8))
    atts2 = np.concatenate([atts, atts[m0[:1]]])
    res = e.aggregate(packed=(atts2, arena2), want_aggregate_pubkeys=True)
    g_dup, g_ovl = int(res["group_of"][m0[0]]), int(res["group_of"][m2[0]])
    flags = res["atts"]["flags"]
    for g in range(res["n_groups"]):
        bad = g in (g_dup, g_ovl)
        assert bool(flags[g] & PE_ATT_FLAG_OVERLAPPING_BITS) == bad
        assert bool(flags[g] & 1) == (not bad)
    # the bits and the aggregate pubkey are still those of the union
    a, b = w["ab"]
    comm, spe = w["comm"], 32
    r = res["atts"][g_ovl]
    c = int((r["slot"] % spe) * (32 // spe) + r["index"])
    mem = comm.members[comm.offsets[c]:comm.offsets[c + 1]]
    assert res["aggpk96"][g_ovl].tobytes() == H.closed_form_sum(mem[res["bits"][g_ovl]], a, b)
    # handlers: the host-buffer path rejects through the cleared verdict, the resident path through the device gate
    for resident in (False, True):
        e2w = _world(engine_factory, 5000, 32, seed=3, parts=2)
        e2 = e2w["e"]
        if resident:
            with e2.pipeline():
                res2 = e2.aggregate(packed=(atts2, arena2))
                assert (res2["atts"]["flags"] & 1).all()   # inside a pipeline the verdict is not on the rows yet
                status, _, count = e2.on_attestation_batch(packed=(res2["atts"], pea.RESIDENT))
                pst, num = e2.process_attestation_batch(e2w["ctx"], packed=(res2["atts"], pea.RESIDENT))
            assert bool(res2["atts"]["flags"][g_ovl] & PE_ATT_FLAG_OVERLAPPING_BITS)
        else:
            res2 = e2.aggregate(packed=(atts2, arena2))
            status, _, count = e2.on_attestation_batch(packed=(res2["atts"], res2["out_arena"]))
            pst, num = e2.process_attestation_batch(e2w["ctx"], packed=(res2["atts"], res2["out_arena"]))
        for g in range(res2["n_groups"]):
            want = 12 if g in (g_dup, g_ovl) else 0   # PE_ATT_BAD_SIGNATURE
            assert status[g] == want and pst[g] == want, (resident, g, status[g], pst[g])
            if want:
                assert count[g] == 0 and num[g] == 0
        # validators of the rejected groups have no latest message and no participation flags (pe:1041)
        _, blk = e2.latest_messages()
        part = e2.participation_get(0) | e2.participation_get(1)
        for g in (g_dup, g_ovl):
            r = res2["atts"][g]
            c = int((r["slot"] % spe) * (32 // spe) + r["index"])
            mem = comm.members[comm.offsets[c]:comm.offsets[c + 1]]
            assert (blk[mem] == NONE32).all() and (part[mem] == 0).all()
        assert (blk != NONE32).sum() > 0


def test_resident_rows_subset_reordered_and_empty_union(engine_factory):
    w = _world(engine_factory, 8000, 64, seed=8, parts=2)
    e = w["e"]
    atts, arena = w["atts"].copy(), w["arena"].copy()
    # empty union: zero every bit of group 5's members
    g_of = e.aggregate(packed=(atts, arena))["group_of"]
    for i in np.nonzero(g_of == 5)[0]:
        o, nb = int(atts[i]["bits_offset"]), int(atts[i]["n_bits"])
        arena[o:o + (nb + 7) // 8] = 0
    ref = _world(engine_factory, 8000, 64, seed=8, parts=2)["e"]
    res_ref = ref.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    sel = np.arange(res_ref["n_groups"])[::-1][::2].copy()          # every other group, reversed
    st_ref, _, cnt_ref = ref.on_attestation_batch(packed=(res_ref["atts"][sel], res_ref["out_arena"]))
    with e.pipeline():
        res = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
        st, _, cnt = e.on_attestation_batch(packed=(res["atts"][sel], pea.RESIDENT))
    assert np.array_equal(st, st_ref) and np.array_equal(cnt, cnt_ref)
    g5 = int(g_of[np.nonzero(g_of == 5)[0][0]])
    if g5 in sel:
        assert st[list(sel).index(g5)] == 11      # PE_ATT_EMPTY_OR_INVALID_INDICES
    assert res["aggpk96"][g5][0] == 0x40           # infinity
    for x, y in zip(e.latest_messages(), ref.latest_messages()):
        assert np.array_equal(x, y)
    # a row that is not a row of the last aggregate is refused
    bogus = res["atts"][:1].copy()
    bogus["bits_offset"] += 1
    with pytest.raises(pea.EngineError):
        e.on_attestation_batch(packed=(bogus, pea.RESIDENT))


def test_unaligned_member_offsets_in_the_arena(engine_factory):
    """k_bits_union reads the members at their byte offsets in the caller's arena (no host re-packing): odd offsets,
    lengths that are not multiples of 8 or 32, a one-bit committee."""
    e = engine_factory()
    n_val = 4096
    e.set_validators(synth.balances(n_val, 2), np.ones(n_val, dtype=np.uint8))
    tree = synth.random_tree(40, 2, "chain")
    H.load_tree(e, tree)
    rng = np.random.default_rng(2)
    sizes = [1, 7, 8, 9, 31, 32, 33, 63, 65, 100, 127, 257]
    rows, want = [], []
    for s_i, size in enumerate(sizes):
        parts = [rng.random(size) < 0.4 for _ in range(3)]
        union = parts[0] | parts[1] | parts[2]
        want.append(union)
        for p in parts:
            rows.append(pea.AttRow(5, s_i, tree.roots[3].tobytes(), 0, tree.roots[0].tobytes(), 0,
                                   tree.roots[0].tobytes(), p))
    carr, arena = pea.pack_attestations(rows)
    arr = np.frombuffer(carr, dtype=synth.ATT_DTYPE, count=len(rows)).copy()
    # shift every member to an odd byte offset
    pad = np.zeros(arena.size * 2 + 64, dtype=np.uint8)
    cur = 1
    for i in range(len(arr)):
        nb = (int(arr[i]["n_bits"]) + 7) // 8
        o = int(arr[i]["bits_offset"])
        pad[cur:cur + nb] = arena[o:o + nb]
        # garbage after the member's last bit must be ignored
        tail = int(arr[i]["n_bits"]) % 8
        if tail:
            pad[cur + nb - 1] |= np.uint8((0xFF << tail) & 0xFF)
        arr[i]["bits_offset"] = cur
        cur += nb + (i % 3)
    res = e.aggregate(packed=(arr, pad))
    assert res["n_groups"] == len(sizes)
    for g, size in enumerate(sizes):
        assert np.array_equal(res["bits"][g], want[g]), size
        assert res["count"][g] == want[g].sum()


def test_arena_in_device_or_pinned_memory(engine_factory):
    """pe_aggregate with the members' bits lying in device memory, in pinned host memory, or at an odd offset inside a
    device allocation: same result as from pageable host memory, synchronous and inside a streaming pipeline."""
    import torch
    w = _world(engine_factory, 20000, 64, seed=77, density=0.7, parts=3)
    e = w["e"]
    ref = e.aggregate(packed=(w["atts"], w["arena"]), want_aggregate_pubkeys=True)
    dev = torch.from_numpy(w["arena"]).cuda()
    pin = torch.from_numpy(w["arena"]).pin_memory()
    shifted = torch.zeros(w["arena"].size + 64, dtype=torch.uint8, device="cuda")
    shifted[13:13 + w["arena"].size] = dev
    atts13 = w["atts"].copy()
    atts13["bits_offset"] += 13
    torch.cuda.synchronize()
    cases = [(w["atts"], pea.DeviceArena(dev.data_ptr(), dev.numel(), keep=dev)),
             (w["atts"], pea.DeviceArena(pin.data_ptr(), pin.numel(), keep=pin)),
             (atts13, pea.DeviceArena(shifted.data_ptr(), shifted.numel(), keep=shifted))]
    for atts, arena in cases:
        got = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
        for k in ("out_arena", "count", "aggpk96", "group_of"):
            assert np.array_equal(got[k], ref[k]), k
        with e.pipeline(lagged=True):
            got = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
            status, _, count = e.on_attestation_batch(packed=(got["atts"], pea.RESIDENT))
            e.get_head()
        e.drain()
        for k in ("out_arena", "count", "aggpk96", "group_of"):
            assert np.array_equal(got[k], ref[k]), k
        assert (status == 0).all() and np.array_equal(count, ref["count"])
    # the handlers re-pack host bits: device memory is refused, not dereferenced
    dev_arena = cases[0][1]
    for call in (lambda: e.on_attestation_batch(packed=(ref["atts"], dev_arena)),
                 lambda: e.process_attestation_batch(w["ctx"], packed=(ref["atts"], dev_arena)),
                 lambda: e.get_indexed_attestations(packed=(ref["atts"], dev_arena))):
        with pytest.raises(AssertionError) as ei:
            call()
        assert ei.value.status == -1


def test_aggregate_across_two_target_epochs(engine_factory):
    """A batch around an epoch boundary: attestations of two target epochs (two committee tables) in one pe_aggregate
    with aggregate pubkeys, rows interleaved.  Every group must equal what per-epoch calls give, synchronously and
    inside a streaming pipeline (where such a batch stays on the engine's stream)."""
    n_val, n_comm, spe = 20000, 64, 32
    w = _world(engine_factory, n_val, n_comm, seed=88, density=0.8, parts=2)
    e, tree, ep = w["e"], w["tree"], w["epoch"]
    comm2 = synth.random_committees(n_val, n_comm, 89)
    e.set_committees(ep + 1, comm2.offsets, comm2.members)
    e.on_tick((ep + 2) * spe * 12)
    atts2, arena2, _ = synth.epoch_attestations(comm2, tree, ep + 1, spe, seed=89, density=0.6, parts=3,
                                                source=(0, tree.roots[0].tobytes()))
    ref1 = e.aggregate(packed=(w["atts"], w["arena"]), want_aggregate_pubkeys=True)
    ref2 = e.aggregate(packed=(atts2, arena2), want_aggregate_pubkeys=True)
    both = np.concatenate([w["atts"], atts2])
    both["bits_offset"][len(w["atts"]):] += w["arena"].size
    arena = np.concatenate([w["arena"], arena2])
    order = np.random.default_rng(3).permutation(len(both))
    both = both[order].copy()

    def key(row):
        return (int(row["target_epoch"]), int(row["slot"]), int(row["index"]), row["beacon_block_root"].tobytes())

    want = {}
    for ref in (ref1, ref2):
        for g in range(ref["n_groups"]):
            want[key(ref["atts"][g])] = (ref["aggpk96"][g].tobytes(), int(ref["count"][g]), ref["bits"][g])

    def check(got):
        assert got["n_groups"] == ref1["n_groups"] + ref2["n_groups"] == len(want)
        for g in range(got["n_groups"]):
            pk, cnt, bits = want[key(got["atts"][g])]
            assert got["aggpk96"][g].tobytes() == pk and int(got["count"][g]) == cnt
            assert np.array_equal(got["bits"][g], bits)

    check(e.aggregate(packed=(both, arena), want_aggregate_pubkeys=True))
    with e.pipeline(lagged=True):
        got = e.aggregate(packed=(both, arena), want_aggregate_pubkeys=True)
        status, _, count = e.on_attestation_batch(packed=(got["atts"], pea.RESIDENT))
        head = e.get_head()
    e.drain()
    check(got)
    # the clock stands in epoch ep + 2: the rows of epoch ep + 1 are "previous epoch" votes, those of ep are too old
    newer = got["atts"]["target_epoch"] == ep + 1
    assert (status[newer] == 0).all() and (status[~newer] != 0).all() and newer.any() and (~newer).any()
    assert np.array_equal(count[newer], got["count"][newer]) and (count[~newer] == 0).all() and len(head) == 32
