"""CPU: L1 (C port) == L0 (literal pyspec) on randomised small inputs -- the condition under which L1 is accepted
as the checker for the BASELINE.json sizes and as the timed CPU baseline."""
import numpy as np
import pytest

from oracle import cport, spec
from tests.scenario import new_world, slot_committee_members

NONE32 = 0xFFFFFFFF


def flatten_store(store):
    """L0 Store -> the flat arrays L1 (and the engine) work on."""
    roots = list(store.blocks.keys())                       # insertion order: parents first
    idx = {r: i for i, r in enumerate(roots)}
    parent = np.array([idx.get(store.blocks[r].parent_root, NONE32) for r in roots], dtype=np.uint32)
    leaf_ok = np.zeros(len(roots), dtype=np.uint8)
    for i, r in enumerate(roots):
        st = store.block_states[r]
        cj = store.justified_checkpoint.epoch == spec.GENESIS_EPOCH or st.current_justified_checkpoint == store.justified_checkpoint
        cf = store.finalized_checkpoint.epoch == spec.GENESIS_EPOCH or st.finalized_checkpoint == store.finalized_checkpoint
        leaf_ok[i] = cj and cf
    state = store.checkpoint_states[store.justified_checkpoint]
    n = len(state.validators)
    vote = np.full(n, NONE32, dtype=np.uint32)
    for v, m in store.latest_messages.items():
        vote[v] = idx[m.root]
    bal = np.array([v.effective_balance for v in state.validators], dtype=np.uint64)
    ep = spec.get_current_epoch(state)
    flags = np.array([(1 if spec.is_active_validator(v, ep) else 0) | (2 if v.slashed else 0) for v in state.validators],
                     dtype=np.uint8)
    for v in store.equivocating_indices:
        flags[v] |= 4
    root_bytes = np.frombuffer(b"".join(roots), dtype=np.uint8).reshape(-1, 32)
    boost = idx[store.proposer_boost_root] if store.proposer_boost_root != spec.Root() else NONE32
    return roots, idx, parent, leaf_ok, root_bytes, vote, bal, flags, boost


@pytest.mark.parametrize("seed", range(6))
def test_get_head_l1_equals_l0(seed):
    rng = np.random.default_rng(seed)
    w = new_world(64, "minimal", PROPOSER_SCORE_BOOST=int(rng.choice([40, 70])))
    # uneven balances and a few inactive / slashed validators
    st = w.store.checkpoint_states[w.store.justified_checkpoint]
    for i, v in enumerate(st.validators):
        v.effective_balance = int(rng.integers(16, 33)) * 10**9
        if i % 13 == 0:
            v.exit_epoch = 0
        if i % 17 == 0:
            v.slashed = True
    roots = [w.store.justified_checkpoint.root]
    slot = 0
    for step in range(25):
        slot += 1
        w.tick_to_slot(slot, offset=int(rng.integers(0, spec.SECONDS_PER_SLOT)))
        parent = roots[int(rng.integers(max(0, len(roots) - 3), len(roots)))]
        if w.store.blocks[parent].slot < slot:
            roots.append(w.block(parent, slot, graffiti=bytes([step])))
        a_slot = slot - 1
        cand = [r for r in roots[-5:] if w.store.blocks[r].slot <= a_slot]
        if cand:
            voters = slot_committee_members(w.store, a_slot)
            w.vote(voters[: 1 + len(voters) // 2], cand[int(rng.integers(0, len(cand)))], a_slot)
        if step == 12:
            w.store.equivocating_indices.update(slot_committee_members(w.store, 2)[:2])
        r_list, idx, parent_a, leaf_ok, root_bytes, vote, bal, flags, boost = flatten_store(w.store)
        head, weights = cport.get_head(parent_a, leaf_ok, root_bytes, vote, bal, flags,
                                       idx[w.store.justified_checkpoint.root], boost,
                                       slots_per_epoch=spec.SLOTS_PER_EPOCH, boost_percent=spec.PROPOSER_SCORE_BOOST)
        assert r_list[head] == spec.get_head(w.store)
        for i, r in enumerate(r_list):
            assert int(weights[i]) == spec.get_latest_attesting_balance(w.store, r)


def test_update_latest_messages_l1_equals_l0():
    rng = np.random.default_rng(7)
    n_val, n_att = 200, 60
    members = rng.permutation(n_val).astype(np.uint32)
    sizes = [10] * 20
    offs = np.cumsum([0] + sizes).astype(np.uint32)
    store = spec.Store(0, 0, spec.Checkpoint(), spec.Checkpoint(), spec.Checkpoint(), spec.Root(), {3, 50, 77})
    roots = [spec.sha256(bytes([i])) for i in range(8)]
    val_flags = np.zeros(n_val, dtype=np.uint8)
    val_flags[[3, 50, 77]] = 4
    vote_epoch = np.zeros(n_val, dtype=np.uint64)
    vote_block = np.full(n_val, NONE32, dtype=np.uint32)
    bit_rows, mo, ep, blk = [], [], [], []
    for a in range(n_att):
        c = int(rng.integers(0, 20))
        bits = rng.random(10) < 0.5
        e = int(rng.integers(0, 4))
        b = int(rng.integers(0, 8))
        att = spec.Attestation(aggregation_bits=list(bits),
                               data=spec.AttestationData(beacon_block_root=roots[b], target=spec.Checkpoint(e, roots[0])))
        indices = sorted(int(members[offs[c] + i]) for i in range(10) if bits[i])
        spec.update_latest_messages(store, indices, att)
        bit_rows.append(bits); mo.append(offs[c]); ep.append(e); blk.append(b)
    arena = np.concatenate([np.packbits(np.asarray(b, dtype=np.uint8), bitorder="little") for b in bit_rows])
    cport.update_latest_messages(mo, [10] * n_att, np.arange(n_att) * 2, ep, blk, arena, members, val_flags,
                                 vote_epoch, vote_block)
    for v in range(n_val):
        if v in store.latest_messages:
            assert roots[vote_block[v]] == store.latest_messages[v].root and vote_epoch[v] == store.latest_messages[v].epoch
        else:
            assert vote_block[v] == NONE32


def test_bits_union_l1_equals_l0():
    rng = np.random.default_rng(3)
    data = spec.AttestationData(slot=5, index=1)
    atts = [spec.Attestation(aggregation_bits=list(rng.random(37) < 0.3), data=data) for _ in range(5)]
    want = spec.aggregate_attestations(atts).aggregation_bits
    arena = np.concatenate([np.packbits(np.asarray(a.aggregation_bits, dtype=np.uint8), bitorder="little") for a in atts])
    out, count = cport.bits_union([0, 5], np.arange(5), np.arange(5) * 5, arena, [37], [0], 5)
    assert list(np.unpackbits(out, bitorder="little")[:37].astype(bool)) == want and count[0] == sum(want)


def test_committee_slices_match_reference_example():
    """K2 (pe:472): 262 144 active validators, 64 committees/slot x 32 slots -> committees of 128;
    and the shuffle is a permutation (compute_shuffled_index pe:513-534)."""
    spec.use_preset("mainnet")
    n = 262144
    count = 64 * 32
    assert max(1, min(spec.MAX_COMMITTEES_PER_SLOT, n // spec.SLOTS_PER_EPOCH // spec.TARGET_COMMITTEE_SIZE)) == 64
    for index in (0, 1, 1000, count - 1):
        start, end = (n * index) // count, (n * (index + 1)) // count
        assert end - start == 128
    spec.use_preset("minimal")
    seed = spec.sha256(b"c1")
    perm = [spec.compute_shuffled_index(i, 100, seed) for i in range(100)]
    assert sorted(perm) == list(range(100))
    spec.use_preset("mainnet")


def test_all_cores_forms_equal_single_thread():
    """bench.py's cpu_baseline times the OpenMP forms (po_*_mt) on all host cores; they are the same loops split over
    independent units and must return what the single-thread checker returns, bit for bit."""
    import pos_evolution_amd.synth as synth
    from tests import helpers as H
    n_val, n_comm, spe = 6000, 64, 32
    tree = synth.random_tree(200, 7, "bushy")
    comm = synth.random_committees(n_val, n_comm, 7)
    atts, arena, _ = synth.epoch_attestations(comm, tree, 9, spe, seed=7, density=0.8, parts=3)
    bal = synth.balances(n_val, 7, mixed=True)
    flags = synth.validator_flags(n_val, 7, inactive_frac=0.02)
    flags[::97] |= 0x04
    pts, _ = H.oracle_points(n_val)
    cport.set_threads(4)
    assert cport.max_threads() >= 1
    # union
    cps = n_comm // spe
    pos = ((atts["slot"] % spe) * cps + atts["index"]).astype(np.int64)
    order = np.argsort(pos, kind="stable").astype(np.uint32)
    gstart = np.concatenate([[0], np.cumsum(np.bincount(pos, minlength=n_comm))]).astype(np.uint32)
    sizes = (comm.offsets[1:] - comm.offsets[:-1]).astype(np.uint32)
    out_off = np.concatenate([[0], np.cumsum((sizes + 7) // 8)]).astype(np.uint32)
    u1, c1 = cport.bits_union(gstart, order, atts["bits_offset"], arena, sizes, out_off[:-1], int(out_off[-1]))
    u2, c2 = cport.bits_union(gstart, order, atts["bits_offset"], arena, sizes, out_off[:-1], int(out_off[-1]), mt=True)
    assert np.array_equal(u1, u2) and np.array_equal(c1, c2)
    # G1 sums
    idx = np.random.default_rng(7).integers(0, n_val, size=5000, dtype=np.uint32)
    offs = np.sort(np.random.default_rng(8).integers(0, 5001, size=40)).astype(np.uint32)
    offs[0], offs[-1] = 0, 5000
    assert np.array_equal(cport.g1_sum_groups(pts, idx, offs), cport.g1_sum_groups(pts, idx, offs, mt=True))
    # ... and straight from the bits (what bench.py's cpu_baseline times) = the index-list form
    want = []
    bits_all = np.unpackbits(u1, bitorder="little")
    for c in range(n_comm):
        sel = bits_all[8 * out_off[c]: 8 * out_off[c] + sizes[c]].astype(bool)
        m = comm.members[comm.offsets[c]:comm.offsets[c + 1]][sel]
        want.append(cport.g1_sum_groups(pts, m, np.array([0, m.size], dtype=np.uint32))[0])
    for mt in (False, True):
        got = cport.g1_sum_attesters(comm.offsets[:-1], sizes, out_off[:-1], u1, comm.members, pts, mt=mt)
        assert np.array_equal(got, np.stack(want))
    # LMD + flags on the aggregates (one per committee: pairwise disjoint, the mt forms' precondition)
    blk = np.random.default_rng(9).integers(0, 200, size=n_comm, dtype=np.uint32)
    te = np.full(n_comm, 9, dtype=np.uint64)
    ve = [np.zeros(n_val, dtype=np.uint64) for _ in range(2)]
    vb = [np.full(n_val, NONE32, dtype=np.uint32) for _ in range(2)]
    for k, mt in enumerate((False, True)):
        cport.update_latest_messages(comm.offsets[:-1], sizes, out_off[:-1], te, blk, u1, comm.members, flags, ve[k], vb[k], mt=mt)
    assert np.array_equal(ve[0], ve[1]) and np.array_equal(vb[0], vb[1])
    nums, parts = [], []
    for mt in (False, True):
        pc, pp = np.zeros(n_val, dtype=np.uint8), np.zeros(n_val, dtype=np.uint8)
        nums.append(cport.process_attestation_flags(comm.offsets[:-1], sizes, out_off[:-1], np.full(n_comm, 7, np.uint8),
                                                    (np.arange(n_comm) % 2).astype(np.uint8), u1, comm.members, bal,
                                                    10**9, 321, pc, pp, mt=mt))
        parts.append((pc, pp))
    assert np.array_equal(nums[0], nums[1])
    assert np.array_equal(parts[0][0], parts[1][0]) and np.array_equal(parts[0][1], parts[1][1])
    # get_head incl. boost and equivocators
    leaf_ok = (np.random.default_rng(3).random(200) > 0.1).astype(np.uint8)
    for boost in (NONE32, 150):
        h1, w1 = cport.get_head(tree.parent, leaf_ok, tree.roots, vb[0], bal, flags, 0, boost_idx=boost)
        h2, w2 = cport.get_head(tree.parent, leaf_ok, tree.roots, vb[0], bal, flags, 0, boost_idx=boost, mt=True)
        assert h1 == h2 and np.array_equal(w1, w2)
