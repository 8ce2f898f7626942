"""-m gpu: the HIP engine through the C ABI vs the L1 C oracle on identical seeded inputs (bit-exact)."""
import numpy as np
import pytest

import pos_evolution_amd.synth as synth
from oracle import cport, g1
from tests import helpers as H

pytestmark = pytest.mark.gpu
NONE32 = 0xFFFFFFFF


# ---------------------------------------------------------------- G1
def test_g1_known_answers(engine_factory):
    e = engine_factory()
    G96 = np.frombuffer(g1.to_bytes96(g1.G), dtype=np.uint8)
    # (i+1)*G: 1G + 2G = 3G, then + 3G forces the doubling branch
    pts = cport.g1_arith_progression(G96.tobytes(), G96.tobytes(), 64)
    out = e.g1_sum([0, 64, 64, 65], index=[*range(64), 0], points96=pts)
    assert out[0].tobytes() == g1.to_bytes96(g1.mul(64 * 65 // 2, g1.G))
    assert out[1].tobytes()[0] == 0x40 and not any(out[1][1:])          # empty group = infinity
    assert out[2].tobytes() == g1.to_bytes96(g1.G)
    # external known answers (compressed 2G, 3G)
    two = e.g1_sum([0, 2], index=[0, 0], points96=pts)[0].tobytes()
    assert g1.compress(g1.from_bytes96(two)).hex().startswith("a572cbea904d6746")
    # P + (-P) = infinity, infinity inputs are skipped
    A = g1.mul(7, g1.G)
    trio = np.stack([np.frombuffer(g1.to_bytes96(p), dtype=np.uint8) for p in (A, g1.neg(A), None, A)])
    out = e.g1_sum([0, 2, 4, 4], points96=trio)
    assert out[0][0] == 0x40
    assert out[1].tobytes() == g1.to_bytes96(A)
    assert out[2][0] == 0x40


@pytest.mark.parametrize("n,groups", [(1000, 7), (5000, 300), (70000, 3), (300000, 2048)])
def test_g1_sum_vs_oracle(engine_factory, n, groups):
    e = engine_factory()
    pts, (a, b) = H.oracle_points(n)
    rng = np.random.default_rng(n)
    index = rng.integers(0, n, size=n, dtype=np.uint32)
    cuts = np.sort(rng.integers(0, n + 1, size=groups - 1))
    offsets = np.concatenate([[0], cuts, [n]]).astype(np.uint32)
    got = e.g1_sum(offsets, index=index, points96=pts)
    want = cport.g1_sum_groups(pts, index, offsets)
    assert np.array_equal(got, want)
    # closed form on one group (independent of the C oracle's adder)
    g = int(np.argmax(np.diff(offsets.astype(np.int64))))
    assert got[g].tobytes() == H.closed_form_sum(index[offsets[g]:offsets[g + 1]], a, b)


def test_g1_single_huge_group(engine_factory):
    e = engine_factory()
    n = 200000
    pts, (a, b) = H.oracle_points(n)
    got = e.g1_sum([0, n], points96=pts)
    assert got[0].tobytes() == H.closed_form_sum(range(n), a, b)


# ---------------------------------------------------------------- get_head
@pytest.mark.parametrize("n_val,n_blocks,kind,boost,mixed", [
    (1024, 40, "branchy", False, False),
    (65536, 2048, "branchy", True, False),
    (262144, 4096, "chain", True, False),
    (262144, 4096, "bushy", False, True),
    (100003, 8192, "bushy", True, True),
])
def test_get_head_vs_oracle(engine_factory, n_val, n_blocks, kind, boost, mixed):
    e = engine_factory()
    seed = n_val + n_blocks
    tree = synth.random_tree(n_blocks, seed, kind)
    rng = np.random.default_rng(seed)
    # leaf checkpoints: ~10 % of blocks carry a non-matching justified checkpoint (filter_block_tree)
    good = (1, tree.roots[0].tobytes())
    bad = (1, tree.roots[min(1, n_blocks - 1)].tobytes())
    leaf_ok = rng.random(n_blocks) > 0.1
    leaf_ok[0] = True
    leaf_cp = [((good if leaf_ok[i] else bad), good) for i in range(n_blocks)]
    H.load_tree(e, tree, leaf_cp)
    e.set_checkpoints(good, good)            # epoch 1 != GENESIS: the leaf test is live
    bal = synth.balances(n_val, seed, mixed)
    flags = synth.validator_flags(n_val, seed, inactive_frac=0.005, slashed_frac=0.01)
    e.set_validators(bal, flags)
    # latest messages are installed through on_attestation (the ABI has no back door)
    comm = synth.random_committees(n_val, 64, seed)
    vote = synth.zipf_votes(n_val, n_blocks, seed)
    rng2 = np.random.default_rng(seed + 1)
    equiv = rng2.choice(n_val, size=max(1, n_val // 100), replace=False)
    e.mark_equivocating(equiv)
    flags_o = flags.copy()
    flags_o[equiv] |= 0x04
    votes_dev = _install_votes(e, tree, comm, vote)
    boost_idx = NONE32
    if boost:
        boost_idx = n_blocks - 1
        e.set_proposer_boost(tree.roots[boost_idx].tobytes())
    parent = tree.parent.copy()
    head_o, w_o = cport.get_head(parent, leaf_ok.astype(np.uint8), tree.roots, votes_dev, bal, flags_o, 0, boost_idx)
    w_e = e.get_weights()
    assert np.array_equal(w_e, w_o)
    assert e.get_head() == tree.roots[head_o].tobytes()
    # inside pipelined calls get_head runs its lean shapes (k_votes<1>; k_tree<512, 4> / <512, 8> from 1025 to 4096
    # blocks): same head, same per-block weights
    for lagged in (False, True):
        with e.pipeline(lagged=lagged):
            assert e.get_head() == tree.roots[head_o].tobytes()
        e.drain()
        assert np.array_equal(e.last_weights(), w_o)


def _install_votes(e, tree, comm, vote):
    """Drive `vote` into the engine through on_attestation batches; returns the vote table actually installed
    (validators whose Zipf block fails validate_on_attestation keep no message)."""
    n_comm = comm.offsets.size - 1
    spe = 32
    cps = n_comm // spe
    n_blocks = tree.roots.shape[0]
    # every attestation: slot = 31 of the block's epoch or later so block.slot <= slot; use a single far epoch
    E = int(tree.slot.max()) // spe + 1
    e.set_committees(E, comm.offsets, comm.members)
    e.on_tick((E + 2) * spe * 12)
    # equivocators' votes are dropped by update_latest_messages (pe:1438): the caller accounts for that
    atts_list, bits_list = [], []
    installed = np.full(vote.shape[0], NONE32, dtype=np.uint32)
    for c in range(n_comm):
        mem = comm.members[comm.offsets[c]:comm.offsets[c + 1]]
        v = vote[mem]
        for blk in np.unique(v[v != NONE32]):
            blk = int(blk)
            a = np.zeros(1, dtype=synth.ATT_DTYPE)[0]
            a["slot"], a["index"] = E * spe + c // cps, c % cps
            a["beacon_block_root"] = tree.roots[blk]
            a["target_epoch"] = E
            a["target_root"] = tree.roots[synth.ancestor_at(tree, blk, E * spe)]
            a["source_root"] = tree.roots[0]
            a["flags"] = 3   # signature valid | is_from_block (no wall-clock epoch check, pe:1423)
            atts_list.append(a)
            bits_list.append(v == blk)
            installed[mem[v == blk]] = blk
    atts = np.array(atts_list, dtype=synth.ATT_DTYPE)
    arena, offs, nb = synth.pack_bit_rows(bits_list)
    atts["bits_offset"], atts["n_bits"] = offs, nb
    status, _, _ = e.on_attestation_batch(packed=(atts, arena))
    assert (status == 0).all(), np.unique(status)
    return installed


@pytest.mark.parametrize("eta", [1, 20, 48, 10**6])
def test_get_head_vote_expiry_vs_oracle(engine_factory, eta):
    """Vote-expiry variant (RLMD-GHOST, pe:1585-1596) at 100 K validators: the engine with vote_expiry_slots = eta
    equals the C oracle run on the vote table with the expired messages (slot + eta < current slot) removed."""
    n_val, n_blocks, spe = 100000, 1024, 32
    e = engine_factory(vote_expiry_slots=eta)
    tree = synth.random_tree(n_blocks, 77, "bushy")
    H.load_tree(e, tree)
    bal = synth.balances(n_val, 77, True)
    flags = synth.validator_flags(n_val, 77, inactive_frac=0.005, slashed_frac=0.01)
    e.set_validators(bal, flags)
    comm = synth.random_committees(n_val, 64, 77)
    vote = synth.zipf_votes(n_val, n_blocks, 77)
    installed = _install_votes(e, tree, comm, vote)       # committee c attests at slot E*32 + c // cps; now = (E+2)*32
    E = int(tree.slot.max()) // spe + 1
    cps = (comm.offsets.size - 1) // spe
    slot_of = np.zeros(n_val, dtype=np.int64)
    for c in range(comm.offsets.size - 1):
        slot_of[comm.members[comm.offsets[c]:comm.offsets[c + 1]]] = E * spe + c // cps
    now = (E + 2) * spe
    alive = installed.copy()
    alive[slot_of + eta < now] = NONE32
    assert (eta >= 64) == np.array_equal(alive, installed)          # eta = 48 keeps slots E*32+16.., eta = 1 keeps none
    leaf_ok = np.ones(n_blocks, dtype=np.uint8)
    head_o, w_o = cport.get_head(tree.parent.copy(), leaf_ok, tree.roots, alive, bal, flags, 0, NONE32)
    assert np.array_equal(e.get_weights(), w_o)
    assert e.get_head() == tree.roots[head_o].tobytes()
    if eta == 1:
        assert not w_o.any()


# ---------------------------------------------------------------- LMD update (ordering rule)
def test_lmd_update_vs_oracle(engine_factory):
    e = engine_factory()
    n_val, n_blocks, spe = 50000, 300, 32
    tree = synth.random_tree(n_blocks, 5, "branchy")
    H.load_tree(e, tree)
    bal = synth.balances(n_val, 5)
    flags = synth.validator_flags(n_val, 5)
    e.set_validators(bal, flags)
    rng = np.random.default_rng(5)
    equiv = rng.choice(n_val, size=500, replace=False)
    e.mark_equivocating(equiv)
    flags_o = flags.copy()
    flags_o[equiv] |= 0x04
    vote_epoch = np.zeros(n_val, dtype=np.uint64)
    vote_block = np.full(n_val, NONE32, dtype=np.uint32)
    last_epoch = int(tree.slot.max()) // spe + 1
    e.on_tick((last_epoch + 3) * spe * 12)
    # three batches; epochs go up and DOWN so that stale votes must lose, duplicates inside a batch so
    # that first-seen must win (pe:1383, pe:1440)
    for batch, epochs in enumerate([(last_epoch + 1,), (last_epoch,), (last_epoch + 2, last_epoch + 1)]):
        all_atts, all_bits, all_comm = [], [], []
        for ep in epochs:
            comm = synth.random_committees(n_val, 64, 100 * batch + ep)
            e.set_committees(ep, comm.offsets, comm.members)
            atts, arena, bit_rows = synth.epoch_attestations(comm, tree, ep, spe, seed=batch, density=0.6, parts=3)
            # overlapping duplicates: re-emit the first 20 rows with different head votes
            atts["flags"] = 3
            dup = atts[:20].copy()
            dup["beacon_block_root"] = tree.roots[0]
            dup["target_root"] = tree.roots[0]
            all_atts += [atts, dup]
            all_bits += bit_rows + bit_rows[:20]
            all_comm += [comm] * (len(atts) + 20)
        atts = np.concatenate(all_atts)
        arena, offs, nb = synth.pack_bit_rows(all_bits)
        atts["bits_offset"], atts["n_bits"] = offs, nb
        status, _, count = e.on_attestation_batch(packed=(atts, arena))
        ok = status == 0
        assert ok.sum() > 0
        # oracle: sequential over the accepted rows, each against its own committee table
        for i in np.nonzero(ok)[0]:
            comm = all_comm[i]
            mo, nbits, bo = H.att_device_rows(atts[i:i + 1], comm, spe)
            blk = e.block_index_of(atts[i]["beacon_block_root"].tobytes())
            cport.update_latest_messages(mo, nbits, bo, atts[i:i + 1]["target_epoch"], [blk], arena, comm.members,
                                         flags_o, vote_epoch, vote_block)
        ep_e, blk_e = e.latest_messages()
        assert np.array_equal(blk_e, vote_block)
        has = vote_block != NONE32
        assert np.array_equal(ep_e[has], vote_epoch[has])


# ---------------------------------------------------------------- aggregation
def test_aggregate_vs_oracle(engine_factory):
    e = engine_factory()
    n_val, spe = 40000, 32
    tree = synth.random_tree(64, 9, "branchy")
    H.load_tree(e, tree)
    pts, (a, b) = H.oracle_points(n_val)
    e.set_validators(synth.balances(n_val, 9), synth.validator_flags(n_val, 9), pts)
    comm = synth.random_committees(n_val, 128, 9)
    e.set_committees(1, comm.offsets, comm.members)
    atts, arena, bit_rows = synth.epoch_attestations(comm, tree, 1, spe, seed=9, density=0.9, parts=4)
    perm = np.random.default_rng(9).permutation(len(atts))     # shuffle: grouping must not rely on adjacency
    atts, bit_rows = atts[perm], [bit_rows[i] for i in perm]
    arena, offs, nb = synth.pack_bit_rows(bit_rows)
    atts["bits_offset"], atts["n_bits"] = offs, nb
    sigs = pts[np.random.default_rng(10).integers(0, n_val, size=len(atts))]
    res = e.aggregate(packed=(atts, arena), sig_points96=sigs, want_aggregate_pubkeys=True)
    assert res["n_groups"] == 128
    for g in range(res["n_groups"]):
        members_i = np.nonzero(res["group_of"] == g)[0]
        want_bits = np.zeros_like(bit_rows[members_i[0]], dtype=bool)
        for i in members_i:
            want_bits |= np.asarray(bit_rows[i], dtype=bool)
        assert np.array_equal(res["bits"][g], want_bits)
        assert res["count"][g] == want_bits.sum()
        out = res["atts"][g]
        cps = 128 // spe
        c = int((out["slot"] % spe) * cps + out["index"])
        mem = comm.members[comm.offsets[c]:comm.offsets[c + 1]]
        assert res["aggpk96"][g].tobytes() == H.closed_form_sum(mem[want_bits], a, b)
    want_sig = cport.g1_sum_groups(sigs, np.argsort(res["group_of"], kind="stable").astype(np.uint32),
                                   np.concatenate([[0], np.cumsum(np.bincount(res["group_of"], minlength=128))]))
    assert np.array_equal(res["sig96"], want_sig)


# ---------------------------------------------------------------- process_attestation flags
def test_process_attestation_vs_oracle(engine_factory):
    from pos_evolution_amd._abi import pe_state_ctx
    e = engine_factory()
    n_val = 30000
    tree = synth.random_tree(80, 11, "chain")
    H.load_tree(e, tree)
    bal = synth.balances(n_val, 11, mixed=True)
    e.set_validators(bal, synth.validator_flags(n_val, 11))
    spe = 32
    comms = {ep: synth.random_committees(n_val, 64, 11 + ep) for ep in (1, 2)}
    parts, rows_comm = [], []
    bit_rows = []
    for ep, seed, dens, np_ in ((1, 11, 0.7, 2), (2, 13, 0.6, 1), (1, 12, 0.95, 1), (2, 14, 0.9, 1)):
        e.set_committees(ep, comms[ep].offsets, comms[ep].members)
        a, _, br = synth.epoch_attestations(comms[ep], tree, ep, spe, seed=seed, density=dens, parts=np_,
                                            source=(0, tree.roots[0].tobytes()))
        parts.append(a)
        bit_rows += br
        rows_comm += [ep] * len(a)
    all_atts = np.concatenate(parts)
    arena, offs, nb = synth.pack_bit_rows(bit_rows)
    all_atts["bits_offset"], all_atts["n_bits"] = offs, nb
    ctx = pe_state_ctx()
    state_slot = 70   # epoch 2; epoch-1 rows with slot < 38 and epoch-2 rows with slot >= 70 fall outside pe:726
    ctx.slot = state_slot
    tip = 69
    ctx.chain_tip_root[:] = tree.roots[tip].tobytes()
    ctx.current_justified_epoch, ctx.previous_justified_epoch = 0, 0
    ctx.current_justified_root[:] = tree.roots[0].tobytes()
    ctx.previous_justified_root[:] = tree.roots[0].tobytes()
    ctx.base_reward_per_increment = 357
    status, num = e.process_attestation_batch(ctx, packed=(all_atts, arena))
    ok = status == 0
    assert ok.sum() > len(all_atts) // 3 and (status == 13).sum() > 0
    # oracle: flag masks recomputed independently in Python from the spec text (A.9)
    cur_epoch = state_slot // spe
    # one members array per epoch table: concatenate both and offset epoch-2 rows
    members = np.concatenate([comms[1].members, comms[2].members])
    mo = np.zeros(len(all_atts), dtype=np.uint32)
    for i, a in enumerate(all_atts):
        ep = rows_comm[i]
        m, _, _ = H.att_device_rows(all_atts[i:i + 1], comms[ep], spe)
        mo[i] = m[0] + (0 if ep == 1 else comms[1].members.size)
    nbits, bo = all_atts["n_bits"].astype(np.uint32), all_atts["bits_offset"].astype(np.uint32)
    masks, which = [], []
    for a in all_atts:
        delay = state_slot - int(a["slot"])
        tgt = synth.ancestor_at(tree, tip, int(a["target_epoch"]) * spe)
        head = synth.ancestor_at(tree, tip, int(a["slot"]))
        mt = a["target_root"].tobytes() == tree.roots[tgt].tobytes()
        mh = mt and a["beacon_block_root"].tobytes() == tree.roots[head].tobytes()
        m = (1 if delay <= 5 else 0) | (2 if mt and delay <= spe else 0) | (4 if mh and delay == 1 else 0)
        masks.append(m)
        which.append(0 if int(a["target_epoch"]) == cur_epoch else 1)
    pc = np.zeros(n_val, dtype=np.uint8)
    pp = np.zeros(n_val, dtype=np.uint8)
    sel = np.nonzero(ok)[0]
    want = cport.process_attestation_flags(mo[sel], nbits[sel], bo[sel], np.array(masks, dtype=np.uint8)[sel],
                                           np.array(which, dtype=np.uint8)[sel], arena, members, bal,
                                           10**9, 357, pc, pp)
    assert np.array_equal(num[sel], want)
    assert np.array_equal(e.participation_get(0), pc)
    assert np.array_equal(e.participation_get(1), pp)


# ---------------------------------------------------------------- get_indexed_attestation
def test_indexed_attestations_sorted_indices(engine_factory):
    """attesting_indices = sorted(committee[i] for i with bits[i]) (A.6) for ragged committees up to 2048 members."""
    e = engine_factory()
    n_val = 6000
    e.set_validators(synth.balances(n_val, 31), np.ones(n_val, dtype=np.uint8))
    sizes = [2048, 0, 1, 2, 63, 64, 65, 1000] + [0] * 23 + [777]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    rng = np.random.default_rng(31)
    members = rng.permutation(n_val)[: offsets[-1]].astype(np.uint32)
    e.set_committees(3, offsets, members)
    rows, want = [], []
    for slot_in_epoch, dens in [(0, 1.0), (0, 0.5), (1, 1.0), (2, 1.0), (3, 0.5), (4, 0.9), (5, 0.0), (6, 0.3), (7, 0.7),
                                (31, 0.5), (9, 1.0)]:
        size = sizes[slot_in_epoch]
        bits = rng.random(size) < dens
        rows.append(pea_row(3 * 32 + slot_in_epoch, bits))
        m = members[offsets[slot_in_epoch]:offsets[slot_in_epoch + 1]]
        want.append(np.sort(m[bits]))
    rows.append(pea_row(3 * 32 + 6, np.ones(10)))        # shorter than the committee (65): rejected
    rows.append(pea_row(4 * 32, np.ones(3)))             # no table for epoch 4
    status, off, idx = e.get_indexed_attestations(rows)
    assert list(status) == [0] * 11 + [10, 8]
    for k, w in enumerate(want):
        assert np.array_equal(idx[off[k]:off[k + 1]], w), k
    assert off[-1] == sum(len(w) for w in want)


def pea_row(slot, bits):
    import pos_evolution_amd as pea
    return pea.AttRow(slot, 0, bytes(32), 0, bytes(32), slot // 32, bytes(32), np.asarray(bits, dtype=np.uint8))


# ---------------------------------------------------------------- BLSPubkey wire format
def test_g1_decompress_vs_oracle(engine_factory):
    """48-byte compressed keys -> affine on the GPU (square root + sign), against oracle/g1.py; malformed and
    off-curve encodings are reported per key."""
    e = engine_factory()
    n = 3000
    pts96, (a, b) = H.oracle_points(n)
    pts = [g1.from_bytes96(pts96[i].tobytes()) for i in range(n)]
    comp = np.frombuffer(b"".join(g1.compress(p) for p in pts), dtype=np.uint8).reshape(n, 48).copy()
    assert np.array_equal(e.g1_compress(pts96), comp)
    out, status = e.g1_decompress(comp)
    assert not status.any() and np.array_equal(out, pts96)
    gen = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
                        "6c55e83ff97a1aeffb3af00adb22c6bb")
    # find an x that is not on the curve
    x = 5
    while pow((x ** 3 + 4) % g1.P, (g1.P - 1) // 2, g1.P) == 1:
        x += 1
    off_curve = bytearray(x.to_bytes(48, "big"))
    off_curve[0] |= 0x80
    flipped = bytes([gen[0] ^ 0x20]) + gen[1:]
    special = [gen, flipped, g1.compress(None), bytes(48), bytes([0xE0]) + bytes(47), bytes([0x9F]) + b"\xff" * 47,
               bytes(off_curve), bytes([0xC0]) + bytes(46) + b"\x01"]
    out, status = e.g1_decompress(np.frombuffer(b"".join(special), dtype=np.uint8).reshape(-1, 48))
    assert list(status) == [0, 0, 0, 1, 1, 1, 2, 1]
    assert out[0].tobytes() == g1.to_bytes96(g1.G) and out[1].tobytes() == g1.to_bytes96(g1.neg(g1.G))
    assert out[2][0] == 0x40 and not out[2][1:].any()
    assert not out[3:].any()


def test_registry_from_compressed_pubkeys(engine_factory):
    """pe_set_pubkeys_compressed loads the same registry as pe_set_validators with uncompressed keys."""
    import pos_evolution_amd as pea
    n = 5000
    pts96, (a, b) = H.oracle_points(n)
    comp = np.frombuffer(b"".join(g1.compress(g1.from_bytes96(pts96[i].tobytes())) for i in range(n)), dtype=np.uint8)
    bal, flags = synth.balances(n, 3), np.ones(n, dtype=np.uint8)
    offsets = [0, 100, 100, 4000, n]
    e1 = engine_factory()
    e1.set_validators(bal, flags, pts96)
    want = e1.g1_sum(offsets)
    e2 = engine_factory()
    e2.set_validators(bal, flags)
    e2.set_pubkeys_compressed(comp)
    assert np.array_equal(e2.g1_sum(offsets), want)
    bad = comp.copy().reshape(n, 48)
    bad[17, 0] &= 0x7F                                  # compression bit cleared
    with pytest.raises(pea.EngineError):
        e2.set_pubkeys_compressed(bad)
    with pytest.raises(pea.EngineError):                 # the failed load leaves no pubkeys behind
        e2.g1_sum(offsets)


# ---------------------------------------------------------------- hypothesis: generated worlds through the engine
def test_get_head_generated_worlds_vs_definition(engine_factory):
    """Small generated block trees / votes / balances / flags / leaf tests / boosts through the ENGINE (votes installed
    by on_attestation) against the definition-level get_head of tests/test_oracle_properties.py: odd shapes a seeded
    generator rarely hits (single-block tree, no validators, all leaves filtered out, boost on a dead branch)."""
    from hypothesis import given, settings, HealthCheck
    from tests.test_oracle_properties import definition_get_head, worlds
    e = engine_factory(max_committee_tables=2)
    spe = 32

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(worlds())
    def check(wd):
        parent, leaf_ok, roots, vote, bal, flags, boost = wd
        if len(set(roots)) != len(roots):
            return
        n, n_val = len(parent), len(vote)
        slot = [0] * n
        for i in range(1, n):
            slot[i] = slot[parent[i]] + 1
        tree = synth.Tree(np.frombuffer(b"".join(roots), dtype=np.uint8).reshape(-1, 32).copy(),
                          np.array(parent, dtype=np.uint32), np.array(slot, dtype=np.uint64))
        good = (1, roots[0])
        bad = (1, roots[min(1, n - 1)] if n > 1 else bytes(32))
        leaf_cp = [((good if leaf_ok[i] else bad), good) for i in range(n)]
        H.load_tree(e, tree, leaf_cp)
        e.set_checkpoints(good, good)
        fl = np.array(flags, dtype=np.uint8)
        e.set_validators(np.array(bal, dtype=np.uint64), fl & 3)
        equiv = np.nonzero(fl & 4)[0]
        if equiv.size:
            e.mark_equivocating(equiv)
        installed = np.full(n_val, NONE32, dtype=np.uint32)
        if n_val:
            comm = synth.random_committees(n_val, spe, 5)
            installed = _install_votes(e, tree, comm, np.array(vote, dtype=np.uint32))
            installed[equiv] = NONE32                      # pe:1438: equivocators' messages are never recorded
        else:
            e.on_tick((max(slot) // spe + 3) * spe * 12)
        if boost != NONE32:
            e.set_proposer_boost(roots[boost])
        head_d, w_d = definition_get_head(parent, leaf_ok, roots, installed, bal, flags, 0, boost)
        assert [int(x) for x in e.get_weights()] == w_d
        assert e.get_head() == roots[head_d]

    check()


# ---------------------------------------------------------------- checkpoint / resume
def test_export_import_state_round_trip(engine_factory):
    """SURVEY.md 5 "checkpoint / resume": a store exported as flat arrays and imported into a fresh handle answers
    every query identically and keeps evolving identically (vote-expiry variant on, so the slots travel too)."""
    n_val, n_blocks, spe = 30000, 400, 32
    cfg = dict(vote_expiry_slots=200, max_committee_tables=4)
    a = engine_factory(**cfg)
    tree = synth.random_tree(n_blocks, 91, "branchy")
    rng = np.random.default_rng(91)
    good = (1, tree.roots[0].tobytes())
    bad = (1, tree.roots[1].tobytes())
    leaf_cp = [((good if rng.random() > 0.1 or i == 0 else bad), good) for i in range(n_blocks)]
    H.load_tree(a, tree, leaf_cp)
    a.set_checkpoints(good, good)
    bal = synth.balances(n_val, 91, True)
    flags = synth.validator_flags(n_val, 91, inactive_frac=0.01, slashed_frac=0.01)
    a.set_validators(bal, flags)
    equiv = rng.choice(n_val, size=300, replace=False)
    a.mark_equivocating(equiv)
    comm = synth.random_committees(n_val, 64, 91)
    _install_votes(a, tree, comm, synth.zipf_votes(n_val, n_blocks, 91))
    a.set_proposer_boost(tree.roots[n_blocks - 1].tobytes())
    a.participation_set(0, rng.integers(0, 8, size=n_val).astype(np.uint8))
    a.participation_set(1, rng.integers(0, 8, size=n_val).astype(np.uint8))
    # the working-state view (what process_attestation / the FFG sums read) differs from the justified state's registry
    sbal = synth.balances(n_val, 93, True)
    sflags = synth.validator_flags(n_val, 93, inactive_frac=0.03, slashed_frac=0.02)
    a.state_set_validators(sbal, sflags)
    # one table computed on the GPU (its members never left the device), besides the host-provided one of _install_votes
    Ec = int(tree.slot.max()) // spe + 3
    a.compute_committees(Ec, bytes(range(32)), n_val, 64, 10, want_result=False)

    st = a.export_state()
    assert st["state_view"] is not None and Ec in st["committees"] and len(st["committees"]) >= 2
    b = engine_factory(**cfg)
    b.import_state(st, bal)
    assert b.store_scalars() == a.store_scalars()
    st_b = b.export_state()
    for k, v in st.items():
        if k == "scalars":
            continue
        if k in ("participation", "state_view"):
            assert all(np.array_equal(x, y) for x, y in zip(v, st_b[k])), k
        elif k == "committees":
            assert sorted(v) == sorted(st_b[k])
            for ep in v:
                assert all(np.array_equal(x, y) for x, y in zip(v[ep], st_b[k][ep])), ep
        else:
            assert np.array_equal(v, st_b[k]), k
    assert a.ffg_balances() == b.ffg_balances()
    assert np.array_equal(a.get_weights(), b.get_weights()) and a.get_head() == b.get_head()

    # both keep evolving identically: a later epoch's attestations, then time moves on until votes start to expire
    E = int(tree.slot.max()) // spe + 1
    comm2 = synth.random_committees(n_val, 64, 92)
    atts, arena, _ = synth.epoch_attestations(comm2, tree, E + 1, spe, seed=92, density=0.5, parts=2,
                                              source=(0, tree.roots[0].tobytes()), vote_recent=30)
    for e in (a, b):
        e.set_committees(E + 1, comm2.offsets, comm2.members)
        e.on_tick((E + 2) * spe * 12)
        status, _, _ = e.on_attestation_batch(packed=(atts, arena))
        assert not status.any()
    assert np.array_equal(a.get_weights(), b.get_weights()) and a.get_head() == b.get_head()
    for e in (a, b):
        e.on_tick((E + 7) * spe * 12)                     # epoch-E votes cast before slot E*32+24 are now older than 200 slots
    wa = a.get_weights()
    assert np.array_equal(wa, b.get_weights()) and a.get_head() == b.get_head() and wa.any()
