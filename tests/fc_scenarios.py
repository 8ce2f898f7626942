"""Fork-choice scenarios shared by the CPU oracle tests and the -m gpu differential tests.

Each scenario takes ``mk(n_validators, **kw) -> World``; with an engine behind the World every handler call
is cross-checked (tests/scenario.py).  Names follow the upstream pyspec tests they recreate
(consensus-specs test_get_head.py / test_on_attestation.py -- not available offline) and the
K1..K10 known answers the reference's prose pins.
"""
from oracle import spec
from tests.scenario import World, slot_committee_members

ETH = 10**9


def genesis_head(mk):
    """K10 / upstream test_genesis: no children -> get_head returns the anchor root (pe:1106, pe:1112-1113)."""
    w = mk(16)
    anchor = w.store.justified_checkpoint.root
    assert w.head() == anchor


def chain_no_attestations(mk):
    """upstream test_chain_no_attestations: the only chain is followed to its tip."""
    w = mk(16)
    anchor = w.store.justified_checkpoint.root
    w.tick_to_slot(1)
    b1 = w.block(anchor, 1)
    w.tick_to_slot(2)
    b2 = w.block(b1, 2)
    assert w.head() == b2


def split_tie_breaker_no_attestations(mk):
    """K4 / upstream test_split_tie_breaker_no_attestations: equal weight -> lexicographically higher root."""
    w = mk(16)
    anchor = w.store.justified_checkpoint.root
    w.tick_to_slot(1, offset=spec.SECONDS_PER_SLOT - 1)  # late: no proposer boost for either
    a = w.block(anchor, 1, graffiti=b"a")
    b = w.block(anchor, 1, graffiti=b"b")
    assert w.store.proposer_boost_root == spec.Root()
    assert w.head() == max(a, b)


def shorter_chain_but_heavier_weight(mk):
    """upstream test_shorter_chain_but_heavier_weight: LMD weight beats chain length."""
    w = mk(64)
    anchor = w.store.justified_checkpoint.root
    w.tick_to_slot(1, offset=spec.SECONDS_PER_SLOT - 1)
    long_tip = anchor
    for s in range(1, 4):
        w.tick_to_slot(s, offset=spec.SECONDS_PER_SLOT - 1)
        long_tip = w.block(long_tip, s, graffiti=b"long")
    short = w.block(anchor, 1, graffiti=b"short")
    assert w.head() == long_tip or w.head() == short  # tie at zero weight, decided by roots
    w.tick_to_slot(5)
    voters = slot_committee_members(w.store, 3)
    w.vote(voters, short, 3)
    assert w.head() == short
    assert spec.get_latest_attesting_balance(w.store, short) == len(voters) * 32 * ETH


def lmd_walkthrough_five_validators(mk):
    """K1 (pe:306-318): at the first fork one side holds 4 latest messages against 1; descend into it, then
    pick the heavier grandchild.  K3 (pe:322): weight = stake voting for B or descendants of B."""
    w = mk(5)  # minimal preset: 8 slots/epoch, committees of 0/1 validators
    anchor = w.store.justified_checkpoint.root
    w.tick_to_slot(1, offset=spec.SECONDS_PER_SLOT - 1)
    top = w.block(anchor, 1, graffiti=b"top")
    bottom = w.block(anchor, 1, graffiti=b"bottom")
    w.tick_to_slot(2, offset=spec.SECONDS_PER_SLOT - 1)
    mid = w.block(bottom, 2, graffiti=b"mid")
    low = w.block(bottom, 2, graffiti=b"low")
    w.tick_to_slot(spec.SLOTS_PER_EPOCH)  # all of epoch 0 is in the past
    # who attests when: each validator sits in exactly one slot committee of the epoch
    slot_of = {}
    for s in range(spec.SLOTS_PER_EPOCH):
        for v in slot_committee_members(w.store, s):
            slot_of[v] = s
    assert sorted(slot_of) == [0, 1, 2, 3, 4]
    late = [v for v in range(5) if slot_of[v] >= 2]
    early = [v for v in range(5) if slot_of[v] < 2]
    # votes: one validator for the top chain, four below the bottom block (3 on mid's side incl. bottom itself)
    plan = {}
    order = late + early
    plan[order[0]] = top
    plan[order[1]] = mid
    plan[order[2]] = mid
    plan[order[3]] = low
    plan[order[4]] = bottom
    applied = {}
    for v, root in plan.items():
        s = max(slot_of[v], w.store.blocks[root].slot)
        if spec.compute_epoch_at_slot(s) != 0 or s != slot_of[v]:
            continue  # a validator can only attest in its own slot, and not to a block from the future
        w.vote([v], root, s)
        applied[v] = root
    n_top = sum(1 for r in applied.values() if r == top)
    n_bottom_subtree = sum(1 for r in applied.values() if r in (bottom, mid, low))
    assert spec.get_latest_attesting_balance(w.store, top) == n_top * 32 * ETH
    assert spec.get_latest_attesting_balance(w.store, bottom) == n_bottom_subtree * 32 * ETH
    if n_bottom_subtree > n_top:
        assert w.head() in (mid, low)
    return n_top, n_bottom_subtree


def lmd_rule_first_seen_and_strictly_later(mk):
    """K5 (pe:1383, pe:1440): a vote is replaced only by a strictly later target epoch; among equals the first
    seen stays."""
    w = mk(32)
    anchor = w.store.justified_checkpoint.root
    w.tick_to_slot(1, offset=spec.SECONDS_PER_SLOT - 1)
    a = w.block(anchor, 1, graffiti=b"a")
    b = w.block(anchor, 1, graffiti=b"b")
    w.tick_to_slot(3)
    voters = slot_committee_members(w.store, 2)
    w.vote(voters, a, 2)
    assert all(w.store.latest_messages[v].root == a for v in voters)
    w.vote(voters, b, 2)  # same epoch: ignored
    assert all(w.store.latest_messages[v].root == a for v in voters)
    # next epoch: strictly later -> replaced
    s2 = spec.SLOTS_PER_EPOCH + 2
    w.tick_to_slot(s2 + 1)
    voters2 = slot_committee_members(w.store, s2)
    w.vote(voters2, b, s2)
    assert all(w.store.latest_messages[v] == spec.LatestMessage(1, b) for v in voters2)


def proposer_boost_correct_head(mk):
    """upstream test_proposer_boost_correct_head + K6 arithmetic (pe:1385-1399): boost = 70 % of one slot's
    committee weight W; 80 votes beat it, 60 do not."""
    w = mk(800, PROPOSER_SCORE_BOOST=70)  # minimal preset: 8 slots -> W = 100 validators x 32 ETH
    anchor = w.store.justified_checkpoint.root
    W = 100 * 32 * ETH
    w.tick_to_slot(1, offset=spec.SECONDS_PER_SLOT - 1)
    left = w.block(anchor, 1, graffiti=b"L")
    w.tick_to_slot(2)
    voters = slot_committee_members(w.store, 1)
    assert len(voters) == 100
    w.vote(voters[:60], left, 1)
    assert spec.get_latest_attesting_balance(w.store, left) == 60 * 32 * ETH
    # timely competing block at slot 2 gets the boost: 0.7 W = 70 votes' worth > 60
    right = w.block(anchor, 2, graffiti=b"R")
    assert w.store.proposer_boost_root == right
    assert spec.get_latest_attesting_balance(w.store, right) == W * 70 // 100
    assert w.head() == right
    # 20 more votes for L: 80 > 70 (K6: "L sees 80 v 0" vs the 70 boost)
    w.vote(voters[60:80], left, 1)
    assert w.head() == left
    # boost also lifts every ancestor of the boosted block and vanishes at the next slot (pe:943-944)
    assert spec.get_latest_attesting_balance(w.store, anchor) == 80 * 32 * ETH + W * 70 // 100
    w.tick_to_slot(3)
    assert w.store.proposer_boost_root == spec.Root()
    assert spec.get_latest_attesting_balance(w.store, right) == 0
    assert w.head() == left


def ex_ante_reorg_arithmetic(mk):
    """K7 (pe:1525-1526): boost 0.8 W, adversary 7 % per slot: 7 + 7 + 80 = 94 > 93 honest votes."""
    w = mk(800, PROPOSER_SCORE_BOOST=80)
    anchor = w.store.justified_checkpoint.root
    w.tick_to_slot(1, offset=spec.SECONDS_PER_SLOT - 1)
    honest = w.block(anchor, 1, graffiti=b"n+1")          # block n+1, gets 93 honest votes of slot 1
    private = w.block(anchor, 1, graffiti=b"adv")          # adversary's withheld sibling
    w.tick_to_slot(2)
    c1 = slot_committee_members(w.store, 1)
    w.vote(c1[:93], honest, 1)
    w.vote(c1[93:100], private, 1)                          # 7 adversarial votes of slot 1
    assert w.head() == honest
    c2 = slot_committee_members(w.store, 2)
    w.tick_to_slot(3)
    w.vote(c2[:7], private, 2)                              # 7 adversarial votes of slot 2
    # timely block n+3 on top of the private block: + 0.8 W boost
    child = w.block(private, 3, graffiti=b"n+3")
    assert w.store.proposer_boost_root == child
    assert spec.get_latest_attesting_balance(w.store, private) == (7 + 7 + 80) * 32 * ETH
    assert spec.get_latest_attesting_balance(w.store, honest) == 93 * 32 * ETH
    assert w.head() == child


def discard_equivocations(mk):
    """K8 (pe:1411-1413, pe:1438) / upstream test_discard_equivocations: equivocators carry no weight,
    present and future."""
    w = mk(64)
    anchor = w.store.justified_checkpoint.root
    w.tick_to_slot(1, offset=spec.SECONDS_PER_SLOT - 1)
    a = w.block(anchor, 1, graffiti=b"a")
    b = w.block(anchor, 1, graffiti=b"b")
    w.tick_to_slot(3)
    voters = slot_committee_members(w.store, 2)
    w.vote(voters, a, 2)
    assert spec.get_latest_attesting_balance(w.store, a) == len(voters) * 32 * ETH
    # slashable pair: same target epoch, different data (double vote)
    tgt = spec.Checkpoint(0, anchor)
    d1 = spec.AttestationData(slot=2, index=0, beacon_block_root=a, source=spec.Checkpoint(), target=tgt)
    d2 = spec.AttestationData(slot=2, index=0, beacon_block_root=b, source=spec.Checkpoint(), target=tgt)
    eq = sorted(voters[: max(1, len(voters) // 2)])
    w.slash(spec.IndexedAttestation(eq, d1), spec.IndexedAttestation(eq, d2))
    assert w.store.equivocating_indices == set(eq)
    assert spec.get_latest_attesting_balance(w.store, a) == (len(voters) - len(eq)) * 32 * ETH
    # future votes of equivocators are ignored too
    s2 = spec.SLOTS_PER_EPOCH + 2
    w.tick_to_slot(s2 + 1)
    for att in w.attestation_for(eq, b, s2):
        w.attest(att)
    assert all(w.store.latest_messages[v].root == a for v in eq if v in w.store.latest_messages)
    # not slashable -> rejected, store untouched (K9)
    w.slash(spec.IndexedAttestation(eq, d1), spec.IndexedAttestation(eq, d1), expect_fail=True)
    # unsorted indices -> invalid indexed attestation
    w.slash(spec.IndexedAttestation(eq[::-1] if len(eq) > 1 else [], d1), spec.IndexedAttestation(eq, d2),
            expect_fail=True)


def invalid_handlers_leave_store_untouched(mk):
    """K9 (pe:1041): every failing assert path of on_attestation / on_block."""
    w = mk(64)
    anchor = w.store.justified_checkpoint.root
    w.tick_to_slot(1)
    b1 = w.block(anchor, 1)
    # block from the future, unknown parent
    w.block(b1, 5, expect_fail=True)
    w.block(spec.Root(b"\x11" * 32), 1, expect_fail=True)
    w.tick_to_slot(2)
    voters = slot_committee_members(w.store, 1)
    good = w.attestation_for(voters, b1, 1)[0]
    # attestation for the current slot: not yet in the past
    w.tick_to_slot(1)
    import dataclasses
    # unknown block root
    bad = dataclasses.replace(good, data=dataclasses.replace(good.data, beacon_block_root=spec.Root(b"\x22" * 32)))
    w.attest(bad, expect_fail=True)
    # unknown target root
    bad = dataclasses.replace(good, data=dataclasses.replace(good.data, target=spec.Checkpoint(0, spec.Root(b"\x33" * 32))))
    w.attest(bad, expect_fail=True)
    # target epoch does not match the slot
    bad = dataclasses.replace(good, data=dataclasses.replace(good.data, target=spec.Checkpoint(1, anchor)))
    w.attest(bad, expect_fail=True)
    # block newer than the attestation slot
    w.tick_to_slot(3)
    b3 = w.block(b1, 3)
    bad = dataclasses.replace(good, data=dataclasses.replace(good.data, beacon_block_root=b3))
    w.attest(bad, expect_fail=True)
    # empty bits, bad signature, short bitlist
    w.attest(dataclasses.replace(good, aggregation_bits=[False] * len(good.aggregation_bits)), expect_fail=True)
    w.attest(dataclasses.replace(good, signature_valid=False), expect_fail=True)
    if len(good.aggregation_bits) > 1:
        w.attest(dataclasses.replace(good, aggregation_bits=good.aggregation_bits[:-1]), expect_fail=True)
    # too old for the wire (two epochs back) but fine from a block (pe:1423)
    w.tick_to_slot(3 * spec.SLOTS_PER_EPOCH)
    w.attest(good, expect_fail=True)
    w.attest(good, is_from_block=True, expect_fail=False)
    assert len(w.store.latest_messages) == sum(good.aggregation_bits)
    assert all(m.root == b1 for m in w.store.latest_messages.values())


def filtered_block_tree(mk):
    """upstream test_filtered_block_tree (A.3, pe:1121-1123): a heavier branch whose leaf disagrees with the
    store's justified checkpoint is not viable; votes on it still count for its viable ancestors."""
    w = mk(64)
    anchor = w.store.justified_checkpoint.root
    spe = spec.SLOTS_PER_EPOCH
    w.tick_to_slot(1)
    b1 = w.block(anchor, 1)
    just = spec.Checkpoint(1, b1)
    # a block whose post-state justifies (1, b1): the store adopts it (early in the epoch, pe:1054)
    w.tick_to_slot(spe)
    good = w.block(b1, spe, scripted=(just, spec.Checkpoint(0, anchor)), graffiti=b"good")
    assert w.store.justified_checkpoint == just
    # sibling branch that never saw the justification
    w.tick_to_slot(spe + 1, offset=spec.SECONDS_PER_SLOT - 1)
    stale = w.block(b1, spe + 1, graffiti=b"stale")
    w.tick_to_slot(spe + 3)
    voters = slot_committee_members(w.store, spe + 2)
    w.vote(voters, stale, spe + 2)
    assert spec.get_latest_attesting_balance(w.store, stale) > spec.get_latest_attesting_balance(w.store, good)
    assert w.head() == good                      # stale is filtered out
    # the votes on the non-viable branch still add weight to the common ancestor
    assert spec.get_latest_attesting_balance(w.store, b1) == len(voters) * 32 * ETH


def justified_checkpoint_promotion_on_tick(mk):
    """on_tick (pe:951-955) + should_update_justified_checkpoint (pe:1046-1061): a better justified checkpoint
    seen late in an epoch waits in best_justified_checkpoint until the next epoch boundary."""
    w = mk(64)
    anchor = w.store.justified_checkpoint.root
    spe = spec.SLOTS_PER_EPOCH
    w.tick_to_slot(1)
    b1 = w.block(anchor, 1)
    w.tick_to_slot(2)
    side = w.block(anchor, 2, graffiti=b"side")
    just = spec.Checkpoint(1, side)            # conflicts with ... nothing yet: justified is still the anchor
    late = spe + spec.SAFE_SLOTS_TO_UPDATE_JUSTIFIED + 1
    w.tick_to_slot(late)
    w.block(side, late, scripted=(just, spec.Checkpoint(0, anchor)))
    # anchor is an ancestor of `side`, so the update is allowed even late (pe:1057-1059)
    assert w.store.justified_checkpoint == just
    just2 = spec.Checkpoint(2, b1)             # conflicting chain, seen late: must wait
    w.tick_to_slot(2 * spe + spec.SAFE_SLOTS_TO_UPDATE_JUSTIFIED + 1)
    w.block(b1, 2 * spe + spec.SAFE_SLOTS_TO_UPDATE_JUSTIFIED + 1, scripted=(just2, spec.Checkpoint(0, anchor)))
    assert w.store.best_justified_checkpoint == just2
    assert w.store.justified_checkpoint == just
    w.tick_to_slot(3 * spe)                    # epoch boundary: promoted
    assert w.store.justified_checkpoint == just2


def justified_state_balances_follow_the_checkpoint(mk):
    """get_latest_attesting_balance weighs votes with checkpoint_states[store.justified_checkpoint] (Appendix A.1):
    when on_block moves the justified checkpoint, effective balances and activity of the NEW justified state count.
    (the engine holds one registry view; the mirror must be handed the new state -- World.check does,
    and pos_evolution_amd.forkchoice.get_head refuses to run on the old one.)"""
    w = mk(64)
    anchor = w.store.justified_checkpoint.root
    spe = spec.SLOTS_PER_EPOCH
    w.tick_to_slot(1)
    b1 = w.block(anchor, 1)
    # the chain's state at b1 differs from genesis: half the validators at 16 ETH, a few exited before epoch 1
    st = w.store.block_states[b1]
    for i, v in enumerate(st.validators):
        if i % 2 == 0:
            v.effective_balance = 16 * ETH
        if i % 7 == 3:
            v.exit_epoch = 1
    w.tick_to_slot(spe)
    just = spec.Checkpoint(1, b1)
    good = w.block(b1, spe, scripted=(just, spec.Checkpoint(0, anchor)), graffiti=b"good")
    assert w.store.justified_checkpoint == just
    w.tick_to_slot(spe + 1, offset=spec.SECONDS_PER_SLOT - 1)
    other = w.block(b1, spe + 1, scripted=(just, spec.Checkpoint(0, anchor)), graffiti=b"other")
    w.tick_to_slot(spe + 3)
    voters = slot_committee_members(w.store, spe + 2)   # epoch-1 committees: from the justified state's active set
    jstate = w.store.checkpoint_states[just]
    assert all(spec.is_active_validator(jstate.validators[v], 1) for v in voters)
    w.vote(voters, other, spe + 2)
    want = sum(jstate.validators[v].effective_balance for v in voters)
    assert want != len(voters) * 32 * ETH
    assert spec.get_latest_attesting_balance(w.store, other) == want
    assert w.head() == other


def rlmd_ghost_vote_expiry(mk):
    """[VARIANT pe:1585-1596, pe:1549] vote expiry period eta: only latest messages from the most recent eta slots
    count.  A heavier but stale branch loses to a lighter fresh one; eta = 0 is the reference's LMD-GHOST."""
    late = spec.SECONDS_PER_SLOT - 1
    for eta in (0, 1, 2, 5):
        w = mk(64, VOTE_EXPIRY_SLOTS=eta)
        anchor = w.store.justified_checkpoint.root
        w.tick_to_slot(1, offset=late)                       # late blocks: no proposer boost
        a = w.block(anchor, 1, graffiti=b"a")
        b = w.block(anchor, 1, graffiti=b"b")
        w.tick_to_slot(2)
        stale = slot_committee_members(w.store, 1)
        w.vote(stale, a, 1)
        assert w.head() == a                                 # slot 2: slot-1 votes are inside every window
        w.tick_to_slot(4)
        fresh = slot_committee_members(w.store, 3)
        fresh = fresh[: len(fresh) // 2]
        assert 0 < len(fresh) < len(stale)
        w.vote(fresh, b, 3)
        # slot 4: eta in (1, 2) has dropped the slot-1 votes (1 + eta < 4); eta = 5 and LMD-GHOST keep them
        assert w.head() == (b if eta in (1, 2) else a)
        assert spec.get_latest_attesting_balance(w.store, a) == (0 if eta in (1, 2) else len(stale) * 32 * ETH)
        w.tick_to_slot(7)
        # slot 7: eta = 5 now drops slot 1 (1 + 5 < 7) but keeps slot 3; eta in (1, 2) has nothing left -> root tie-break
        assert w.head() == {0: a, 5: b}.get(eta, max(a, b))
    spec.use_preset(spec.PRESET_NAME)                        # leave the module constants at eta = 0


ALL = [
    genesis_head, chain_no_attestations, split_tie_breaker_no_attestations, shorter_chain_but_heavier_weight,
    lmd_walkthrough_five_validators, lmd_rule_first_seen_and_strictly_later, proposer_boost_correct_head,
    ex_ante_reorg_arithmetic, discard_equivocations, invalid_handlers_leave_store_untouched, filtered_block_tree,
    justified_checkpoint_promotion_on_tick, justified_state_balances_follow_the_checkpoint, rlmd_ghost_vote_expiry,
]
