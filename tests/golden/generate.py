#!/usr/bin/env python
"""Regenerates the fixtures in tests/golden/ by RUNNING THE REFERENCE'S OWN pyspec text.

``oracle/ref_extract.py`` pulls the fenced Python out of /root/reference/pos-evolution.md and executes it inside
``oracle.spec``'s namespace (``spec.ORACLE_OF_RECORD == "reference"``): compute_shuffled_index / compute_committee
(pe:495-534), get_head (pe:1102-1116), on_tick / on_block / on_attestation (pe:934-1036, pe:1423), update_latest_messages
(pe:1435-1441) below are the reference's code objects; only the callees the reference never defines
(get_ancestor, get_latest_attesting_balance, get_filtered_block_tree, validate_on_attestation, get_beacon_committee:
SURVEY.md Appendix A) and the G1/G2 arithmetic (oracle/g1.py, oracle/g2.py: no BLS code in the reference at all) are
this repo's restatements.  The script refuses to write fork-choice / shuffle fixtures from the transcription.

    python tests/golden/generate.py        # rewrites the *.json next to this script (needs /root/reference)
    python tests/golden/generate.py --digest   # prints sha256 of the fork-choice trace + shuffle vectors it would write
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

from oracle import g1, g2, spec  # noqa: E402
from tests.scenario import new_world, slot_committee_members  # noqa: E402


def g1_vectors():
    out = {"generator_compressed": g1.compress(g1.G).hex(), "scalar_mul": [], "subset_sums": []}
    for k in [1, 2, 3, 4, 5, 7, 0xDEADBEEF, 2**64 + 1, g1.R_ORDER - 1]:
        out["scalar_mul"].append({"k": hex(k), "compressed": g1.compress(g1.mul(k, g1.G)).hex(),
                                  "uncompressed": g1.to_bytes96(g1.mul(k, g1.G)).hex()})
    a, b = 0x1234567, 0x89ABCDE
    rng = np.random.default_rng(11)
    for size in (0, 1, 2, 17, 300):
        idx = sorted(int(i) for i in rng.choice(100000, size=size, replace=False))
        k = (len(idx) * a + sum(idx) * b) % g1.R_ORDER
        out["subset_sums"].append({"a": hex(a), "b": hex(b), "indices": idx,
                                   "sum_uncompressed": g1.to_bytes96(g1.mul(k, g1.G)).hex()})
    return out


def g2_vectors():
    """G2 sums (bls.Aggregate over signature points): frozen oracle/g2.py answers, incl. the edge cases."""
    out = {"generator_compressed": g2.compress(g2.G2).hex(), "two_g_compressed": g2.compress(g2.double(g2.G2)).hex(),
           "sums": []}
    A = g2.mul(7, g2.G2)
    prog = g2.synthetic_points(24, 1, 1)          # (i+1)*G2: 1G + 2G, + 3G doubles
    rnd = g2.synthetic_points(40, 0x1234567, 0x89ABCDE)
    cases = [("empty", []), ("single", [A]), ("doubling", [g2.G2, g2.G2]), ("cancel", [A, g2.neg(A)]),
             ("with_infinity", [None, A, None]), ("progression_24", prog), ("random_40", rnd),
             ("cancel_then_more", [A, g2.neg(A), rnd[0], rnd[1]])]
    for name, pts in cases:
        out["sums"].append({"name": name, "points": [g2.to_bytes192(p).hex() for p in pts],
                            "sum": g2.to_bytes192(g2.sum_points(pts)).hex()})
    return out


def shuffle_vectors():
    out = []
    for preset in ("minimal", "mainnet"):
        spec.use_preset(preset)
        for n in (1, 2, 7, 100, 333):
            seed = spec.sha256(f"{preset}-{n}".encode())
            out.append({"preset": preset, "rounds": spec.SHUFFLE_ROUND_COUNT, "index_count": n, "seed": seed.hex(),
                        "shuffled": [spec.compute_shuffled_index(i, n, seed) for i in range(n)]})
    spec.use_preset("mainnet")
    return out


def lm_digest(latest_messages) -> str:
    """sha256 over the sorted (validator, epoch, root) triples of a latest-message table."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(latest_messages):
        m = latest_messages[k]
        h.update(int(k).to_bytes(8, "little") + int(m.epoch).to_bytes(8, "little") + bytes(m.root))
    return h.hexdigest()


def forkchoice_trace(seed=5, steps=40, n_val=96):
    """A random event stream on the minimal preset with the oracle's answer after every event."""
    rng = np.random.default_rng(seed)
    w = new_world(n_val, "minimal")
    events = []
    anchor = w.store.justified_checkpoint.root
    roots = [anchor]
    slot = 0

    tables = {}

    def snap():
        return {"head": bytes(spec.get_head(w.store)).hex(), "lm_digest": lm_digest(w.store.latest_messages),
                "n_messages": len(w.store.latest_messages), "boost": bytes(w.store.proposer_boost_root).hex()}

    for step in range(steps):
        slot += int(rng.integers(1, 3))
        t = slot * spec.SECONDS_PER_SLOT + int(rng.integers(0, spec.SECONDS_PER_SLOT))
        w.tick(t)
        events.append({"op": "tick", "time": t, "expect": snap()})
        parent = roots[int(rng.integers(max(0, len(roots) - 4), len(roots)))]
        if w.store.blocks[parent].slot < slot:
            r = w.block(parent, slot, graffiti=bytes([step]))
            roots.append(r)
            events.append({"op": "block", "root": bytes(r).hex(), "parent": bytes(parent).hex(), "slot": slot,
                           "expect": snap()})
        a_slot = slot - 1
        cand = [r for r in roots[-6:] if w.store.blocks[r].slot <= a_slot]
        if cand:
            blk = cand[int(rng.integers(0, len(cand)))]
            voters = slot_committee_members(w.store, a_slot)
            rng.shuffle(voters)
            for att in w.attestation_for(voters[: max(1, len(voters) // 2)], blk, a_slot):
                ok = w.attest(att)
                st = w.store.checkpoint_states[att.data.target]
                ep = att.data.target.epoch
                if ep not in tables:
                    cps = spec.get_committee_count_per_slot(st, ep)
                    tables[ep] = [spec.get_beacon_committee(st, ep * spec.SLOTS_PER_EPOCH + s_, i_)
                                  for s_ in range(spec.SLOTS_PER_EPOCH) for i_ in range(cps)]
                events.append({"op": "attestation", "slot": att.data.slot, "index": att.data.index,
                               "beacon_block_root": bytes(att.data.beacon_block_root).hex(),
                               "target_epoch": att.data.target.epoch, "target_root": bytes(att.data.target.root).hex(),
                               "bits": [int(b) for b in att.aggregation_bits], "accepted": ok,
                               "expect": snap()})
    return {"preset": "minimal", "n_validators": n_val, "anchor_root": bytes(anchor).hex(),
            "committees_epoch": {str(k): v for k, v in tables.items()}, "events": events}


def ref_pins():
    from oracle import ref_extract
    return ref_extract.pins_from_reference()


def digest():
    import hashlib
    blob = json.dumps({"trace": forkchoice_trace(), "shuffle": shuffle_vectors()}, sort_keys=True).encode()
    return hashlib.sha256(blob).hexdigest()


if __name__ == "__main__":
    if "--digest" in sys.argv:
        print(spec.ORACLE_OF_RECORD, digest())
        sys.exit(0)
    assert spec.ORACLE_OF_RECORD == "reference", "fixtures are generated from the reference's own text only"
    json.dump(ref_pins(), open(os.path.join(HERE, "ref_pins.json"), "w"), indent=1)
    json.dump(g1_vectors(), open(os.path.join(HERE, "g1_vectors.json"), "w"), indent=1)
    json.dump(g2_vectors(), open(os.path.join(HERE, "g2_vectors.json"), "w"), indent=1)
    json.dump(shuffle_vectors(), open(os.path.join(HERE, "shuffle_vectors.json"), "w"))
    json.dump(forkchoice_trace(), open(os.path.join(HERE, "forkchoice_trace.json"), "w"))
    print("golden fixtures written")
