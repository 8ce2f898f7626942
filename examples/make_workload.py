"""Writes the workload file examples/step_client.c reads: a synthetic registry, a block tree, and per step one epoch's
committee table (the reference's swap-or-not shuffle, run on the GPU) and attestations.  Needs the GPU (the registry's
pubkeys and the committee tables are made by the engine).

    python examples/make_workload.py /tmp/workload.bin --validators 1048576 --committees 2048 --blocks 4096 --steps 40
"""
import argparse
import hashlib
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MAGIC = 0x30764F5645534F50  # "POSEVOv0"


def build(validators: int, committees: int, blocks: int, steps: int, parts: int = 4, density: float = 0.99, rounds: int = 90):
    """-> (engine with the store loaded, dict of everything the file holds)."""
    import pos_evolution_amd as pea
    import pos_evolution_amd.synth as synth
    from pos_evolution_amd._abi import pe_state_ctx

    spe = 32
    e = pea.Engine(max_committee_tables=steps + 1)
    tree = synth.random_tree(blocks, 4, "bushy")
    e.store_init(0, 0, tree.roots[0].tobytes())
    for i in range(1, blocks):
        e.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
    bal = synth.balances(validators, 4, mixed=False)
    flags = synth.validator_flags(validators, 4, inactive_frac=0.005)
    pts = synth.registry_points(e, validators)
    e.set_validators(bal, flags, pts)
    epoch0 = int(tree.slot.max()) // spe + 1
    out = []
    for s in range(steps):
        ep = epoch0 + s
        seed = hashlib.sha256(b"client-seed" + ep.to_bytes(8, "little")).digest()
        off, mem = e.compute_committees(ep, seed, validators, committees, rounds)
        comm = synth.Committees(off, mem)
        atts, arena, _ = synth.epoch_attestations(comm, tree, ep, spe, seed=4, density=density, parts=parts,
                                                  source=(0, tree.roots[0].tobytes()), vote_recent=64, vote_seed=4)
        ctx = pe_state_ctx()
        ctx.slot = (ep + 1) * spe
        ctx.chain_tip_root[:] = tree.roots[blocks - 1].tobytes()
        ctx.current_justified_root[:] = tree.roots[0].tobytes()
        ctx.previous_justified_root[:] = tree.roots[0].tobytes()
        ctx.base_reward_per_increment = 2264
        out.append(dict(epoch=ep, tick=(ep + 1) * spe * 12, ctx=ctx, offsets=off, members=mem, atts=atts, arena=arena))
    return e, dict(spe=spe, tree=tree, bal=bal, flags=flags, pts=pts, steps=out)


def write(path: str, w: dict):
    tree, steps = w["tree"], w["steps"]
    n_val, n_blocks = w["bal"].size, tree.roots.shape[0]
    n_comm = steps[0]["offsets"].size - 1
    with open(path, "wb") as f:
        # steps that carry "sigs" ((n_atts, 96) uint8: one compressed BLSSignature per attestation) set header flag 1; the
        # client then runs pe_aggregate_signed and hands back every aggregate's compressed signature
        has_sigs = all("sigs" in st for st in steps)
        f.write(struct.pack("<8Q", MAGIC, n_val, n_comm, n_blocks, len(steps), w["spe"], 0, 1 if has_sigs else 0))
        f.write(np.ascontiguousarray(tree.roots, dtype=np.uint8).tobytes())
        f.write(np.ascontiguousarray(tree.parent, dtype=np.uint32).tobytes())
        f.write(np.ascontiguousarray(tree.slot, dtype=np.uint64).tobytes())
        f.write(np.ascontiguousarray(w["bal"], dtype=np.uint64).tobytes())
        f.write(np.ascontiguousarray(w["flags"], dtype=np.uint8).tobytes())
        f.write(np.ascontiguousarray(w["pts"], dtype=np.uint8).tobytes())
        for st in steps:
            f.write(struct.pack("<4Q", st["epoch"], len(st["atts"]), st["arena"].size, st["tick"]))
            f.write(bytes(st["ctx"]))
            f.write(np.ascontiguousarray(st["offsets"], dtype=np.uint32).tobytes())
            f.write(np.ascontiguousarray(st["members"], dtype=np.uint32).tobytes())
            f.write(np.ascontiguousarray(st["atts"]).tobytes())
            f.write(np.ascontiguousarray(st["arena"], dtype=np.uint8).tobytes())
            if has_sigs:
                assert st["sigs"].shape == (len(st["atts"]), 96)
                f.write(np.ascontiguousarray(st["sigs"], dtype=np.uint8).tobytes())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--validators", type=int, default=1 << 20)
    ap.add_argument("--committees", type=int, default=2048)
    ap.add_argument("--blocks", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    _, w = build(a.validators, a.committees, a.blocks, a.steps)
    write(a.path, w)
    print(f"{a.path}: {os.path.getsize(a.path) / 1e6:.1f} MB, {a.steps} steps of {len(w['steps'][0]['atts'])} attestations")
