"""Shared builders for the differential tests (engine vs oracle on identical seeded inputs)."""
import numpy as np

import pos_evolution_amd.synth as synth
from oracle import cport, g1

NONE32 = 0xFFFFFFFF
ZERO = bytes(32)


def load_tree(engine, tree, leaf_cp=None, genesis_time=0):
    """store_init + add_block for a synthetic tree; leaf_cp[i] = (justified, finalized) checkpoint tuples."""
    engine.store_init(genesis_time, int(tree.slot[0]), tree.roots[0].tobytes())
    for i in range(1, tree.roots.shape[0]):
        j, f = leaf_cp[i] if leaf_cp is not None else ((0, tree.roots[0].tobytes()), (0, tree.roots[0].tobytes()))
        engine.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]), j, f)


def oracle_points(n, a=0x1234567, b=0x89ABCDE):
    """(n, 96) u8 with P_i = A + i*B (closed-form sums)."""
    A = g1.mul(a, g1.G)
    B = g1.mul(b, g1.G)
    return cport.g1_arith_progression(g1.to_bytes96(A), g1.to_bytes96(B), n), (a, b)


def closed_form_sum(indices, a, b):
    """sum_{i in S} (A + i*B) = (|S|*a + (sum i)*b) * G."""
    idx = [int(i) for i in indices]
    k = (len(idx) * a + sum(idx) * b) % g1.R_ORDER
    return g1.to_bytes96(g1.mul(k, g1.G))


def att_device_rows(atts, comm, slots_per_epoch):
    """Flat per-attestation arrays the C oracle consumes, from ATT_DTYPE rows + a committee table."""
    n_comm = comm.offsets.size - 1
    cps = n_comm // slots_per_epoch
    pos = (atts["slot"] % slots_per_epoch) * cps + atts["index"]
    member_off = comm.offsets[pos.astype(np.int64)]
    return member_off.astype(np.uint32), atts["n_bits"].astype(np.uint32), atts["bits_offset"].astype(np.uint32)
