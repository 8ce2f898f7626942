"""-m gpu: committee computation on the GPU (pe_compute_committees) against the reference's own functions
compute_committee (pe:495-504) / compute_shuffled_index (pe:513-534) as transcribed in the L0 oracle -- this part
of the path is defined verbatim by the reference, so parity here is pinned by the reference text."""
import json
import os

import numpy as np
import pytest

from oracle import spec

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("preset,n_active,n_val,cps", [
    ("minimal", 1, 10, 1), ("minimal", 2, 10, 1), ("minimal", 33, 64, 2), ("minimal", 257, 300, 4),
    ("minimal", 5000, 5000, 4), ("mainnet", 100, 128, 1), ("mainnet", 2500, 4000, 2),
])
def test_compute_committees_vs_reference_functions(engine_factory, preset, n_active, n_val, cps):
    spec.use_preset(preset)
    try:
        e = engine_factory(slots_per_epoch=spec.SLOTS_PER_EPOCH)
        e.set_validators(np.full(n_val, 32 * 10**9, dtype=np.uint64), np.full(n_val, 1, dtype=np.uint8))
        rng = np.random.default_rng(n_active)
        active = np.sort(rng.choice(n_val, size=n_active, replace=False)).astype(np.uint32)  # increasing, like the spec
        seed = spec.sha256(f"seed-{preset}-{n_active}".encode())
        count = cps * spec.SLOTS_PER_EPOCH
        off, mem = e.compute_committees(7, seed, active, count, spec.SHUFFLE_ROUND_COUNT)
        for c in range(count):
            want = spec.compute_committee([int(x) for x in active], seed, c, count)
            assert list(mem[off[c]:off[c + 1]]) == want, (c, want)
        assert off[-1] == n_active
    finally:
        spec.use_preset("mainnet")


def test_golden_shuffle_vectors(engine_factory):
    for row in json.load(open(os.path.join(HERE, "golden", "shuffle_vectors.json"))):
        n = row["index_count"]
        e = engine_factory()
        e.set_validators(np.full(n, 32 * 10**9, dtype=np.uint64), np.full(n, 1, dtype=np.uint8))
        _, mem = e.compute_committees(1, bytes.fromhex(row["seed"]), np.arange(n, dtype=np.uint32), 32, row["rounds"])
        assert list(mem) == row["shuffled"]


def test_million_validator_epoch_is_a_partition_and_drives_on_attestation(engine_factory):
    import time
    import pos_evolution_amd.synth as synth
    from tests import helpers as H
    n = 1 << 20
    e = engine_factory()
    tree = synth.random_tree(40, 3, "branchy")
    H.load_tree(e, tree)
    e.set_validators(synth.balances(n, 3), synth.validator_flags(n, 3))
    seed = spec.sha256(b"million")
    epoch = int(tree.slot.max()) // 32 + 1
    t0 = time.perf_counter()
    off, mem = e.compute_committees(epoch, seed, np.arange(n, dtype=np.uint32), 2048, 90)
    dt = time.perf_counter() - t0
    print(f"\n1M-validator epoch shuffle (90 rounds) + table registration: {dt * 1e3:.2f} ms")
    assert np.array_equal(np.sort(mem), np.arange(n, dtype=np.uint32))           # a permutation
    assert np.array_equal(np.diff(off.astype(np.int64)), np.full(2048, 512))     # K2-style slice sizes
    # spot-check 64 positions against the literal per-index function
    for i in np.random.default_rng(0).integers(0, n, size=64):
        assert mem[i] == spec.compute_shuffled_index(int(i), n, seed)
    # the GPU-computed table is live: attestations resolve against it
    comm = synth.Committees(off, mem)
    atts, arena, _ = synth.epoch_attestations(comm, tree, epoch, 32, seed=4, density=0.5, parts=1, from_block=True)
    e.on_tick((epoch + 2) * 32 * 12)
    status, _, count = e.on_attestation_batch(packed=(atts, arena))
    assert (status == 0).all()
    _, blk = e.latest_messages()
    assert (blk != 0xFFFFFFFF).sum() == count.sum()


def test_compute_committees_identity_active_set_and_no_readback(engine_factory):
    """active_indices = n ("validators 0 .. n - 1 are active") must give the table of the explicit index array, and
    want_result=False keeps it on the device: a following aggregation resolves against it."""
    import pos_evolution_amd as pea
    n, count, rounds = 30000, 64, 90
    seed = spec.sha256(b"identity")
    a = engine_factory()
    b = engine_factory()
    for e in (a, b):
        e.set_validators(np.full(n, 32 * 10**9, dtype=np.uint64), np.ones(n, dtype=np.uint8))
    off_a, mem_a = a.compute_committees(3, seed, np.arange(n, dtype=np.uint32), count, rounds)
    off_b, mem_b = b.compute_committees(3, seed, n, count, rounds)
    assert np.array_equal(off_a, off_b) and np.array_equal(mem_a, mem_b)
    assert sorted(mem_b.tolist()) == list(range(n))
    assert b.compute_committees(4, seed, n, count, rounds, want_result=False) is None
    # an unsorted or duplicated active set is still validated
    with pytest.raises(pea.EngineError):
        a.compute_committees(5, seed, np.array([1, 1, 2], dtype=np.uint32), 32, rounds)
    perm = np.random.default_rng(1).permutation(n).astype(np.uint32)
    off_p, mem_p = a.compute_committees(6, seed, perm, count, rounds)
    assert np.array_equal(mem_p, perm[mem_a])      # members[i] = indices[shuffled(i)]
