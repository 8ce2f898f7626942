"""-m gpu, opt-in (POSEVO_TEST_S29=1): the accumulation kernel over the S29 field form (POSEVO_G1_S29=1,
k_g1_accumulate_s29 in g1_kernels.hip; the field and point arithmetic itself is held against Python integers and
oracle/g1.py on the CPU by tests/test_host_fp29.py) must give the aggregate pubkeys the default kernel gives -- random
keys, the structured keys (i + 1) G that hit the doubling branch, P / -P pairs, rows that hold no point.

Opt-in because the kernel was written after the round's GPU budget was spent: it has compiled for gfx950 (195 VGPRs, no
scratch, 5050 VALU instructions per mixed add against 6697 + 500 s_nop) but has not run on hardware yet.  Run with
    POSEVO_TEST_S29=1 python -m pytest tests/test_gpu_g1_s29.py -m gpu -q
and make it unconditional once it is green."""
import os

import numpy as np
import pytest

import pos_evolution_amd as pea
import pos_evolution_amd.synth as synth
from oracle import g1
from tests import helpers as H

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.environ.get("POSEVO_TEST_S29"), reason="opt-in: POSEVO_TEST_S29=1 (see the docstring)")]


def _engine(s29, **cfg):
    old = os.environ.get("POSEVO_G1_S29")
    os.environ["POSEVO_G1_S29"] = "1" if s29 else "0"
    try:
        return pea.Engine(**cfg)           # the knob is read per engine, at creation
    finally:
        if old is None:
            del os.environ["POSEVO_G1_S29"]
        else:
            os.environ["POSEVO_G1_S29"] = old


def _aggregate(e, tree, bal, flags, pts, comm, atts, arena, epoch):
    H.load_tree(e, tree)
    e.set_validators(bal, flags, pts)
    e.set_committees(epoch, comm.offsets, comm.members)
    e.on_tick((epoch + 1) * 32 * 12)
    sync = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    with e.pipeline():
        piped = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    assert np.array_equal(sync["aggpk96"], piped["aggpk96"])
    return sync


@pytest.mark.parametrize("n_val,n_comm,density", [(4096, 32, 0.9), (30000, 64, 0.5), (20000, 32, 1.0)])
def test_aggregate_pubkeys_equal_the_default_kernel_and_the_closed_form(n_val, n_comm, density):
    tree = synth.random_tree(80, 7, "bushy")
    pts, (a, b) = H.oracle_points(n_val)
    bal = synth.balances(n_val, 7, mixed=True)
    flags = synth.validator_flags(n_val, 7, inactive_frac=0.01)
    comm = synth.random_committees(n_val, n_comm, 7)
    epoch = int(tree.slot.max()) // 32 + 1
    atts, arena, bit_rows = synth.epoch_attestations(comm, tree, epoch, 32, seed=7, density=density, parts=2)
    ref = _aggregate(_engine(False), tree, bal, flags, pts, comm, atts, arena, epoch)
    got = _aggregate(_engine(True), tree, bal, flags, pts, comm, atts, arena, epoch)
    assert got["n_groups"] == ref["n_groups"] and np.array_equal(got["aggpk96"], ref["aggpk96"])
    assert np.array_equal(got["out_arena"], ref["out_arena"])
    # and both against the closed form of the synthetic registry, through the union bits
    spe, cps = 32, n_comm // 32
    for k in range(got["n_groups"]):
        row = got["atts"][k]
        c = int((row["slot"] % spe) * cps + row["index"])
        members = comm.members[comm.offsets[c]:comm.offsets[c + 1]]
        idx = members[np.asarray(got["bits"][k], dtype=bool)]
        assert got["aggpk96"][k].tobytes() == H.closed_form_sum(idx, a, b)


def test_structured_keys_doubling_negatives_and_empty_rows():
    """(i + 1) G keys in index order (accumulator meets an equal point), P followed by -P (accumulator returns to
    infinity and goes on), validators without a key (all-zero rows)."""
    n = 512
    keys = [g1.mul(i + 1, g1.G) for i in range(n // 4)]
    pts = np.zeros((n, 96), dtype=np.uint8)
    pattern = []
    for i in range(n):
        k = keys[(i // 4) % len(keys)]
        pt = [k, k, g1.neg(k), None][i % 4]       # runs of: P, P, -P, (no key)
        pattern.append(pt)
        if pt is not None:
            pts[i] = np.frombuffer(g1.to_bytes96(pt), dtype=np.uint8)
    offsets = np.arange(0, n + 1, 16, dtype=np.uint32)       # 32 committees of 16 consecutive validators
    members = np.arange(n, dtype=np.uint32)
    index = members
    for s29 in (False, True):
        e = _engine(s29)
        e.store_init(0, 0, b"\x01" * 32)
        e.set_validators(np.full(n, 32 * 10**9, dtype=np.uint64), np.ones(n, dtype=np.uint8), pts)
        out = e.g1_sum(offsets, index=index)      # pe_g1_sum over the registry: launch_g1_planned, the same kernels
        for c in range(offsets.size - 1):
            exp = g1.sum_points([p for p in pattern[offsets[c]:offsets[c + 1]] if p is not None])
            assert out[c].tobytes() == g1.to_bytes96(exp), (s29, c)
