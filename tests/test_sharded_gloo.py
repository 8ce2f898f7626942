"""CPU, world_size 2, gloo: the N>1 exchange path of pos_evolution_amd.sharded (same code the GPUs run, with the
collectives on CPU tensors).  The per-shard compute is a test double built on the C oracle -- allowed here because
tests/ may use the oracle as the checker; the product never does."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardEngine:
    """Stands in for pos_evolution_amd.Engine on a CPU: same exchange-facing methods, host pointers."""

    def __init__(self, tree, vote, bal, flags, pts, comm):
        self.tree, self.vote, self.bal, self.flags, self.pts, self.comm = tree, vote, bal, flags, pts, comm
        self.num_blocks = tree.parent.size

    def set_stream(self, s):
        pass

    def votes_partial(self, ptr):
        from pos_evolution_amd import _abi
        n = self.num_blocks
        buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(n + _abi.PE_EXCHANGE_EXTRA,))
        buf[:] = 0
        ok = (self.flags & 1).astype(bool) & ~(self.flags & 4).astype(bool) & (self.vote != 0xFFFFFFFF)
        np.add.at(buf, self.vote[ok].astype(np.int64), self.bal[ok])
        act = (self.flags & 1).astype(bool)
        buf[n] = self.bal[act].sum()
        buf[n + 1] = act.sum()

    def head_from_weights(self, ptr):
        from oracle import cport
        from pos_evolution_amd import _abi
        n = self.num_blocks
        buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(n + _abi.PE_EXCHANGE_EXTRA,))
        extra = buf[n:].reshape(-1, 2)
        # feed the reduced direct weights back through the oracle's tree phase: one synthetic "validator" per block
        head, _ = cport.get_head(self.tree.parent, np.ones(n, np.uint8), self.tree.roots,
                                 np.arange(n, dtype=np.uint32), buf[:n].copy(), np.full(n, 1, np.uint8), 0)
        self.totals = (int(extra[:, 0].sum()), int(extra[:, 1].sum()))
        return self.tree.roots[head].tobytes()

    def aggregate_partial(self, ptr, rows=None, packed=None, capacity_groups=0):
        from oracle import cport
        offs = self.comm.offsets
        g = offs.size - 1
        assert g <= capacity_groups, "PE_ERR_CAPACITY"
        from pos_evolution_amd import _abi
        pb = _abi.PE_G1_PARTIAL_BYTES          # the exchange slot size; the oracle's Jacobian partial (144 B) sits in it
        out = cport.g1_partial_groups(self.pts, self.comm.members, offs)      # all bits set: whole committees
        dst = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(g, pb))
        dst[:, :144] = out
        return dict(n_groups=g)

    def g1_finish(self, ptr, n_ranks, n_groups):
        from oracle import cport
        from pos_evolution_amd import _abi
        pb = _abi.PE_G1_PARTIAL_BYTES
        src = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n_ranks * n_groups, pb))
        return cport.g1_finish_partials(np.ascontiguousarray(src[:, :144]), n_ranks, n_groups)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pos_evolution_amd.synth as synth
    from oracle import cport, g1
    from pos_evolution_amd.sharded import ShardedForkChoice

    V, B, Cn = 4000, 200, 32
    tree = synth.random_tree(B, 3, "bushy")
    # global workload, then this rank's validator range
    bal_g = synth.balances(V, 3, mixed=True)
    flags_g = synth.validator_flags(V, 3, inactive_frac=0.02)
    vote_g = synth.zipf_votes(V, B, 3)
    A, Bp = g1.to_bytes96(g1.mul(5, g1.G)), g1.to_bytes96(g1.mul(7, g1.G))
    pts_g = cport.g1_arith_progression(A, Bp, V)
    comm_g = synth.random_committees(V, Cn, 3)
    lo, hi = rank * V // world, (rank + 1) * V // world
    loc_members, loc_offs = [], [0]
    for c in range(Cn):
        m = comm_g.members[comm_g.offsets[c]:comm_g.offsets[c + 1]]
        m = m[(m >= lo) & (m < hi)] - lo
        loc_members.append(m)
        loc_offs.append(loc_offs[-1] + m.size)
    comm_l = synth.Committees(np.array(loc_offs, dtype=np.uint32), np.concatenate(loc_members).astype(np.uint32))
    eng = OracleShardEngine(tree, vote_g[lo:hi], bal_g[lo:hi], flags_g[lo:hi], pts_g[lo:hi], comm_l)
    sh = ShardedForkChoice(eng, n_groups_max=Cn, device=torch.device("cpu"))
    head = sh.get_head()
    agg = sh.aggregate()
    # unsharded truth
    head_o, _ = cport.get_head(tree.parent, np.ones(B, np.uint8), tree.roots, vote_g, bal_g, flags_g, 0)
    want_pk = cport.g1_sum_groups(pts_g, comm_g.members, comm_g.offsets)
    act = (flags_g & 1).astype(bool)
    ok = (head == tree.roots[head_o].tobytes() and np.array_equal(agg["aggpk96"], want_pk)
          and eng.totals == (int(bal_g[act].sum()), int(act.sum())))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]
