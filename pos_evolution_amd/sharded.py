"""Validator-range sharding over N GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI) for the two exchange steps.

Every rank owns a contiguous range of validators (records, latest messages, participation, pubkeys) and a full copy
of the (small) block table.  Only two things ever cross ranks:

  * get_head:   ONE all-reduce(sum) of (B + PE_EXCHANGE_EXTRA) u64 -- per-block direct vote weight plus the
                active-balance partials the proposer boost needs.  Integer sums: bit-exact for any order.
  * aggregate:  ONE all-gather of C x 192 B XYZZ G1 partials (RCCL has no EC-add reduction op), then every rank
                runs the finishing add + normalisation locally.

Messages are tens of KiB: latency-bound, so ring-vs-tree and per-link bandwidth are irrelevant here.
LMD updates and participation flags are local to the shard that owns the validator.

The exchange buffers are torch tensors (plumbing: device memory + collectives); the engine reads and writes them
through raw device pointers on the SAME stream torch issues the collectives on, so no host synchronisation sits
between kernels and collectives.

``use_engine_rccl=True`` is the thin face of the C ABI's own exchange (pe_dist_init / pe_get_head_sharded /
pe_aggregate_sharded, include/posevo.h): the engine owns the RCCL communicator and issues the collectives between its
kernels on its own stream; torch.distributed only carries the 256-byte id (two RCCL unique ids) to the other ranks.
"""
from __future__ import annotations

import numpy as np

from . import _abi


class ShardedForkChoice:
    def __init__(self, engine, n_groups_max: int = 2048, group=None, device=None, use_engine_rccl: bool = False,
                 single_comm: bool = False, collectives=None):
        """use_engine_rccl: the exchange lives behind the C ABI (pe_get_head_sharded / pe_aggregate_sharded) -- over the
        engine's own RCCL communicators (single_comm: PE_DIST_SINGLE_COMM), or, with ``collectives`` (an object with
        all_reduce_u64 / all_gather, e.g. HostStagedCollectives), over the caller's (pe_dist_init_custom)."""
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.engine = engine
        self.group = group
        self.use_engine_rccl = use_engine_rccl
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        self._wbuf = None
        self._pw = _abi.PE_G1_PARTIAL_BYTES // 4  # int32 words per partial
        self._partial = torch.zeros(n_groups_max * self._pw, dtype=torch.int32, device=device)
        self._gathered = torch.zeros(self.world * n_groups_max * self._pw, dtype=torch.int32, device=device)
        self.n_groups_max = n_groups_max
        if use_engine_rccl:
            if collectives is not None:
                self.collectives = collectives
                engine.dist_init_custom(self.rank, self.world, collectives.all_reduce_u64, collectives.all_gather)
                return
            ids = [engine.dist_unique_id() if self.rank == 0 else None]
            if dist.is_initialized() and self.world > 1:
                dist.broadcast_object_list(ids, src=0, group=group)
            engine.dist_init_ex(ids[0], self.rank, self.world, single_comm=single_comm)
            return
        if device.type == "cuda":
            # engine kernels and RCCL ordered on one (non-null) stream
            if torch.cuda.current_stream().cuda_stream == 0:
                torch.cuda.set_stream(torch.cuda.Stream())
            engine.set_stream(torch.cuda.current_stream().cuda_stream)

    def _weights_buffer(self):
        n = self.engine.num_blocks + _abi.PE_EXCHANGE_EXTRA
        if self._wbuf is None or self._wbuf.numel() != n:
            self._wbuf = self.torch.zeros(n, dtype=self.torch.int64, device=self.device)
        return self._wbuf

    def get_head(self) -> bytes:
        """get_head (pe:1102-1116) over all shards; every rank returns the same root."""
        if self.use_engine_rccl:
            return self.engine.get_head_sharded()
        buf = self._weights_buffer()
        self.engine.votes_partial(buf.data_ptr())
        if self.dist.is_initialized():  # also with world == 1: keeps the RCCL path exercised on one GPU
            if self._staged():
                host = buf.cpu()
                self.dist.all_reduce(host, group=self.group)
                buf.copy_(host)
            else:
                self.dist.all_reduce(buf, group=self.group)
        return self.engine.head_from_weights(buf.data_ptr())

    def _staged(self) -> bool:
        """gloo with device buffers (dry runs of the N > 1 path on one GPU): collectives go through host copies."""
        return self.device.type == "cuda" and self.dist.get_backend(self.group) != "nccl"

    def aggregate(self, rows=None, packed=None):
        """pe_aggregate over all shards: rank-local bitfield unions, global aggregate pubkeys.
        Every rank must pass attestations that form the SAME groups in the SAME order (group g of every rank =
        that rank's members of committee g)."""
        if self.use_engine_rccl:
            return self.engine.aggregate_sharded(rows=rows, packed=packed)
        # the engine refuses (PE_ERR_CAPACITY, before writing anything) a batch that forms more groups than the
        # exchange buffers were sized for
        res = self.engine.aggregate_partial(self._partial.data_ptr(), rows=rows, packed=packed,
                                            capacity_groups=self.n_groups_max)
        g = res["n_groups"]
        part = self._partial[: g * self._pw]
        gathered = self._gathered[: self.world * g * self._pw]
        if self.dist.is_initialized():
            if self.dist.get_backend(self.group) == "nccl":  # RCCL: one flat collective, no staging copies
                self.dist.all_gather_into_tensor(gathered, part, group=self.group)
            elif self._staged():                             # gloo + device buffers: through the host
                host = [self.torch.empty(part.numel(), dtype=part.dtype) for _ in range(self.world)]
                self.dist.all_gather(host, part.cpu(), group=self.group)
                gathered.copy_(self.torch.cat(host))
            else:                                            # gloo (CPU tests)
                self.dist.all_gather(list(gathered.chunk(self.world)), part, group=self.group)
        else:
            gathered.copy_(part)
        res["aggpk96"] = self.engine.g1_finish(gathered.data_ptr(), self.world, g)
        return res


class HostStagedCollectives:
    """The two exchange steps as host-staged collectives over ANY torch.distributed backend (gloo included): the pair of
    callables ``Engine.dist_init_custom`` takes (pe_dist_init_custom, include/posevo.h).  Each call synchronises the
    stream it is ordered on, copies the device buffer to the host, runs the collective there and copies the result back
    -- slow, and exactly what is needed to run the engine-owned sharded step with several ranks that SHARE one GPU (RCCL
    refuses two ranks on one device), or on a node whose GPUs have no peer links.  Not a performance path."""

    def __init__(self, group=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group, self.C = torch, dist, group, C
        self.world = dist.get_world_size(group)
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.calls = {"all_reduce": 0, "all_gather": 0}

    def _sync(self, stream):
        if self.hip.hipStreamSynchronize(self.C.c_void_p(stream)) != 0:
            raise RuntimeError("hipStreamSynchronize failed")

    def _copy(self, dst, src, nbytes, kind):
        if self.hip.hipMemcpy(self.C.c_void_p(dst), self.C.c_void_p(src), nbytes, kind) != 0:
            raise RuntimeError("hipMemcpy failed")

    def all_reduce_u64(self, buf, count, stream):
        self._sync(stream)
        host = self.torch.empty(count, dtype=self.torch.int64)  # two's-complement sums = u64 sums
        self._copy(host.data_ptr(), buf, 8 * count, 2)          # hipMemcpyDeviceToHost
        self.dist.all_reduce(host, group=self.group)
        self._copy(buf, host.data_ptr(), 8 * count, 1)          # hipMemcpyHostToDevice
        self.calls["all_reduce"] += 1
        return 0

    def all_gather(self, send, recv, nbytes, stream):
        self._sync(stream)
        mine = self.torch.empty(nbytes, dtype=self.torch.uint8)
        self._copy(mine.data_ptr(), send, nbytes, 2)
        parts = [self.torch.empty(nbytes, dtype=self.torch.uint8) for _ in range(self.world)]
        self.dist.all_gather(parts, mine, group=self.group)
        allb = self.torch.cat(parts)
        self._copy(recv, allb.data_ptr(), nbytes * self.world, 1)
        self.calls["all_gather"] += 1
        return 0
