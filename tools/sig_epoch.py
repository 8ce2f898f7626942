#!/usr/bin/env python
"""pe_aggregate_signatures over one epoch's unaggregated signatures (1 048 576 compressed BLSSignatures, 2048 committees of
512 attesters), a few calls in a row: the command the profiler passes of the signature leg run
(tools/r06_calls.sh: rocprofv3 --kernel-trace --stats, then --pmc passes).  Prints ms per call and the first aggregate."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--validators", type=int, default=1 << 20)
    ap.add_argument("--committees", type=int, default=2048)
    ap.add_argument("--calls", type=int, default=3)
    args = ap.parse_args()
    import torch

    import pos_evolution_amd as pea
    import pos_evolution_amd.synth as synth
    from pos_evolution_amd import DeviceArena

    V, C = args.validators, args.committees
    e = pea.Engine(device=0)
    base = synth.signature_points(e, min(V, 16384))
    sigs = np.ascontiguousarray(np.tile(base, (-(-V // base.shape[0]), 1))[:V])
    sig_t = torch.from_numpy(sigs.reshape(-1)).cuda()
    sig_dev = DeviceArena(sig_t.data_ptr(), sig_t.numel(), keep=sig_t)
    index = np.random.Generator(np.random.PCG64(1)).permutation(V).astype(np.uint32)
    offsets = (np.arange(C + 1, dtype=np.uint64) * (V // C)).astype(np.uint32)
    idx_t = torch.from_numpy(index).cuda()
    idx_dev = DeviceArena(idx_t.data_ptr(), idx_t.numel() * 4, keep=idx_t)
    torch.cuda.synchronize()
    for k in range(args.calls):
        t0 = time.perf_counter()
        agg, status, bad = e.aggregate_signatures(sig_dev, offsets, index=idx_dev)
        dt = time.perf_counter() - t0
        print(f"call {k}: {dt * 1e3:.2f} ms, {V / dt / 1e6:.2f} M signatures/s, bad {int(bad.sum())}, agg[0] {bytes(agg[0]).hex()[:16]}")
    e.close()


if __name__ == "__main__":
    main()
