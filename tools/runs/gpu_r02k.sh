#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r02k_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02k_tests.log
tail -n 25 gpurun_out/r02k_tests.log
# all-cores CPU baseline scaling probe
python - <<'PY' > gpurun_out/r02k_cpu_scaling.txt 2>&1
import os, time, numpy as np, sys
sys.path.insert(0, ".")
from oracle import cport
import pos_evolution_amd.synth as synth
from tests import helpers as H
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp max", cport.max_threads())
n=1<<18
pts,_=H.oracle_points(n)
comm=synth.random_committees(n, 2048, 1)
sizes=(comm.offsets[1:]-comm.offsets[:-1]).astype(np.uint32)
out_off=np.concatenate([[0],np.cumsum((sizes+7)//8)]).astype(np.uint32)
union=np.full(int(out_off[-1]),0xFF,dtype=np.uint8)
for t in (1,2,4,8,16,32,64,128,256):
    cport.set_threads(t)
    t0=time.perf_counter(); cport.g1_sum_attesters(comm.offsets[:-1], sizes, out_off[:-1], union, comm.members, pts, mt=True); dt=time.perf_counter()-t0
    print(t, "threads: %.1f ms  %.2f M points/s"%(dt*1e3, n/dt/1e6))
PY
cat gpurun_out/r02k_cpu_scaling.txt
