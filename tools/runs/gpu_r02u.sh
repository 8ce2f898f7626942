#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
POSEVO_HOST_TRACE=1 POSEVO_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 1 --steps 100 --warmup 6 --no-cpu-baseline > gpurun_out/r02u_engine_rccl.json 2> gpurun_out/r02u_engine_rccl_hosttrace.txt
grep "posevo host" gpurun_out/r02u_engine_rccl_hosttrace.txt | grep -v "comm\." 
mkdir -p gpurun_out/prof_r02u
POSEVO_FORCE_DIST=1 rocprofv3 --kernel-trace -d gpurun_out/prof_r02u -o tl -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 1 --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/r02u_under_rocprof.json 2> gpurun_out/prof_r02u/err.log
ls gpurun_out/prof_r02u/
for f in $(find gpurun_out/prof_r02u -name "*.db"); do python tools/rocpd_timeline.py $f 20 3 > gpurun_out/r02u_timeline.txt 2>&1 && cat gpurun_out/r02u_timeline.txt | tail -45; done
rm -rf gpurun_out/prof_r02u
