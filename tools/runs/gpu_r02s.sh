#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r02s_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02s_tests.log
tail -n 5 gpurun_out/r02s_tests.log
timeout 1500 bash tools/profile_round.sh r02s
POSEVO_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 1 --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r02s_bench_engine_rccl_world1.json 2> gpurun_out/r02s_bench_engine_rccl_world1.err
head -c 400 gpurun_out/r02s_bench_engine_rccl_world1.json; tail -n 3 gpurun_out/r02s_bench_engine_rccl_world1.err
