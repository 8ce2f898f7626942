#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_c_client.py -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -15
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -Iinclude examples/step_client.c -Lpos_evolution_amd -lposevo -Wl,-rpath,$GRAFT_REPO_ROOT/pos_evolution_amd -o /tmp/step_client || exit 1
timeout 600 python examples/make_workload.py /tmp/workload.bin --steps 60 2>&1 | tail -2
for mode in streaming streaming pipelined sync streaming; do POSEVO_HOST_TRACE=1 timeout 120 /tmp/step_client /tmp/workload.bin $mode 6 nohash 2> /tmp/trace_$mode.txt | tail -1; done; grep "posevo host" /tmp/trace_streaming.txt | grep -v "comm\."  | tee gpurun_out/r03l_c_client.txt
timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py same box: ms/step', d['ms_per_step'], 'p50', d['step_ms_p50'])" | tee -a gpurun_out/r03l_c_client.txt
