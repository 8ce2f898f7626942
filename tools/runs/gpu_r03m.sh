#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_pipeline_robust.py tests/test_gpu_sharded.py tests/test_gpu_c_client.py -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -4
for i in 1 2 3; do
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline 2> gpurun_out/r03m_trace_$i.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.3f p50 %.3f min %.3f frac %.4f'%(d['ms_per_step'], d['step_ms_p50'], d['step_ms_min'], d['roofline']['frac']))"
done
grep "posevo host" gpurun_out/r03m_trace_3.txt | grep "head\.\|pipe\."
