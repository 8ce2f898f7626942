#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for v in 131072 262144; do
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --validators $v --steps 200 --warmup 6 --no-cpu-baseline --head-calls 20 2> gpurun_out/r03q_trace_$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v ms/step %.3f p50 %.3f min %.3f p90 %.3f'%(d['ms_per_step'], d['step_ms_p50'], d['step_ms_min'], d['step_ms_p90']), {k:(round(v,4) if v else v) for k,v in d['kernel_avg_ms'].items() if v})"
grep "posevo host" gpurun_out/r03q_trace_$v.txt | grep -v "comm\.\|agg.3e\|agg.4\|att.3\|proc.3\|att.1b\|launch_g1" | awk '{print $3, $7, $11, $13}' | tr '\n' ';'; echo
done
