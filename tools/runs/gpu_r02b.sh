#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02b_tests.log
tail -5 gpurun_out/r02b_tests.log
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r02b_bench_pipe.json 2> gpurun_out/r02b_bench_pipe.err
POSEVO_G1_SIDE_STREAM=0 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r02b_bench_pipe_noside.json 2> gpurun_out/r02b_bench_pipe_noside.err
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02b_bench_trace.json 2> gpurun_out/r02b_hosttrace.txt
POSEVO_BREAKDOWN=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02b_bench_breakdown.json 2> /dev/null
for f in pipe pipe_noside breakdown; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02b_bench_$f.json"))
    print("$f", "ms/step %.3f"%d["ms_per_step"], "head p50 %.1f"%d["get_head_p50_us"], {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items()}, d.get("host_breakdown_ms_per_step"))
except Exception as ex:
    print("$f FAILED", ex)
PY
done
grep "posevo host" gpurun_out/r02b_hosttrace.txt | tail -24
