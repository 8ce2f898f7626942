#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_engine.py -m gpu -x -q > gpurun_out/r02c_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02c_tests.log
tail -5 gpurun_out/r02c_tests.log
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r02c_bench_lag.json 2> gpurun_out/r02c_bench_lag.err
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-lag > gpurun_out/r02c_bench_nolag.json 2> gpurun_out/r02c_bench_nolag.err
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02c_bench_trace.json 2> gpurun_out/r02c_hosttrace.txt
POSEVO_BREAKDOWN=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02c_bench_breakdown.json 2> /dev/null
for f in lag nolag breakdown; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02c_bench_$f.json"))
    print("$f", "ms/step %.3f"%d["ms_per_step"], "head p50 %.1f"%d["get_head_p50_us"], {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items()}, d.get("host_breakdown_ms_per_step"))
except Exception as ex:
    print("$f FAILED", ex)
PY
done
tail -3 gpurun_out/r02c_bench_lag.err
grep "posevo host" gpurun_out/r02c_hosttrace.txt | tail -24
