#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r02d_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02d_tests.log
tail -3 gpurun_out/r02d_tests.log
for i in 1 2; do
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r02d_bench_lag$i.json 2> gpurun_out/r02d_bench_lag$i.err
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-lag > gpurun_out/r02d_bench_nolag$i.json 2> gpurun_out/r02d_bench_nolag$i.err
done
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/r02d_bench_trace.json 2> gpurun_out/r02d_hosttrace.txt
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-lag > gpurun_out/r02d_bench_trace_nolag.json 2> gpurun_out/r02d_hosttrace_nolag.txt
for f in lag1 nolag1 lag2 nolag2 trace trace_nolag; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02d_bench_$f.json"))
    print("$f", "ms/step %.3f"%d["ms_per_step"], "head p50 %.1f"%d["get_head_p50_us"], {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items()})
except Exception as ex:
    print("$f FAILED", ex)
PY
done
echo LAGGED; grep "posevo host" gpurun_out/r02d_hosttrace.txt | tail -32
echo NOLAG; grep "posevo host" gpurun_out/r02d_hosttrace_nolag.txt | tail -32
