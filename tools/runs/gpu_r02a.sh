#!/bin/bash
# round 2, call A: GPU parity suite + A/B of the pipelined C ABI
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a_tests.log
tail -5 gpurun_out/r02a_tests.log
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r02a_bench_pipe.json 2> gpurun_out/r02a_bench_pipe.err
POSEVO_G1_SIDE_STREAM=0 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r02a_bench_pipe_noside.json 2> gpurun_out/r02a_bench_pipe_noside.err
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-pipeline > gpurun_out/r02a_bench_sync.json 2> gpurun_out/r02a_bench_sync.err
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02a_bench_trace.json 2> gpurun_out/r02a_hosttrace.txt
for f in pipe pipe_noside sync; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02a_bench_$f.json"))
    print("$f", "ms/step %.3f"%d["ms_per_step"], "head p50 %.1f"%d["get_head_p50_us"], {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items()})
except Exception as ex:
    print("$f FAILED", ex)
PY
done
grep "posevo host" gpurun_out/r02a_hosttrace.txt | tail -20
