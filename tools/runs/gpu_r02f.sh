#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02f_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02f_tests.log
tail -n 3 gpurun_out/r02f_tests.log
run() { # name, args...
  name=$1; shift
  timeout 300 python bench.py --steps 100 --warmup 6 --no-cpu-baseline "$@" > gpurun_out/r02f_bench_$name.json 2> gpurun_out/r02f_bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02f_bench_$name.json"))
    print("$name", "ms/step %.3f"%d["ms_per_step"], "head p50 %.1f"%d["get_head_p50_us"], {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v})
except Exception as ex:
    print("$name FAILED", ex); print(open("gpurun_out/r02f_bench_$name.err").read()[-800:])
PY
}
run lag1
run nolag1 --no-lag
run lag2
run sync --no-pipeline
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > gpurun_out/r02f_bench_trace.json 2> gpurun_out/r02f_hosttrace.txt
grep "posevo host" gpurun_out/r02f_hosttrace.txt | grep -v "agg.2\|att.1[abd]\|wait_outputs \|wait_d2h" | tail -32
