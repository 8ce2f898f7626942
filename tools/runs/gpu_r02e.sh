#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r02e_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02e_tests.log
POSEVO_FC_CUS=32 timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r02e_tests_fc32.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02e_tests_fc32.log
tail -3 gpurun_out/r02e_tests.log gpurun_out/r02e_tests_fc32.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 100 --warmup 6 --no-cpu-baseline > gpurun_out/r02e_bench_$name.json 2> gpurun_out/r02e_bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02e_bench_$name.json"))
    print("$name", "ms/step %.3f"%d["ms_per_step"], "head p50 %.1f"%d["get_head_p50_us"], {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v})
except Exception as ex:
    print("$name FAILED", ex); print(open("gpurun_out/r02e_bench_$name.err").read()[-800:])
PY
}
run fc0 POSEVO_FC_CUS=0
run fc16 POSEVO_FC_CUS=16
run fc32 POSEVO_FC_CUS=32
run fc48 POSEVO_FC_CUS=48
run fc32ns POSEVO_FC_CUS=32 POSEVO_FC_CUS_SPREAD=0
run fc0b POSEVO_FC_CUS=0
POSEVO_FC_CUS=32 POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > gpurun_out/r02e_bench_trace.json 2> gpurun_out/r02e_hosttrace.txt
grep "posevo host" gpurun_out/r02e_hosttrace.txt | tail -32
