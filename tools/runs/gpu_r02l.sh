#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline "$@" > gpurun_out/r02l_bench_$name.json 2> gpurun_out/r02l_bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02l_bench_$name.json"))
    print("$name", "ms/step %.3f"%d["ms_per_step"], "p50 %.3f"%d["step_ms_p50"], {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v})
except Exception as ex:
    print("$name FAILED", ex); print(open("gpurun_out/r02l_bench_$name.err").read()[-600:])
PY
}
for V in 131072 262144; do
for K in 1 2 4 8; do
run v${V}_k${K}_sync POSEVO_G1_MIN_K=$K -- --validators $V --no-pipeline
done
run v${V}_k4_stream POSEVO_G1_MIN_K=4 -- --validators $V
run v${V}_k2_stream POSEVO_G1_MIN_K=2 -- --validators $V
done
