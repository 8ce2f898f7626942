#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof_r02r
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02r_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02r_tests.log
tail -n 4 gpurun_out/r02r_tests.log
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline "$@" > gpurun_out/r02r_bench_$name.json 2> gpurun_out/r02r_bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02r_bench_$name.json"))
    print("$name ms/step %.3f p50 %.3f min %.3f"%(d["ms_per_step"], d["step_ms_p50"], d["step_ms_min"]), {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v})
except Exception as ex:
    print("$name FAILED", ex); print(open("gpurun_out/r02r_bench_$name.err").read()[-600:])
PY
}
run tree3 A=1 --
run tree2 POSEVO_LIB_PATH=$GRAFT_REPO_ROOT/pos_evolution_amd/libposevo_tree2.so --
run tree3b A=1 --
run tree2b POSEVO_LIB_PATH=$GRAFT_REPO_ROOT/pos_evolution_amd/libposevo_tree2.so --
run tree3_sync A=1 -- --no-pipeline
run tree2_sync POSEVO_LIB_PATH=$GRAFT_REPO_ROOT/pos_evolution_amd/libposevo_tree2.so -- --no-pipeline
rocprofv3 --kernel-trace -d gpurun_out/prof_r02r -o tl -- python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/r02r_bench_under_rocprof.json 2> gpurun_out/prof_r02r/err.log
python tools/rocpd_timeline.py gpurun_out/prof_r02r/tl_results.db 20 3 > gpurun_out/r02r_timeline.txt 2>&1
cat gpurun_out/r02r_timeline.txt
rm -rf gpurun_out/prof_r02r/*.db
tools/g1_phases 8 512 | head -14
