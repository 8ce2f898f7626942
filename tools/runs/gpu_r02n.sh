#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof_r02n
for i in 1 2; do
timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r02n_bench$i.json 2> gpurun_out/r02n_bench$i.err
python - <<PY
import json
d=json.load(open("gpurun_out/r02n_bench$i.json"))
print("run$i ms/step %.3f p50 %.3f min %.3f"%(d["ms_per_step"], d["step_ms_p50"], d["step_ms_min"]), {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v})
PY
done
rocprofv3 --kernel-trace -d gpurun_out/prof_r02n -o tl -- python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/r02n_bench_under_rocprof.json 2> gpurun_out/prof_r02n/err.log
python tools/rocpd_timeline.py gpurun_out/prof_r02n/tl_results.db 20 3 > gpurun_out/r02n_timeline.txt 2>&1
cat gpurun_out/r02n_timeline.txt
rm -rf gpurun_out/prof_r02n/*.db
