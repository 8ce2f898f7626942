#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03h_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03h_tests.log
tail -n 4 gpurun_out/r03h_tests.log
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step %.3f p50 %.3f min %.3f frac %.4f head %.1f"%(d["ms_per_step"], d["step_ms_p50"], d["step_ms_min"], d["roofline"]["frac"], d["get_head_p50_us"]), {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v})
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
for i in 1 2 3; do
timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r03h_default_$i.json 2> gpurun_out/r03h.err
show gpurun_out/r03h_default_$i.json
done
POSEVO_G1_STREAM_ONE_WAVE=1 timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r03h_k16.json 2> gpurun_out/r03h.err
show gpurun_out/r03h_k16.json
mkdir -p gpurun_out/prof_r03h
rocprofv3 --kernel-trace -d gpurun_out/prof_r03h -o tl -- python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/r03h_under_rocprof.json 2> gpurun_out/prof_r03h/err.log
for f in $(find gpurun_out/prof_r03h -name "*.db"); do python tools/rocpd_timeline.py $f 20 3 > gpurun_out/r03h_timeline.txt 2>&1; tail -34 gpurun_out/r03h_timeline.txt; done
rm -rf gpurun_out/prof_r03h
