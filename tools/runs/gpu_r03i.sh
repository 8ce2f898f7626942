#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
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
timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r03i_prio_$i.json 2> gpurun_out/r03i.err
show gpurun_out/r03i_prio_$i.json
POSEVO_LIB_PATH=$GRAFT_REPO_ROOT/pos_evolution_amd/libposevo_noprio.so timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r03i_noprio_$i.json 2> gpurun_out/r03i.err
show gpurun_out/r03i_noprio_$i.json
done
mkdir -p gpurun_out/prof_r03i
rocprofv3 --kernel-trace -d gpurun_out/prof_r03i -o tl -- python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/r03i_under_rocprof.json 2> gpurun_out/prof_r03i/err.log
for f in $(find gpurun_out/prof_r03i -name "*.db"); do python tools/rocpd_timeline.py $f 20 3 > gpurun_out/r03i_timeline.txt 2>&1; tail -24 gpurun_out/r03i_timeline.txt; done
rm -rf gpurun_out/prof_r03i
