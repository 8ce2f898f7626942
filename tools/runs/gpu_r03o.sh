#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step %.3f p50 %.3f min %.3f frac %.4f"%(d["ms_per_step"], d["step_ms_p50"], d["step_ms_min"], d["roofline"]["frac"]), {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v})
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
for i in 1 2; do
for m in 0 2 8; do
POSEVO_FIN_CU_MASK=$m timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r03o_mask${m}_$i.json 2> gpurun_out/r03o.err
show gpurun_out/r03o_mask${m}_$i.json
done; done
