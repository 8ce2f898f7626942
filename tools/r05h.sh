#!/bin/bash
# round 5, call h: the streaming step's timeline with the new defaults (exclusive accumulation workgroups, flag passes on the
# tree's stream); the signature leg behind / beside the accumulation on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05h; mkdir -p $O
timeout 200 python tools/engine_timeline.py --steps 30 --show 4 > $O/engine_timeline.txt 2>&1; tail -16 $O/engine_timeline.txt
timeout 200 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20.txt 2>&1; head -26 $O/engine_timeline_cold20.txt
for b in 1 0; do
  POSEVO_SIG_BEHIND=$b timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant > $O/bench_sig_behind$b.json 2> $O/bench_sig_behind$b.err
  python - <<PY
import json
d=json.load(open("$O/bench_sig_behind$b.json"))
print("[r05h] sig_behind=$b: ms/step", round(d["ms_per_step"],4), "signed", round(d["ms_per_step_with_signatures"],4), d["with_signatures"].get("steps_verified"), "unaggregated ms/epoch", round(d["with_unaggregated_signatures"]["ms_per_epoch"],2))
PY
done
