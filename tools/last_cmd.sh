K="forkchoice or head or engine or pipeline or edge or shapes or full_size or golden" bash tools/gpu.sh r03q tests
export STEPS=10
for q in 2 4; do
POSEVO_VOTES_QUADS=$q BENCH_ARGS="--no-verify-steps --no-shuffle-variant --head-calls 400" bash tools/gpu.sh r03q label:q$q quick > /dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r03q/bench_quick_q$q.json").read().strip().splitlines()[-1])
print("quads $q", "p50", round(d["get_head_p50_us"],1), "p99", round(d["get_head_p99_us"],1), "ms/step", round(d["ms_per_step"],3))
PY
done
