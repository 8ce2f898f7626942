#!/bin/bash
# round 5, call d: sharded steps with held launches, the signature leg behind the accumulation, slot cadence diagnosis, per-rank load of range shards
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05d; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dist_custom.py tests/test_gpu_sharded.py tests/test_gpu_pairing.py tests/test_gpu_g2.py -x -q > $O/pytest_sharded.log 2>&1; echo "[r05d] sharded/pairing/g2 tests rc $?"; tail -12 $O/pytest_sharded.log
POSEVO_SLOT_TIMELINE=$O/slot_timeline.txt POSEVO_HOST_TRACE=1 timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > $O/bench_legs.json 2> $O/bench_legs.err; echo "[r05d] bench legs rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05d/bench_legs.json"))
print("ms/step", d["ms_per_step"], "signed", d.get("ms_per_step_with_signatures"), "shuffle", d.get("ms_per_step_with_shuffle"))
print("slot", {k:v for k,v in d.get("slot_cadence",{}).items() if "us" in k})
print("unagg", {k:v for k,v in d.get("with_unaggregated_signatures",{}).items() if k in ("ms_per_epoch","signatures_per_s","error")})
PY
grep "posevo host" $O/bench_legs.err | cut -c1-120 | tail -40
sed -n 1,4p $O/slot_timeline.txt; awk 'NR>400 && NR<470' $O/slot_timeline.txt
for pair in 1 0; do for shape in "configs3 131072" "configs4 524288"; do set -- $shape
  POSEVO_PAIR=$pair POSEVO_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --shape $1 --validators $2 --steps 100 --warmup 6 --no-cpu-baseline --no-signed-steps --no-slot-cadence --no-shuffle-variant > $O/rank_${1}_pair$pair.json 2> $O/rank_${1}_pair$pair.err
  echo "[r05d] per-rank step of an 8-way range shard, $1, pair=$pair: rc $? $(timeout 20 python tools/benchline.py < $O/rank_${1}_pair$pair.json 2>/dev/null | cut -c1-150)"; done; done
bash tools/gpu.sh r05d label:all tests
