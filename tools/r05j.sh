#!/bin/bash
# round 5, call j: the accumulation alone and in the step on ONE box (tools/accbench, then the bench line and the driver's
# command); the N > 1 bench path with one rank over the engine's own RCCL (POSEVO_FORCE_DIST) at the per-rank sizes of 8 shards
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05j; mkdir -p $O
timeout 120 tools/accbench > $O/accbench.txt 2>&1; echo "[r05j] accbench rc $?"; tail -8 $O/accbench.txt
bash tools/gpu.sh r05j label:box5 quick driver
for spec in "configs3 131072 engine" "configs4 524288 engine" "configs3 1048576 committee"; do set -- $spec
  POSEVO_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --shape $1 --validators $2 --sharded-mode $3 --steps 100 --warmup 6 --no-cpu-baseline --no-signed-steps --no-slot-cadence --no-shuffle-variant > $O/rank_${1}_$3.json 2> $O/rank_${1}_$3.err
  echo "[r05j] one rank over the engine's RCCL, $1 x $2 validators, $3 shards: rc $? $(timeout 20 python tools/benchline.py < $O/rank_${1}_$3.json 2>/dev/null | cut -c1-170)"; tail -2 $O/rank_${1}_$3.err | cut -c1-200
done
