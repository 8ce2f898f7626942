#!/bin/bash
# round 5, call k: the N > 1 bench path with ONE rank over the engine's own RCCL (POSEVO_FORCE_DIST) at the per-rank sizes of
# 8 shards -- what bench.py --gpus 8 runs on every rank, minus the other seven
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05k; mkdir -p $O
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
for spec in "configs3 131072 engine" "configs4 524288 engine" "configs3 1048576 committee"; do set -- $spec
  POSEVO_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --shape $1 --validators $2 --sharded-mode $3 --steps 100 --warmup 6 --no-cpu-baseline --no-signed-steps --no-slot-cadence --no-shuffle-variant > $O/rank_${1}_$3.json 2> $O/rank_${1}_$3.err
  echo "[r05k] one rank over the engine's RCCL, $1 x $2 validators, $3 shards: rc $? $(timeout 20 python tools/benchline.py < $O/rank_${1}_$3.json 2>/dev/null | cut -c1-170)"; tail -2 $O/rank_${1}_$3.err | cut -c1-200
done
