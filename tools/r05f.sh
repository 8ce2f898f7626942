#!/bin/bash
# round 5, call f: the scheduling knobs of the streaming G1 chain A/B on one box (tools/sweep.py), the streaming tests with all
# of them switched on, the per-rank load of the sharded divisions
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05f; mkdir -p $O
timeout 560 python tools/sweep.py $O --budget 470 2>&1 | tee $O/sweep.log | grep "^\[sweep\]"
( export POSEVO_ACC_EXCLUSIVE=1 POSEVO_TREE_ROTATE=1 POSEVO_ACC_DONE_EVENT=1 POSEVO_ROWS_EVENT=1 POSEVO_STATE_ON=1 POSEVO_SIDE_STREAMS=2
  timeout 400 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_pipeline.py tests/test_gpu_pipeline_robust.py tests/test_gpu_g1_accumulate.py tests/test_gpu_shapes.py -x -q > $O/pytest_knobs.log 2>&1
  echo "[r05f] streaming tests with every knob on: rc $?"; tail -5 $O/pytest_knobs.log )
for shape in configs3 configs4; do
  timeout 300 python bench.py --emulate-ranks 8 --shape $shape --steps 100 --warmup 6 --no-signed-steps --no-slot-cadence > $O/emulate8_$shape.json 2> $O/emulate8_$shape.err
  echo "[r05f] committee shards, per-rank step of 8, $shape: rc $? $(timeout 20 python tools/benchline.py < $O/emulate8_$shape.json 2>/dev/null | cut -c1-160)"
done
