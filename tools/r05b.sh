#!/bin/bash
# round 5, call b: the paired launches -- their tests first, then the whole -m gpu suite, then A/B of the step period
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pairing.py -x -q > $O/pytest_pairing.log 2>&1; echo "[r05b] pairing tests rc $?"; tail -15 $O/pytest_pairing.log
export BENCH_ARGS=""
bash tools/gpu.sh r05b label:pair quick driver env:POSEVO_PAIR=0 label:nopair quick driver
unset POSEVO_PAIR
timeout 300 python tools/engine_timeline.py --steps 24 --show 2 > $O/engine_timeline_pair.txt 2>&1; tail -45 $O/engine_timeline_pair.txt
timeout 300 python tools/engine_timeline.py --cold 20 > $O/engine_timeline_cold20_pair.txt 2>&1; head -30 $O/engine_timeline_cold20_pair.txt
bash tools/gpu.sh r05b label:all tests
