#!/bin/bash
# round 5, call c: pairing after the first-use fixes; the new G2 leg; the whole suite; the full bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pairing.py -q > $O/pytest_pairing.log 2>&1; echo "[r05c] pairing tests rc $?"; tail -8 $O/pytest_pairing.log
timeout 900 python -m pytest tests/test_gpu_g2.py -q -k "aggregate_signatures" > $O/pytest_g2.log 2>&1; echo "[r05c] g2 tests rc $?"; tail -8 $O/pytest_g2.log
bash tools/gpu.sh r05c label:pair driver quick
timeout 300 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20.txt 2>&1; head -26 $O/engine_timeline_cold20.txt
timeout 300 python tools/engine_timeline.py --steps 24 --show 2 > $O/engine_timeline.txt 2>&1; tail -12 $O/engine_timeline.txt
bash tools/gpu.sh r05c label:full bench
bash tools/gpu.sh r05c label:all tests
