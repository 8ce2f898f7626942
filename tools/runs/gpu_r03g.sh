#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03g_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03g_tests.log
tail -n 5 gpurun_out/r03g_tests.log
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 100 --warmup 6 --no-cpu-baseline > gpurun_out/r03g_trace.json 2> gpurun_out/r03g_hosttrace.txt
grep "posevo host" gpurun_out/r03g_hosttrace.txt | grep "agg\.\|pipe\."
timeout 1500 bash tools/profile_round.sh r03g 2>&1 | tail -40
