#!/bin/bash
# round 5, call g: the winners of tools/r05f.sh together (named sets), the driver's command shape for the default and the best
# set, tools/mfmabench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05g; mkdir -p $O
timeout 330 python tools/sweep.py $O --budget 300 --sets "exclusive+state_on_fin,exclusive+state_on_fin+tree_rotate,exclusive+state_on_fin+rows_event+done_event,exclusive+state_on_fin+tree_rotate+rows_event+done_event,exclusive+state_on_fin+lag6,two_side+state_on_fin,two_side+state_on_fin+rows_event+done_event" 2>&1 | tee $O/sweep.log | grep "^\[sweep\]"
timeout 120 tools/mfmabench > $O/mfmabench.txt 2>&1; echo "[r05g] mfmabench rc $?"; cat $O/mfmabench.txt
