#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in "1 64" "2 64" "4 64" "4 128" "2 128" "8 512"; do
  set -- $cfg
  echo "=== k=$1 size=$2" 
  timeout 60 tools/g1_phases $1 $2 | head -14
done > gpurun_out/r02m_g1_phases_small.txt 2>&1
cat gpurun_out/r02m_g1_phases_small.txt
