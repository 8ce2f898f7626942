K="committee_sharded_dry_run" bash tools/gpu.sh r03s tests
grep -n "Error\|error\|assert" gpurun_out/r03s/pytest.log | head -20
