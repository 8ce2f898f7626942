bash tools/gpu.sh r03D tests
grep -n "Error\|error\|assert \|FAILED" gpurun_out/r03D/pytest.log | head -20
