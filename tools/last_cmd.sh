bash tools/gpu.sh r03x tests
grep -n "Error\|error\|assert \|FAILED" gpurun_out/r03x/pytest.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
