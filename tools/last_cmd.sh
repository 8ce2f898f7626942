K="sharded" bash tools/gpu.sh r03i tests
