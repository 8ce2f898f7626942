K="dist_custom or sharded" bash tools/gpu.sh r03h tests
