export STEPS=150 BENCH_ARGS="--no-verify-steps --lag 4"
bash tools/gpu.sh r03g label:base quick
POSEVO_G1_TREE_SERIAL=1 bash tools/gpu.sh r03g label:serial quick
POSEVO_G1_TREE_SERIAL=1 POSEVO_G1_NORM_STREAM=0 bash tools/gpu.sh r03g label:serial_nonorm quick
POSEVO_LIB_PATH=$PWD/build/variants/libposevo_t3.so bash tools/gpu.sh r03g label:t3 quick
POSEVO_LIB_PATH=$PWD/build/variants/libposevo_t3.so POSEVO_G1_TREE_PAD_LDS=0 bash tools/gpu.sh r03g label:t3_nopad quick
POSEVO_G1_TREE_PAD_LDS=0 bash tools/gpu.sh r03g label:nopad quick
POSEVO_G1_STREAM_ONE_WAVE=1 bash tools/gpu.sh r03g label:onewave quick
bash tools/gpu.sh r03g label:base_b quick
