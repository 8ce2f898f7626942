#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <name> <extra hipcc flags...> -> build/variants/libposevo_<name>.so
# (travels to the GPU box with the snapshot; select it with POSEVO_LIB_PATH).  Only the kernel files are rebuilt with the
# extra flags; host objects come from the in-tree build.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/pos_evolution_amd/csrc
OUT=$ROOT/build/variants; mkdir -p "$OUT/$NAME"
make -C "$SRC" -j8 > /dev/null
for k in att_kernels g1_kernels g2_kernels fc_kernels shuffle_kernels pair_kernels; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c "$SRC/$k.hip" -o "$OUT/$NAME/$k.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libposevo_$NAME.so" "$SRC"/engine_*.o "$OUT/$NAME"/*.o
echo "$OUT/libposevo_$NAME.so"
