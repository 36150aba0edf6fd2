#!/bin/bash
# Kernel experiment: builds ablated variants of the fused Bottleneck kernel (conv_bneck.h, -DY5_BNECK_ABL=<bits>) as separate libraries
# under build_variants/ and times each with scripts/bneck_bench.py (Y5_LIB_PATH selects the library).  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
SRC=yolov5_amd/csrc
OBJS=$(ls $SRC/_build/*.o | grep -v bneck.o)
for abl in 0 1 2 4 8 16 3 31; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DY5_BNECK_ABL=$abl -c $SRC/bneck.hip -o build_variants/bneck_$abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/lib_$abl.so $OBJS build_variants/bneck_$abl.o
  echo "== ablation $abl"
  Y5_LIB_PATH=$PWD/build_variants/lib_$abl.so python scripts/bneck_bench.py 2>&1 | grep "stages 1, grid cap 768\|stages 2, grid cap 512"
done
