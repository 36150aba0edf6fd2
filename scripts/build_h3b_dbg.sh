#!/bin/bash
# kernel-experiment build: the library with -DY5_H3B_TIMING in bneck.hip only (phase stamps of conv_h3b.h) -> yolov5_amd/libyolov5_hip_h3bdbg.so
set -e
cd "$(dirname "$0")/../yolov5_amd/csrc"
make -j8 > /dev/null
mkdir -p _build_dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DY5_H3B_TIMING -c bneck.hip -o _build_dbg/bneck_h3b.o
objs=$(ls _build/*.o | grep -v "/bneck.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyolov5_hip_h3bdbg.so $objs _build_dbg/bneck_h3b.o
ls -la ../libyolov5_hip_h3bdbg.so
