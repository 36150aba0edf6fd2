#!/bin/bash
# kernel-experiment build: the library with -DY5_G8_TIMING (+ extra -D flags given as arguments) in convg8.hip only -> yolov5_amd/libyolov5_hip_g8dbg.so
set -e
cd "$(dirname "$0")/../yolov5_amd/csrc"
make -j8 > /dev/null
mkdir -p _build_dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DY5_G8_TIMING "$@" -c convg8.hip -o _build_dbg/convg8_dbg.o
objs=$(ls _build/*.o | grep -v "/convg8.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyolov5_hip_g8dbg.so $objs _build_dbg/convg8_dbg.o
ls -la ../libyolov5_hip_g8dbg.so
