#!/bin/bash
# kernel-experiment build: the library with the -D flags given as arguments in convg8.hip only -> yolov5_amd/libyolov5_hip_${OUT:-g8dbg}.so
#   scripts/build_g8_dbg.sh -DY5_G8_TIMING                 phase stamps (scripts/g8_timing.py)
#   OUT=g8nt scripts/build_g8_dbg.sh -DY5_G8_ST_AUX=2      epilogue stores with the nt policy (scripts/r6_stream_probe.sh)
set -e
cd "$(dirname "$0")/../yolov5_amd/csrc"
make -j8 > /dev/null
mkdir -p _build_dbg
O=${OUT:-g8dbg}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c convg8.hip -o _build_dbg/convg8_$O.o
objs=$(ls _build/*.o | grep -v "/convg8.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyolov5_hip_$O.so $objs _build_dbg/convg8_$O.o
ls -la ../libyolov5_hip_$O.so
