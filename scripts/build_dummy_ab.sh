#!/bin/bash
# kernel-experiment build: the library as it was before the y5_bglds16_dummy fix (identical dummies mergeable by the compiler) -> yolov5_amd/libyolov5_hip_mergedummy.so,
# for the same-box timing A/B of the fix (profiles/r05/r05_dummy_dma_merge.log)
set -e
cd "$(dirname "$0")/../yolov5_amd/csrc"
make -j8 > /dev/null
mkdir -p _build_md
pids=()
for f in conv head bneck convh3 front sppf; do
  extra=""; [ $f = head ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DY5_DUMMY_MERGEABLE $extra -c $f.hip -o _build_md/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
objs=$(ls _build/*.o | grep -v "/conv.o\|/head.o\|/bneck.o\|/convh3.o\|/front.o\|/sppf.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyolov5_hip_mergedummy.so $objs _build_md/*.o
ls -la ../libyolov5_hip_mergedummy.so
