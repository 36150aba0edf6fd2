#!/bin/bash
# Kernel-experiment aid: what each part of the fused backbone front (csrc/conv_front.h) costs -- variant libraries built with one part removed
# (results are wrong by construction; only the time is read).  Build here (hipcc cross-compiles), run on the GPU box:
#   bash scripts/front_ablate.sh build ; gpurun -- 'bash scripts/front_ablate.sh run'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V="${FRV:-BASE NOSILU NOSTEMMFMA NOSTEM NOL1}"
if [ "$1" = "build" ]; then
  cd yolov5_amd/csrc
  for v in $V; do
    D=""; [ "$v" != "BASE" ] && D="-DY5_FR_$v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -c front.hip -o _build/front_abl.o 2>/dev/null || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyolov5_hip_fr_$v.so $(ls _build/*.o | grep -v "front") _build/front_abl.o || exit 1
  done
  rm -f _build/front_abl.o
else
  for v in $V; do
    echo -n "$v: "; Y5_LIB_PATH=yolov5_amd/libyolov5_hip_fr_$v.so timeout 100 python scripts/front_bench.py --blocks 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['front_us_blocks0'], 'us (two-launch', d['two_launch_us'], ')')"
  done
fi
