#!/bin/bash
# Kernel-experiment aid: what each part of the patch-staged 3x3 weight-gradient kernel (csrc/wgrad3.h) costs -- variant libraries with one part
# removed (results wrong by construction; only the time is read).  bash scripts/wgrad3_ablate.sh build ; gpurun -- 'bash scripts/wgrad3_ablate.sh run'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V="${WGV:-BASE NOSTAGE NOREAD NOMFMA NOATOM}"
if [ "$1" = "build" ]; then
  cd yolov5_amd/csrc
  for v in $V; do
    D=""; [ "$v" != "BASE" ] && D="-DY5_WG_$v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -c wgrad.hip -o _build/wgrad_abl.o 2>/dev/null || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyolov5_hip_wg_$v.so $(ls _build/*.o | grep -v "wgrad") _build/wgrad_abl.o || exit 1
  done
  rm -f _build/wgrad_abl.o
else
  for v in $V; do
    echo "== $v"; Y5_LIB_PATH=yolov5_amd/libyolov5_hip_wg_$v.so timeout 100 python scripts/wgrad_bench.py --k3-ab --cfgs 3 --only ${WGONLY:-160,80,40,20} --iters 6 2>&1 | grep -v amdgpu.ids
  done
fi
