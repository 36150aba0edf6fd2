#!/bin/bash
# 3x3 weight gradients per yolov5s layer: general gather kernel (cfg 1) vs the patch-staged family (cfg 3: full channel tile; 322 / 341: capped tiles),
# atomic and deterministic forms, one gpurun call (profiles/r03/r03_wgrad3_ab.log was produced by an earlier form of this script that selected the
# caps through an environment knob of that build).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for mode in "" "--det"; do
  for cfgs in 1,3 322 341; do
    echo "=== cfgs $cfgs mode '$mode'"
    python scripts/wgrad_bench.py --k3-ab --cfgs $cfgs --iters 6 $mode 2>&1 | grep -v amdgpu.ids
  done
done
