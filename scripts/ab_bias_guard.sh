#!/bin/bash
# A/B inside one gpurun call (same box, fresh tile-choice caches each): the conv kernels BEFORE the bias-in-LDS change (variant library built from
# commit 4fb5a3d's conv.hip / conv_igemm.h / conv_h3.h / convh3.hip) against the current ones with Y5_BIAS_LDS = 0 (global bias reads), 1 (LDS bias
# unless it costs a resident workgroup by LDS size), 2 (LDS bias whenever it fits).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {  # name, lib, bias mode
  export Y5_TUNE_CACHE=/tmp/y5_tune_$1.json
  echo -n "$1: "
  Y5_LIB_PATH=$2 Y5_BIAS_LDS=$3 timeout 300 python bench.py --steps 20 --warmup 5 --no-train --no-cpu-baseline --no-configs --no-selfcheck --op-table gpurun_out/op_ab_$1.json 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; forward', d.get('forward_ms'), 'frac', d['roofline']['frac'])"
}
for rep in 1 2; do
  run old yolov5_amd/libyolov5_hip_oldbias.so 0
  run oldigemm_newh3 yolov5_amd/libyolov5_hip_oldigemm.so 1
  run newigemm_oldh3 yolov5_amd/libyolov5_hip_oldh3.so 1
  run new1 yolov5_amd/libyolov5_hip.so 1
done
