cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for L in old new old new; do
  P=yolov5_amd/libyolov5_hip.so; [ $L = old ] && P=yolov5_amd/libyolov5_hip_old.so
  Y5_LIB_PATH=$P Y5_TUNE_CACHE=/tmp/tc_$L.json timeout 300 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'], d['forward_ms'], d['nms_device_us_per_img'], d['nms_us_per_img'], d['selfcheck']['ok'])"
done 2>&1 | tee gpurun_out/r04_ab_nms_windows.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plans.py -q -k "nms or C2" 2>&1 | tail -2
