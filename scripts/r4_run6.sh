cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for L in old new old new; do
  P=yolov5_amd/libyolov5_hip.so; [ $L = old ] && P=yolov5_amd/libyolov5_hip_old.so
  echo "== $L"; Y5_LIB_PATH=$P timeout 300 python scripts/r4_layers.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04_ab_loader_layers.log 2>&1
cat gpurun_out/r04_ab_loader_layers.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "conv" 2>&1 | tail -3
for L in old new old new; do
  P=yolov5_amd/libyolov5_hip.so; [ $L = old ] && P=yolov5_amd/libyolov5_hip_old.so
  Y5_LIB_PATH=$P Y5_TUNE_CACHE=/tmp/tc_$L.json timeout 300 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'], d['forward_ms'], d['selfcheck']['ok'], d['roofline']['stack_frac'], d['roofline']['dominant_kernel']['ms_per_step'])"
done 2>&1 | tee gpurun_out/r04_ab_loader_bench.log
