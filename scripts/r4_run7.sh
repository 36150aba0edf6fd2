cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 1 2 3; do for L in base frp1 frp2; do
  P=yolov5_amd/libyolov5_hip.so; [ $L != base ] && P=yolov5_amd/libyolov5_hip_$L.so
  echo -n "$L: "; Y5_LIB_PATH=$P timeout 200 python scripts/front_bench.py --iters 40 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['front_us_blocks0'], d['two_launch_us'], d['max_abs_diff'])"
done; done 2>&1 | tee gpurun_out/r04_front_prio_ab.log
