cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/r4_ablate.py > gpurun_out/r04_ablate_v0.log 2>&1; cat gpurun_out/r04_ablate_v0.log
Y5_LIB_PATH=yolov5_amd/libyolov5_hip_h3dbg.so timeout 300 python scripts/h3_timing.py 76,70,63 > gpurun_out/r04_h3_timing_v0.log 2>&1; cat gpurun_out/r04_h3_timing_v0.log
