#!/bin/bash
# Round 6: (1) phase stamps of id 95 on the stride-2 layers (class tap order); (2) same-box A/B of the Detect heads on the plan's second stream
# (Y5_EXPERIMENTAL=head_branch) now that the neck's 8-phase launches leave CUs idle (200 / 230 tiles on 256 CUs).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
Y5_LIB_PATH=$PWD/yolov5_amd/libyolov5_hip_g8dbg.so timeout 300 python scripts/g8_timing.py 3 18 5 7 6cv3 > gpurun_out/g8_timing_v6.log 2>&1
O=gpurun_out/r06_ab_head_branch; rm -rf $O; mkdir -p $O
COMMON="--no-train --no-pipeline --no-cpu-baseline --no-selfcheck --no-configs --steps 50 --warmup 10"
export Y5_TUNE_CACHE=/tmp/tc_hb.json
for pass in 1 2 3; do
  timeout 900 python bench.py $COMMON > $O/a$pass.log 2>&1; grep '^{' $O/a$pass.log | tail -1 > $O/a$pass.json
  Y5_EXPERIMENTAL=head_branch timeout 900 python bench.py $COMMON > $O/b$pass.log 2>&1; grep '^{' $O/b$pass.log | tail -1 > $O/b$pass.json
done
python - <<PY | tee $O/summary.log
import json
for n in ("a1","b1","a2","b2","a3","b3"):
    try:
        d=json.load(open("$O/%s.json"%n))
        print(n, "value", round(d["value"]), "ms_per_step", d.get("ms_per_step"), "fwd_ms", d.get("forward_ms"), "sustained", d.get("config",{}).get("gpu_state",{}).get("mfma_sustained_tflops"))
    except Exception as e:
        print(n, "failed", e)
PY
