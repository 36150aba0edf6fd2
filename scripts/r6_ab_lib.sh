#!/bin/bash
# Round 6: same-box A/B of two builds of the kernel library on the headline workload -- arm A: yolov5_amd/libyolov5_hip_old.so (the previous commit's kernels),
# arm B: the tree's library; alternating passes, one tune cache per arm (a cache is bound to the library's hash anyway).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06_ab_lib; rm -rf $O; mkdir -p $O
COMMON="--no-train --no-pipeline --no-cpu-baseline --no-selfcheck --no-configs --steps 50 --warmup 10"
for pass in 1 2 3; do
  Y5_LIB_PATH=$PWD/yolov5_amd/libyolov5_hip_old.so Y5_TUNE_CACHE=/tmp/tc_a.json timeout 900 python bench.py $COMMON $([ $pass = 1 ] && echo --op-table $O/op_table_a.json) > $O/a$pass.log 2>&1; grep '^{' $O/a$pass.log | tail -1 > $O/a$pass.json
  Y5_TUNE_CACHE=/tmp/tc_b.json timeout 900 python bench.py $COMMON $([ $pass = 1 ] && echo --op-table $O/op_table_b.json) > $O/b$pass.log 2>&1; grep '^{' $O/b$pass.log | tail -1 > $O/b$pass.json
done
python - <<PY | tee $O/summary.log
import json
for n in ("a1","b1","a2","b2","a3","b3"):
    try:
        d=json.load(open("$O/%s.json"%n))
        print(n, "value", round(d["value"]), "ms_per_step", d.get("ms_per_step"), "fwd_ms", d.get("forward_ms"), "sustained", d.get("config",{}).get("gpu_state",{}).get("mfma_sustained_tflops"))
    except Exception as e:
        print(n, "failed", e)
try:
    ta={r["op"]+str(i):r for i,r in enumerate(json.load(open("$O/op_table_a.json")))}
    tb={r["op"]+str(i):r for i,r in enumerate(json.load(open("$O/op_table_b.json")))}
    for k in ta:
        if k in tb and abs(ta[k]["ms"]-tb[k]["ms"])*1e3 > 1.5: print(f"{k:44s} {str(ta[k]['cfg']):>6s} {ta[k]['ms']*1e3:7.1f} -> {str(tb[k]['cfg']):>6s} {tb[k]['ms']*1e3:7.1f} us")
except Exception as e:
    print("op tables:", e)
PY
