#!/bin/bash
# Round 6: same-box A/B of the 8-phase implicit-GEMM family (ids 95, 96) in the tuner's race -- arm A: ids kept out (Y5_AUTOTUNE_SKIP=95-96), arm B: in; alternating
# passes, one tune cache per arm; bench line incl. configs C4 / C5 (no training leg); arm B's first pass writes the op tables.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06_ab_g8; rm -rf $O; mkdir -p $O
COMMON="--no-train --no-pipeline --no-cpu-baseline --no-selfcheck --steps 50 --warmup 10"
for pass in 1 2; do
  Y5_AUTOTUNE_SKIP=95-96 Y5_TUNE_CACHE=/tmp/tc_a.json timeout 900 python bench.py $COMMON > $O/a$pass.log 2>&1; grep '^{' $O/a$pass.log | tail -1 > $O/a$pass.json
  Y5_TUNE_CACHE=/tmp/tc_b.json timeout 900 python bench.py $COMMON $([ $pass = 1 ] && echo --op-table $O/op_table_b.json) > $O/b$pass.log 2>&1; grep '^{' $O/b$pass.log | tail -1 > $O/b$pass.json
done
python - <<PY
import json
for n in ("a1","b1","a2","b2"):
    try:
        d=json.load(open("$O/%s.json"%n))
        c=d.get("configs",{})
        print(n, "value", round(d["value"]), "fwd_ms", d.get("forward_ms"), "stack_frac", d["roofline"].get("stack_frac"), "sustained", d.get("gpu_state",{}).get("mfma_sustained_tflops"),
              "C4", {k:c.get("C4",{}).get(k) for k in ("value","forward_ms","forward_mfma_frac")}, "C5", {k:c.get("C5",{}).get(k) for k in ("value","ms_per_step","forward_ms")})
    except Exception as e:
        print(n, "failed", e)
PY
