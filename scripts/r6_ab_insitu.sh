#!/bin/bash
# Round 6: (1) the GPU test of the in-situ refinement; (2) same-box A/B on the headline workload -- arm A: Y5_DISABLE=insitu_tune (the isolated race's winners),
# arm B: default (runner-ups timed in place on the first forward); ONE tile-choice cache, so both arms start from the same isolated races.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06_ab_insitu; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plans.py -q -x -s -k in_situ > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
COMMON="--no-train --no-pipeline --no-cpu-baseline --no-selfcheck --no-configs --steps 50 --warmup 10"
export Y5_TUNE_CACHE=/tmp/tc_insitu.json
for pass in 1 2 3; do
  Y5_DISABLE=insitu_tune timeout 900 python bench.py $COMMON $([ $pass = 2 ] && echo --op-table $O/op_table_a.json) > $O/a$pass.log 2>&1; grep '^{' $O/a$pass.log | tail -1 > $O/a$pass.json
  timeout 900 python bench.py $COMMON $([ $pass = 2 ] && echo --op-table $O/op_table_b.json) > $O/b$pass.log 2>&1; grep '^{' $O/b$pass.log | tail -1 > $O/b$pass.json
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
    ta=json.load(open("$O/op_table_a.json")); tb=json.load(open("$O/op_table_b.json"))
    for x,y in zip(ta,tb):
        if x["cfg"]!=y["cfg"]: print(f"{x['op']:40s} {str(x['cfg']):>6s} {x['ms']*1e3:7.1f} us (isolated {(x.get('ms_isolated') or 0)*1e3:6.1f}) -> {str(y['cfg']):>6s} {y['ms']*1e3:7.1f} us (isolated {(y.get('ms_isolated') or 0)*1e3:6.1f})")
    print("sum of in-situ op times: a %.1f us, b %.1f us" % (sum(r["ms"] for r in ta)*1e3, sum(r["ms"] for r in tb)*1e3))
except Exception as e:
    print("op tables:", e)
PY
tail -5 $O/pytest.log
