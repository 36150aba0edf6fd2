#!/bin/bash
# Round 6: same-box A/B of the in-situ refinement on the headline workload.  One pass with the refinement off fills the isolated races (and their front-runner
# lists) into a base cache; every arm starts from its own copy of it.  Arms: off = Y5_DISABLE=insitu_tune, a = the runner-up only (Y5_INSITU_ALTS=1),
# b = up to three alternatives per race -- an experimental build of engine.py that also timed the races' third and fourth configuration in place (measured equal
# to arm a, profiles/r06/r06_ab_insitu_three_arms.log, and not kept: at HEAD Y5_INSITU_ALTS is not read and arms a and b are the same).  Alternating passes; the first pass of an arm refines, the later ones apply its stored decisions.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06_ab_insitu; rm -rf $O; mkdir -p $O
COMMON="--no-train --no-pipeline --no-cpu-baseline --no-selfcheck --no-configs --steps 50 --warmup 10"
Y5_DISABLE=insitu_tune Y5_TUNE_CACHE=/tmp/tc_base.json timeout 900 python bench.py $COMMON > $O/base.log 2>&1
cp /tmp/tc_base.json /tmp/tc_off.json; cp /tmp/tc_base.json /tmp/tc_a.json; cp /tmp/tc_base.json /tmp/tc_b.json
for pass in 1 2 3; do
  Y5_DISABLE=insitu_tune Y5_TUNE_CACHE=/tmp/tc_off.json timeout 900 python bench.py $COMMON > $O/off$pass.log 2>&1; grep '^{' $O/off$pass.log | tail -1 > $O/off$pass.json
  Y5_INSITU_ALTS=1 Y5_TUNE_CACHE=/tmp/tc_a.json timeout 900 python bench.py $COMMON > $O/a$pass.log 2>&1; grep '^{' $O/a$pass.log | tail -1 > $O/a$pass.json
  Y5_TUNE_CACHE=/tmp/tc_b.json timeout 900 python bench.py $COMMON > $O/b$pass.log 2>&1; grep '^{' $O/b$pass.log | tail -1 > $O/b$pass.json
done
python - <<PY | tee $O/summary.log
import json
for n in ("off1","a1","b1","off2","a2","b2","off3","a3","b3"):
    try:
        d=json.load(open("$O/%s.json"%n))
        print(n, "value", round(d["value"]), "ms_per_step", d.get("ms_per_step"), "fwd_ms", d.get("forward_ms"), "sustained", d.get("config",{}).get("gpu_state",{}).get("mfma_sustained_tflops"),
              "swaps", d.get("config",{}).get("tile_choices",{}).get("in_situ_swaps"))
    except Exception as e:
        print(n, "failed", e)
PY
