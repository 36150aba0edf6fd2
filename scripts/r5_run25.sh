#!/bin/bash
# Round 5, GPU call: C3-tail fusion at c_ = 128 (conv_h3b.h CV3 form: last Bottleneck + cv3 in one launch) -- parity, bench A/B (same box, alternating arms)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run25; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "cv3 or c128 or bneck128" > $O/pytest_cv3.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_cv3.log; tail -3 $O/pytest_cv3.log
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --no-selfcheck --op-table $O/op_$tag.json 2>$O/bench_$tag.err | grep '^{' > $O/bench_$tag.json; python -c "
import json,sys
d=json.loads(open('$O/bench_$tag.json').read()); print('$tag', d['value'], d['ms_per_step'], d['forward_ms'], d['roofline'].get('stack_frac'), d.get('launches_per_forward'))"; }
run off1 Y5_FUSED_CV3_128=0 Y5_TUNE_CACHE=/tmp/tc_off.json
run on1 Y5_FUSED_CV3_128=1 Y5_TUNE_CACHE=/tmp/tc_on.json
run off2 Y5_FUSED_CV3_128=0 Y5_TUNE_CACHE=/tmp/tc_off.json
run on2 Y5_FUSED_CV3_128=1 Y5_TUNE_CACHE=/tmp/tc_on.json
run auto1 Y5_FUSED_CV3_128=auto Y5_TUNE_CACHE=/tmp/tc_auto.json
python - <<'PY' | tee $O/op_compare.log
import json
O="gpurun_out/r05_run25"
for tag in ("off2","on2","auto1"):
    try:
        d=json.load(open(f"{O}/op_{tag}.json"))
    except Exception as e:
        print(tag, "no op table", e); continue
    rows = d["ops"] if isinstance(d, dict) and "ops" in d else d
    print("==", tag)
    for r in rows:
        n = r.get("name") or r.get("op")
        if any(k in str(n) for k in ("6.C3", "13.C3", "20.C3")) and ("cv3" in str(n) or "bneck128" in str(n)):
            print("  ", n, r.get("cfg"), r.get("ms_in_situ", r.get("ms")))
PY
