#!/bin/bash
# Round 5, GPU call: (1) probe of dummy LDS-DMA loads (compiler merge + retire order); (2) parity of every kernel that issues dummies after the
# y5_bglds16_dummy fix; (3) forward bench A/B for the C3-tail fusion at c_ = 128 with the fixed kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run27; rm -rf $O; mkdir -p $O
timeout 120 scripts/probes/oob_retire 2>&1 | tee $O/oob_retire.log
timeout 300 python scripts/cv3_dbg.py > $O/cv3_dbg.log 2>&1; grep -c "^ok" $O/cv3_dbg.log; grep "^FAIL" $O/cv3_dbg.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_head.py -q -x > $O/pytest_parity.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_parity.log; tail -3 $O/pytest_parity.log
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --no-selfcheck --op-table $O/op_$tag.json 2>$O/bench_$tag.err | grep '^{' > $O/bench_$tag.json; python -c "
import json,sys
d=json.loads(open('$O/bench_$tag.json').read()); print('$tag', d['value'], d['ms_per_step'], d['forward_ms'], d['roofline'].get('stack_frac'))"; }
run off1 Y5_FUSED_CV3_128=0 Y5_TUNE_CACHE=/tmp/tc_off.json
run on1 Y5_FUSED_CV3_128=1 Y5_TUNE_CACHE=/tmp/tc_on.json
run off2 Y5_FUSED_CV3_128=0 Y5_TUNE_CACHE=/tmp/tc_off.json
run on2 Y5_FUSED_CV3_128=1 Y5_TUNE_CACHE=/tmp/tc_on.json
