#!/bin/bash
# Round 5: Detect conv + decode fused at P4 / P5 (conv_headk.h) -- GPU parity, bench A/B (Y5_FUSED_HEAD_DEEP = 0 / 1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run17; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_head.py -q -x > $O/pytest_head.log 2>&1; tail -3 $O/pytest_head.log
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --no-selfcheck --op-table $O/op_$tag.json 2>$O/bench_$tag.err | grep '^{' > $O/bench_$tag.json; python -c "
import json,sys
d=json.loads(open('$O/bench_$tag.json').read()); print('$tag', d['value'], d['ms_per_step'], d['forward_ms'], d['roofline'].get('stack_frac'))
t=json.load(open('$O/op_$tag.json')); print('   ', [(r['op'][:26], round(r['ms']*1e3,1)) for r in t if 'detect' in r['op'] or 'decode' in r['op']])"; }
run off1 Y5_FUSED_HEAD_DEEP=0 Y5_TUNE_CACHE=/tmp/tc_off.json
run on1 Y5_FUSED_HEAD_DEEP=1 Y5_TUNE_CACHE=/tmp/tc_on.json
run off2 Y5_FUSED_HEAD_DEEP=0 Y5_TUNE_CACHE=/tmp/tc_off.json
run on2 Y5_FUSED_HEAD_DEEP=1 Y5_TUNE_CACHE=/tmp/tc_on.json
