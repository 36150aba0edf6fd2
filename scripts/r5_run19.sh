#!/bin/bash
# Round 5: K-streamed pointwise kernel (conv_pwk.h, ids 93 / 94) on the deep 1x1 layers -- per-configuration timing, GPU parity, bench A/B (Y5_AUTOTUNE_SKIP=93-94)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run19; rm -rf $O; mkdir -p $O
timeout 400 python scripts/conv_bench.py --only "6.cv1+cv2,14.Conv,9.SPPF.cv2,8.cv3,8.cv1+cv2,17.cv1+cv2,4.cv3" > $O/conv_bench_pwk.log 2>&1; grep -v amdgpu $O/conv_bench_pwk.log | cut -c1-140; grep -v amdgpu $O/conv_bench_pwk.log | sed 's/.*| //' | tr ' ' '\n' | grep -E "^(43|8|39|93|94|84):" | tr '\n' ' '; echo
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "test_conv_matches_torch_fp32_reference" > $O/pytest_conv.log 2>&1; tail -2 $O/pytest_conv.log
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --no-selfcheck --op-table $O/op_$tag.json 2>$O/bench_$tag.err | grep '^{' > $O/bench_$tag.json; python -c "
import json,sys
d=json.loads(open('$O/bench_$tag.json').read()); print('$tag', d['value'], d['ms_per_step'], d['forward_ms'], d['roofline'].get('stack_frac'))
t=json.load(open('$O/op_$tag.json')); print('   ', [(r['op'][5:22], r.get('cfg'), round(r['ms']*1e3,1)) for r in t if r.get('cfg') in (93, 94)])"; }
run off1 Y5_AUTOTUNE_SKIP=93-94 Y5_TUNE_CACHE=/tmp/tc_off.json
run on1 Y5_TUNE_CACHE=/tmp/tc_on.json
run off2 Y5_AUTOTUNE_SKIP=93-94 Y5_TUNE_CACHE=/tmp/tc_off.json
run on2 Y5_TUNE_CACHE=/tmp/tc_on.json
