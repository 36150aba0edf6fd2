#!/bin/bash
# Round 5: halo-resident 3x3 at stride 2 (conv_h3.h SD = 2, ids 61..77 + the small-tile ids 90..92) on the down-sampling layers -- per-configuration
# timing, GPU parity of the new ids, bench A/B (Y5_H3_S2=0 rejects stride 2 in the halo family: the tuner then picks what round 4 had)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run15; rm -rf $O; mkdir -p $O
timeout 400 python scripts/conv_bench.py --only "3.Conv,5.Conv,7.Conv,18.Conv,21.Conv" > $O/conv_bench_s2.log 2>&1; grep -v amdgpu $O/conv_bench_s2.log | cut -c1-1300
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "test_conv_matches_torch_fp32_reference" > $O/pytest_conv.log 2>&1; tail -3 $O/pytest_conv.log
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --no-selfcheck --op-table $O/op_$tag.json 2>$O/bench_$tag.err | grep '^{' > $O/bench_$tag.json; python -c "
import json,sys
d=json.loads(open('$O/bench_$tag.json').read()); print('$tag', d['value'], d['ms_per_step'], d['forward_ms'], d['roofline'].get('stack_frac'))"; }
run off1 Y5_H3_S2=0 Y5_TUNE_CACHE=/tmp/tc_off.json
run on1 Y5_H3_S2=1 Y5_TUNE_CACHE=/tmp/tc_on.json
run off2 Y5_H3_S2=0 Y5_TUNE_CACHE=/tmp/tc_off.json
run on2 Y5_H3_S2=1 Y5_TUNE_CACHE=/tmp/tc_on.json
