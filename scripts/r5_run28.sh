#!/bin/bash
# Round 5, GPU call: timing A/B of the y5_bglds16_dummy fix (same box, same plan, alternating arms): library with un-mergeable dummies (product) against
# the pre-fix build (scripts/build_dummy_ab.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run28; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --no-selfcheck --op-table $O/op_$tag.json 2>$O/bench_$tag.err | grep '^{' > $O/bench_$tag.json; python -c "
import json,sys
d=json.loads(open('$O/bench_$tag.json').read()); print('$tag', d['value'], d['ms_per_step'], d['forward_ms'], d['roofline'].get('stack_frac'))"; }
run fixed1 Y5_TUNE_CACHE=/tmp/tc.json
run merge1 Y5_TUNE_CACHE=/tmp/tc.json Y5_LIB_PATH=yolov5_amd/libyolov5_hip_mergedummy.so
run fixed2 Y5_TUNE_CACHE=/tmp/tc.json
run merge2 Y5_TUNE_CACHE=/tmp/tc.json Y5_LIB_PATH=yolov5_amd/libyolov5_hip_mergedummy.so
run fixed3 Y5_TUNE_CACHE=/tmp/tc.json
run merge3 Y5_TUNE_CACHE=/tmp/tc.json Y5_LIB_PATH=yolov5_amd/libyolov5_hip_mergedummy.so
