#!/bin/bash
# Round 5, GPU call: conv_h3b.h after the bias / select fix -- phase stamps, parity, isolated timing, bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run6; rm -rf $O; mkdir -p $O
Y5_LIB_PATH=yolov5_amd/libyolov5_hip_h3bdbg.so timeout 300 python scripts/h3b_timing.py 2>&1 | tee $O/h3b_timing.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "c128 or bneck128" > $O/pytest_h3b.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_h3b.log; tail -3 $O/pytest_h3b.log
BNECK_ONLY128=1 timeout 300 python scripts/bneck_bench.py > $O/bneck_bench.log 2>&1; tail -6 $O/bneck_bench.log
timeout 300 python scripts/conv_bench.py --only "8.b.cv2,6.b.cv2" > $O/conv_bench_h3.log 2>&1; tail -3 $O/conv_bench_h3.log | cut -c1-700
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --no-selfcheck --op-table $O/op_$tag.json 2>$O/bench_$tag.err | grep '^{' > $O/bench_$tag.json; python -c "
import json,sys
d=json.loads(open('$O/bench_$tag.json').read()); print('$tag', d['value'], d['ms_per_step'], d['forward_ms'], d['roofline'].get('stack_frac'))"; }
run off1 Y5_FUSED_BNECK128=0 Y5_TUNE_CACHE=/tmp/tc_off.json
run on1 Y5_FUSED_BNECK128=1 Y5_TUNE_CACHE=/tmp/tc_on.json
run off2 Y5_FUSED_BNECK128=0 Y5_TUNE_CACHE=/tmp/tc_off.json
run on2 Y5_FUSED_BNECK128=1 Y5_TUNE_CACHE=/tmp/tc_on.json
