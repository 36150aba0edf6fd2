#!/bin/bash
# Round 5: DetectPipeline with the NMS chain on a high-priority side stream (default) vs on the caller's stream (Y5_PIPE_OVERLAP=0); same box, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run13; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; env "$@" Y5_TUNE_CACHE=/tmp/tc_on.json timeout 400 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --no-selfcheck 2>$O/bench_$tag.err | grep '^{' > $O/bench_$tag.json; python -c "
import json,sys
d=json.loads(open('$O/bench_$tag.json').read()); print('$tag', d['value'], d['ms_per_step'], d['forward_ms'], d.get('alt_step_mode',{}).get('ms_per_step'))"; }
run side1 Y5_PIPE_OVERLAP=1
run same1 Y5_PIPE_OVERLAP=0
run side2 Y5_PIPE_OVERLAP=1
run same2 Y5_PIPE_OVERLAP=0
run side3 Y5_PIPE_OVERLAP=1
run same3 Y5_PIPE_OVERLAP=0
