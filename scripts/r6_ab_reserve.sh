#!/bin/bash
# Round 6 experiment: CUs left free by the forward plan's persistent grids for the NMS chain of the previous batch (DetectPipeline's side stream).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06_ab_reserve; rm -rf $O; mkdir -p $O
COMMON="--no-train --no-pipeline --no-cpu-baseline --no-selfcheck --no-configs --steps 50 --warmup 10"
for pass in 1 2; do
  for r in 0 8 16; do
    Y5_BENCH_RESERVE_CUS=$r Y5_TUNE_CACHE=/tmp/tc_r$r.json timeout 600 python bench.py $COMMON > $O/r${r}_$pass.log 2>&1
    grep '^{' $O/r${r}_$pass.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('reserve $r pass $pass: value', d['value'], 'ms_per_step', d['ms_per_step'], 'forward_ms', d['forward_ms'], 'alt', d['alt_step_mode']['ms_per_step'])"
  done
done | tee $O/summary.log
