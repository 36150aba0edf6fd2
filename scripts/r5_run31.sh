#!/bin/bash
# Round 5, GPU call: SPPF pool backward, fixed-point scatter form -- parity + repeatability, timing of the forms, deterministic training step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run31; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_train_ops.py -q -k sppf_pool_bwd 2>&1 | tail -3 | tee $O/pytest_poolbwd.log
for v in "Y5_SPPF_BWD_GV=1" "Y5_SPPF_BWD_GV=2" "Y5_SPPF_BWD_GV=4" "Y5_SPPF_BWD_GATHER=1" "Y5_SPPF_BWD_GATHER=1 Y5_SPPF_BWD_GV=1"; do env $v timeout 120 python scripts/poolbwd_bench.py 2>&1 | grep "per launch" | tee -a $O/poolbwd_bench.log; done
RUNS=3 timeout 600 python scripts/r5_det_check.py > $O/det_check.log 2>&1; grep "^losses\|^run" $O/det_check.log
timeout 600 python bench.py --no-cpu-baseline --no-configs --no-pipeline --no-selfcheck 2>$O/bench.err | grep '^{' > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['forward_ms'], 'train', d['train']['ms_per_step'])"
