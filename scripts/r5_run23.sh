#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run23; mkdir -p $O
T=$(date +%s)
BISECT_TAG=box$T BISECT_PLAN=$O/plan_$T.json timeout 300 python scripts/r5_train_bisect.py 2>&1 | grep "gradient rel" | cut -c1-200 | tee $O/line_$T.log
grep -m1 "model name" /proc/cpuinfo | tee -a $O/line_$T.log
