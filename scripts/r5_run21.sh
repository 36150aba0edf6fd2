#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run21; mkdir -p $O
run() { tag=$1; shift; env BISECT_TAG=$tag "$@" timeout 300 python scripts/r5_train_bisect.py 2>&1 | grep "gradient rel" | cut -c1-300; }
run d1; run d2; run d3
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -s > $O/pytest_train.log 2>&1; grep -E "train plan parity|passed|failed|outside" $O/pytest_train.log | cut -c1-700
