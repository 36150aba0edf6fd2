#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run24; mkdir -p $O
export Y5_TUNE_CACHE=/tmp/tc_b.json
run() { tag=$1; shift; env BISECT_TAG=$tag "$@" timeout 300 python scripts/r5_train_bisect.py 2>&1 | grep "gradient rel" | cut -c1-230; }
run s1024 BISECT_SCALE=1024
run s8192 BISECT_SCALE=8192
run s65536 BISECT_SCALE=65536
run s128 BISECT_SCALE=128
