#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run20; mkdir -p $O
run() { tag=$1; shift; env BISECT_TAG=$tag "$@" timeout 300 python scripts/r5_train_bisect.py 2>&1 | grep "gradient rel" | cut -c1-420; }
run default
run stats0 Y5_BN_FUSED_STATS=0
run skipnew Y5_AUTOTUNE_SKIP=90-94 Y5_H3_S2=0
run stats0_skipnew Y5_BN_FUSED_STATS=0 Y5_AUTOTUNE_SKIP=90-94 Y5_H3_S2=0
run hyst0 Y5_TUNE_HYST=0
