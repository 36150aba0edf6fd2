#!/bin/bash
# Round 5: training step A/B -- BatchNorm statistics passes walking the tensor from its end (Y5_BN_REV=1) vs front to back (0); same box, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run7; rm -rf $O; mkdir -p $O
export Y5_TUNE_CACHE=/tmp/tc_train.json
timeout 300 python scripts/train_bench.py --steps 3 --warmup 2 > $O/train_warm.log 2>&1; tail -1 $O/train_warm.log | cut -c1-300
for i in 1 2 3; do for rev in 0 1; do
  Y5_BN_REV=$rev timeout 300 python scripts/train_bench.py --steps 15 --warmup 4 > $O/train_rev${rev}_$i.log 2>&1; echo "rev=$rev $(tail -1 $O/train_rev${rev}_$i.log | cut -c1-200)"
done; done
