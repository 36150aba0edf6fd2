#!/bin/bash
# Round 5: BatchNorm statistics from the convolution epilogue (Y5_BN_FUSED_STATS=1) vs the separate statistics pass (0); training step, same box, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run14; rm -rf $O; mkdir -p $O
export Y5_TUNE_CACHE=/tmp/tc_train.json
timeout 300 python scripts/train_bench.py --steps 3 --warmup 2 > $O/train_warm.log 2>&1; tail -1 $O/train_warm.log | cut -c1-300
for i in 1 2 3; do for f in 0 1; do
  Y5_BN_FUSED_STATS=$f timeout 300 python scripts/train_bench.py --steps 15 --warmup 4 > $O/train_f${f}_$i.log 2>&1; echo "fused_stats=$f $(tail -1 $O/train_f${f}_$i.log | cut -c1-200)"
done; done
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -k "matches_oracle or fp32_training" > $O/pytest_train.log 2>&1; tail -3 $O/pytest_train.log
Y5_STATS_DEBUG=1 timeout 300 python scripts/train_bench.py --steps 1 --warmup 1 2>&1 | grep "stats" | head -70 > $O/stats_layers.log; wc -l $O/stats_layers.log; head -60 $O/stats_layers.log
