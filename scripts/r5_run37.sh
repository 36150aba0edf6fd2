#!/bin/bash
# Round 5, last GPU call: pool-backward test after the non-finite propagation change, then the PMC passes again (they carry the source hash of HEAD)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_run37; rm -rf $O; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_train_ops.py -q -k sppf_pool_bwd 2>&1 | tail -3 | tee $O/pytest_poolbwd.log
timeout 120 python scripts/poolbwd_bench.py 2>&1 | grep "per launch" | tee $O/poolbwd_bench.log
bash scripts/pmc_forward.sh > $O/pmc_forward.log 2>&1; cp gpurun_out/pmc_forward.json $O/ 2>/dev/null; tail -4 $O/pmc_forward.log
bash scripts/pmc_issue_mix.sh > $O/pmc_issue_mix.log 2>&1; cp gpurun_out/pmc_issue_mix.json $O/ 2>/dev/null; grep mfma_busy_frac $O/pmc_issue_mix.log | head -1 | cut -c1-200
rm -rf gpurun_out/pmcm_* gpurun_out/pmcf_*
