#!/bin/bash
# Round-5 measurement pass (one gpurun call): GPU tests, bench line + op table, PMC traffic + issue mix, rocprofv3 kernel statistics of the bench and of the
# training step, per-configuration timing of the Bottleneck.cv2 layers.  Everything lands in gpurun_out/r05_final/ ; copy what is to be judged to profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_final; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --op-table $O/op_table.json > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 > $O/bench.json
bash scripts/pmc_forward.sh > $O/pmc_forward.log 2>&1; cp gpurun_out/pmc_forward.json $O/ 2>/dev/null; tail -12 $O/pmc_forward.log
bash scripts/pmc_issue_mix.sh > $O/pmc_issue_mix.log 2>&1; cp gpurun_out/pmc_issue_mix.json $O/ 2>/dev/null; grep mfma_busy_frac $O/pmc_issue_mix.log | head -2 | cut -c1-300
bash scripts/gpu_check.sh prof > $O/prof.log 2>&1; cp gpurun_out/prof/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null; cp gpurun_out/prof/rocprof_frac.json gpurun_out/prof/bench_line.json $O/ 2>/dev/null
find gpurun_out/prof -name "*kernel_stats.csv" | head -2; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv
bash scripts/gpu_check.sh train > $O/train.log 2>&1; cp gpurun_out/train_kernel_stats.csv gpurun_out/train_bench.log gpurun_out/train_line_under_rocprof.json $O/ 2>/dev/null; tail -2 $O/train.log | cut -c1-300
timeout 400 python scripts/conv_bench.py --only "8.b.cv2,5.Conv,7.Conv" > $O/conv_bench_bcv2.log 2>&1; tail -4 $O/conv_bench_bcv2.log | cut -c1-400
rm -rf gpurun_out/prof gpurun_out/pmcm_* gpurun_out/pmcf_*
Y5_TUNE_CACHE=/tmp/tc_r.json timeout 900 python scripts/r5_ddp_reserve.py none0 all0 none0 all0 2>&1 | grep "plain\|reserve" > $O/ddp_exchange.log; cat $O/ddp_exchange.log | cut -c1-160
BNECK_ONLY128=1 timeout 300 python scripts/bneck_bench.py 2>&1 | grep -v amdgpu > $O/bneck_bench.log; tail -3 $O/bneck_bench.log
