cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export Y5_TUNE_CACHE=/tmp/tc_main.json
timeout 600 python bench.py --op-table gpurun_out/r04_op_table_v0.json > gpurun_out/r04_bench_v0.log 2>&1; echo "bench rc=$?"
grep '^{' gpurun_out/r04_bench_v0.log | tail -1 | cut -c1-600
Y5_SPLIT=2 timeout 300 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline > gpurun_out/r04_bench_split2.log 2>&1; echo "split rc=$?"
grep '^{' gpurun_out/r04_bench_split2.log | tail -1 | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline > gpurun_out/r04_bench_nosplit.log 2>&1
grep '^{' gpurun_out/r04_bench_nosplit.log | tail -1 | cut -c1-400
bash scripts/pmc_issue_mix.sh > gpurun_out/r04_pmc_issue_mix_v0.log 2>&1; tail -16 gpurun_out/r04_pmc_issue_mix_v0.log | cut -c1-400
