cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2j; mkdir -p $O
bash scripts/gpu_check.sh prof > $O/prof_stdout.log 2>&1
cp -r gpurun_out/prof $O/prof; cp gpurun_out/prof.log $O/prof.log
timeout 600 python bench.py --op-table $O/op_table.json > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-600
bash scripts/pmc_forward.sh > $O/pmc_stdout.log 2>&1; cp gpurun_out/pmc_forward.json $O/; tail -12 $O/pmc_stdout.log
timeout 600 bash scripts/bneck_ablate.sh > $O/bneck_ablation.txt 2>&1; tail -20 $O/bneck_ablation.txt
timeout 600 bash scripts/nms_ablate.sh > $O/nms_ablation.txt 2>&1; tail -8 $O/nms_ablation.txt
