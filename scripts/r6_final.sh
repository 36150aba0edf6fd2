#!/bin/bash
# Round-6 measurement pass (one gpurun call): GPU tests, smoke, bench line + op table (yolov5s) + yolov5x op table, PMC traffic + issue mix of the TIMED
# (export-mode) plan, rocprofv3 kernel statistics of the bench and of the training step, the 8-phase family's layer bench.
# Everything lands in gpurun_out/r06_final/ ; copy what is to be judged to profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06_final; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
export Y5_TUNE_CACHE=/tmp/tc_final.json
timeout 900 python bench.py --op-table $O/op_table.json > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 > $O/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1; grep '^{' $O/bench_driver_cmd.log | tail -1 > $O/bench_driver_cmd.json
timeout 600 python bench.py --model yolov5x --batch 16 --imgsz 1280 --no-train --no-pipeline --no-configs --no-cpu-baseline --no-selfcheck --steps 20 --warmup 5 --op-table $O/op_table_yolov5x.json > $O/bench_yolov5x.log 2>&1; grep '^{' $O/bench_yolov5x.log | tail -1 > $O/bench_yolov5x.json
cp /tmp/tc_final.json $O/tune_db.json   # -> yolov5_amd/tune_db.json (engine._load_tune_cache): the choices of the plans timed above, bound to this library build
export Y5_SEED_TUNE_CACHE=/tmp/tc_final.json
unset Y5_TUNE_CACHE
bash scripts/pmc_forward.sh > $O/pmc_forward.log 2>&1; cp gpurun_out/pmc_forward.json $O/ 2>/dev/null; tail -14 $O/pmc_forward.log
bash scripts/pmc_issue_mix.sh > $O/pmc_issue_mix.log 2>&1; cp gpurun_out/pmc_issue_mix.json $O/ 2>/dev/null; grep mfma_busy_frac $O/pmc_issue_mix.log | head -2 | cut -c1-300
bash scripts/gpu_check.sh prof > $O/prof.log 2>&1; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv
cp gpurun_out/prof/rocprof_frac.json gpurun_out/prof/bench_line.json $O/ 2>/dev/null
bash scripts/gpu_check.sh train > $O/train.log 2>&1; cp gpurun_out/train_kernel_stats.csv gpurun_out/train_bench.log gpurun_out/train_line_under_rocprof.json $O/ 2>/dev/null; tail -2 $O/train.log | cut -c1-300
timeout 600 python scripts/g8_bench.py --out $O/g8_bench.json 2>&1 | grep -v amdgpu.ids > $O/g8_bench.log; tail -3 $O/g8_bench.log | cut -c1-250
rm -rf gpurun_out/prof gpurun_out/pmcm_* gpurun_out/pmcf_*
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("value", d["value"], "at ref clock", d.get("value_at_ref_clock"), "fwd", d["forward_ms"], "stack_frac", d["roofline"]["stack_frac"], "dominant", d["roofline"]["kernel"][:60], d["roofline"]["frac"])
print("traffic", json.dumps(d["roofline"]["traffic"])[:300]); print("busy", json.dumps(d["roofline"]["mfma_busy_frac"])[:300])
print("configs", json.dumps(d.get("configs"))[:900]); print("train", json.dumps(d.get("train"))[:500])
PY
