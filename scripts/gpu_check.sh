#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel-trace summary.  Logs -> gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT="${1:-all}"
if [ "$WHAT" = "all" ] || [ "$WHAT" = "test" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
  tail -25 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "bench" ]; then
  timeout 600 python bench.py --op-table gpurun_out/op_table.json > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/bench.log
  tail -5 gpurun_out/bench.log
  timeout 300 python scripts/metrics_bench.py > gpurun_out/metrics_bench.log 2>&1; tail -2 gpurun_out/metrics_bench.log
  timeout 300 python scripts/letterbox_bench.py > gpurun_out/letterbox_bench.log 2>&1; tail -2 gpurun_out/letterbox_bench.log
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "prof" ]; then
  rm -rf gpurun_out/prof
  # tile choices cached by a first plain run: the profiled run launches no autotune timing kernels
  export Y5_TUNE_CACHE=/tmp/y5_tune_prof.json
  [ -n "$Y5_SEED_TUNE_CACHE" ] && [ -f "$Y5_SEED_TUNE_CACHE" ] && cp "$Y5_SEED_TUNE_CACHE" "$Y5_TUNE_CACHE"   # (scripts/r6_final.sh: the timed run's choices)
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-configs --no-selfcheck > /dev/null 2>&1
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o r -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-configs --no-selfcheck > "$OLDPWD/gpurun_out/prof.log" 2>&1); echo "prof rc=$?"
  find gpurun_out/prof -name "*stats*" | head; 
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
  t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
  gb=$(python -c "import json,re,sys; l=[x for x in open('gpurun_out/prof.log') if x.startswith('{')][-1]; print(json.loads(l)['roofline']['algorithmic_gbytes_per_step'])")
  gf=$(python -c "import json,re,sys; l=[x for x in open('gpurun_out/prof.log') if x.startswith('{')][-1]; print(json.loads(l)['roofline']['algorithmic_gflop_per_step'])")
  [ -n "$t" ] && python scripts/rocprof_frac.py "$t" --gbytes "$gb" --gflop "$gf" --out gpurun_out/prof/rocprof_frac.json | head -30
  grep '^{' gpurun_out/prof.log | tail -1 > gpurun_out/prof/bench_line.json
  # keep the merged-back directory small: drop the raw per-dispatch trace, keep the summaries
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "train" ]; then
  # training step: per-kernel summary with warm tile / split choices (the first plain run fills the cache)
  export Y5_TUNE_CACHE=/tmp/y5_tune_train.json
  [ -n "$Y5_SEED_TUNE_CACHE" ] && [ -f "$Y5_SEED_TUNE_CACHE" ] && cp "$Y5_SEED_TUNE_CACHE" "$Y5_TUNE_CACHE"
  timeout 300 python scripts/train_bench.py --steps 5 --warmup 3 > gpurun_out/train_bench.log 2>&1; tail -1 gpurun_out/train_bench.log
  rm -rf gpurun_out/trainprof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/trainprof" -o t -- python "$OLDPWD/scripts/train_bench.py" --steps 10 --warmup 3 > "$OLDPWD/gpurun_out/train_prof.log" 2>&1); echo "train prof rc=$?"
  f=$(find gpurun_out/trainprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/train_kernel_stats.csv && head -12 "$f" | cut -c1-150
  grep images/sec gpurun_out/train_prof.log | tail -1 > gpurun_out/train_line_under_rocprof.json
  rm -rf gpurun_out/trainprof
fi
