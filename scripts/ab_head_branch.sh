mkdir -p gpurun_out; export Y5_TUNE_CACHE=/tmp/tune_ab.json
timeout 300 python bench.py --no-train --no-pipeline --no-cpu-baseline > /dev/null 2>&1
for i in 1 2; do for hb in 1 0; do
  Y5_HEAD_BRANCH=$hb timeout 300 python bench.py --no-train --no-pipeline --no-cpu-baseline --steps 100 > gpurun_out/bench_hb$hb.log 2>&1
  echo "branch=$hb $(tail -1 gpurun_out/bench_hb$hb.log | grep -o "\"value\": [0-9.]*\|\"forward_ms\": [0-9.]*\|\"conv_ms_per_step\": [0-9.]*" | tr "\n" " ")"
done; done
export TMPDIR=/tmp; R=$PWD
(cd /tmp && Y5_HEAD_BRANCH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_nb -o r -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-pipeline > $R/gpurun_out/prof_nb.log 2>&1)
t=$(find gpurun_out/prof_nb -name "*kernel_trace.csv" | head -1)
python scripts/rocprof_frac.py $t --gbytes 7.314 --out gpurun_out/prof_nb/rocprof_frac.json | head -9
grep "^{" gpurun_out/prof_nb.log | tail -1 | grep -o "\"forward_ms\": [0-9.]*\|\"conv_ms_per_step\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "
find gpurun_out/prof_nb -name "*kernel_trace.csv" -delete
