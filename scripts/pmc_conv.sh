#!/bin/bash
# PMC passes (counters only; never combined with tracing domains other than kernel-trace) over scripts/conv_bench.py for one layer.
# usage: pmc_conv.sh "<layer substring>"   -> gpurun_out/pmc_<n>/...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L="$1"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OLDPWD/gpurun_out/pmc_$i" -o p -- python "$OLDPWD/scripts/conv_bench.py" --only "$L" --iters 3 > "$OLDPWD/gpurun_out/pmc_$i.log" 2>&1)
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
        if r['Counter_Name'] in ('SQ_WAVES','SQ_INSTS_LDS','TCC_HIT_sum','FETCH_SIZE','WRITE_SIZE','GRBM_GUI_ACTIVE'): cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if 'conv' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        n = max(cnt.get((k, c), 0), 1)
        print(f"   {c:28s} total={v:.4g}")
PY
