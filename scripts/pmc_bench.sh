#!/bin/bash
# PMC passes over bench.py (counters + kernel-trace only).  Per-kernel per-launch averages -> gpurun_out/pmc_bench.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcb_$i
  (cd /tmp && Y5_AUTOTUNE_ITERS=2 timeout 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OLDPWD/gpurun_out/pmcb_$i" -o p -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-train > "$OLDPWD/gpurun_out/pmcb_$i.log" 2>&1)
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob('gpurun_out/pmcb_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not k.startswith(('y5_', '_Z', 'void y5_')) or 'y5_' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
out = {k: {c: v / n[k][c] for c, v in d.items()} | {"launches_profiled": max(n[k].values())} for k, d in agg.items()}
json.dump(out, open('gpurun_out/pmc_bench.json', 'w'), indent=1)
for k, d in sorted(out.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0))[:14]:
    wc = d.get('SQ_WAVE_CYCLES', 1)
    print(f"{k[:60]:60s} wait={d.get('SQ_WAIT_ANY',0)/wc:.2f} istall={d.get('SQ_WAIT_INST_ANY',0)/wc:.2f} act={d.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} "
          f"valu/mfma={d.get('SQ_INSTS_VALU',0)/max(d.get('SQ_INSTS_MFMA',1),1):.1f} fetchMB={d.get('FETCH_SIZE',0)/1e3:.1f} writeMB={d.get('WRITE_SIZE',0)/1e3:.1f}")
PY
# keep only the summaries
find gpurun_out/pmcb_* -name "*.csv" -size +2M -delete 2>/dev/null
