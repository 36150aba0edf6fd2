#!/bin/bash
# SQ counters of the fused backbone front (and of the stem / conv+pw pair it replaces) from scripts/front_bench.py under rocprofv3 (separate --pmc
# passes, --kernel-trace only) -> gpurun_out/pmc_front.json: instructions per wave by class, wait / issue-stall / active shares, LDS bank conflicts.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcfr_$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OLDPWD/gpurun_out/pmcfr_$i" -o p -- python "$OLDPWD/scripts/front_bench.py" --iters 3 > "$OLDPWD/gpurun_out/pmcfr_$i.log" 2>&1)
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmcfr_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'y5_conv' not in k: continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVES', 'SQ_INSTS_VALU', 'GRBM_GUI_ACTIVE'): cnt[(k, r['Counter_Name'])] += 1
for f in glob.glob('gpurun_out/pmcfr_1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'y5_conv' in r['Kernel_Name']: dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = []
for k, d in tot.items():
    n1 = cnt[(k, 'SQ_WAVES')] or 1; n2 = cnt[(k, 'SQ_INSTS_VALU')] or 1; n3 = cnt[(k, 'GRBM_GUI_ACTIVE')] or 1
    w = d['SQ_WAVES'] / n1 or 1; wc = d['SQ_WAVE_CYCLES'] / n1 or 1
    o = {"kernel": k[:90], "launches": n1, "avg_us_under_pmc": round(sum(dur[k]) / max(len(dur[k]), 1), 1), "waves": round(w)}
    for c in ('SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SALU'): o[c.lower() + "_per_wave"] = round(d[c] / n2 / w, 1)
    for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU'): o[c.lower() + "_over_wave_cycles"] = round(d[c] / n1 / wc, 3)
    o["wave_quad_cycles_per_wave"] = round(wc / w)
    o["mfma_busy_cycles_per_simd"] = round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / n1 / 1024)
    o["busy_cycles"] = round(d['SQ_BUSY_CYCLES'] / n1)
    o["lds_bank_conflict_over_idx_active"] = round(d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1), 3)
    o["lds_idx_active_per_cu"] = round(d['SQ_LDS_IDX_ACTIVE'] / n2 / 256)
    o["gui_active_cycles"] = round(d['GRBM_GUI_ACTIVE'] / n3)
    o["wait_inst_lds_over_wave_cycles"] = round(d['SQ_WAIT_INST_LDS'] / n3 / wc, 3)
    out.append(o)
json.dump(out, open('gpurun_out/pmc_front.json', 'w'), indent=1)
for r in out: print(r)
PY
find gpurun_out/pmcfr_* -name "*.csv" -size +2M -delete
