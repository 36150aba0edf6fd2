#!/bin/bash
# Instruction-issue mix of every forward kernel (two SQ counter passes, --kernel-trace only) -> gpurun_out/pmc_issue_mix.json
# per kernel: launches per forward, VALU / MFMA / LDS / VMEM instructions per wave, share of busy cycles with a VALU or an MFMA instruction in flight.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp Y5_TUNE_CACHE=/tmp/tc_fwd.json Y5_GRAPH=0
export Y5_FUSED_K3PW=1 Y5_FUSED_CV3=1 Y5_FUSED_HEAD=1
N=5
python scripts/forward_only.py 2 > /dev/null 2>&1
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcm_$i
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OLDPWD/gpurun_out/pmcm_$i" -o p -- python "$OLDPWD/scripts/forward_only.py" $N > "$OLDPWD/gpurun_out/pmcm_$i.log" 2>&1)
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections, json
N = $N
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmcm_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'y5_' not in k: continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
out = []
for k, d in tot.items():
    w = d.get('SQ_WAVES', 0) or 1
    busy = d.get('SQ_BUSY_CYCLES', 0) or 1
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1
    out.append({"kernel": k[:110], "launches_per_forward": round(cnt[k] / N, 1), "waves_per_launch": round(w / max(cnt[k], 1)),
                "valu_per_wave": round(d.get('SQ_INSTS_VALU', 0) / w), "mfma_per_wave": round(d.get('SQ_INSTS_MFMA', 0) / w),
                "lds_per_wave": round(d.get('SQ_INSTS_LDS', 0) / w), "vmem_per_wave": round(d.get('SQ_INSTS_VMEM', 0) / w),
                "salu_per_wave": round(d.get('SQ_INSTS_SALU', 0) / w),
                "valu_per_mfma": round(d.get('SQ_INSTS_VALU', 0) / max(d.get('SQ_INSTS_MFMA', 0), 1), 2),
                "active_valu_over_wave_cycles": round(d.get('SQ_ACTIVE_INST_VALU', 0) / wc, 3),
                "mfma_busy_over_busy_cycles": round(d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy, 3),
                "wait_inst_over_wave_cycles": round(d.get('SQ_WAIT_INST_ANY', 0) / wc, 3)})
out.sort(key=lambda r: -r["launches_per_forward"] * r["waves_per_launch"] * (r["valu_per_wave"] + 8 * r["mfma_per_wave"]))
json.dump(out, open('gpurun_out/pmc_issue_mix.json', 'w'), indent=1)
for r in out[:14]: print(r)
PY
find gpurun_out/pmcm_* -name "*.csv" -size +5M -delete
