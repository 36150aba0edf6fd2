#!/bin/bash
# Round 5: what the one-rank RCCL exchange step actually launches (rocprofv3 kernel trace of scripts/r5_ddp_reserve.py, reserve = 0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_run9; rm -rf $O; mkdir -p $O
export Y5_TUNE_CACHE=/tmp/tc_r.json
timeout 600 python scripts/r5_ddp_reserve.py 0 > $O/warm.log 2>&1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $GRAFT_REPO_ROOT/scripts/r5_ddp_reserve.py 0 > $O/prof.log 2>&1); echo "prof rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -1 $f; grep -i "nccl\|rccl\|copy\|memcpy" $f | cut -c1-250 | head; grep -c . $f
t=$(find $O/prof -name "*kernel_trace.csv" | head -1); python - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
n=[r for r in rows if 'nccl' in r['Kernel_Name'].lower() or 'rccl' in r['Kernel_Name'].lower()]
print(len(rows),'dispatches;',len(n),'nccl kernels')
for r in n[:12]:
    print(r['Kernel_Name'][:80], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'us', 'grid',r.get('Grid_Size'),'wg',r.get('Workgroup_Size'))
PY
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.csv" -size +2M -delete
