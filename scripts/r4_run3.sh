cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export Y5_TUNE_CACHE=/tmp/tc_main.json
timeout 600 python bench.py --no-configs --no-pipeline --no-cpu-baseline > gpurun_out/r04_bench_v1.log 2>&1; echo "bench rc=$?"
grep '^{' gpurun_out/r04_bench_v1.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], d['forward_ms']); print({k:r[k] for k in ('kernel','achieved','frac','stack_frac','mfma_busy_frac','dominant_kernel')}); print(d['train']); print(d['config'].get('plan_sha16'))
" || tail -20 gpurun_out/r04_bench_v1.log
bash scripts/pmc_issue_mix.sh > gpurun_out/r04_pmc_issue_mix_v1.log 2>&1; tail -18 gpurun_out/r04_pmc_issue_mix_v1.log | cut -c1-420
cp gpurun_out/pmc_issue_mix.json gpurun_out/r04_pmc_issue_mix_v1.json
# plain kernel trace of the graph replay: launch gaps
rm -rf gpurun_out/trace1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/trace1" -o t -- python "$OLDPWD/scripts/forward_only.py" 12 > "$OLDPWD/gpurun_out/trace1.log" 2>&1)
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/trace1/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'y5_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'front' in r['Kernel_Name']]
s,e=idx[-2],idx[-1]
tot=gaps=0; prev=None; out=[]
for r in rows[s:e]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    g=(st-prev) if prev else 0
    out.append((r['Kernel_Name'][:60],(en-st)/1e3,g/1e3)); tot+=en-st; gaps+=g; prev=max(prev or 0,en)
for o in out: print(f"{o[0]:60s} {o[1]:7.1f} gap {o[2]:6.1f}")
print('n',e-s,'sum dur us',tot/1e3,'sum gaps us',gaps/1e3,'span us',(int(rows[e-1]['End_Timestamp'])-int(rows[s]['Start_Timestamp']))/1e3)
PY
find gpurun_out/trace1 -name "*.csv" -size +5M -delete
