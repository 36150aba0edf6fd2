cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "virtual_upsample" 2>&1 | tail -3
for v in 0 1 0 1; do
Y5_TUNE_CACHE=/tmp/tc_v$v.json Y5_VIRTUAL_UP=$v timeout 300 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --op-table gpurun_out/r04_op_table_vup$v.json 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('VIRTUAL_UP=$v', d['value'], d['ms_per_step'], d['forward_ms'], d['selfcheck']['ok'], d['roofline']['stack_frac'])"
done
python - <<'PY'
import json
a=json.load(open('gpurun_out/r04_op_table_vup0.json')); b=json.load(open('gpurun_out/r04_op_table_vup1.json'))
for x,y in zip(a,b):
    if abs(x['ms']-y['ms'])>0.004 or x['cfg']!=y['cfg']: print(f"{x['op']:36s} cfg {x['cfg']}->{y['cfg']}  {x['ms']*1e3:7.1f} -> {y['ms']*1e3:7.1f} us")
print(sum(x['ms'] for x in a), sum(y['ms'] for y in b))
PY
