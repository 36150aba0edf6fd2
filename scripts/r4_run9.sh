cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1 0 1; do
Y5_TUNE_CACHE=/tmp/tc_b$v.json Y5_FUSED_BNECK64=$v timeout 300 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --op-table gpurun_out/r04_op_table_bn64_$v.json 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BNECK64=$v', d['value'], d['ms_per_step'], d['forward_ms'], d['selfcheck']['ok'], d['roofline']['stack_frac'], d['selfcheck'].get('box_rel_err_mean'))"
done 2>&1 | tee gpurun_out/r04_ab_bneck64.log
python - <<'PY'
import json
a=json.load(open('gpurun_out/r04_op_table_bn64_0.json')); b=json.load(open('gpurun_out/r04_op_table_bn64_1.json'))
for name,t in (('off',a),('on',b)):
    print(name, [(x['op'].split(':',1)[1][:22], round(x['ms']*1e3,1)) for x in t if ('4.C3' in x['op'] or '17.C3' in x['op'] or x['op'] in ('conv:b.cv1','conv:b.cv2')) ][:24])
print(sum(x['ms'] for x in a), sum(y['ms'] for y in b))
PY
timeout 900 python -m pytest tests/test_gpu_plans.py -q -x -k "C2" 2>&1 | tail -2
