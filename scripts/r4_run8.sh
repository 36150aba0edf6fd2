cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1 0 1; do
Y5_TUNE_CACHE=/tmp/tc_c$v.json Y5_VIRTUAL_UP=$v timeout 600 python bench.py --no-cpu-baseline --no-train --no-pipeline --no-selfcheck 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['configs']; print('VIRTUAL_UP=$v', d['value'], 'C4', c['C4'].get('images_per_sec'), c['C4'].get('forward_ms'), 'C5', c['C5'].get('images_per_sec'), c['C5'].get('forward_ms'))"
done 2>&1 | tee gpurun_out/r04_ab_virtual_up_c4c5.log
