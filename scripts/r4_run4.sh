cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export Y5_TUNE_CACHE=/tmp/tc_main.json
for v in 0 1 0 1; do
Y5_HEAD_BRANCH=$v timeout 300 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --no-selfcheck 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('HEAD_BRANCH=$v', d['value'], d['ms_per_step'], d['forward_ms'])"
done
