cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
run() { env "$@" Y5_TUNE_CACHE=/tmp/tc_env.json timeout 300 python bench.py --no-cpu-baseline --no-train --no-configs --no-pipeline --no-selfcheck 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['forward_ms'])"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run Y5_GRAPH=0
