#!/bin/bash
# A/B inside one gpurun call (same box): training step with the general weight-gradient kernel only (Y5_WGRAD_CFG=1) against the timed choice between
# it and the patch-staged 3x3 family (default), each with its own fresh tune cache; then the training-path GPU tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 1 -2 1 -2; do
  export Y5_TUNE_CACHE=/tmp/y5_tune_train_$v.json
  echo -n "Y5_WGRAD_CFG=$v: "; Y5_WGRAD_CFG=$v timeout 300 python scripts/train_bench.py --steps 10 --warmup 3 2>&1 | grep images/sec | tail -1 | cut -c1-220
done
python - <<'P'
import json
d = json.load(open('/tmp/y5_tune_train_-2.json'))
for k, v in d.items():
    if k.startswith('(-7002') or k.startswith('[-7002') or '-7002' in k[:8]:
        c = v[0] if isinstance(v, list) else v
        print(k, '-> cfg', (c >> 20) - 1, 'splits', c & 0xFFFFF)
P
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_ops.py tests/test_gpu_ddp.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -4
