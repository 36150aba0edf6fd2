#!/bin/bash
# Round 5: full GPU test suite + the default bench line (what the driver runs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05_full; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py --op-table $O/op_table.json > $O/bench.log 2>&1 ); echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['forward_ms'], d['roofline']['frac'], d['roofline'].get('stack_frac'), d['roofline'].get('traffic'), d.get('train',{}).get('ms_per_step'), d.get('train',{}).get('exchange',{}).get('exposed_us'))"
