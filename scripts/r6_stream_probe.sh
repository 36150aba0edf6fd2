#!/bin/bash
# Round 6: timing sweep + per-dispatch memory counters of the stride-2 3x3 layers (scripts/r6_stream_probe.py).  -> gpurun_out/stream_probe*.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scripts/r6_stream_probe.py > gpurun_out/stream_probe.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  T=$(echo $C | tr ' ' '_')
  rm -rf gpurun_out/sp_$T
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/gpurun_out/sp_$T" -o p -- python "$OLDPWD/scripts/r6_stream_probe.py" --pmc-run --blocks 0 > "$OLDPWD/gpurun_out/sp_$T.log" 2>&1)
  echo "pass $T rc=$?"
done
python - <<'PY' | tee gpurun_out/stream_probe_pmc.log
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/sp_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'y5_conv' not in k: continue
        key = (int(r['Dispatch_Id']), k[:60])
        rows.setdefault((f.split('/')[1], key), {})[r['Counter_Name']] = float(r['Counter_Value'])
for (p, key), v in rows.items():
    print(p, key, {n: round(x, 1) for n, x in v.items()})
PY
find gpurun_out/sp_* -name "*.csv" -size +5M -delete
