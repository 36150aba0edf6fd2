#!/bin/bash
# Round 6: the stride-2 3x3 layers on the 8-phase kernels (scripts/r6_stream_probe.py) -- natural tap order against the class order of Y5ConvParams::tap_seq,
# default stores against nt stores (OUT=g8nt scripts/build_g8_dbg.sh -DY5_G8_ST_AUX=2), with the memory-side counters of each.  -> gpurun_out/stream_probe2*.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/stream_probe2.log; : > $L
for round in 1 2; do
  echo "== round $round: class order" >> $L;   timeout 300 python scripts/r6_stream_probe.py --cfgs 96,95 --blocks 0,192 >> $L 2>&1
  echo "== round $round: natural order" >> $L; Y5_G8_NATURAL_TAPS=1 timeout 300 python scripts/r6_stream_probe.py --cfgs 96,95 --blocks 0,192 >> $L 2>&1
  echo "== round $round: class order + nt stores" >> $L; Y5_LIB_PATH=$PWD/yolov5_amd/libyolov5_hip_g8nt.so timeout 300 python scripts/r6_stream_probe.py --cfgs 96,95 --blocks 0,192 >> $L 2>&1
done
for ARM in class natural nt; do
  for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    T=${ARM}_$(echo $C | tr ' ' '_')
    rm -rf gpurun_out/sp2_$T
    ( [ $ARM = natural ] && export Y5_G8_NATURAL_TAPS=1; [ $ARM = nt ] && export Y5_LIB_PATH=$PWD/yolov5_amd/libyolov5_hip_g8nt.so
      cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/gpurun_out/sp2_$T" -o p -- python "$OLDPWD/scripts/r6_stream_probe.py" --pmc-run --cfgs 96,95 --blocks 0 > "$OLDPWD/gpurun_out/sp2_$T.log" 2>&1 )
    echo "pass $T rc=$?"
  done
done
python - <<'PY' | tee gpurun_out/stream_probe2_pmc.log
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/sp2_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'y5_conv' not in k: continue
        rows.setdefault((f.split('/')[1], int(r['Dispatch_Id']), k[:44]), {})[r['Counter_Name']] = float(r['Counter_Value'])
for key, v in rows.items():
    print(*key, {n: round(x, 1) for n, x in v.items()})
PY
find gpurun_out/sp2_* -name "*.csv" -size +5M -delete
