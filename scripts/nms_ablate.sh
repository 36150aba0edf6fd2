#!/bin/bash
# Kernel experiment: ablated variants of the greedy NMS kernel (-DY5_NMS_ABL=<bits>), timed with rocprofv3 kernel stats over scripts/nms_probe.py.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
export TMPDIR=/tmp
SRC=yolov5_amd/csrc
OBJS=$(ls $SRC/_build/*.o | grep -v "/nms.o")
for abl in 0 1 2 4 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=off -DY5_NMS_ABL=$abl -c $SRC/nms.hip -o build_variants/nms_$abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/libnms_$abl.so $OBJS build_variants/nms_$abl.o
  rm -rf /tmp/nmsprof_$abl
  Y5_LIB_PATH=$PWD/build_variants/libnms_$abl.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nmsprof_$abl -- python scripts/nms_probe.py > /dev/null 2>&1 || true
  python - <<EOF
import csv, glob
f=glob.glob("/tmp/nmsprof_$abl/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])) if f else []:
    if "greedy" in r["Name"]: print("ablation $abl greedy avg us", float(r["AverageNs"])/1e3)
EOF
done
