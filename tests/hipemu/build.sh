#!/bin/bash
# TEST INFRASTRUCTURE ONLY: build liby5emu.so = the product kernel sources compiled for the HOST on top of the
# fiber-based HIP emulator.  Loaded only by tests/hipemu/emu.py.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../yolov5_amd/csrc"
OUT="$HERE/_build"
CXX=/opt/rocm/lib/llvm/bin/clang++
mkdir -p "$OUT"
FLAGS="-x c++ -std=c++17 -O1 -g -fPIC -mf16c -I$HERE/include -Wno-unused-value -Wno-unknown-attributes -Wno-ignored-attributes -Wno-psabi"
for f in core conv misc nms loss mask bn wgrad train_misc optim metrics preprocess head augment bneck convh3 front sppf convg8; do
  extra=""
  if [ "$f" = "nms" ] || [ "$f" = "misc" ] || [ "$f" = "loss" ] || [ "$f" = "optim" ] || [ "$f" = "metrics" ] || [ "$f" = "preprocess" ] || [ "$f" = "head" ] || [ "$f" = "augment" ]; then extra="-ffp-contract=off"; fi
  if [ ! -f "$OUT/$f.o" ] || [ "$SRC/$f.hip" -nt "$OUT/$f.o" ] || [ -n "$(find "$SRC" "$HERE/include" -name '*.h' -newer "$OUT/$f.o" 2>/dev/null)" ]; then
    $CXX $FLAGS $extra -c "$SRC/$f.hip" -o "$OUT/$f.o" &
  fi
done
if [ ! -f "$OUT/emu_runtime.o" ] || [ "$HERE/emu_runtime.cpp" -nt "$OUT/emu_runtime.o" ] || [ "$HERE/include/hip/hip_runtime.h" -nt "$OUT/emu_runtime.o" ]; then
  $CXX -std=c++17 -O1 -g -fPIC -I"$HERE/include" -c "$HERE/emu_runtime.cpp" -o "$OUT/emu_runtime.o" &
fi
wait
$CXX -shared -fPIC -o "$OUT/liby5emu.so" "$OUT"/core.o "$OUT"/conv.o "$OUT"/misc.o "$OUT"/nms.o "$OUT"/loss.o "$OUT"/mask.o "$OUT"/bn.o "$OUT"/wgrad.o "$OUT"/train_misc.o "$OUT"/optim.o "$OUT"/metrics.o "$OUT"/preprocess.o "$OUT"/head.o "$OUT"/augment.o "$OUT"/bneck.o "$OUT"/convh3.o "$OUT"/front.o "$OUT"/sppf.o "$OUT"/convg8.o "$OUT"/emu_runtime.o
echo "$OUT/liby5emu.so"
