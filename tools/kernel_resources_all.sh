#!/bin/bash
# Kernel resources of the multi-unit library: every translation unit compiled to assembly (8 at a time), one table.
# usage: tools/kernel_resources_all.sh > profiles/<tag>_kernel_resources.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
UNITS=$(python3 - <<PY
import importlib, sys
sys.path.insert(0, "$ROOT")
m = importlib.import_module("lc3-codec_amd.api")
print(" ".join("%d_%d" % u for u in m._translation_units()))
PY
)
echo "kernel resources of liblc3gpu.so, unit by unit (hipcc --offload-arch=gfx950 -O3 -DLC3_TU_KIND=k -DLC3_TU_INDEX=i; tools/kernel_resources.py on the compiler's assembly metadata): VGPRs, SGPRs, static LDS bytes, scratch bytes per lane, spilled registers"
echo $UNITS | tr ' ' '\n' | xargs -P 8 -I{} bash -c 'k=${0%_*}; i=${0#*_}; hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only -DLC3_TU_KIND=$k -DLC3_TU_INDEX=$i -o '$TMP'/u_$0.s '$ROOT'/lc3-codec_amd/csrc/lc3gpu.hip 2>/dev/null' {}
for u in $UNITS; do
  echo "== unit kind ${u%_*} index ${u#*_}"
  python3 $ROOT/tools/kernel_resources.py $TMP/u_$u.s | tail -n +2
done
rm -rf $TMP
