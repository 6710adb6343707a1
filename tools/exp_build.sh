#!/bin/bash
# Build experiment variants of the library side by side: only the MAIN translation unit (the headline kernels live there) is compiled with the
# extra flags, every other object is the production one (lc3-codec_amd/api.py::build_native, LC3_HIPCC_EXTRA_MAIN).
# usage: tools/exp_build.sh name=-DFLAG[,-DFLAG2] ...      ->  lc3-codec_amd/lib/liblc3gpu_<name>.so
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}; defs=${defs//,/ }
  ( cd $ROOT && LC3GPU_LIB=liblc3gpu_$name.so LC3_HIPCC_EXTRA_MAIN="$defs" LC3_BUILD_JOBS=1 python3 -c "
import importlib; m = importlib.import_module('lc3-codec_amd'); print(m.build_native())" ) &
done
wait
