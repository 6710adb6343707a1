#!/bin/bash
# GPU box: kernel times (every kernel alone on the chip: --arrangement single) of experiment builds of the library -- knock-out builds,
# whose output is garbage, and repeat builds (lc3_dev_experiments.h).  usage: tools/exp_ko.sh <prefix> <tag> <tag> ...
# library of a tag: lc3-codec_amd/lib/liblc3gpu_<prefix><tag>.so (tools/exp_build.sh); an empty prefix and tag = the production library
P=$1; shift
for ko in "$@"; do
  lib=liblc3gpu_$P$ko.so; [ -z "$P$ko" ] && lib=liblc3gpu.so
  LC3GPU_LIB=$lib python3 bench.py --arrangement single --no-parity --no-cpu-baseline --no-overlap-probe --sustain-seconds 0 --steps 24 --warmup 4 2>/dev/null | python3 -c "
import json, sys
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('${P}${ko:-production}', {k[4:-7]: round(v, 4) for k, v in j['kernel_ms'].items() if v > 0}, round(j['ms_per_step'], 4))"
done
