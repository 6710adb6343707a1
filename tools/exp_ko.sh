#!/bin/bash
# GPU box: kernel times of knock-out builds (timing only: their output is garbage).  usage: tools/exp_ko.sh <prefix> <ko> <ko> ...
# a build: LC3GPU_LIB=liblc3gpu_<prefix><ko>.so LC3_HIPCC_EXTRA="-D...=<ko>" python -c "...build_native(force=True)"
P=$1; shift
for ko in "$@"; do
  LC3GPU_LIB=liblc3gpu_$P$ko.so python3 bench.py --arrangement single --no-parity --no-cpu-baseline --no-overlap-probe --sustain-seconds 0 --steps 24 --warmup 4 2>/dev/null | python3 -c "
import json, sys
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$P$ko', {k[4:-7]: round(v, 4) for k, v in j['kernel_ms'].items() if v > 0}, round(j['ms_per_step'], 4))"
done
