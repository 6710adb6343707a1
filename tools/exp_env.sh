#!/bin/bash
# GPU box: the default bench line (both caller arrangements) under environment settings.  usage: tools/exp_env.sh "A=1 B=2" "C=3" ...
for e in "$@"; do
  env $e python3 bench.py --no-parity --no-cpu-baseline --sustain-seconds 0 --steps 32 --warmup 4 2>/dev/null | python3 -c "
import json, sys
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); o = j['other_arrangement']
print('%-60s pipelined %6.2f M  single %6.2f M  kernels alone %s' % ('$e', j['value'] / 1e6, o['value'] / 1e6, {k[4:-7]: round(v, 3) for k, v in o['kernel_ms'].items() if v > 0}))"
done
