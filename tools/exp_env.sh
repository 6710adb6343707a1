#!/bin/bash
# GPU box: the default bench line (every caller arrangement) under environment settings.  usage: tools/exp_env.sh "A=1 B=2" "C=3" ...
# prints each arrangement's rate and the kernels' durations in it (one stream: alone; two streams: stretched, beside the other handle's)
for e in "$@"; do
  env $e python3 bench.py --no-parity --no-cpu-baseline --sustain-seconds 0 --steps 32 --warmup 4 2>/dev/null | python3 -c "
import json, sys
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
f = lambda d: {k[4:-7]: round(v, 3) for k, v in d.items() if v > 0}
print('$e')
for a in [dict(arrangement=j['config']['arrangement'], value=j['value'], kernel_ms=j['kernel_ms'])] + j['other_arrangements']:
    print('    %-10s %6.2f M  %s' % (a['arrangement'], a['value'] / 1e6, f(a['kernel_ms'])))"
done
