#!/bin/bash
# GPU box: the sustained leg (seconds of back-to-back steps) of each caller arrangement.  usage: tools/exp_sustained.sh "ENV=.. --arrangement x" ...
for e in "$@"; do
  envs=$(echo "$e" | tr ' ' '\n' | grep '=' | grep -v '^--' | tr '\n' ' ')
  args=$(echo "$e" | tr ' ' '\n' | grep -v '=' | tr '\n' ' ')
  env $envs python3 bench.py --no-parity --no-cpu-baseline --no-overlap-probe --sustain-seconds 2.5 --steps 32 --warmup 4 $args 2>/dev/null | python3 -c "
import json, sys
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s = j['sustained']
print('%-50s burst %6.2f M  sustained %6.2f M over %.1f s at %.0f MHz' % ('$e', j['value'] / 1e6, s['value'] / 1e6, s['seconds'], s['shader_clock_MHz']['median']))"
done
