#!/bin/bash
# GPU box: the configurations beside the headline one under several builds of the library: 48 kHz / 7.5 ms, 24 kHz (run-time view), the mixed
# batch, and the default bench line.  usage: tools/exp_views.sh <lib file under lib/> ...
for lib in "$@"; do
  echo "== $lib"
  for cfg in "48000 7500 113" "24000 10000 60" "44100 10000 110"; do
    LC3GPU_LIB=$lib python3 tools/uniform_batch.py $cfg 2>/dev/null | tail -1 | python3 -c "
import json, sys
j = json.loads(sys.stdin.read()); print('  %-46s %6.2f M  %s  exact %s' % (j['config'][:44], j['frames_per_s'] / 1e6, j['kernel_ms'], j['bitstream_exact_64_streams'] and j['pcm_exact_64_streams']))"
  done
  LC3GPU_LIB=$lib python3 tools/mixed_batch.py 2>/dev/null | tail -1 | python3 -c "
import json, sys
j = json.loads(sys.stdin.read()); print('  mixed batch %6.2f M  %.3f ms/step  parity %s' % (j['frames_per_s'] / 1e6, j['ms_per_step'], j['parity_all_streams_first_step']))"
  LC3GPU_LIB=$lib python3 bench.py --no-parity --no-cpu-baseline --sustain-seconds 0 --steps 32 --warmup 4 2>/dev/null | python3 -c "
import json, sys
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); o = j['other_arrangement']
print('  headline: pipelined %6.2f M  single %6.2f M  front alone %.3f' % (j['value'] / 1e6, o['value'] / 1e6, o['kernel_ms']['lc3_enc_front_kernel']))"
done
