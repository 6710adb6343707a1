#!/bin/bash
# GPU box: vector / scalar / LDS instruction counts per frame of the encoder's kernels under experiment builds (knock-out or repeat builds,
# tools/exp_build.sh): one rocprofv3 --pmc pass per library on a short one-stream bench run.  usage: tools/exp_ko_insts.sh <tag> <tag> ...
# (empty tag "" = the production library); prints one line per library.
export TMPDIR=/tmp
ROOT=$(pwd)
for tag in "$@"; do
  lib=liblc3gpu_$tag.so; [ -z "$tag" ] && lib=liblc3gpu.so
  out=$ROOT/gpurun_out/_pmc_$tag
  rm -rf $out
  ( cd /tmp && LC3GPU_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $out -o pmc --output-format csv -- python3 $ROOT/bench.py --arrangement single --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-overlap-probe --sustain-seconds 0 > /dev/null 2>&1 )
  python3 tools/pmc_summary.py $out | python3 -c "
import csv, sys
r = {}
for row in csv.DictReader(sys.stdin):
    k = row['kernel'].split('<')[0][4:].replace('_kernel', '')
    if k in ('enc_front', 'enc_back', 'sns_vq'):
        r.setdefault(k, {})[row['counter'][9:]] = round(float(row['per_frame']))
print('%-10s' % ('${tag:-production}'), r)"
  rm -rf $out
done
