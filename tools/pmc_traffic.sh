#!/bin/bash
out=gpurun_out/pmc_traffic
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $grp -d $out/$grp --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $out/$grp.log 2>&1
done
python3 tools/pmc_summary.py $out
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'])"
