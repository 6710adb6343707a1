#!/bin/bash
out=gpurun_out/pmc_ic
mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $out/pass$i --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $out/pass$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $out/pass$i.log)"
done
python3 tools/pmc_summary.py $out 
