#!/bin/bash
# Profiles of the default bench.py run on the GPU box (run from the repository root through gpurun):
#   tools/run_profiles.sh <tag>      e.g. r02_v3
# writes gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), gpurun_out/<tag>_pmc_summary.csv (five --pmc
# passes, counters only: never combined with a trace domain) and gpurun_out/<tag>_bench.json (the un-profiled bench line).
# Copy what should be judged into profiles/ and run tools/pmc_json.py on the summary (on the same tree).
set -u
TAG=${1:-prof}
OUT=gpurun_out
mkdir -p $OUT/$TAG
export TMPDIR=/tmp
ROOT=$(pwd)
BENCH="python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-overlap-probe --no-other-modes --sustain-seconds 0"
TRACE="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-overlap-probe --no-other-modes --sustain-seconds 0"  # (averages over 25 steps)
python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# the default command (caller arrangement `pipeline`, the library's object: up to four kernels share the chip, durations include that) ...
( cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/$TAG/trace -o trace --output-format csv -- $TRACE > $ROOT/$OUT/$TAG/trace.log 2>&1 )
find $OUT/$TAG/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
# ... and the same steps on ONE caller stream: every kernel alone on the chip
( cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/$TAG/trace1 -o trace --output-format csv -- $TRACE --arrangement single > $ROOT/$OUT/$TAG/trace1.log 2>&1 )
find $OUT/$TAG/trace1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats_single.csv
BENCH="$BENCH --arrangement single"   # counters: per launch, the kernels serialised by the profiler anyway
i=0
for CTRS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_THREAD_CYCLES_VALU" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
            "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    ( cd /tmp && rocprofv3 --pmc $CTRS -d $ROOT/$OUT/$TAG/pmc$i -o pmc --output-format csv -- $BENCH > $ROOT/$OUT/$TAG/pmc$i.log 2>&1 )
done
python3 tools/pmc_summary.py $OUT/$TAG > $OUT/${TAG}_pmc_summary.csv
rm -rf $OUT/$TAG/trace/*/*.db $OUT/$TAG/trace1/*/*.db 2>/dev/null
du -sh $OUT/$TAG | tail -1
head -3 $OUT/${TAG}_kernel_stats.csv
