#!/bin/bash
# rocprofv3 counter passes over a short bench run (one --pmc group per run, no tracing options; MI355X_MICROARCH.md).
# Usage (on the GPU box, from the repo root): bash tools/pmc_run.sh <outdir>
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp -d "$out/pass$i" --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > "$out/pass$i.log" 2>&1 || echo "pass $i ($grp) failed"
done
python3 tools/pmc_summary.py "$out" > "$out/summary.csv"
cat "$out/summary.csv"
