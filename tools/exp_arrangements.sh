#!/bin/bash
# GPU box: the default bench line under the split / stagger settings, both caller arrangements each (gpurun_out/exp_arr_*.json)
OUT=gpurun_out
mkdir -p $OUT
run() {  # tag, env...
    tag=$1; shift
    env "$@" python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 --sustain-seconds 0 > $OUT/exp_arr_$tag.json 2> $OUT/exp_arr_$tag.err
    python3 - "$tag" $OUT/exp_arr_$tag.json <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    o = j["other_arrangement"]
    print("%-28s single %6.2f M (%.3f ms)   pipelined %6.2f M (%.3f ms)   parity %s %s   kernels %s" % (
        sys.argv[1], j["value"] / 1e6, j["ms_per_step"], o["value"] / 1e6, o["ms_per_step"], j["parity"]["bitstream_exact"], o["parity"]["bitstream_exact"],
        {k[4:-7]: round(v, 3) for k, v in j["kernel_ms"].items() if v > 0}))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run nosplit LC3GPU_SPLIT=0
run split_stagger1 LC3GPU_SPLIT_STAGGER=1
run split_stagger0 LC3GPU_SPLIT_STAGGER=0
run nosplit_again LC3GPU_SPLIT=0
run split_stagger1_again LC3GPU_SPLIT_STAGGER=1
