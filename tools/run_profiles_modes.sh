#!/bin/bash
# HBM traffic per frame of the other shapes of the 65 536-frame batch (SURVEY 8d Config 2: 65 536 x 1 cold, 65 536 x 1 carried, 4 096 x 16;
# the headline 16 384 x 4 for comparison): two rocprofv3 --pmc passes per shape (FETCH_SIZE, WRITE_SIZE -- counters only, never with a
# trace domain), summarised per kernel and frame.   usage (through gpurun, repository root): tools/run_profiles_modes.sh [tag=r06]
TAG=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT/${TAG}_modes
export TMPDIR=/tmp
ROOT=$(pwd)
for SHAPE in "65536 1 cold" "65536 1" "4096 16" "16384 4"; do
    NAME=$(echo $SHAPE | tr ' ' '_')
    for C in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && rocprofv3 --pmc $C -d $ROOT/$OUT/${TAG}_modes/$NAME/$C -o pmc --output-format csv -- python3 $ROOT/tools/shape_run.py $SHAPE > $ROOT/$OUT/${TAG}_modes/$NAME.$C.log 2>&1 )
    done
    python3 tools/pmc_summary.py $OUT/${TAG}_modes/$NAME 65536 > $OUT/${TAG}_modes_${NAME}.csv
done
python3 - $OUT $TAG > $OUT/${TAG}_modes_traffic.json <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB (1024 B) per FRAME, per kernel, for shapes S x T of the 65 536-frame batch "
               "(tools/shape_run.py: one caller stream, 6 steps; `cold` = every step from fresh state); raw counter values (gfx950 counts 16-byte-per-lane "
               "reads at half their size: the wave-per-stream kernels' FETCH is a lower bound, MI355X_MICROARCH.md)", "shapes": {}}
for name in ("65536_1_cold", "65536_1", "4096_16", "16384_4"):
    k = {}
    with open("%s/%s_modes_%s.csv" % (out, tag, name)) as f:
        for row in csv.DictReader(f):
            kern = row["kernel"].split("<")[0]
            # every launch of the kernel in a step counts (cold: the decoder's reset is a zero-frame launch of the synthesis kernel): six steps
            k.setdefault(kern, {})[row["counter"]] = float(row["mean_per_launch"]) * float(row["launches"]) / (6.0 * 65536.0)
    tot_f = sum(v.get("FETCH_SIZE", 0.0) for v in k.values())
    tot_w = sum(v.get("WRITE_SIZE", 0.0) for v in k.values())
    res["shapes"][name] = {"kernels_KB_per_frame": k, "fetch_KB_per_frame": tot_f, "write_KB_per_frame": tot_w, "total_GB_per_step": (tot_f + tot_w) * 1024 * 65536 / 1e9}
print(json.dumps(res, indent=1))
PY
cat $OUT/${TAG}_modes_traffic.json | head -50
