#!/bin/bash
# The default bench.py arrangement (the library's pipeline object) in N consecutive FRESH processes: how the runtime deals the pipeline's
# four HIP streams onto hardware queues is decided per process, so the spread over processes is the figure that says whether the headline
# depends on luck (round-5 review, weak point 6).  usage (through gpurun, from the repository root): tools/repeat_bench.sh [N=10] [tag=r06]
# -> gpurun_out/<tag>_repeat_bench.json  {values, min, median, max, min_over_median}
N=${1:-10}
TAG=${2:-r06}
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/${TAG}_repeat_lines.jsonl
for i in $(seq 1 $N); do
    python3 bench.py --no-cpu-baseline --no-overlap-probe --no-other-modes --sustain-seconds 0 >> $OUT/${TAG}_repeat_lines.jsonl 2>> $OUT/${TAG}_repeat.err
done
python3 - "$OUT/${TAG}_repeat_lines.jsonl" > $OUT/${TAG}_repeat_bench.json <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
v = sorted(r["value"] for r in rows)
med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
print(json.dumps({"what": "python bench.py (default arrangement: lc3gpu_pipeline_submit) in %d consecutive fresh processes, %d timed steps each" % (len(v), rows[0]["steps"]),
                  "arrangement": rows[0]["config"]["arrangement"], "values_frames_per_s": [r["value"] for r in rows],
                  "parity_mismatches": [r["parity_mismatches_all_ranks"] for r in rows],
                  "min": v[0], "median": med, "max": v[-1], "min_over_median": v[0] / med}))
PY
cat $OUT/${TAG}_repeat_bench.json
