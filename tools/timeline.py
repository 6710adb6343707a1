#!/usr/bin/env python3
"""What ran beside what: a `rocprofv3 --kernel-trace` CSV of a bench.py run read as a timeline.

  (cd /tmp && rocprofv3 --kernel-trace -d out -o t --output-format csv -- python3 bench.py --arrangement pipelined --steps 12 ...)
  python tools/timeline.py out/*/t_kernel_trace.csv [--skip-steps 4] [--json gpurun_out/x.json]

The codec's kernels are grouped into steps by counting launches of lc3_enc_front_kernel per queue; for the steps after the warm-up it prints
  * per kernel: launches, mean duration, the hardware queues it ran on
  * the encoder chain's three boundaries (front -> quantiser, quantiser -> back, back -> packer) and the decoder's (parser -> synthesis):
    the idle time on the chain's own queue between the end of one kernel and the start of the next, and how much of that gap some kernel of
    ANOTHER queue covered (a gap nothing covers is chip time lost to the boundary)
  * concurrency: the share of the wall time with 0 / 1 / 2 / 3+ of the codec's kernels in flight, and wall time per step
Timestamps are the profiler's (ns)."""
import csv
import json
import sys
from collections import defaultdict

SHORT = [("lc3_enc_front", "front"), ("lc3_sns_vq", "vq"), ("lc3_enc_back", "back"), ("lc3_pack", "pack"), ("lc3_parse", "parse"),
         ("lc3_decode", "synth"), ("lc3_recon", "recon"), ("lc3_tns_", "tns")]


def short(name):
    for k, v in SHORT:
        if k in name:
            return v
    return None


def main():
    args = sys.argv[1:]
    path = args[0]
    skip = int(args[args.index("--skip-steps") + 1]) if "--skip-steps" in args else 4
    out_json = args[args.index("--json") + 1] if "--json" in args else None
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            ev.append({"k": k, "q": int(r["Queue_Id"]), "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])})
    ev.sort(key=lambda x: x["s"])
    if not ev:
        raise SystemExit("no codec kernels in " + path)
    # steps: the n-th front-half launch of a queue opens that queue's n-th step; a step's window = from its earliest front start
    fronts = defaultdict(list)
    for x in ev:
        if x["k"] == "front":
            fronts[x["q"]].append(x["s"])
    n_steps = min(len(v) for v in fronts.values())
    # (several encoder queues -- the split arrangements -- are not tied to each other and one may run ahead: a step has begun when the LAST
    # queue has begun it)
    t0 = max(v[skip] for v in fronts.values()) if n_steps > skip + 1 else ev[0]["s"]
    t1 = max(v[n_steps - 1] for v in fronts.values())  # up to the start of the last step (its tail is cut off by the end of the run)
    steps = n_steps - 1 - skip
    live = [x for x in ev if x["s"] >= t0 and x["s"] < t1]
    res = {"file": path, "steps_measured": steps, "wall_ms_per_step": (t1 - t0) / 1e6 / max(1, steps), "queues": sorted({x["q"] for x in live})}
    per = defaultdict(lambda: {"n": 0, "ms": 0.0, "queues": set()})
    for x in live:
        p = per[x["k"]]
        p["n"] += 1
        p["ms"] += (x["e"] - x["s"]) / 1e6
        p["queues"].add(x["q"])
    res["kernels"] = {k: {"launches": p["n"], "mean_ms": p["ms"] / p["n"], "ms_per_step": p["ms"] / max(1, steps), "queues": sorted(p["queues"])}
                      for k, p in per.items()}
    # boundaries on a queue: consecutive kernels of one chain
    byq = defaultdict(list)
    for x in live:
        byq[x["q"]].append(x)
    gaps = defaultdict(lambda: {"n": 0, "gap_ms": 0.0, "covered_ms": 0.0})
    for q, xs in byq.items():
        others = [y for y in live if y["q"] != q]
        for a, b in zip(xs, xs[1:]):
            name = a["k"] + "->" + b["k"]
            g0, g1 = a["e"], b["s"]
            if g1 <= g0:
                continue
            cov = 0
            # time inside [g0, g1) during which at least one kernel of another queue runs
            segs = sorted((max(g0, y["s"]), min(g1, y["e"])) for y in others if y["s"] < g1 and y["e"] > g0)
            cur = g0
            for s, e in segs:
                s = max(s, cur)
                if e > s:
                    cov += e - s
                    cur = e
            d = gaps[name]
            d["n"] += 1
            d["gap_ms"] += (g1 - g0) / 1e6
            d["covered_ms"] += cov / 1e6
    res["boundaries"] = {k: {"count": d["n"], "mean_gap_us": 1e3 * d["gap_ms"] / d["n"], "gap_ms_per_step": d["gap_ms"] / max(1, steps),
                             "covered_by_another_queue_ms_per_step": d["covered_ms"] / max(1, steps)} for k, d in sorted(gaps.items())}
    # concurrency histogram
    pts = []
    for x in live:
        pts.append((x["s"], 1))
        pts.append((min(x["e"], t1), -1))
    pts.sort()
    hist, lvl, last = defaultdict(int), 0, t0
    for t, d in pts:
        if t > last:
            hist[min(lvl, 3)] += t - last
            last = t
        lvl += d
    tot = float(sum(hist.values())) or 1.0
    res["in_flight_share"] = {("3+" if k == 3 else str(k)): hist[k] / tot for k in sorted(hist)}
    print(json.dumps(res, indent=1))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
