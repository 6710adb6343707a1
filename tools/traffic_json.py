#!/usr/bin/env python3
"""Turn a tools/pmc_summary.py summary.csv into profiles/hbm_traffic_latest.json (what bench.py's roofline.traffic reads).

usage: python tools/traffic_json.py <summary.csv> <name of the committed copy under profiles/>
"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, committed = sys.argv[1], sys.argv[2]
    kernels = {}
    with open(src) as f:
        for row in csv.DictReader(f):
            key = {"FETCH_SIZE": "fetch_size_kb_per_launch", "WRITE_SIZE": "write_size_kb_per_launch"}.get(row["counter"])
            if key:
                kernels.setdefault(row["kernel"].split("<")[0], {})[key] = float(row["mean_per_launch"])  # drop the <view>
    out = {
        "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, KB per launch of 65536 "
                  "frames; raw, FETCH_SIZE not doubled: the accesses are not wide streaming reads, "
                  "MI355X_MICROARCH.md HBM section)" % committed,
        "frames_per_launch": 65536,
        "kernels": dict(sorted(kernels.items())),
    }
    with open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
