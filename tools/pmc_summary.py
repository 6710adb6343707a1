#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: mean counter value per launch for the big (65536-frame) launches of each kernel.
Usage: python tools/pmc_summary.py <dir with *_counter_collection.csv files (searched recursively)> [frames_per_launch]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 65536.0
acc = defaultdict(list)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            if name.startswith("void "):  # template instantiations: "void kernel<view>(...)"; the view stays in the name
                name = name[5:]
            if not name.startswith("lc3_"):
                continue
            acc[(name, row["Counter_Name"], int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
# keep, per kernel, the largest grid only (the bench launches)
big = {}
for (name, ctr, grid) in acc:
    big[name] = max(big.get(name, 0), grid)
print("kernel,counter,launches,mean_per_launch,per_frame")
for (name, ctr, grid), vals in sorted(acc.items()):
    if grid != big[name]:
        continue
    m = sum(vals) / len(vals)
    print(f"{name},{ctr},{len(vals)},{m:.6g},{m / frames:.4g}")
