#!/usr/bin/env python3
"""Kernel resources from the compiler's assembly metadata (hipcc -S --cuda-device-only): VGPRs, SGPRs, LDS, scratch bytes per lane, spills.
usage: python tools/kernel_resources.py /tmp/lc3gpu.s [substring ...]"""
import re
import sys

rows, cur, pending_lds = [], None, 0
for ln in open(sys.argv[1]):
    m = re.match(r"\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", ln)
    if not m:
        continue
    k, v = m.groups()
    if k == "group_segment_fixed_size":  # (the metadata's keys are in alphabetical order: this one precedes the kernel's .name)
        pending_lds = int(v)
    elif k == "name":
        cur = {"name": v, "group_segment_fixed_size": pending_lds}
        rows.append(cur)
    elif cur is not None:
        cur[k] = int(v)
sel = sys.argv[2:]
print("%-100s %5s %5s %8s %8s %10s %10s" % ("kernel", "vgpr", "sgpr", "lds", "scratch", "vgpr_spill", "sgpr_spill"))
for r in rows:
    if sel and not any(s in r["name"] for s in sel):
        continue
    if "vgpr_count" not in r:
        continue
    print("%-100s %5d %5d %8d %8d %10d %10d" % (r["name"][:100], r["vgpr_count"], r.get("sgpr_count", 0), r.get("group_segment_fixed_size", 0),
                                              r.get("private_segment_fixed_size", 0), r.get("vgpr_spill_count", 0), r.get("sgpr_spill_count", 0)))
