#!/usr/bin/env python3
"""Turn a tools/pmc_summary.py summary.csv into profiles/pmc_latest.json: the per-kernel counters bench.py's roofline
object quotes (HBM traffic, VALU instructions, active lanes), stamped with the sha256 of the kernel sources they were
measured on.  bench.py recomputes that hash and withholds the counters (null) when the sources have changed since.

usage (on the tree the passes were run on): python tools/pmc_json.py <summary.csv> <name of the committed copy under profiles/>
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha  # noqa: E402

KEYS = {"FETCH_SIZE": "fetch_size_kb", "WRITE_SIZE": "write_size_kb", "SQ_INSTS_VALU": "sq_insts_valu",
        "SQ_THREAD_CYCLES_VALU": "sq_thread_cycles_valu", "SQ_INSTS_SALU": "sq_insts_salu", "SQ_WAVE_CYCLES": "sq_wave_cycles",
        "SQ_WAIT_ANY": "sq_wait_any", "SQ_BUSY_CYCLES": "sq_busy_cycles", "SQ_LDS_BANK_CONFLICT": "sq_lds_bank_conflict",
        "SQ_LDS_IDX_ACTIVE": "sq_lds_idx_active"}


def main():
    src, committed = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    kernels = {}
    with open(src) as f:
        for row in csv.DictReader(f):
            key = KEYS.get(row["counter"])
            if key:
                kernels.setdefault(row["kernel"].split("<")[0], {})[key] = float(row["mean_per_launch"])  # drop the <view>
    out = {
        "source": "profiles/%s (rocprofv3 --pmc, separate passes, mean per launch of %d frames; FETCH_SIZE / WRITE_SIZE in KB, raw: "
                  "FETCH_SIZE not doubled, the accesses are not wide streaming reads, MI355X_MICROARCH.md HBM section)" % (committed, frames),
        "kernel_source_sha256": kernel_source_sha(),
        "frames_per_launch": frames,
        "kernels": dict(sorted(kernels.items())),
    }
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
