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
        "SQ_LDS_IDX_ACTIVE": "sq_lds_idx_active", "SQ_ACTIVE_INST_VALU": "sq_active_inst_valu", "SQ_INSTS_LDS": "sq_insts_lds",
        "SQ_INSTS_VMEM_RD": "sq_insts_vmem_rd", "SQ_INSTS_VMEM_WR": "sq_insts_vmem_wr", "SQ_INSTS_SMEM": "sq_insts_smem"}


# the producer / consumer forms of the two lane-per-frame kernels report under the names bench.py's timer slots carry
ALIAS = {"lc3_parse_pc_kernel": "lc3_parse_kernel", "lc3_pack_pc_kernel": "lc3_pack_kernel"}
# kernels whose HBM reads are 16-byte-per-lane coalesced copies (planes, state blobs, PCM of the wave-per-stream kernels):
# gfx950's FETCH_SIZE counts such a read at half its size (MI355X_MICROARCH.md, HBM / rocprofv3 section), so the corrected figure doubles
# it -- an upper bound, not every read of these kernels is that wide.  The lane-per-frame kernels read their own column per lane: raw
WIDE_READS = ("lc3_enc_front_kernel", "lc3_enc_back_kernel", "lc3_decode_kernel")


def main():
    src, committed = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    kernels = {}
    with open(src) as f:
        for row in csv.DictReader(f):
            key = KEYS.get(row["counter"])
            if key:
                name = row["kernel"].split("<")[0]  # drop the <view>
                kernels.setdefault(ALIAS.get(name, name), {})[key] = float(row["mean_per_launch"])
    for name, k in kernels.items():
        if "fetch_size_kb" in k:
            k["fetch_size_kb_corrected"] = k["fetch_size_kb"] * (2.0 if name in WIDE_READS else 1.0)
    out = {
        "source": "profiles/%s (rocprofv3 --pmc, separate passes, mean per launch of %d frames; FETCH_SIZE / WRITE_SIZE in KB (1024 bytes); "
                  "fetch_size_kb_corrected = FETCH_SIZE x 2 for the wave-per-stream kernels, whose reads are 16-byte-per-lane coalesced and "
                  "counted at half their size on gfx950 (MI355X_MICROARCH.md, HBM section), raw for the lane-per-frame kernels; lc3_parse_kernel / "
                  "lc3_pack_kernel are the producer / consumer kernels lc3_parse_pc_kernel / lc3_pack_pc_kernel)" % (committed, frames),
        "kernel_source_sha256": kernel_source_sha(),
        "frames_per_launch": frames,
        "kernels": dict(sorted(kernels.items())),
    }
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
