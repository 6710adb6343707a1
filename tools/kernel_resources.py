#!/usr/bin/env python3
"""Kernel resources -- VGPRs, SGPRs, static LDS, scratch bytes per lane, spilled registers -- from the AMDGPU code-object metadata.

  python tools/kernel_resources.py /tmp/lc3gpu.s [substring ...]            the compiler's assembly (hipcc -S --cuda-device-only)
  python tools/kernel_resources.py lc3-codec_amd/lib/liblc3gpu.so [...]     a BUILT library: its gfx950 code objects are unbundled
                                                                            (llvm-objdump --offloading) and their notes read (llvm-readelf)

tests/test_kernel_resources.py imports `from_library` and holds the headline kernels to their register budgets."""
import os
import re
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("LC3_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
_KEY = re.compile(r"\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)")


def parse(text):
    """rows of {name, vgpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size}
    from metadata text (the `.amdgpu_metadata` block of an assembly file, or `llvm-readelf --notes` of a code object: same keys)"""
    rows, cur, pending_lds = [], None, 0
    for ln in text.splitlines():
        m = _KEY.match(ln)
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
    return [r for r in rows if "vgpr_count" in r]


def from_library(path):
    """every kernel of every gfx950 code object bundled in a built shared library"""
    path = os.path.abspath(path)
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        link = os.path.join(tmp, "lib.so")
        os.symlink(path, link)
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True,
                                   stdout=subprocess.PIPE, universal_newlines=True).stdout
            rows += parse(notes)
    return rows


def main():
    src = sys.argv[1]
    rows = from_library(src) if src.endswith(".so") else parse(open(src).read())
    sel = sys.argv[2:]
    print("%-100s %5s %5s %8s %8s %10s %10s" % ("kernel", "vgpr", "sgpr", "lds", "scratch", "vgpr_spill", "sgpr_spill"))
    for r in rows:
        if sel and not any(s in r["name"] for s in sel):
            continue
        print("%-100s %5d %5d %8d %8d %10d %10d" % (r["name"][:100], r["vgpr_count"], r.get("sgpr_count", 0), r.get("group_segment_fixed_size", 0),
                                                  r.get("private_segment_fixed_size", 0), r.get("vgpr_spill_count", 0), r.get("sgpr_spill_count", 0)))


if __name__ == "__main__":
    main()
