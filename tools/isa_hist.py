#!/usr/bin/env python3
"""Instruction histogram of the loops of a gfx950 kernel, from the compiler's assembly.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only \
        -o /tmp/lc3gpu.s lc3-codec_amd/csrc/lc3gpu.hip
  python tools/isa_hist.py /tmp/lc3gpu.s _Z16lc3_parse_kernelI13lc3_cfg_48k10E [--min 60] [--json out.json]

A loop = the label range [target, branch] of a backward branch (s_branch / s_cbranch_* to a label defined earlier in the
function).  For every loop with at least --min instructions the instructions between the two are counted by class:
  valu_alu    v_* arithmetic / logic / compare / select (everything v_* that is not below)
  v_mov       v_mov_b32 / v_mov_b64 / v_accvgpr_* (register moves)
  v_cndmask   selects (subset of valu_alu, also reported alone)
  v_cmp       compares (subset of valu_alu, also reported alone)
  v_readlane  v_readlane / v_readfirstlane / v_writelane
  s_nop       hazard padding (the operand + 1 wait states are summed as nop_states)
  s_waitcnt   waits
  salu        other s_* instructions (branches included, reported alone as s_branch)
  lds         ds_*
  vmem        global_* / buffer_* / flat_* / scratch_*
  smem        s_load_* / s_buffer_load_*
Nested loops are counted inside their parents as well; `inner` lists the loops a loop contains.
(The hot loops of the lane-per-frame kernels are identified by what they contain: the parser's symbol loop is the loop with two
ds_read_b128 and v_mul_u32_u24; the packer's the one with ds_write_b8 and two v_mul_u32_u24.)"""
import json
import re
import sys


def classify(op):
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime", "s_memrealtime")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("v_mov_b", "v_accvgpr")):
        return "v_mov"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "v_readlane"
    if op.startswith("v_"):
        return "valu_alu"
    return "other"


def function_body(lines, name):
    start = None
    for i, ln in enumerate(lines):
        if ln.startswith(name) and ln.split(";")[0].rstrip().endswith(":"):
            start = i
            break
    if start is None:
        raise SystemExit("no function starting with %r" % name)
    body = []
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith(".Lfunc_end"):
            break
        if s.startswith(".section") or s.startswith(".amdhsa_kernel") or s.startswith(".rodata"):
            break
        body.append(ln.rstrip("\n"))
    return lines[start].split(";")[0].strip().rstrip(":"), body


def main():
    args = sys.argv[1:]
    if len(args) < 2:
        raise SystemExit(__doc__)
    path, name = args[0], args[1]
    min_len = int(args[args.index("--min") + 1]) if "--min" in args else 60
    out_json = args[args.index("--json") + 1] if "--json" in args else None
    with open(path) as f:
        lines = f.readlines()
    fname, body = function_body(lines, name)
    insts, labels = [], {}
    for ln in body:
        s = ln.split(";")[0].strip() if not ln.strip().startswith(";") else ""
        if not s:
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        insts.append((parts[0], parts[1] if len(parts) > 1 else ""))
    loops = []
    for i, (op, rest) in enumerate(insts):
        if op.startswith(("s_branch", "s_cbranch")):
            tgt = rest.strip()
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    loops.sort(key=lambda l: (l[0], -l[1]))
    report = {"function": fname, "instructions": len(insts), "loops": []}
    for lo, hi, tgt in loops:
        n = hi - lo + 1
        if n < min_len:
            continue
        h, ops, nop_states = {}, {}, 0
        for op, rest in insts[lo:hi + 1]:
            c = classify(op)
            h[c] = h.get(c, 0) + 1
            ops[op] = ops.get(op, 0) + 1
            if c == "s_nop":
                try:
                    nop_states += int(rest.strip()) + 1
                except ValueError:
                    nop_states += 1
        sub = lambda pre: sum(v for k, v in ops.items() if k.startswith(pre))
        entry = {
            "label": tgt, "first": lo, "last": hi, "instructions": n, "classes": dict(sorted(h.items())),
            "vector_total": h.get("valu_alu", 0) + h.get("v_mov", 0) + h.get("v_readlane", 0),
            "v_cndmask": sub("v_cndmask"), "v_cmp": sub("v_cmp"), "s_branch": sub("s_branch") + sub("s_cbranch"),
            "nop_states": nop_states, "sdwa_dpp": sum(v for k, v in ops.items() if k.endswith(("_sdwa", "_dpp"))),
            "inner": [t for (a, b, t) in loops if a >= lo and b <= hi and (a, b) != (lo, hi) and b - a + 1 >= min_len],
            "top_ops": sorted(ops.items(), key=lambda kv: -kv[1])[:24],
        }
        report["loops"].append(entry)
    for e in report["loops"]:
        print("%-14s [%6d..%6d] %5d instr  vector %4d (alu %4d of which cndmask %3d cmp %3d; mov %3d; lane %2d)  salu %4d (branch %2d)  "
              "nop %2d (%2d states)  wait %3d  lds %3d  vmem %3d  inner %s" % (
                  e["label"], e["first"], e["last"], e["instructions"], e["vector_total"], e["classes"].get("valu_alu", 0), e["v_cndmask"],
                  e["v_cmp"], e["classes"].get("v_mov", 0), e["classes"].get("v_readlane", 0), e["classes"].get("salu", 0), e["s_branch"],
                  e["classes"].get("s_nop", 0), e["nop_states"], e["classes"].get("s_waitcnt", 0), e["classes"].get("lds", 0),
                  e["classes"].get("vmem", 0), ",".join(e["inner"]) or "-"))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
