#!/usr/bin/env python3
"""Static instruction histogram of gfx950 functions by ISSUE CLASS (tools/instr_cost.hip measures what each class costs).

    hipcc -DLC3_TU_KIND=0 -DLC3_TU_INDEX=0 --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only \
          -o /tmp/lc3gpu_main.s lc3-codec_amd/csrc/lc3gpu.hip
    python tools/isa_classes.py /tmp/lc3gpu_main.s lc3_enc_mdctI13lc3_cfg_48k10 lc3_enc_front_kernelI13lc3_cfg_48k10 ... [--json out.json]

Classes (a wave64 vector instruction's encoding decides what the SIMD can do with it, profiles/r06_instr_cost.json):
  vop2_f32      v_add/sub/subrev/mul/max/min_f32 in the VOP2 (e32) encoding, no modifiers
  vop2_int      v_add/sub_u32, shifts, and/or/xor in the VOP2 encoding
  vop2_other    the rest of the e32 encodings (v_cndmask_e32, v_min/max_*32, v_fmac, v_mul_*24 ...)
  vop1          v_mov, v_cvt, v_rcp ... (one source)
  vop3          e64 encodings and native VOP3 operations (v_fma, v_add3, v_bfi, v_lshl_add, v_mad, v_cndmask_e64, v_mul_lo ...)
  vopc          compares
  packed        v_pk_*
  dpp / sdwa    instructions with a dpp / sdwa operand
  lane          v_readlane / v_readfirstlane / v_writelane
and the non-vector rest: salu, s_nop, s_waitcnt, smem, lds, vmem.  Counts are STATIC (a loop body counts once); the kernels here are
mostly straight-line code per frame, so the shares are close to the dynamic ones -- SQ_INSTS_VALU per wave-frame says by how much."""
import json
import re
import sys

VOP2_F32 = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_max_f32", "v_min_f32"}
VOP2_INT = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_and_b32", "v_or_b32", "v_xor_b32"}
VOP3_ONLY = ("v_fma_", "v_add3_", "v_bfi_", "v_bfe_", "v_lshl_add", "v_lshl_or", "v_and_or", "v_or3", "v_xad", "v_mad_", "v_mul_lo", "v_mul_hi", "v_alignbit",
             "v_perm", "v_med3", "v_min3", "v_max3", "v_add_lshl", "v_div_", "v_ldexp", "v_cvt_pk", "v_lerp", "v_sad", "v_readlane", "v_writelane",
             "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_add_co_u32", "v_sub_co_u32", "v_addc_co", "v_subb_co", "v_mbcnt", "v_bcnt", "v_trig", "v_cubeid")


def classify(op, operands):
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
    if not op.startswith("v_"):
        return "other"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if op.startswith("v_pk_"):
        return "packed"
    if "_dpp" in op or "row_" in operands or "quad_perm" in operands:
        return "dpp"
    if "_sdwa" in op:
        return "sdwa"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "vopc"
    base = op.replace("_e32", "").replace("_e64", "")
    e64 = op.endswith("_e64") or base.startswith(VOP3_ONLY) or "|" in operands or "neg(" in operands or "clamp" in operands or " mul:" in operands or " div:" in operands
    if not e64:
        # a negated / absolute source or an SGPR pair as the select mask forces VOP3 even when the assembler prints no suffix
        if re.search(r"(^|, )-[vs\d]", operands) or re.search(r"(^|, )-\|", operands):
            e64 = True
        if base == "v_cndmask_b32" and not operands.rstrip().endswith("vcc"):
            e64 = True
        # VOP2 needs src1 in a VGPR: an SGPR or constant second source is VOP3
    if e64:
        return "vop3"
    if base in VOP2_F32:
        return "vop2_f32"
    if base in VOP2_INT:
        return "vop2_int"
    if base.startswith(("v_mov_", "v_cvt_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_floor_", "v_ceil_", "v_trunc_", "v_rndne_", "v_fract_", "v_not_",
                        "v_bfrev_", "v_ffbh_", "v_ffbl_", "v_frexp_", "v_accvgpr", "v_sin_", "v_cos_", "v_nop", "v_swap")):
        return "vop1"
    return "vop2_other"


def function_body(lines, name):
    start = None
    for i, ln in enumerate(lines):
        head = ln.split(";")[0].rstrip()
        if head.endswith(":") and name in head and not head.startswith((".", "\t")):
            start = i
            break
    if start is None:
        raise SystemExit("no function label containing %r" % name)
    body = []
    for ln in lines[start + 1:]:
        if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end") or ln.startswith("\t.section") or ln.startswith("\ts_endpgm"):
            if ln.startswith("\ts_endpgm"):
                body.append(ln)
            break
        body.append(ln)
    return body


def histogram(body):
    h, ops = {}, {}
    for ln in body:
        t = ln.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        parts = t.split(None, 1)
        op, operands = parts[0], (parts[1] if len(parts) > 1 else "")
        c = classify(op, operands)
        h[c] = h.get(c, 0) + 1
        ops.setdefault(c, {})
        ops[c][op] = ops[c].get(op, 0) + 1
    return h, ops


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if out_json in args:
        args.remove(out_json)
    lines = open(args[0]).read().splitlines()
    result = {}
    for name in args[1:]:
        h, ops = histogram(function_body(lines, name))
        valu = sum(v for k, v in h.items() if k in ("vop2_f32", "vop2_int", "vop2_other", "vop1", "vop3", "vopc", "packed", "dpp", "sdwa", "lane"))
        result[name] = {"vector_instructions": valu, "classes": dict(sorted(h.items())),
                        "top": {c: dict(sorted(o.items(), key=lambda kv: -kv[1])[:8]) for c, o in sorted(ops.items()) if c not in ("salu", "s_nop", "s_waitcnt", "smem", "lds", "vmem")}}
        print(name, "vector", valu, {k: v for k, v in sorted(h.items())})
    if out_json:
        with open(out_json, "w") as f:
            json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
