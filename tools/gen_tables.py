#!/usr/bin/env python3
"""Generate tables/lc3_tables.h from the LC3 constant tables.

The LC3 codec is defined by a set of constant tables from the Bluetooth LC3
specification (MDCT windows, band edges, SNS codebooks, arithmetic-coder
models, LTPF filters).  The reference carries them as typed constants in
`src/tables/*.rs`.  This script reads those *data* files (run in the build
container only, where /root/reference exists), converts every value with the
reference's own typing rules and emits one C header that both the CPU oracle
and the HIP product include:

  * f32 tables: every decimal literal is rounded ONCE, decimal -> binary32
    (exact rational arithmetic, ties-to-even), and emitted as its IEEE-754 bit
    pattern so no compiler can re-round it.  `a / b` entries (SNS gains) are
    evaluated as an f32 division of two exactly representable integers.
  * integer tables keep their values; the C type is the narrowest that holds
    the table (documented per table in the header).

Nothing but numbers is taken from the reference.  The output is committed; the
GPU box never needs /root/reference.
"""
import re
import struct
import sys
from fractions import Fraction
from pathlib import Path

REF = Path("/root/reference/src/tables")
OUT = Path(__file__).resolve().parent.parent / "tables" / "lc3_tables.h"


def f32_bits_from_fraction(q: Fraction) -> int:
    """Correctly rounded (nearest-even) binary32 bit pattern of an exact rational."""
    if q == 0:
        return 0
    sign = 0
    if q < 0:
        sign = 0x80000000
        q = -q
    # start from the double approximation, then fix up exactly
    approx = struct.unpack("<I", struct.pack("<f", float(q)))[0]
    best = None
    for cand in (approx - 1, approx, approx + 1):
        if cand < 0 or cand >= 0x7F800000:
            continue
        v = Fraction(struct.unpack("<f", struct.pack("<I", cand))[0])
        err = abs(v - q)
        key = (err, cand & 1)  # ties -> even mantissa
        if best is None or key < best[0]:
            best = (key, cand)
    return sign | best[1]


def parse_float_token(tok: str) -> int:
    tok = tok.strip().replace("_", "")
    if "/" in tok:
        a, b = tok.split("/")
        fa = Fraction(a.strip())
        fb = Fraction(b.strip())
        # both operands are small integers, exactly representable in f32
        return f32_bits_from_fraction(fa / fb)
    return f32_bits_from_fraction(Fraction(tok))


def strip_comments(text: str) -> str:
    return re.sub(r"//[^\n]*", "", text)


def find_consts(text: str):
    """Yield (name, type_str, body_str) for each `pub const NAME: TYPE = BODY;`."""
    text = strip_comments(text)
    for m in re.finditer(r"pub const (\w+):\s*([^=]+?)\s*=\s*", text):
        name, ty = m.group(1), m.group(2).strip()
        i = m.end()
        if text[i] != "[":
            j = text.index(";", i)
            yield name, ty, text[i:j].strip()
            continue
        depth = 0
        j = i
        while True:
            c = text[j]
            if c == "[":
                depth += 1
            elif c == "]":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        yield name, ty, text[i : j + 1]


def parse_nested(body: str):
    """Parse a (possibly nested) bracketed list of tokens into nested python lists of strings."""
    pos = 0

    def parse():
        nonlocal pos
        assert body[pos] == "["
        pos += 1
        items = []
        tok = ""
        while True:
            c = body[pos]
            if c == "[":
                items.append(parse())
            elif c == "]":
                if tok.strip():
                    items.append(tok.strip())
                pos += 1
                return items
            elif c == ",":
                if tok.strip():
                    items.append(tok.strip())
                tok = ""
                pos += 1
            else:
                tok += c
                pos += 1

    return parse()


def dims_of(ty: str):
    # e.g. [[Scaler; 8]; 32] -> base Scaler, dims [32, 8]
    dims = []
    t = ty
    while t.startswith("["):
        inner, n = t[1:-1].rsplit(";", 1)
        dims.append(int(n.strip()))
        t = inner.strip()
    return t, dims


def flatten(x):
    if isinstance(x, list):
        for y in x:
            yield from flatten(y)
    else:
        yield x


def emit_table(name, base, dims, flat_tokens, out):
    n = 1
    for d in dims:
        n *= d
    assert len(flat_tokens) == n, (name, len(flat_tokens), n)
    dimstr = "".join(f"[{d}]" for d in dims)
    if base == "Scaler":
        bits = [parse_float_token(t) for t in flat_tokens]
        out.append(f"/* f32 bit patterns */\nLC3_TABLE_QUAL uint32_t LC3T_{name}_BITS{dimstr} LC3_TABLE_ALIGN = {{")
        vals = [f"0x{b:08x}u" for b in bits]
        per = 8
    else:
        ints = [int(t.replace("_", ""), 0) for t in flat_tokens]
        lo, hi = min(ints), max(ints)
        if base == "u8":
            cty = "uint8_t"
        elif base in ("i16",):
            cty = "int16_t"
        elif base == "u16":
            cty = "uint16_t"
        else:  # usize
            cty = "uint16_t" if hi < 65536 else "uint32_t"
            assert lo >= 0
        out.append(f"LC3_TABLE_QUAL {cty} LC3T_{name}{dimstr} LC3_TABLE_ALIGN = {{")
        vals = [str(v) for v in ints]
        per = 16
    # nested braces are optional in C for multi-dim arrays; emit flat rows
    lines = []
    for i in range(0, len(vals), per):
        lines.append("  " + ", ".join(vals[i : i + per]) + ",")
    out.extend(lines)
    out.append("};\n")


def main():
    out = []
    out.append("/* GENERATED by tools/gen_tables.py -- do not edit.")
    out.append(" * LC3 specification constant tables (data only), typed as the reference types them")
    out.append(" * (reference: src/tables/{mdct_windows,band_index_tables,spec_noise_shape_quant_tables,")
    out.append(" * spectral_data_tables,temporal_noise_shaping_tables,long_term_post_filter_coef}.rs).")
    out.append(" * f32 tables are stored as IEEE-754 bit patterns (suffix _BITS) so that the single")
    out.append(" * decimal->binary32 rounding the reference's compiler performs is reproduced exactly.")
    out.append(" * Define LC3_TABLE_QUAL before including (e.g. `static const` or `__device__ const`). */")
    out.append("#ifndef LC3_TABLES_H_\n#define LC3_TABLES_H_\n#include <stdint.h>\n")
    out.append("#ifndef LC3_TABLE_QUAL\n#define LC3_TABLE_QUAL static const\n#endif\n")
    out.append("/* every table is 16-byte aligned so that byte/short tables can also be fetched as 32-bit words */")
    out.append("#define LC3_TABLE_ALIGN __attribute__((aligned(16)))\n")
    files = [
        "mdct_windows.rs",
        "band_index_tables.rs",
        "spec_noise_shape_quant_tables.rs",
        "spectral_data_tables.rs",
        "temporal_noise_shaping_tables.rs",
        "long_term_post_filter_coef.rs",
    ]
    count = 0
    for fn in files:
        text = (REF / fn).read_text()
        out.append(f"/* ---- {fn} ---- */")
        for name, ty, body in find_consts(text):
            base, dims = dims_of(ty)
            if not dims:
                out.append(f"#define LC3T_{name} {int(body)}\n")
                continue
            toks = list(flatten(parse_nested(body)))
            emit_table(name, base, dims, toks, out)
            count += 1
    out.append("#endif /* LC3_TABLES_H_ */")
    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text("\n".join(out) + "\n")
    print(f"wrote {OUT} ({count} tables)", file=sys.stderr)


if __name__ == "__main__":
    main()
