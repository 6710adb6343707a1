#!/usr/bin/env python3
"""Lift the reference's in-file known-answer vectors into tests/golden/ref_kats.json.

The reference (ninjasource/lc3-codec v0.2.0) has no tests/ directory: every
golden vector is a numeric array literal inside a `#[test]` function at the
bottom of a source module (SURVEY.md section 8c lists them).  This script
copies the *numbers* (inputs and expected outputs) - never code - into one JSON
fixture so that the GPU box, which has no /root/reference, can check the oracle
and the HIP path against them.

Layout of the fixture:
  { "<path relative to src/>::<test fn>": [ {"name": <let-binding or "">,
                                               "line": <1-based line of the literal>,
                                               "values": [...]}, ... in source order ] }
Float literals are stored as their decimal text converted by Python's float();
every literal in the reference's tests is a shortest round-trip f32 decimal, so
np.float32(value) recovers the exact f32 the Rust compiler produced.
Run in the build container only:  python tools/extract_goldens.py
"""
import json
import re
import sys
from pathlib import Path

SRC = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "ref_kats.json"

NUM = r"[-+]?(?:0x[0-9a-fA-F_]+|\d[\d_]*\.?[\d_]*(?:[eE][-+]?\d+)?)"
TOKEN_OK = re.compile(rf"^\s*(?:{NUM}|true|false)\s*$")


def strip_comments(text):
    # keep line structure so that line numbers survive
    return re.sub(r"//[^\n]*", "", text)


def parse_list(text, pos):
    """Parse a bracketed literal starting at text[pos] == '['.  Returns (value|None, end)."""
    assert text[pos] == "["
    pos += 1
    items = []
    tok = ""
    ok = True
    while pos < len(text):
        c = text[pos]
        if c == "[":
            sub, pos = parse_list(text, pos)
            if sub is None:
                ok = False
            else:
                items.append(sub)
            continue
        if c == "]":
            if tok.strip():
                if TOKEN_OK.match(tok):
                    items.append(conv(tok))
                else:
                    ok = False
            return (items if ok else None), pos + 1
        if c == ",":
            if tok.strip():
                if TOKEN_OK.match(tok):
                    items.append(conv(tok))
                else:
                    ok = False
            tok = ""
        elif c == ";":
            ok = False  # `[0.0; 480]` style fill, not a vector
            tok = ""
        else:
            tok += c
        pos += 1
    return None, pos


def conv(tok):
    t = tok.strip().replace("_", "")
    if t == "true":
        return True
    if t == "false":
        return False
    if t.lower().startswith("0x") or t.lower().startswith("-0x"):
        return int(t, 16)
    if re.match(r"^[-+]?\d+$", t):
        return int(t)
    return float(t)


def extract_file(path):
    text = strip_comments(path.read_text())
    out = {}
    tests = [m for m in re.finditer(r"#\[test\]\s*(?:#\[[^\]]*\]\s*)*fn\s+(\w+)\s*\(", text)]
    for idx, m in enumerate(tests):
        start = m.end()
        end = tests[idx + 1].start() if idx + 1 < len(tests) else len(text)
        body = text[start:end]
        entries = []
        pos = 0
        while True:
            i = body.find("[", pos)
            if i < 0:
                break
            # skip attribute brackets `#[...]` and index expressions `x[...]`
            prev = body[:i].rstrip()
            if prev.endswith("#") or (prev and (prev[-1].isalnum() or prev[-1] in "_)]")):
                pos = i + 1
                continue
            val, endp = parse_list(body, i)
            if val is not None and len(val) >= 1:
                mm = re.search(r"let\s+(?:mut\s+)?(\w+)\s*(?::[^=]+)?=\s*&?(?:mut\s+)?$", body[:i])
                name = mm.group(1) if mm else ""
                line = text[: start + i].count("\n") + 1
                entries.append({"name": name, "line": line, "values": val})
                pos = endp
            else:
                pos = i + 1
        if entries:
            out[m.group(1)] = entries
    return out


def main():
    fixture = {}
    for path in sorted(SRC.rglob("*.rs")):
        if "tables" in path.parts or path.name == "wav.rs":
            continue
        rel = path.relative_to(SRC).as_posix()
        for fn, entries in extract_file(path).items():
            fixture[f"{rel}::{fn}"] = entries
    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text(json.dumps(fixture, separators=(",", ":")))
    n = sum(len(v) for v in fixture.values())
    print(f"wrote {OUT}: {len(fixture)} tests, {n} vectors, {OUT.stat().st_size} bytes", file=sys.stderr)
    for k, v in fixture.items():
        print(k, [(e["name"], e["line"], len(e["values"])) for e in v])


if __name__ == "__main__":
    main()
