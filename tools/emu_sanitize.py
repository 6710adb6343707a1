#!/usr/bin/env python3
"""Device headers under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; GPU sanitizers are not available on the
pool).  Builds the CPU wave emulator (tests/emu/lc3_emu.cpp, which compiles lc3-codec_amd/csrc/lc3_dev_*.h unchanged) with
-fsanitize=address,undefined and runs encode + decode + random-garbage decode of seven configurations against the oracle.
Any out-of-bounds LDS / plane index or misaligned vector access in the device code aborts the run.

usage (the sanitizer runtimes have to be preloaded into python):
    LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \\
    ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 python tools/emu_sanitize.py
"""
import importlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu_lib  # noqa: E402
import oracle_lib as O  # noqa: E402

synth = importlib.import_module("lc3-codec_amd.synth")


def main():
    out = os.path.join(tempfile.mkdtemp(prefix="lc3emu_asan_"), "liblc3emu_asan.so")
    # integer wrap-around and float->int saturation are part of the reference's arithmetic: not flagged
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                           "-fno-strict-aliasing", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-sanitize=float-cast-overflow,shift-base,shift-exponent,signed-integer-overflow",
                           "-Wno-unknown-pragmas", "-Wno-attributes", "-o", out,
                           os.path.join(ROOT, "tests", "emu", "lc3_emu.cpp"), "-lpthread"])
    emu_lib.LIB = out
    emu_lib._lib = None
    emu_lib.build = lambda: out
    rng = np.random.default_rng(5)
    for fs, us, nb, nf in [(48000, 10000, 150, 480), (48000, 10000, 40, 480), (48000, 7500, 113, 360), (32000, 10000, 80, 320),
                           (16000, 7500, 30, 120), (24000, 10000, 60, 240), (44100, 10000, 110, 480)]:
        pcm = synth.make_pcm(6, 3, nf, fs, seed=3)
        ref = O.encode_batch(pcm, nb, fs, us)
        assert np.array_equal(ref, emu_lib.encode(pcm, nb, fs, us)), (fs, us, nb)
        assert np.array_equal(O.decode_batch(ref, nf, fs, us), emu_lib.decode(ref, nf, fs, us)), (fs, us, nb)
        junk = rng.integers(0, 256, size=(6, 3, nb), dtype=np.uint8)
        assert np.array_equal(O.decode_batch(junk, nf, fs, us), emu_lib.decode(junk, nf, fs, us)), ("garbage", fs, us)
        print("clean:", fs, us, nb, flush=True)


if __name__ == "__main__":
    main()
