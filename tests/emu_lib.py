"""Loader for the CPU wave emulator (tests/emu/lc3_emu.cpp): the product's device headers compiled with g++
and run as 64 host threads per wavefront.  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
LIB = os.path.join(EMU_DIR, "liblc3emu.so")
_lib = None


def build():
    deps = [os.path.join(EMU_DIR, "lc3_emu.cpp"), os.path.join(ROOT, "tables", "lc3_tables.h")]
    csrc = os.path.join(ROOT, "lc3-codec_amd", "csrc")
    deps += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                           "-fno-strict-aliasing", "-Wno-unknown-pragmas", "-Wno-attributes", "-o", LIB,
                           os.path.join(EMU_DIR, "lc3_emu.cpp"), "-lpthread"])
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        for n in ("pow10f", "log2f", "log10f", "exp2f", "asinf", "sinf_small", "exp2_raw"):
            f = getattr(_lib, "lc3emu_" + n)
            f.restype = ctypes.c_float
            f.argtypes = [ctypes.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def encode(pcm, nbytes, fs_hz=48000, frame_us=10000, dbg=None, spec_flags=0):
    pcm = np.ascontiguousarray(pcm, np.int16)
    S, T, _ = pcm.shape
    out = np.zeros((S, T, nbytes), np.uint8)
    rc = lib().lc3emu_encode_spec(fs_hz, frame_us, nbytes, S, T, _p(pcm), _p(out), _p(dbg), int(spec_flags))
    assert rc == 0
    return out


def decode(data, nf, fs_hz=48000, frame_us=10000, bad=None, late=0):
    data = np.ascontiguousarray(data, np.uint8)
    S, T, nbytes = data.shape
    out = np.zeros((S, T, nf), np.int16)
    if bad is not None:
        bad = np.ascontiguousarray(bad, np.uint8)
    rc = lib().lc3emu_decode_late(fs_hz, frame_us, nbytes, S, T, _p(data), _p(bad), _p(out), int(late))
    assert rc == 0
    return out
