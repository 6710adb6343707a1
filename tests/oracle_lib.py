"""ctypes loader for the CPU oracle (oracle/liblc3oracle.so).

Test infrastructure only: the product package never imports this module.
The library is built on demand with oracle/Makefile (plain gcc)."""
import ctypes
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liblc3oracle.so")

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        if not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.lc3o_encoder_new.restype = ctypes.c_void_p
        _lib.lc3o_decoder_new.restype = ctypes.c_void_p
        _lib.lc3o_encoder_free.argtypes = [ctypes.c_void_p]
        _lib.lc3o_decoder_free.argtypes = [ctypes.c_void_p]
        _lib.lc3o_kat_powf.restype = ctypes.c_float
        _lib.lc3o_kat_powf.argtypes = [ctypes.c_float, ctypes.c_float]
    return _lib


_native = None


def lib_native():
    """the oracle built with -O3 -march=native on THIS host (oracle/Makefile `native`; bench.py's second cpu_baseline leg); None when it
    cannot be built here"""
    global _native
    if _native is None:
        path = os.path.join(ORACLE_DIR, "liblc3oracle_native.so")
        try:
            subprocess.check_call(["make", "-s", "-B", "-C", ORACLE_DIR, "native"])  # -B: a copy built on another host is not this host's
            _native = ctypes.CDLL(path)
        except (subprocess.CalledProcessError, OSError):
            _native = False
    return _native or None


def P(a):
    """pointer to a numpy array's data"""
    return a.ctypes.data_as(ctypes.c_void_p)


_kats = None


def kats():
    global _kats
    if _kats is None:
        with open(os.path.join(ROOT, "tests", "golden", "ref_kats.json")) as f:
            _kats = json.load(f)
    return _kats


def kat(test, name=None, index=None, dtype=None):
    """Fetch one vector of a reference test: by let-binding name (n-th occurrence) or by position."""
    entries = kats()[test]
    if name is not None:
        sel = [e for e in entries if e["name"] == name]
        e = sel[index or 0]
    else:
        e = entries[index]
    v = e["values"]
    return np.array(v, dtype=dtype) if dtype is not None else v


class Encoder:
    """One channel of the oracle encoder (mirrors Lc3Encoder with num_channels = 1)."""

    def __init__(self, fs_hz=48000, frame_us=10000):
        self.L = lib()
        self.h = ctypes.c_void_p(self.L.lc3o_encoder_new(fs_hz, frame_us))
        assert self.h, "unsupported configuration"
        cfg = np.zeros(7, np.int32)
        self.L.lc3o_kat_config(fs_hz, frame_us, P(cfg))
        self.nf = int(cfg[5])
        self.ne = int(cfg[2])

    def encode_frame(self, pcm, nbytes):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        assert pcm.size == self.nf
        out = np.zeros(nbytes, np.uint8)
        self.L.lc3o_encode_frame(self.h, P(pcm), P(out), nbytes)
        return out

    def __del__(self):
        try:
            self.L.lc3o_encoder_free(self.h)
        except Exception:
            pass


class Decoder:
    def __init__(self, fs_hz=48000, frame_us=10000):
        self.L = lib()
        self.h = ctypes.c_void_p(self.L.lc3o_decoder_new(fs_hz, frame_us))
        assert self.h, "unsupported configuration"
        cfg = np.zeros(7, np.int32)
        self.L.lc3o_kat_config(fs_hz, frame_us, P(cfg))
        self.nf = int(cfg[5])
        self.ne = int(cfg[2])

    def decode_frame(self, buf, bits_per_sample=16):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        out = np.zeros(self.nf, np.int16)
        rc = self.L.lc3o_decode_frame(self.h, bits_per_sample, P(buf), int(buf.size), P(out))
        return rc, out

    def last_was_plc(self):
        return bool(self.L.lc3o_decoder_last_plc(self.h))

    def __del__(self):
        try:
            self.L.lc3o_decoder_free(self.h)
        except Exception:
            pass


def encode_batch(pcm, nbytes, fs_hz=48000, frame_us=10000, threads=1, spec_flags=0, library=None):
    """pcm int16[S][T][nf] -> uint8[S][T][nbytes]; every stream starts from a fresh encoder.  spec_flags: LC3O_SPEC_* bits
    (corrections of the reference's deviations from the specification; 0 = the reference's behaviour)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    S, T, nf = pcm.shape
    out = np.zeros((S, T, nbytes), np.uint8)
    rc = (library or lib()).lc3o_encode_batch_spec(fs_hz, frame_us, nbytes, S, T, P(pcm), P(out), threads, int(spec_flags))
    assert rc == 0
    return out


def decode_batch(data, nf, fs_hz=48000, frame_us=10000, threads=1, library=None):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    S, T, nbytes = data.shape
    out = np.zeros((S, T, nf), np.int16)
    rc = (library or lib()).lc3o_decode_batch(fs_hz, frame_us, nbytes, S, T, P(data), P(out), threads)
    assert rc == 0
    return out


def ltpf_transition_counts(reset=False):
    """How often each decoder LTPF transition case (index 1..5) ran inside the oracle since the last reset."""
    import ctypes
    cnt = (ctypes.c_long * 6).in_dll(lib(), "lc3o_ltpf_trans_count")
    ctypes.c_int.in_dll(lib(), "lc3o_ltpf_trans_counting").value = 1  # the statistic is off until a test asks for it
    out = [int(v) for v in cnt]
    if reset:
        for i in range(6):
            cnt[i] = 0
    return out


def encoder_path_counts(reset=False):
    """[frames through the quantiser, frames that take a second quantise + bit-count pass, frames with an active TNS filter, lsb_mode frames]
    inside the oracle since the last reset (tools/quantiser_paths.py)."""
    cnt = (ctypes.c_long * 4).in_dll(lib(), "lc3o_enc_path_count")
    ctypes.c_int.in_dll(lib(), "lc3o_enc_path_counting").value = 1
    out = [int(v) for v in cnt]
    if reset:
        for i in range(4):
            cnt[i] = 0
    return out


def timed_run(pcm, nbytes, fs_hz=48000, frame_us=10000, threads=1, roundtrip=True, seconds=5.0, library=None):
    """bench.py's cpu_baseline leg: `threads` host threads, each with ONE persistent encoder (and decoder) that codes its own stream
    of consecutive frames (pcm[t % S], cycled) for `seconds`; thread start-up, allocation and initialisation lie outside the timed
    region.  -> (frames coded, elapsed seconds)"""
    pcm = np.ascontiguousarray(pcm, np.int16)
    S, T, nf = pcm.shape
    frames = ctypes.c_double(0.0)
    elapsed = ctypes.c_double(0.0)
    f = (library or lib()).lc3o_timed_run
    f.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                                                                 ctypes.POINTER(ctypes.c_double)]
    rc = f(fs_hz, frame_us, nbytes, T, P(pcm), S, int(threads), int(bool(roundtrip)), float(seconds), ctypes.byref(frames),
           ctypes.byref(elapsed))
    assert rc == 0, rc
    return frames.value, elapsed.value
