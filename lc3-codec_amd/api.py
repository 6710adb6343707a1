"""ctypes binding of liblc3gpu.so and the Python mirror of the reference API.

Names, argument meaning and error behaviour follow the reference:
  Lc3Encoder.calc_working_buffer_lengths / new / encode_frame   (encoder/lc3_encoder.rs:117-209)
  Lc3Decoder.calc_working_buffer_lengths / new / decode_frame   (decoder/lc3_decoder.rs:181-244)
plus the batch entry points (`encode` / `decode`) that take DEVICE pointers
(e.g. torch tensors' data_ptr()) and a HIP stream.  No torch types cross the ABI.
"""
import ctypes
import enum
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# LC3GPU_PROFILE=1 selects the diagnostic build with in-kernel stage stamps (never used for timing claims)
_PROFILE = os.environ.get("LC3GPU_PROFILE", "0") == "1"
# LC3GPU_LIB=<file name under lib/> selects another build of the same sources (compiler / tuning experiments, e.g. one made with
# LC3_HIPCC_EXTRA="-DLC3_RECON_WAVES=6"); the default is the one build() produces
_LIB = os.path.join(_HERE, "lib", os.environ.get("LC3GPU_LIB") or ("liblc3gpu_prof.so" if _PROFILE else "liblc3gpu.so"))
_SRC = os.path.join(_HERE, "csrc", "lc3gpu.hip")
# host-only sources of the library on top of its own C ABI (the pipeline object): plain C++ beside the HIP translation units
_HOST_SRCS = [os.path.join(_HERE, "csrc", "lc3gpu_pipeline.cpp")]

class Lc3GpuError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = _strerror(code)
        super().__init__(f"lc3gpu error {code} ({msg}){': ' + what if what else ''}")


class Lc3EncoderError(Lc3GpuError):
    """The reference's Lc3EncoderError is an empty enum (lc3_encoder.rs:29-30): encode never returns Err.
    Raised here only where the reference panics (bad channel index / slice length)."""


class Lc3DecoderError(Lc3GpuError):
    """Lc3DecoderError::Only16BitsPerAudioSampleSupported (lc3_decoder.rs:36-42,80-82) and the panics."""


class SamplingFrequency(enum.IntEnum):  # common/config.rs:1-9
    Hz8000 = 8000
    Hz16000 = 16000
    Hz24000 = 24000
    Hz32000 = 32000
    Hz44100 = 44100
    Hz48000 = 48000


class FrameDuration(enum.IntEnum):  # common/config.rs:11-15 (microseconds)
    SevenPointFiveMs = 7500
    TenMs = 10000


_lib = None


def library_path():
    return _LIB


def _translation_units():
    """(kind, index) of every translation unit of the multi-unit build (csrc/lc3gpu.hip, "translation units"): the main unit (0: host
    side + the whole-source module's device code), the encoder kernels (1) and the decoder kernels (3) of every configuration view beyond
    the base ones, one unit per mixed-configuration kernel (2)"""
    import re

    with open(os.path.join(_HERE, "csrc", "lc3_cfg_views.h")) as f:
        text = f.read()
    n_all = int(re.search(r"#define LC3_N_VIEWS_ALL (\d+)", text).group(1))
    n_base = int(re.search(r"#define LC3_N_VIEWS_BASE (\d+)", text).group(1))
    return [(0, 0)] + [(k, v) for v in range(n_base + 1, n_all + 1) for k in (1, 3)] + [(2, k) for k in range(8)]


def _sources_hash(srcs):
    """sha256 over the CONTENTS of every source a library is built from.  Freshness is decided by content, not by time stamps: an object
    whose compile started before an edit and ended after it is newer than the sources and still stale (seen in round 5: a test session's
    build running while the sources were being edited left exactly that)"""
    import hashlib

    h = hashlib.sha256()
    for path in sorted(srcs):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _build_signature(single, extra, extra_main, units, src_hash):
    """what a built library is a function of: the sources' contents and the build switches (advisor, round 4: switching LC3_SINGLE_TU /
    LC3_HIPCC_EXTRA used to leave a library of the other flavour in place)"""
    return "sources=%s single=%d extra=%s extra_main=%s units=%s profile=%d\n" % (
        src_hash, single, " ".join(extra), " ".join(extra_main), ",".join("%d_%d" % u for u in units), _PROFILE)


def build_native(force=False, verbose=False):
    """Compile csrc/lc3gpu.hip for gfx950 into lib/liblc3gpu.so (hipcc cross-compiles without a GPU).  The default library is built from
    several translation units of the same source side by side (device code generation is serial inside one unit; LC3_BUILD_JOBS, default
    the CPUs this process may use) and carries a compile-time view of every standard configuration; LC3_SINGLE_TU=1, the diagnostic build
    (LC3GPU_PROFILE=1) and builds with LC3_HIPCC_EXTRA compile it whole, with the four views of the round-1..4 library.
    LC3_HIPCC_EXTRA_MAIN="-D..." (timing experiments on the headline kernels): the multi-unit build with those flags on the MAIN unit only;
    the other units' objects are the production ones.  A library remembers the switches it was built with (<lib>.stamp) and is rebuilt
    when they differ; experiment builds need their own file name (LC3GPU_LIB) so that they never replace the default library."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc")) if f.endswith((".h", ".hip", ".cpp"))]
    srcs += [os.path.join(_ROOT, "include", "lc3gpu.h"), os.path.join(_ROOT, "tables", "lc3_tables.h")]
    src_hash = _sources_hash(srcs)  # taken BEFORE anything is compiled: what the objects below are guaranteed to be at least as old as
    # (the HIP translation units do not see the host-only sources: an edit there relinks the library without recompiling them)
    dev_hash = _sources_hash([p_ for p_ in srcs if not p_.endswith(".cpp")])
    extra = os.environ.get("LC3_HIPCC_EXTRA", "").split()  # compiler experiments
    extra_main = os.environ.get("LC3_HIPCC_EXTRA_MAIN", "").split()
    if (extra or extra_main) and not os.environ.get("LC3GPU_LIB"):
        raise RuntimeError("LC3_HIPCC_EXTRA / LC3_HIPCC_EXTRA_MAIN builds are experiments: name their library with LC3GPU_LIB=liblc3gpu_<what>.so")
    single = _PROFILE or bool(extra) or os.environ.get("LC3_SINGLE_TU", "0") == "1"
    units = [] if single else _translation_units()
    sig = _build_signature(single, extra, extra_main, units, src_hash)
    stamp = _LIB + ".stamp"

    def stamped(path, text):
        try:
            with open(path) as f:
                return f.read() == text
        except OSError:
            return False

    if not force and os.path.exists(_LIB) and stamped(stamp, sig):
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
             "-Wno-missing-braces"]
    if single:
        cmd = ["hipcc"] + extra + (["-DLC3_PROFILE"] if _PROFILE else []) + flags + ["-shared", "-o", _LIB, _SRC] + _HOST_SRCS
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(sig)
        return _LIB
    from concurrent.futures import ThreadPoolExecutor

    objdir = os.path.join(os.path.dirname(_LIB), "obj")
    os.makedirs(objdir, exist_ok=True)
    tag = os.path.splitext(os.path.basename(_LIB))[0]

    def unit_obj(u):  # an experiment's main unit gets an object of its own; every other object is shared with the production build
        return os.path.join(objdir, ("%s_0_0.o" % tag) if (extra_main and u == (0, 0)) else "lc3gpu_%d_%d.o" % u)

    def compile_unit(u):
        obj, tmp = unit_obj(u), unit_obj(u) + ".tmp%d" % os.getpid()
        cmd = ["hipcc", "-DLC3_TU_KIND=%d" % u[0], "-DLC3_TU_INDEX=%d" % u[1]] + (extra_main if u == (0, 0) else []) + flags + ["-c", "-o", tmp, _SRC]
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, obj)  # (a unit that fails leaves no partial object behind)
            with open(obj + ".stamp", "w") as f:
                f.write(unit_sig(u))
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        return obj

    def unit_sig(u):
        return "sources=%s flags=%s\n" % (dev_hash, " ".join(extra_main) if u == (0, 0) else "")

    def fresh(u):
        o = unit_obj(u)
        return os.path.exists(o) and stamped(o + ".stamp", unit_sig(u))

    try:
        jobs = len(os.sched_getaffinity(0))
    except AttributeError:
        jobs = os.cpu_count() or 1
    jobs = max(1, int(os.environ.get("LC3_BUILD_JOBS", jobs)))
    # (the main unit and the mixed-kernel units are the long ones: first).  Objects stamped with these sources' hash are kept unless forced;
    # an experiment build never recompiles the shared ones it finds up to date
    order = sorted(units, key=lambda u: (u[0] != 0, u[0] != 2))
    todo = [u for u in order if (force and (not extra_main or u == (0, 0))) or not fresh(u)]
    if verbose:
        print("hipcc -DLC3_TU_KIND=k -DLC3_TU_INDEX=i %s -c %s   x %d of %d translation units, %d at a time" % (
            " ".join(flags), _SRC, len(todo), len(units), jobs))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(compile_unit, todo))
    host_objs = []
    for src in _HOST_SRCS:  # (seconds each: always recompiled)
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-fPIC", "-c", "-o", obj, src])
        host_objs.append(obj)
    cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", _LIB] + sorted(unit_obj(u) for u in units) + host_objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(sig)
    return _LIB


def tool_path():
    return os.path.join(os.path.dirname(_LIB), "lc3gpu-tool")


def build_tool(force=False, verbose=False):
    """Compile the file-driver command line tool (host/lc3_files.cpp, host/lc3gpu_tool.cpp) against liblc3gpu.so."""
    host = os.path.join(_HERE, "host")
    srcs = [os.path.join(host, f) for f in ("lc3_files.cpp", "lc3gpu_tool.cpp")]
    deps = srcs + [os.path.join(host, "lc3_files.hpp"), os.path.join(_ROOT, "include", "lc3gpu.h"),
                   os.path.join(_ROOT, "include", "lc3gpu.hpp"), _LIB]
    out = tool_path()
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = ["hipcc", "-O2", "-std=c++17", "-Wno-unused-result", "-o", out] + srcs + [
        "-L" + os.path.dirname(_LIB), "-llc3gpu", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def load_library():
    """Load liblc3gpu.so; fails loudly if the native extension is missing (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64
    # under torch/lib.  If the system runtime (/opt/rocm) is mapped first through this library, a later
    # `import torch` binds to the wrong runtime and torch.cuda reports no device.  When torch is installed
    # (it is only used by callers for device memory / streams), let it bring in its runtime first.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for the C ABI
        pass
    if not os.path.exists(_LIB):
        raise ImportError(f"{_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the LC3 engine has no non-native path)")
    L = ctypes.CDLL(_LIB)
    vp, i, pi64 = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)
    L.lc3gpu_strerror.restype = ctypes.c_char_p
    L.lc3gpu_strerror.argtypes = [i]
    L.lc3gpu_config.argtypes = [i, i, vp]
    L.lc3gpu_encoder_working_buffer_lengths.argtypes = [i, i, i, pi64]
    L.lc3gpu_decoder_working_buffer_lengths.argtypes = [i, i, i, pi64]
    L.lc3gpu_encoder_create.argtypes = [ctypes.POINTER(vp), i, i, i]
    L.lc3gpu_encoder_create_spec.argtypes = [ctypes.POINTER(vp), i, i, i, i]
    L.lc3gpu_encoder_create_mixed_spec.argtypes = [ctypes.POINTER(vp), i, vp, i]
    L.lc3gpu_encoder_destroy.argtypes = [vp]
    L.lc3gpu_encoder_reset.argtypes = [vp]
    L.lc3gpu_encode_frame.argtypes = [vp, i, vp, i, vp, i]
    L.lc3gpu_encode_frame_debug.argtypes = [vp, vp, i, vp, i, vp]
    L.lc3gpu_encode.argtypes = [vp, vp, vp, i, i, vp]
    L.lc3gpu_encode_range.argtypes = [vp, i, i, vp, vp, i, i, vp]
    L.lc3gpu_encoder_state_size.restype = ctypes.c_size_t
    L.lc3gpu_encoder_state_size.argtypes = [vp]
    L.lc3gpu_encoder_state_save.argtypes = [vp, vp, ctypes.c_size_t]
    L.lc3gpu_encoder_state_load.argtypes = [vp, vp, ctypes.c_size_t]
    L.lc3gpu_encode_layout.argtypes = [vp, i, vp, vp, i, i, vp]
    L.lc3gpu_encoder_create_mixed.argtypes = [ctypes.POINTER(vp), i, vp]
    L.lc3gpu_encode_mixed.argtypes = [vp, vp, vp, i, vp]
    L.lc3gpu_decoder_create.argtypes = [ctypes.POINTER(vp), i, i, i]
    L.lc3gpu_decoder_destroy.argtypes = [vp]
    L.lc3gpu_decoder_reset.argtypes = [vp]
    L.lc3gpu_decode_frame.argtypes = [vp, i, i, vp, i, vp, i]
    L.lc3gpu_decode.argtypes = [vp, vp, vp, vp, i, i, vp]
    L.lc3gpu_decode_range.argtypes = [vp, i, i, vp, vp, vp, i, i, vp]
    L.lc3gpu_decoder_state_size.restype = ctypes.c_size_t
    L.lc3gpu_decoder_state_size.argtypes = [vp]
    L.lc3gpu_decoder_state_save.argtypes = [vp, vp, ctypes.c_size_t]
    L.lc3gpu_decoder_state_load.argtypes = [vp, vp, ctypes.c_size_t]
    L.lc3gpu_decode_layout.argtypes = [vp, i, vp, vp, vp, i, i, vp]
    L.lc3gpu_decoder_create_mixed.argtypes = [ctypes.POINTER(vp), i, vp]
    L.lc3gpu_decode_mixed.argtypes = [vp, vp, vp, vp, i, vp]
    L.lc3gpu_decoder_plc_events.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.lc3gpu_encoder_pair_timeouts.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.lc3gpu_decoder_pair_timeouts.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.lc3gpu_kernel_info.argtypes = [i, vp]
    L.lc3gpu_encoder_timing.argtypes = [vp, i, vp]
    L.lc3gpu_decoder_timing.argtypes = [vp, i, vp]
    L.lc3gpu_decoder_timing_kernels.argtypes = [vp, i, vp]
    L.lc3gpu_decode_frame_debug.argtypes = [vp, i, vp, i, vp, i, vp]
    L.lc3gpu_decoder_synth_debug.argtypes = [vp, i, vp, i, i, i, i, vp, i, vp]
    L.lc3gpu_selftest_math.argtypes = [i, vp, vp, i, vp]
    L.lc3gpu_clock_probe.argtypes = [vp, vp, i]
    L.lc3gpu_encoder_stage_event.argtypes = [vp, i, vp]
    L.lc3gpu_decoder_stage_event.argtypes = [vp, i, vp]
    L.lc3gpu_encoder_bind_stream.argtypes = [vp, vp, i]
    L.lc3gpu_decoder_bind_stream.argtypes = [vp, vp, i]
    L.lc3gpu_encoder_debug_pair_giveup.argtypes = [vp]
    L.lc3gpu_decoder_debug_pair_giveup.argtypes = [vp]
    L.lc3gpu_encode_host.argtypes = [vp, vp, vp, i, i]
    L.lc3gpu_decode_host.argtypes = [vp, vp, vp, vp, i, i]
    L.lc3gpu_host_alloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
    L.lc3gpu_host_free.argtypes = [vp]
    L.lc3gpu_pipeline_create.argtypes = [ctypes.POINTER(vp), i, i, i, i]
    L.lc3gpu_pipeline_create_mixed.argtypes = [ctypes.POINTER(vp), i, vp, i, vp]
    L.lc3gpu_pipeline_submit_mixed.argtypes = [vp, vp, vp, vp, i]
    L.lc3gpu_pipeline_encode_mixed.argtypes = [vp, vp, vp, i]
    L.lc3gpu_pipeline_decode_mixed.argtypes = [vp, vp, vp, vp, i]
    L.lc3gpu_pipeline_destroy.argtypes = [vp]
    L.lc3gpu_pipeline_reset.argtypes = [vp]
    L.lc3gpu_pipeline_submit.argtypes = [vp, vp, vp, vp, i, i]
    L.lc3gpu_pipeline_encode.argtypes = [vp, vp, vp, i, i]
    L.lc3gpu_pipeline_decode.argtypes = [vp, vp, vp, vp, i, i]
    L.lc3gpu_pipeline_wait.argtypes = [vp]
    L.lc3gpu_pipeline_join.argtypes = [vp, vp]
    L.lc3gpu_pipeline_follow.argtypes = [vp, vp]
    L.lc3gpu_pipeline_mark.argtypes = [vp, vp]
    L.lc3gpu_pipeline_groups.argtypes = [vp]
    L.lc3gpu_pipeline_group.argtypes = [vp, i, ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.lc3gpu_pipeline_last_hip_error.argtypes = [vp]
    _lib = L
    return L


# every symbol include/lc3gpu.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "lc3gpu_version", "lc3gpu_strerror", "lc3gpu_last_hip_error", "lc3gpu_device_count", "lc3gpu_config",
    "lc3gpu_encoder_working_buffer_lengths", "lc3gpu_decoder_working_buffer_lengths", "lc3gpu_encoder_create",
    "lc3gpu_encoder_destroy", "lc3gpu_encoder_reset", "lc3gpu_encode_frame", "lc3gpu_encode", "lc3gpu_encode_range",
    "lc3gpu_encoder_state_size", "lc3gpu_encoder_state_save", "lc3gpu_encoder_state_load", "lc3gpu_decoder_create",
    "lc3gpu_decoder_destroy", "lc3gpu_decoder_reset", "lc3gpu_decode_frame", "lc3gpu_decode", "lc3gpu_decode_range",
    "lc3gpu_decoder_state_size", "lc3gpu_decoder_state_save", "lc3gpu_decoder_state_load",
    "lc3gpu_decoder_plc_events", "lc3gpu_encode_frame_debug", "lc3gpu_kernel_info", "lc3gpu_prof_read", "lc3gpu_encoder_timing", "lc3gpu_decoder_timing",
    "lc3gpu_decoder_timing_kernels", "lc3gpu_decode_frame_debug", "lc3gpu_decoder_synth_debug", "lc3gpu_selftest_math", "lc3gpu_encode_layout", "lc3gpu_decode_layout", "lc3gpu_encoder_create_mixed", "lc3gpu_decoder_create_mixed", "lc3gpu_encode_mixed",
    "lc3gpu_decode_mixed", "lc3gpu_encoder_create_spec", "lc3gpu_encoder_create_mixed_spec", "lc3gpu_clock_probe",
    "lc3gpu_encoder_stage_event", "lc3gpu_decoder_stage_event", "lc3gpu_encoder_pair_timeouts", "lc3gpu_decoder_pair_timeouts",
    "lc3gpu_encoder_debug_pair_giveup", "lc3gpu_decoder_debug_pair_giveup", "lc3gpu_encoder_bind_stream", "lc3gpu_decoder_bind_stream", "lc3gpu_encode_host", "lc3gpu_decode_host", "lc3gpu_host_alloc",
    "lc3gpu_host_free", "lc3gpu_pipeline_create", "lc3gpu_pipeline_create_mixed", "lc3gpu_pipeline_submit_mixed", "lc3gpu_pipeline_encode_mixed",
    "lc3gpu_pipeline_decode_mixed", "lc3gpu_pipeline_destroy", "lc3gpu_pipeline_reset", "lc3gpu_pipeline_submit",
    "lc3gpu_pipeline_encode", "lc3gpu_pipeline_decode", "lc3gpu_pipeline_wait", "lc3gpu_pipeline_join", "lc3gpu_pipeline_follow", "lc3gpu_pipeline_mark",
    "lc3gpu_pipeline_groups", "lc3gpu_pipeline_group", "lc3gpu_pipeline_last_hip_error",
]

# LC3GPU_SPEC_*: opt-in corrections of the reference's deviations from the LC3 specification (default 0 = reference behaviour)
SPEC_8KHZ_ENCODE, SPEC_TNS_SSWB_STOP, SPEC_BW_CUTOFF_DB, SPEC_SNS_LAST_GAIN, SPEC_NBITS_SPEC_OLD, SPEC_ALL = 1, 2, 4, 8, 16, 31

LAYOUT_PLANAR, LAYOUT_INTERLEAVED = 0, 1
# stage events (LC3GPU_ENC_STAGE_* / LC3GPU_DEC_STAGE_*)
ENC_STAGE_FRONT, ENC_STAGE_VQ, ENC_STAGE_BACK, DEC_STAGE_PARSE = 0, 1, 2, 0


def _event_handle(event):
    """a hipEvent_t as an integer: None, an integer, or an object with `cuda_event` (torch.cuda.Event -- record it once before passing it:
    torch creates the HIP event lazily)"""
    if event is None:
        return None
    h = getattr(event, "cuda_event", event)
    if not h:
        raise ValueError("the event has no HIP handle yet (torch.cuda.Event: record it once first)")
    return ctypes.c_void_p(int(h))
# stage dumps of Lc3Decoder.decode_frame_debug / synth_debug (LC3GPU_DBG_*)
DBG_INT, DBG_SPEC, DBG_IMDCT, DBG_LTPF, DBG_GAIN, DBG_TNS, DBG_FLOATS = 0, 400, 800, 1280, 1760, 2160, 2560
RECON_LANE, RECON_LATE, RECON_WAVE = 0, 1, 2
# stage dumps of Lc3Encoder.encode_frame_debug (LC3GPU_ENC_DBG_*)
ENC_DBG_SCALARS, ENC_DBG_EB, ENC_DBG_ATTACK, ENC_DBG_FLOATS = 1440, 1472, 1536, 1600


class StreamDesc(ctypes.Structure):
    """lc3gpu_stream_desc: one stream of a mixed-configuration handle"""
    _fields_ = [("fs_hz", ctypes.c_int), ("frame_us", ctypes.c_int), ("nbytes", ctypes.c_int)]


def _desc_array(descs):
    arr = (StreamDesc * len(descs))()
    for k, d in enumerate(descs):
        arr[k].fs_hz, arr[k].frame_us, arr[k].nbytes = int(d[0]), int(d[1]), int(d[2])
    return arr


def _layout(layout):
    if layout in (LAYOUT_PLANAR, "planar", None):
        return LAYOUT_PLANAR
    if layout in (LAYOUT_INTERLEAVED, "interleaved"):
        return LAYOUT_INTERLEAVED
    raise ValueError("layout must be 'planar' or 'interleaved'")



def _strerror(code):
    try:
        return load_library().lc3gpu_strerror(int(code)).decode()
    except Exception:
        return "?"


def selftest_math(which, x=None, d=None, n=None):
    """tests only: a float routine evaluated on the device (lc3gpu_selftest_math) -> float32[n]"""
    L = load_library()
    if x is not None:
        x = np.ascontiguousarray(x, np.float32)
        n = x.size
    if d is not None:
        d = np.ascontiguousarray(d, np.float32)
    out = np.zeros(int(n), np.float32)
    rc = L.lc3gpu_selftest_math(int(which), _ptr(x), _ptr(d), int(n), _ptr(out))
    if rc:
        raise Lc3GpuError(rc, "selftest_math")
    return out


def clock_probe(d_out, stream=None, spin=50000):
    """measurement aid: a one-wave kernel on `stream` leaves {shader cycles, 100 MHz ticks} of its spin in d_out (device uint64[3]);
    asynchronous.  clock = 100 MHz x cycles / ticks"""
    rc = load_library().lc3gpu_clock_probe(_ptr(stream), _ptr(d_out), int(spin))
    if rc:
        raise Lc3GpuError(rc, "clock_probe")


def device_count():
    return int(load_library().lc3gpu_device_count())


def _ptr(x):
    """device pointer (int) or host numpy array -> c_void_p"""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(ctypes.c_void_p)
    if hasattr(x, "data_ptr"):
        return ctypes.c_void_p(x.data_ptr())
    return ctypes.c_void_p(int(x))


class Lc3Config:
    """common/config.rs:18-100"""

    def __init__(self, sampling_frequency, frame_duration):
        out = (ctypes.c_int * 7)()
        rc = load_library().lc3gpu_config(int(frame_duration), int(sampling_frequency), out)
        if rc:
            raise Lc3GpuError(rc, "Lc3Config")
        self.fs_ind, self.fs, self.ne, n10, self.nb, self.nf, self.z = list(out)
        self.n_ms = FrameDuration.TenMs if n10 else FrameDuration.SevenPointFiveMs


class Lc3Encoder:
    """`num_channels` independent encoder channels resident on the current HIP device."""

    @staticmethod
    def calc_working_buffer_lengths(num_channels, frame_duration, sampling_frequency):
        """-> (integer_len, scaler_len, complex_len), lc3_encoder.rs:194-209"""
        out = (ctypes.c_int64 * 3)()
        rc = load_library().lc3gpu_encoder_working_buffer_lengths(num_channels, int(frame_duration),
                                                                  int(sampling_frequency), out)
        if rc:
            raise Lc3EncoderError(rc)
        return tuple(int(v) for v in out)

    def __init__(self, num_channels, frame_duration, sampling_frequency, spec_flags=0):
        self._L = load_library()
        self.config = Lc3Config(sampling_frequency, frame_duration)
        self.num_channels = int(num_channels)
        h = ctypes.c_void_p()
        rc = self._L.lc3gpu_encoder_create_spec(ctypes.byref(h), self.num_channels, int(frame_duration),
                                                int(sampling_frequency), int(spec_flags))
        if rc:
            raise Lc3EncoderError(rc, "Lc3Encoder::new")
        self._h = h

    new = classmethod(lambda cls, *a, **k: cls(*a, **k))

    @classmethod
    def mixed(cls, descs, spec_flags=0):
        """one handle for streams of different configurations: descs = [(fs_hz, frame_us, nbytes), ...] (lc3gpu_encoder_create_mixed)"""
        self = cls.__new__(cls)
        self._L = load_library()
        self.descs = [(int(d[0]), int(d[1]), int(d[2])) for d in descs]
        self.configs = [Lc3Config(d[0], d[1]) for d in self.descs]
        self.config = None
        self.num_channels = len(self.descs)
        h = ctypes.c_void_p()
        rc = self._L.lc3gpu_encoder_create_mixed_spec(ctypes.byref(h), self.num_channels, _desc_array(self.descs), int(spec_flags))
        if rc:
            raise Lc3EncoderError(rc, "Lc3Encoder::mixed")
        self._h = h
        return self

    def encode_mixed(self, d_pcm, d_out, n_frames, stream=None):
        """ragged DEVICE buffers, streams in descriptor order (include/lc3gpu.h); one launch per kernel"""
        rc = self._L.lc3gpu_encode_mixed(self._h, _ptr(d_pcm), _ptr(d_out), int(n_frames), _ptr(stream))
        if rc:
            raise Lc3EncoderError(rc, "encode_mixed")

    def encode_frame(self, channel_index, samples_in, buf_out):
        """encode one frame of one channel; len(buf_out) selects the bitrate (lc3_encoder.rs:65,175-191)"""
        samples_in = np.ascontiguousarray(samples_in, dtype=np.int16)
        if not (isinstance(buf_out, np.ndarray) and buf_out.dtype == np.uint8 and buf_out.flags.c_contiguous):
            raise TypeError("buf_out must be a contiguous uint8 numpy array")
        rc = self._L.lc3gpu_encode_frame(self._h, int(channel_index), _ptr(samples_in), int(samples_in.size),
                                         _ptr(buf_out), int(buf_out.size))
        if rc:
            raise Lc3EncoderError(rc, "encode_frame")

    def encode_frame_debug(self, samples_in, nbytes):
        samples_in = np.ascontiguousarray(samples_in, dtype=np.int16)
        out = np.zeros(nbytes, np.uint8)
        dbg = np.zeros(ENC_DBG_FLOATS, np.float32)
        rc = self._L.lc3gpu_encode_frame_debug(self._h, _ptr(samples_in), int(samples_in.size), _ptr(out), nbytes,
                                               _ptr(dbg))
        if rc:
            raise Lc3EncoderError(rc, "encode_frame_debug")
        return out, dbg

    def encode(self, d_pcm, d_out, nbytes, n_frames, stream=None, first_channel=None, n_channels=None, layout="planar"):
        """batch: DEVICE int16[S][T][nf] -> DEVICE uint8[S][T][nbytes] (layout "interleaved": int16[T][nf][S] ->
        uint8[T][S][nbytes]), asynchronous on `stream`"""
        if first_channel is None:
            rc = self._L.lc3gpu_encode_layout(self._h, _layout(layout), _ptr(d_pcm), _ptr(d_out), int(nbytes), int(n_frames),
                                              _ptr(stream))
        else:
            rc = self._L.lc3gpu_encode_range(self._h, int(first_channel), int(n_channels), _ptr(d_pcm), _ptr(d_out),
                                             int(nbytes), int(n_frames), _ptr(stream))
        if rc:
            raise Lc3EncoderError(rc, "encode")

    def stage_event(self, stage, event):
        """every batch call from now on records `event` (the caller's; None clears) behind the kernels of `stage` (ENC_STAGE_*)"""
        rc = self._L.lc3gpu_encoder_stage_event(self._h, int(stage), _event_handle(event))
        if rc:
            raise Lc3EncoderError(rc, "stage_event")
        self._stage_events = getattr(self, "_stage_events", {})
        self._stage_events[int(stage)] = event  # keeps the event alive while it is set

    def timing(self, enable=True):
        """-> (front ms, vector-quantiser ms, back ms, pack ms, batch calls) since the last call; (re)arms recording (enable = n > 1: every n-th batch call)"""
        out = (ctypes.c_double * 5)()
        rc = self._L.lc3gpu_encoder_timing(self._h, int(enable), out)
        if rc:
            raise Lc3EncoderError(rc, "timing")
        return float(out[0]), float(out[1]), float(out[2]), float(out[3]), int(out[4])

    def reset(self):
        rc = self._L.lc3gpu_encoder_reset(self._h)
        if rc:
            raise Lc3EncoderError(rc, "reset")

    def encode_host(self, pcm, out, nbytes, n_frames):
        """HOST int16[S][T][nf] -> HOST uint8[S][T][nbytes] (numpy arrays or raw addresses, e.g. of PinnedBuffer); synchronous (lc3gpu_encode_host)"""
        rc = self._L.lc3gpu_encode_host(self._h, _ptr(pcm), _ptr(out), int(nbytes), int(n_frames))
        if rc:
            raise Lc3EncoderError(rc, "encode_host")

    def bind_stream(self, stream, bind=True):
        """every later batch call comes on `stream` (which outlives the handle): no per-call event of the handle's own (lc3gpu_encoder_bind_stream)"""
        rc = self._L.lc3gpu_encoder_bind_stream(self._h, _ptr(stream), int(bool(bind)))
        if rc:
            raise Lc3EncoderError(rc, "bind_stream")

    def debug_pair_giveup(self):
        """tests only: the device does what a pair half that gives up does (count + host flag)"""
        rc = self._L.lc3gpu_encoder_debug_pair_giveup(self._h)
        if rc:
            raise Lc3EncoderError(rc, "debug_pair_giveup")

    def pair_timeouts(self):
        """producer / consumer pair halves of the packer that ever gave up on their partner (include/lc3gpu.h); 0 unless a wave died"""
        v = ctypes.c_uint64()
        rc = self._L.lc3gpu_encoder_pair_timeouts(self._h, ctypes.byref(v))
        if rc:
            raise Lc3EncoderError(rc, "pair_timeouts")
        return int(v.value)

    def state_save(self):
        n = self._L.lc3gpu_encoder_state_size(self._h) * self.num_channels
        buf = np.zeros(n, np.uint8)
        rc = self._L.lc3gpu_encoder_state_save(self._h, _ptr(buf), buf.size)
        if rc:
            raise Lc3EncoderError(rc, "state_save")
        return buf

    def state_load(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        if buf.size != self._L.lc3gpu_encoder_state_size(self._h) * self.num_channels:
            raise ValueError("state blob size does not match this handle (state_size * num_channels)")
        rc = self._L.lc3gpu_encoder_state_load(self._h, _ptr(buf), buf.size)
        if rc:
            raise Lc3EncoderError(rc, "state_load")

    @classmethod
    def _borrowed(cls, handle, num_channels, frame_duration, sampling_frequency):
        """a handle somebody else owns (a pipeline's): every method works, close() does not destroy it (frame_duration None: a mixed handle)"""
        self = cls.__new__(cls)
        self._L = load_library()
        self.config = Lc3Config(sampling_frequency, frame_duration) if frame_duration is not None else None
        self.num_channels = int(num_channels)
        self._h = ctypes.c_void_p(handle)
        self._borrowed_handle = True
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed_handle", False):
                self._L.lc3gpu_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Lc3Decoder:
    @staticmethod
    def calc_working_buffer_lengths(num_channels, frame_duration, sampling_frequency):
        """-> (scaler_len, complex_len), lc3_decoder.rs:236-244"""
        out = (ctypes.c_int64 * 2)()
        rc = load_library().lc3gpu_decoder_working_buffer_lengths(num_channels, int(frame_duration),
                                                                  int(sampling_frequency), out)
        if rc:
            raise Lc3DecoderError(rc)
        return tuple(int(v) for v in out)

    def __init__(self, num_channels, frame_duration, sampling_frequency):
        self._L = load_library()
        self.config = Lc3Config(sampling_frequency, frame_duration)
        self.num_channels = int(num_channels)
        h = ctypes.c_void_p()
        rc = self._L.lc3gpu_decoder_create(ctypes.byref(h), self.num_channels, int(frame_duration),
                                           int(sampling_frequency))
        if rc:
            raise Lc3DecoderError(rc, "Lc3Decoder::new")
        self._h = h

    new = classmethod(lambda cls, *a, **k: cls(*a, **k))

    @classmethod
    def mixed(cls, descs):
        """descs = [(fs_hz, frame_us, nbytes), ...] (lc3gpu_decoder_create_mixed)"""
        self = cls.__new__(cls)
        self._L = load_library()
        self.descs = [(int(d[0]), int(d[1]), int(d[2])) for d in descs]
        self.configs = [Lc3Config(d[0], d[1]) for d in self.descs]
        self.config = None
        self.num_channels = len(self.descs)
        h = ctypes.c_void_p()
        rc = self._L.lc3gpu_decoder_create_mixed(ctypes.byref(h), self.num_channels, _desc_array(self.descs))
        if rc:
            raise Lc3DecoderError(rc, "Lc3Decoder::mixed")
        self._h = h
        return self

    def decode_mixed(self, d_in, d_pcm, n_frames, stream=None, d_bad_frame=None):
        rc = self._L.lc3gpu_decode_mixed(self._h, _ptr(d_in), _ptr(d_bad_frame), _ptr(d_pcm), int(n_frames), _ptr(stream))
        if rc:
            raise Lc3DecoderError(rc, "decode_mixed")

    def decode_frame(self, num_bits_per_audio_sample, channel_index, buf_in, samples_out):
        """lc3_decoder.rs:217-234; corrupt frames are concealed, not reported (":138-141")"""
        buf_in = np.ascontiguousarray(buf_in, dtype=np.uint8)
        if not (isinstance(samples_out, np.ndarray) and samples_out.dtype == np.int16 and samples_out.flags.c_contiguous):
            raise TypeError("samples_out must be a contiguous int16 numpy array")
        rc = self._L.lc3gpu_decode_frame(self._h, int(num_bits_per_audio_sample), int(channel_index), _ptr(buf_in),
                                         int(buf_in.size), _ptr(samples_out), int(samples_out.size))
        if rc:
            raise Lc3DecoderError(rc, "decode_frame")

    def decode(self, d_in, d_pcm, nbytes, n_frames, stream=None, d_bad_frame=None, first_channel=None,
               n_channels=None, layout="planar"):
        """batch: DEVICE uint8[S][T][nbytes] -> DEVICE int16[S][T][nf] (layout "interleaved": uint8[T][S][nbytes] ->
        int16[T][nf][S]), asynchronous on `stream`"""
        if first_channel is None:
            rc = self._L.lc3gpu_decode_layout(self._h, _layout(layout), _ptr(d_in), _ptr(d_bad_frame), _ptr(d_pcm), int(nbytes),
                                              int(n_frames), _ptr(stream))
        else:
            rc = self._L.lc3gpu_decode_range(self._h, int(first_channel), int(n_channels), _ptr(d_in),
                                             _ptr(d_bad_frame), _ptr(d_pcm), int(nbytes), int(n_frames), _ptr(stream))
        if rc:
            raise Lc3DecoderError(rc, "decode")

    def stage_event(self, stage, event):
        """every batch call from now on records `event` (the caller's; None clears) behind the kernels of `stage` (DEC_STAGE_PARSE)"""
        rc = self._L.lc3gpu_decoder_stage_event(self._h, int(stage), _event_handle(event))
        if rc:
            raise Lc3DecoderError(rc, "stage_event")
        self._stage_events = getattr(self, "_stage_events", {})
        self._stage_events[int(stage)] = event

    def timing(self, enable=True):
        """-> (parse-kernel ms, synthesis-kernel ms, batch calls) since the last call; (re)arms recording (enable = n > 1: every n-th batch call)"""
        out = (ctypes.c_double * 3)()
        rc = self._L.lc3gpu_decoder_timing(self._h, int(enable), out)
        if rc:
            raise Lc3DecoderError(rc, "timing")
        return float(out[0]), float(out[1]), int(out[2])

    def decode_frame_debug(self, buf_in, recon_form=1):
        """one frame of channel 0 with stage dumps -> (pcm int16[nf], dbg float32[DBG_FLOATS]); recon_form 0 lane, 1 late, 2 wave"""
        buf_in = np.ascontiguousarray(buf_in, np.uint8)
        out = np.zeros(self.config.nf, np.int16)
        dbg = np.zeros(DBG_FLOATS, np.float32)
        rc = self._L.lc3gpu_decode_frame_debug(self._h, int(recon_form), _ptr(buf_in), int(buf_in.size), _ptr(out), int(out.size), _ptr(dbg))
        if rc:
            raise Lc3DecoderError(rc, "decode_frame_debug")
        return out, dbg

    def synth_debug(self, data, ltpf_active, pitch_index, nbytes, time_in=False):
        """the synthesis half alone on channel 0: a spectrum (ne floats) or, with time_in, the post-filter's input samples (nf floats)
        -> (pcm int16[nf], dbg float32[DBG_FLOATS])"""
        data = np.ascontiguousarray(data, np.float32)
        out = np.zeros(self.config.nf, np.int16)
        dbg = np.zeros(DBG_FLOATS, np.float32)
        rc = self._L.lc3gpu_decoder_synth_debug(self._h, int(bool(time_in)), _ptr(data), int(data.size), int(ltpf_active), int(pitch_index),
                                                int(nbytes), _ptr(out), int(out.size), _ptr(dbg))
        if rc:
            raise Lc3DecoderError(rc, "synth_debug")
        return out, dbg

    def timing_kernels(self, enable=True):
        """-> (parse ms, reconstruction-kernel ms, TNS-kernel ms, synthesis ms, batch calls) since the last call; (re)arms recording (enable = n > 1: every n-th batch call)"""
        out = (ctypes.c_double * 5)()
        rc = self._L.lc3gpu_decoder_timing_kernels(self._h, int(enable), out)
        if rc:
            raise Lc3DecoderError(rc, "timing")
        return float(out[0]), float(out[1]), float(out[2]), float(out[3]), int(out[4])

    def reset(self):
        rc = self._L.lc3gpu_decoder_reset(self._h)
        if rc:
            raise Lc3DecoderError(rc, "reset")

    def plc_events(self):
        v = ctypes.c_uint64()
        rc = self._L.lc3gpu_decoder_plc_events(self._h, ctypes.byref(v))
        if rc:
            raise Lc3DecoderError(rc, "plc_events")
        return int(v.value)

    def decode_host(self, data, pcm, nbytes, n_frames, bad_frame=None):
        """HOST uint8[S][T][nbytes] -> HOST int16[S][T][nf]; synchronous (lc3gpu_decode_host)"""
        rc = self._L.lc3gpu_decode_host(self._h, _ptr(data), _ptr(bad_frame), _ptr(pcm), int(nbytes), int(n_frames))
        if rc:
            raise Lc3DecoderError(rc, "decode_host")

    def bind_stream(self, stream, bind=True):
        rc = self._L.lc3gpu_decoder_bind_stream(self._h, _ptr(stream), int(bool(bind)))
        if rc:
            raise Lc3DecoderError(rc, "bind_stream")

    def debug_pair_giveup(self):
        rc = self._L.lc3gpu_decoder_debug_pair_giveup(self._h)
        if rc:
            raise Lc3DecoderError(rc, "debug_pair_giveup")

    def pair_timeouts(self):
        """producer / consumer pair halves of the parser that ever gave up on their partner (include/lc3gpu.h); 0 unless a wave died"""
        v = ctypes.c_uint64()
        rc = self._L.lc3gpu_decoder_pair_timeouts(self._h, ctypes.byref(v))
        if rc:
            raise Lc3DecoderError(rc, "pair_timeouts")
        return int(v.value)

    def state_save(self):
        n = self._L.lc3gpu_decoder_state_size(self._h) * self.num_channels
        buf = np.zeros(n, np.uint8)
        rc = self._L.lc3gpu_decoder_state_save(self._h, _ptr(buf), buf.size)
        if rc:
            raise Lc3DecoderError(rc, "state_save")
        return buf

    def state_load(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        if buf.size != self._L.lc3gpu_decoder_state_size(self._h) * self.num_channels:
            raise ValueError("state blob size does not match this handle (state_size * num_channels)")
        rc = self._L.lc3gpu_decoder_state_load(self._h, _ptr(buf), buf.size)
        if rc:
            raise Lc3DecoderError(rc, "state_load")

    _borrowed = Lc3Encoder.__dict__["_borrowed"]

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed_handle", False):
                self._L.lc3gpu_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedBuffer:
    """page-locked host memory from lc3gpu_host_alloc as a numpy array (`.array`): what lc3gpu_encode_host / lc3gpu_decode_host copy at the
    PCIe link's rate"""

    def __init__(self, shape, dtype):
        self._L = load_library()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        rc = self._L.lc3gpu_host_alloc(ctypes.byref(p), max(1, n))
        if rc:
            raise Lc3GpuError(rc, "host_alloc")
        self._p = p
        self.array = np.frombuffer((ctypes.c_uint8 * max(1, n)).from_address(p.value), dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            self._L.lc3gpu_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Lc3Pipeline:
    """lc3gpu_pipeline: the caller loop (examples/encode.rs:97-115, examples/decode.rs:93-112) in the arrangement that measured best --
    groups of the channels, each with an encoder handle and a decoder handle of its own on HIP streams of its own"""

    def __init__(self, num_channels, frame_duration, sampling_frequency, n_groups=0):
        self._L = load_library()
        self.config = Lc3Config(sampling_frequency, frame_duration)
        self.num_channels = int(num_channels)
        h = ctypes.c_void_p()
        rc = self._L.lc3gpu_pipeline_create(ctypes.byref(h), self.num_channels, int(frame_duration), int(sampling_frequency), int(n_groups))
        if rc:
            raise Lc3GpuError(rc, "Lc3Pipeline")
        self._h = h
        self._collect_groups(frame_duration, sampling_frequency)

    @classmethod
    def mixed(cls, descs, n_groups=0, group_first=None):
        """streams of different configurations, descs = [(fs_hz, frame_us, nbytes), ...]; group g takes descs[group_first[g]:group_first[g + 1]]
        (None: equal shares of the list); ragged buffers as for Lc3Encoder.mixed (lc3gpu_pipeline_create_mixed)"""
        self = cls.__new__(cls)
        self._L = load_library()
        self.descs = [(int(d[0]), int(d[1]), int(d[2])) for d in descs]
        self.config = None
        self.num_channels = len(self.descs)
        h = ctypes.c_void_p()
        gf = (ctypes.c_int * len(group_first))(*[int(v) for v in group_first]) if group_first is not None else None
        rc = self._L.lc3gpu_pipeline_create_mixed(ctypes.byref(h), self.num_channels, _desc_array(self.descs),
                                                  int(len(group_first) if group_first is not None else n_groups), gf)
        if rc:
            raise Lc3GpuError(rc, "Lc3Pipeline.mixed")
        self._h = h
        self._collect_groups(None, None)
        return self

    def submit_mixed(self, d_pcm, d_bytes, d_pcm_out, n_frames):
        self._check(self._L.lc3gpu_pipeline_submit_mixed(self._h, _ptr(d_pcm), _ptr(d_bytes), _ptr(d_pcm_out), int(n_frames)), "pipeline_submit_mixed")

    def encode_mixed(self, d_pcm, d_bytes, n_frames):
        self._check(self._L.lc3gpu_pipeline_encode_mixed(self._h, _ptr(d_pcm), _ptr(d_bytes), int(n_frames)), "pipeline_encode_mixed")

    def decode_mixed(self, d_bytes, d_pcm_out, n_frames, d_bad_frame=None):
        self._check(self._L.lc3gpu_pipeline_decode_mixed(self._h, _ptr(d_bytes), _ptr(d_bad_frame), _ptr(d_pcm_out), int(n_frames)), "pipeline_decode_mixed")

    def _collect_groups(self, frame_duration, sampling_frequency):
        h = self._h
        self.groups = []
        for g in range(self._L.lc3gpu_pipeline_groups(h)):
            first, n, e, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_void_p(), ctypes.c_void_p()
            rc = self._L.lc3gpu_pipeline_group(h, g, ctypes.byref(first), ctypes.byref(n), ctypes.byref(e), ctypes.byref(d))
            if rc:
                raise Lc3GpuError(rc, "pipeline_group")
            self.groups.append({"first": first.value, "n": n.value,
                                "enc": Lc3Encoder._borrowed(e.value, n.value, frame_duration, sampling_frequency),
                                "dec": Lc3Decoder._borrowed(d.value, n.value, frame_duration, sampling_frequency)})

    def _check(self, rc, what):
        if rc:
            raise Lc3GpuError(rc, what)

    def submit(self, d_pcm, d_bytes, d_pcm_out, nbytes, n_frames):
        self._check(self._L.lc3gpu_pipeline_submit(self._h, _ptr(d_pcm), _ptr(d_bytes), _ptr(d_pcm_out), int(nbytes), int(n_frames)), "pipeline_submit")

    def encode(self, d_pcm, d_bytes, nbytes, n_frames):
        self._check(self._L.lc3gpu_pipeline_encode(self._h, _ptr(d_pcm), _ptr(d_bytes), int(nbytes), int(n_frames)), "pipeline_encode")

    def decode(self, d_bytes, d_pcm_out, nbytes, n_frames, d_bad_frame=None):
        self._check(self._L.lc3gpu_pipeline_decode(self._h, _ptr(d_bytes), _ptr(d_bad_frame), _ptr(d_pcm_out), int(nbytes), int(n_frames)), "pipeline_decode")

    def wait(self):
        self._check(self._L.lc3gpu_pipeline_wait(self._h), "pipeline_wait")

    def join(self, stream=None):
        self._check(self._L.lc3gpu_pipeline_join(self._h, _ptr(stream)), "pipeline_join")

    def follow(self, stream=None):
        self._check(self._L.lc3gpu_pipeline_follow(self._h, _ptr(stream)), "pipeline_follow")

    def mark(self, event):
        """records `event` (torch.cuda.Event that has been recorded once, or a hipEvent_t) behind the last group's latest work"""
        self._check(self._L.lc3gpu_pipeline_mark(self._h, _event_handle(event)), "pipeline_mark")

    def reset(self):
        self._check(self._L.lc3gpu_pipeline_reset(self._h), "pipeline_reset")

    def close(self):
        if getattr(self, "_h", None):
            for g in self.groups:
                g["enc"].close()
                g["dec"].close()
            self._L.lc3gpu_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prof_read():
    """diagnostic build only: per-slot cycles accumulated since the previous call"""
    out = (ctypes.c_ulonglong * 64)()
    rc = load_library().lc3gpu_prof_read(out)
    if rc:
        raise Lc3GpuError(rc, "prof_read")
    return list(out)


def kernel_info(which):
    out = (ctypes.c_int * 5)()
    rc = load_library().lc3gpu_kernel_info(int(which), out)
    if rc:
        raise Lc3GpuError(rc, "kernel_info")
    return dict(zip(["lds_bytes", "vgprs", "sgprs", "scratch_bytes", "max_threads"], list(out)))
