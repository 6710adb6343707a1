"""lc3-codec_amd -- MI355X-native batched LC3 encode/decode engine.

Host-side mirror of the reference's `Lc3Encoder` / `Lc3Decoder` interface
(ninjasource/lc3-codec v0.2.0, src/encoder/lc3_encoder.rs:117-209,
src/decoder/lc3_decoder.rs:181-244) over the C ABI of liblc3gpu.so
(include/lc3gpu.h).  The compute path is hand-written HIP for gfx950; there is no
CPU fallback: constructing a codec without the native library or without a GPU
raises.  The directory name contains '-', import it with
importlib.import_module("lc3-codec_amd")."""
from .api import (  # noqa: F401
    DEC_STAGE_PARSE,
    ENC_STAGE_BACK,
    ENC_STAGE_FRONT,
    ENC_STAGE_VQ,
    FrameDuration,
    Lc3Config,
    Lc3Decoder,
    Lc3DecoderError,
    Lc3Encoder,
    Lc3EncoderError,
    LAYOUT_INTERLEAVED,
    LAYOUT_PLANAR,
    Lc3GpuError,
    Lc3Pipeline,
    PinnedBuffer,
    SPEC_8KHZ_ENCODE,
    SPEC_ALL,
    SPEC_BW_CUTOFF_DB,
    SPEC_NBITS_SPEC_OLD,
    SPEC_SNS_LAST_GAIN,
    SPEC_TNS_SSWB_STOP,
    SamplingFrequency,
    StreamDesc,
    build_native,
    build_tool,
    clock_probe,
    device_count,
    library_path,
    load_library,
    selftest_math,
    tool_path,
)
