#!/usr/bin/env python3
"""Writes bindings/lc3gpu.rs: the Rust side of the drop-in boundary.

The reference is a Rust crate (ninjasource/lc3-codec v0.2.0); BASELINE.json's north star asks for "Rust host code calling a thin C-ABI
HIP layer".  There is no Rust toolchain in this image, so the file is shipped UNCOMPILED -- but it is not hand-copied either: the
`extern "C"` block is generated from include/lc3gpu.h declaration by declaration (tests/test_abi.py regenerates it and compares, and
checks that every symbol the header declares is bound), and the safe wrappers below it are the reference's own surface
(`Lc3Encoder::new / encode_frame`, src/encoder/lc3_encoder.rs:117-191; `Lc3Decoder::new / decode_frame`, src/decoder/lc3_decoder.rs:181-234)
plus the batch, host-batch and pipeline calls.

usage: gen_rust_binding.py [out.rs]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lc3gpu.h")

OPAQUE = {"lc3gpu_encoder": "Lc3GpuEncoder", "lc3gpu_decoder": "Lc3GpuDecoder", "lc3gpu_pipeline": "Lc3GpuPipeline",
          "lc3gpu_stream_desc": "Lc3GpuStreamDesc"}
SCALAR = {"int": "i32", "unsigned": "u32", "unsigned int": "u32", "float": "f32", "double": "f64", "size_t": "usize",
          "int16_t": "i16", "uint8_t": "u8", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64",
          "unsigned long long": "u64", "char": "c_char", "void": "c_void"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def rust_type(ctype):
    """'const int16_t *' -> '*const i16', 'lc3gpu_encoder **' -> '*mut *mut Lc3GpuEncoder', 'int' -> 'i32'"""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").strip()
    const = False
    words = [w for w in t.split() if w]
    if words and words[0] == "const":
        const, words = True, words[1:]
    if words and words[0] == "struct":
        words = words[1:]
    base = " ".join(words)
    r = OPAQUE.get(base) or SCALAR.get(base)
    if r is None:
        raise ValueError("unmapped C type: %r" % ctype)
    for i in range(stars):
        r = ("*const " if (const and i == 0) else "*mut ") + r  # (the innermost pointer carries the const of `const T *`)
    if stars == 0 and base == "void":
        return None
    return r


def parse_param(p):
    p = p.strip()
    if p == "void" or not p:
        return None
    m = re.match(r"^(.*?)(\w+)\s*(\[\s*\d*\s*\])?$", p)
    ctype, name, arr = m.group(1), m.group(2), m.group(3)
    if arr:
        ctype += "*"
    if name in ("in", "out", "type", "fn", "ref", "box", "move", "match", "loop", "where", "dyn"):
        name += "_"
    return name, rust_type(ctype)


def declarations():
    text = strip_comments(open(HEADER).read())
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    out = []
    for m in re.finditer(r"([\w\s\*]+?)\b(lc3gpu_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        ps = [parse_param(p) for p in " ".join(params.split()).split(",")]
        ps = [p for p in ps if p]
        out.append((name, ps, rust_type(ret)))
    return out


def constants():
    text = strip_comments(open(HEADER).read())
    out = []
    for m in re.finditer(r"^#define\s+(LC3GPU_\w+)\s+(-?\d+)\s*$", text, flags=re.M):
        if m.group(1) != "LC3GPU_H_":
            out.append((m.group(1), int(m.group(2))))
    return out


WRAPPERS = r'''
// ------------------------------------------------------------------------------------------------------------------
// Safe surface: what a maintainer adds beside src/encoder/lc3_encoder.rs:175-180 / src/decoder/lc3_decoder.rs:217-223 under
// `#[cfg(feature = "mi355x")]`.  Same names and argument meaning as the CPU types; the working buffers the CPU objects borrow
// (lc3_encoder.rs:117-173) are not needed -- device memory belongs to the handle.
// ------------------------------------------------------------------------------------------------------------------
use crate::common::config::{FrameDuration, SamplingFrequency};
use crate::decoder::lc3_decoder::Lc3DecoderError;
use crate::encoder::lc3_encoder::Lc3EncoderError;

fn frame_us(d: FrameDuration) -> i32 {
    match d {
        FrameDuration::SevenPointFiveMs => 7500,
        FrameDuration::TenMs => 10000,
    }
}
fn fs_hz(f: SamplingFrequency) -> i32 {
    match f {
        SamplingFrequency::Hz8000 => 8000,
        SamplingFrequency::Hz16000 => 16000,
        SamplingFrequency::Hz24000 => 24000,
        SamplingFrequency::Hz32000 => 32000,
        SamplingFrequency::Hz44100 => 44100,
        SamplingFrequency::Hz48000 => 48000,
    }
}

/// `Lc3Encoder` (encoder/lc3_encoder.rs:33-40,117-191) on the GPU: `num_channels` independent channels resident on the current HIP device.
pub struct Lc3EncoderGpu {
    h: *mut Lc3GpuEncoder,
    num_channels: usize,
}
impl Lc3EncoderGpu {
    /// lc3_encoder.rs:194-209 (kept for API parity: the numbers are the reference's, the buffers are not used)
    pub fn calc_working_buffer_lengths(num_channels: usize, d: FrameDuration, f: SamplingFrequency) -> (usize, usize, usize) {
        let mut out = [0i64; 3];
        let rc = unsafe { lc3gpu_encoder_working_buffer_lengths(num_channels as i32, frame_us(d), fs_hz(f), out.as_mut_ptr()) };
        assert_eq!(rc, LC3GPU_OK);
        (out[0] as usize, out[1] as usize, out[2] as usize)
    }
    /// lc3_encoder.rs:117-173.  Panics where the reference's constructor panics (8 kHz: encoder/bandwidth_detector.rs:36-37).
    pub fn new(num_channels: usize, d: FrameDuration, f: SamplingFrequency) -> Self {
        let mut h = core::ptr::null_mut();
        let rc = unsafe { lc3gpu_encoder_create(&mut h, num_channels as i32, frame_us(d), fs_hz(f)) };
        assert_eq!(rc, LC3GPU_OK, "lc3gpu_encoder_create: {}", rc);
        Self { h, num_channels }
    }
    /// lc3_encoder.rs:175-191: `buf_out.len()` selects the bit rate; cannot fail (`Lc3EncoderError` is empty), panics where the
    /// reference panics (channel index, slice length).
    pub fn encode_frame(&mut self, channel_index: usize, samples_in: &[i16], buf_out: &mut [u8]) -> Result<(), Lc3EncoderError> {
        let rc = unsafe {
            lc3gpu_encode_frame(self.h, channel_index as i32, samples_in.as_ptr(), samples_in.len() as i32, buf_out.as_mut_ptr(), buf_out.len() as i32)
        };
        if rc != LC3GPU_OK {
            panic!("encode_frame: lc3gpu error {}", rc)
        }
        Ok(())
    }
    /// The caller loop of examples/encode.rs:97-115 over host buffers in one call: `pcm` = [channel][frame][nf], `out` = [channel][frame][nbytes].
    pub fn encode_host(&mut self, pcm: &[i16], out: &mut [u8], nbytes: usize, n_frames: usize) -> i32 {
        assert_eq!(out.len(), self.num_channels * n_frames * nbytes);
        unsafe { lc3gpu_encode_host(self.h, pcm.as_ptr(), out.as_mut_ptr(), nbytes as i32, n_frames as i32) }
    }
    /// The same loop over DEVICE buffers, asynchronous on `hip_stream`.
    ///
    /// # Safety
    /// `d_pcm` / `d_out` must be device allocations of [channel][frame][nf] i16 / [channel][frame][nbytes] u8 that outlive the call's work.
    pub unsafe fn encode_device(&mut self, d_pcm: *const i16, d_out: *mut u8, nbytes: usize, n_frames: usize, hip_stream: *mut c_void) -> i32 {
        lc3gpu_encode(self.h, d_pcm, d_out, nbytes as i32, n_frames as i32, hip_stream)
    }
}
impl Drop for Lc3EncoderGpu {
    fn drop(&mut self) {
        unsafe {
            lc3gpu_encoder_destroy(self.h);
        }
    }
}

/// `Lc3Decoder` (decoder/lc3_decoder.rs:51-69,181-234) on the GPU.
pub struct Lc3DecoderGpu {
    h: *mut Lc3GpuDecoder,
    num_channels: usize,
}
impl Lc3DecoderGpu {
    /// lc3_decoder.rs:236-244
    pub fn calc_working_buffer_lengths(num_channels: usize, d: FrameDuration, f: SamplingFrequency) -> (usize, usize) {
        let mut out = [0i64; 2];
        let rc = unsafe { lc3gpu_decoder_working_buffer_lengths(num_channels as i32, frame_us(d), fs_hz(f), out.as_mut_ptr()) };
        assert_eq!(rc, LC3GPU_OK);
        (out[0] as usize, out[1] as usize)
    }
    /// lc3_decoder.rs:181-215
    pub fn new(num_channels: usize, d: FrameDuration, f: SamplingFrequency) -> Self {
        let mut h = core::ptr::null_mut();
        let rc = unsafe { lc3gpu_decoder_create(&mut h, num_channels as i32, frame_us(d), fs_hz(f)) };
        assert_eq!(rc, LC3GPU_OK, "lc3gpu_decoder_create: {}", rc);
        Self { h, num_channels }
    }
    /// lc3_decoder.rs:217-234: corrupt frames are concealed and `Ok(())` comes back (:138-141); the only error is the sample width.
    pub fn decode_frame(&mut self, num_bits_per_audio_sample: usize, channel_index: usize, buf_in: &[u8], samples_out: &mut [i16]) -> Result<(), Lc3DecoderError> {
        let rc = unsafe {
            lc3gpu_decode_frame(self.h, num_bits_per_audio_sample as i32, channel_index as i32, buf_in.as_ptr(), buf_in.len() as i32,
                                samples_out.as_mut_ptr(), samples_out.len() as i32)
        };
        match rc {
            LC3GPU_OK => Ok(()),
            LC3GPU_EBITS => Err(Lc3DecoderError::Only16BitsPerAudioSampleSupported),
            e => panic!("decode_frame: lc3gpu error {}", e),
        }
    }
    /// The caller loop of examples/decode.rs:93-112 over host buffers; `bad_frame` = one flag per (channel, frame), non-zero = lost.
    pub fn decode_host(&mut self, data: &[u8], bad_frame: Option<&[u8]>, pcm: &mut [i16], nbytes: usize, n_frames: usize) -> i32 {
        assert_eq!(data.len(), self.num_channels * n_frames * nbytes);
        let bad = bad_frame.map_or(core::ptr::null(), |b| b.as_ptr());
        unsafe { lc3gpu_decode_host(self.h, data.as_ptr(), bad, pcm.as_mut_ptr(), nbytes as i32, n_frames as i32) }
    }
    /// frames concealed so far (decoder/packet_loss_concealment.rs:63-85)
    pub fn plc_events(&mut self) -> u64 {
        let mut v = 0u64;
        unsafe { lc3gpu_decoder_plc_events(self.h, &mut v) };
        v
    }
}
impl Drop for Lc3DecoderGpu {
    fn drop(&mut self) {
        unsafe {
            lc3gpu_decoder_destroy(self.h);
        }
    }
}

/// Encode + decode of many channels in the arrangement that measured fastest (include/lc3gpu.h, "pipeline"): device buffers, asynchronous.
pub struct Lc3PipelineGpu {
    h: *mut Lc3GpuPipeline,
}
impl Lc3PipelineGpu {
    pub fn new(num_channels: usize, d: FrameDuration, f: SamplingFrequency) -> Self {
        let mut h = core::ptr::null_mut();
        let rc = unsafe { lc3gpu_pipeline_create(&mut h, num_channels as i32, frame_us(d), fs_hz(f), 0) };
        assert_eq!(rc, LC3GPU_OK, "lc3gpu_pipeline_create: {}", rc);
        Self { h }
    }
    /// # Safety
    /// device pointers of [channel][frame][nf] i16, [channel][frame][nbytes] u8, [channel][frame][nf] i16 that stay valid until `wait`.
    pub unsafe fn submit(&mut self, d_pcm: *const i16, d_bytes: *mut u8, d_pcm_out: *mut i16, nbytes: usize, n_frames: usize) -> i32 {
        lc3gpu_pipeline_submit(self.h, d_pcm, d_bytes, d_pcm_out, nbytes as i32, n_frames as i32)
    }
    pub fn wait(&mut self) -> i32 {
        unsafe { lc3gpu_pipeline_wait(self.h) }
    }
}
impl Drop for Lc3PipelineGpu {
    fn drop(&mut self) {
        unsafe {
            lc3gpu_pipeline_destroy(self.h);
        }
    }
}
'''


def generate():
    lines = [
        "// bindings/lc3gpu.rs -- Rust binding of liblc3gpu.so (include/lc3gpu.h), GENERATED by tools/gen_rust_binding.py: do not edit.",
        "//",
        "// Drop it into the reference crate as src/gpu.rs behind `#[cfg(feature = \"mi355x\")]`; build.rs:",
        "//   println!(\"cargo:rustc-link-search=native=<repo>/lc3-codec_amd/lib\"); println!(\"cargo:rustc-link-lib=dylib=lc3gpu\");",
        "// Not compiled in this repository (the image has no Rust toolchain); tests/test_abi.py checks that the extern block binds every",
        "// symbol include/lc3gpu.h declares, with the header's parameter lists, and that this file is what the generator writes.",
        "#![allow(dead_code, non_camel_case_types, clippy::too_many_arguments, clippy::missing_safety_doc)]",
        "use core::ffi::{c_char, c_void};",
        "",
    ]
    for c, r in sorted(OPAQUE.items()):
        if c == "lc3gpu_stream_desc":
            continue
        lines += ["#[repr(C)]", "pub struct %s {" % r, "    _private: [u8; 0],", "}"]
    lines += ["/// one stream of a mixed-configuration handle (lc3gpu_stream_desc)", "#[repr(C)]", "#[derive(Clone, Copy, Debug)]",
              "pub struct Lc3GpuStreamDesc {", "    pub fs_hz: i32,", "    pub frame_us: i32,", "    pub nbytes: i32,", "}", ""]
    for name, val in constants():
        lines.append("pub const %s: i32 = %d;" % (name, val))
    lines += ["", "#[link(name = \"lc3gpu\")]", "extern \"C\" {"]
    for name, ps, ret in declarations():
        sig = ", ".join("%s: %s" % p for p in ps)
        lines.append("    pub fn %s(%s)%s;" % (name, sig, (" -> " + ret) if ret else ""))
    lines.append("}")
    return "\n".join(lines) + "\n" + WRAPPERS


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "bindings", "lc3gpu.rs")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        f.write(generate())


if __name__ == "__main__":
    main()
