"""File drivers: lc3-codec_amd/host/lc3_files.{hpp,cpp} behind lc3gpu-tool.
CPU part: WAV header reader against the reference's own test vector (src/common/wav.rs:131-148), error paths and the
compare tool.  GPU part (-m gpu): WAV -> .lc3 -> WAV through the tool against the oracle, including the reference
drivers' quirks (zero-size header written up front, last channel frame never decoded)."""
import importlib
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pkg = importlib.import_module("lc3-codec_amd")
synth = importlib.import_module("lc3-codec_amd.synth")


@pytest.fixture(scope="module")
def tool():
    pkg.build_native()
    return pkg.build_tool()


def run(tool, *args):
    return subprocess.run([tool, *map(str, args)], capture_output=True, text=True)


REF_HEADER = bytes([  # src/common/wav.rs:133-137
    0x52, 0x49, 0x46, 0x46, 0x16, 0x29, 0x0B, 0x00, 0x57, 0x41, 0x56, 0x45, 0x66, 0x6D, 0x74, 0x20, 0x10, 0x00,
    0x00, 0x00, 0x01, 0x00, 0x02, 0x00, 0x44, 0xAC, 0x00, 0x00, 0x10, 0xB1, 0x02, 0x00, 0x04, 0x00, 0x10, 0x00,
    0x64, 0x61, 0x74, 0x61, 0x70, 0x28, 0x0B, 0x00, 0x00])


def test_wav_header_reference_vector(tool, tmp_path):
    f = tmp_path / "h.wav"
    f.write_bytes(REF_HEADER)
    r = run(tool, "wavinfo", f)
    assert r.returncode == 0, r.stderr
    fields = dict(zip(r.stdout.split()[0::2], map(int, r.stdout.split()[1::2])))
    assert fields["num_channels"] == 2 and fields["sample_rate"] == 44100 and fields["byte_rate"] == 176400
    assert fields["block_align"] == 4 and fields["bits_per_sample"] == 16 and fields["data_size"] == 731248
    assert fields["data_start_position"] == 44


@pytest.mark.parametrize("mutate,code", [
    (lambda b: b[:20], 2),                                  # ReadHeaderInvalidHeaderLength
    (lambda b: b"RIFX" + b[4:], 3),                         # ReadHeaderChunkIdNotRIFF
    (lambda b: b[:8] + b"WAVX" + b[12:], 4),                # ReadHeaderFormatNotWAVE
    (lambda b: b[:12] + b"fmtx" + b[16:], 5),               # ReadHeaderSubChunk1IdNotFmt
    (lambda b: b[:16] + b"\x12\x00\x00\x00" + b[20:], 6),  # ReadHeaderInvalidPcmHeaderLength
    (lambda b: b[:20] + b"\x03\x00" + b[22:], 7),          # ReadHeaderAudioFormatNotPcm
    (lambda b: b[:36] + b"junk" + b[40:], 8),               # ReadHeaderMissingDataSection
])
def test_wav_header_errors(tool, tmp_path, mutate, code):
    f = tmp_path / "bad.wav"
    f.write_bytes(mutate(REF_HEADER))
    r = run(tool, "wavinfo", f)
    assert r.returncode == 1 and r.stdout.strip() == f"error {code}"


def test_wav_header_list_chunk(tool, tmp_path):  # wav.rs:104-108: size from the LIST chunk, data four bytes further on
    f = tmp_path / "list.wav"
    f.write_bytes(REF_HEADER[:36] + b"LIST" + struct.pack("<I", 26) + b"\0" * 8)
    r = run(tool, "wavinfo", f)
    assert r.returncode == 0 and "data_size 26 data_start_position 48" in r.stdout


def test_compare_tool(tool, tmp_path):  # examples/compare.rs:6-36
    a = bytes(range(150)) * 3
    (tmp_path / "a.lc3").write_bytes(a)
    (tmp_path / "b.lc3").write_bytes(a)
    r = run(tool, "compare", tmp_path / "a.lc3", tmp_path / "b.lc3")
    assert r.returncode == 0 and "no difference" in r.stdout
    b = bytearray(a)
    b[150 + 7] ^= 0x10
    (tmp_path / "b.lc3").write_bytes(bytes(b))
    r = run(tool, "compare", tmp_path / "a.lc3", tmp_path / "b.lc3")
    assert r.returncode == 1 and "Diff at frame 2 byte index 7: left: 7 right: 23" in r.stdout


def _wav_bytes(pcm_interleaved, fs, channels):
    data = pcm_interleaved.astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, channels, fs,
                                                                                  fs * channels * 2, channels * 2, 16)
    return hdr + b"data" + struct.pack("<I", len(data)) + data


@pytest.mark.gpu
@pytest.mark.parametrize("fs,us,channels,nbytes,n_samples", [
    (48000, 10000, 2, 150, 48000 * 73 // 100),   # stereo, last PCM frame partial
    (48000, 7500, 1, 113, 360 * 9),               # mono 7.5 ms, whole frames
    (32000, 10000, 2, 80, 3200 + 17),
])
def test_file_round_trip_matches_oracle(tool, tmp_path, fs, us, channels, nbytes, n_samples):
    nf = {48000: 480, 32000: 320}[fs] * us // 10000
    n_frames = -(-n_samples // nf)
    planar = synth.make_pcm(channels, n_frames, nf, fs)          # [ch][frame][nf]
    inter = planar.reshape(channels, n_frames * nf).T[:n_samples]  # [sample][ch], truncated like a real file
    wav, lc3, back = tmp_path / "in.wav", tmp_path / "out.lc3", tmp_path / "back.wav"
    wav.write_bytes(_wav_bytes(inter.reshape(-1), fs, channels))
    r = run(tool, "encode", wav, lc3, fs, channels, us, nbytes, "--frames-per-launch", 7)
    assert r.returncode == 0, r.stderr
    # expected: zero-padded last frame, de-interleaved, frames in time order with channels inside (examples/encode.rs)
    padded = np.zeros((n_frames * nf, channels), np.int16)
    padded[:n_samples] = inter
    ref_in = np.ascontiguousarray(padded.T.reshape(channels, n_frames, nf))
    ref = O.encode_batch(ref_in, nbytes, fs, us)                  # [ch][frame][nbytes]
    expect = np.ascontiguousarray(ref.transpose(1, 0, 2)).tobytes()
    got = lc3.read_bytes()
    assert got == expect
    # decode with the reference driver's behaviour: header first with zero sizes, last channel frame never decoded
    r = run(tool, "decode", lc3, back, fs, channels, us, nbytes, "--frames-per-launch", 5)
    assert r.returncode == 0, r.stderr
    out = back.read_bytes()
    units = -(-len(got) // nbytes) - 1
    frames = units // channels
    fs_hdr = 44100 if fs == 44100 else fs
    assert out[:44] == (b"RIFF" + struct.pack("<I", 36) + b"WAVEfmt " +
                        struct.pack("<IHHIIHH", 16, 1, channels, fs_hdr, fs_hdr * channels * 2, 4, 16) + b"data" + struct.pack("<I", 0))
    ref_pcm = O.decode_batch(ref, nf, fs, us)[:, :frames]           # [ch][frame][nf]
    expect_pcm = np.ascontiguousarray(ref_pcm.reshape(channels, frames * nf).T).astype("<i2").tobytes()
    assert out[44:] == expect_pcm
    # the same with both quirks switched off
    r = run(tool, "decode", lc3, back, fs, channels, us, nbytes, "--fix-header", "--keep-last-frame")
    assert r.returncode == 0, r.stderr
    out = back.read_bytes()
    full = O.decode_batch(ref, nf, fs, us)
    assert struct.unpack("<I", out[40:44])[0] == n_frames * nf * channels * 2 and out[32:34] == struct.pack("<H", channels * 2)
    assert out[44:] == np.ascontiguousarray(full.reshape(channels, n_frames * nf).T).astype("<i2").tobytes()


def test_static_channel_api_buffer_lengths(tool):
    """The no_std API shape (channel count as a template argument, lc3_encoder.rs:286-303, lc3_decoder.rs:296-310) through
    the C++ facade: the reference's numbers for 48 kHz / 10 ms (1900/1106/960 and 4971/960 per channel), twice
    that for the default two channels.  No device needed."""
    r1 = run(tool, "buffer-lengths", 1, 48000, 10000)
    r2 = run(tool, "buffer-lengths", 2, 48000, 10000)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    one = list(map(int, r1.stdout.split()))
    assert one == [1900, 1106, 960, 4971, 960]
    assert list(map(int, r2.stdout.split())) == [2 * v for v in one]
    assert run(tool, "buffer-lengths", 1, 12345, 10000).returncode == 1  # unsupported rate: an error, not an abort


@pytest.mark.gpu
def test_single_frame_timing_harness(tool):
    """examples/arm/src/main.rs:39-112 shape: one channel, one encode_frame and one decode_frame call, host-timed."""
    r = run(tool, "frame-timing", 48000, 10000, 150, 20)
    assert r.returncode == 0, r.stderr
    assert "Encoded in" in r.stdout and "Decoded in" in r.stdout
    assert "(1900, 1106, 960)" in r.stdout


@pytest.mark.gpu
def test_pipeline_throughput_from_cpp(tool):
    """`lc3gpu-tool throughput`: the headline arrangement from a C++ caller -- lc3gpu_pipeline_create / _submit / _wait on hipMalloc'ed
    buffers, no Python in the process.  It checks its first submission against the single-frame calls itself (exit code 3 on a
    difference); here the rate must be the benchmark's (the pipeline object owns the arrangement, the caller only submits)."""
    import re

    r = subprocess.run([tool, "throughput", "64", "2", "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([tool, "throughput", "16384", "4", "200"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"([\d.]+) M frames/s", r.stdout)
    assert m and float(m.group(1)) > 50.0, r.stdout
