// File drivers around the batched codec: the callers of the hot path in the reference
// (examples/encode.rs:36-120, examples/decode.rs:36-124, examples/compare.rs:6-36) and its minimal WAV header
// reader/writer (src/common/wav.rs:45-127), rebuilt on top of the C ABI (include/lc3gpu.h).  SURVEY.md section 8,
// row (f1).  Host-side C++ (the reference's drivers are compiled code); the codec work itself is liblc3gpu.so.
//
// Reference behaviours kept by default (each can be switched off, see Lc3FileOptions):
//  * the decoder driver writes the 44-byte WAV header BEFORE decoding, with data_size = 0, chunk size 36 and
//    block_align = 4, and never patches it (examples/decode.rs:71-84);
//  * it stops when `in_cursor + nbytes >= file length`, so the last channel frame of a file is never decoded
//    (examples/decode.rs:100-104), and a frame whose channels were not all decoded is not written;
//  * the encoder driver zero-pads the last, partial PCM frame (examples/encode.rs:84-95);
//  * the `.lc3` container is headerless: frames in time order, channels inside a frame in index order, each
//    `num_bytes_per_channel` bytes.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lc3gpu.h"

namespace lc3files {

enum class WavError {  // src/common/wav.rs:9-18
    Ok = 0,
    WriteHeaderBufferTooSmall,
    ReadHeaderInvalidHeaderLength,
    ReadHeaderChunkIdNotRIFF,
    ReadHeaderFormatNotWAVE,
    ReadHeaderSubChunk1IdNotFmt,
    ReadHeaderInvalidPcmHeaderLength,
    ReadHeaderAudioFormatNotPcm,
    ReadHeaderMissingDataSection,
};

constexpr size_t RIFF_HEADER_ONLY_LEN = 8;   // wav.rs:20
constexpr size_t FULL_WAV_HEADER_LEN = 44;   // wav.rs:21

struct WavHeader {  // wav.rs:27-43
    size_t num_channels = 0, sample_rate = 0, byte_rate = 0, block_align = 0, bits_per_sample = 0;
    size_t data_size = 0, data_start_position = 0, data_with_header_size = 0;
};

inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)((uint16_t)p[0] | ((uint16_t)p[1] << 8)); }
inline void wr32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
inline void wr16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

// wav.rs:45-67
inline WavError write_header(const WavHeader &h, uint8_t *buf, size_t buf_len, size_t *written) {
    if (buf_len < FULL_WAV_HEADER_LEN) return WavError::WriteHeaderBufferTooSmall;
    std::memcpy(buf, "RIFF", 4);
    wr32(buf + 4, (uint32_t)h.data_with_header_size);
    std::memcpy(buf + 8, "WAVE", 4);
    std::memcpy(buf + 12, "fmt ", 4);
    wr32(buf + 16, 16);  // PCM header length
    wr16(buf + 20, 1);   // PCM
    wr16(buf + 22, (uint16_t)h.num_channels);
    wr32(buf + 24, (uint32_t)h.sample_rate);
    wr32(buf + 28, (uint32_t)h.byte_rate);
    wr16(buf + 32, (uint16_t)h.block_align);
    wr16(buf + 34, (uint16_t)h.bits_per_sample);
    std::memcpy(buf + 36, "data", 4);
    wr32(buf + 40, (uint32_t)h.data_size);
    *written = FULL_WAV_HEADER_LEN;
    return WavError::Ok;
}

// wav.rs:69-127 (including its treatment of a "LIST" chunk in the data position: size taken from the LIST chunk,
// data assumed four bytes further on)
inline WavError read_header(const uint8_t *buf, size_t len, WavHeader *out) {
    if (len < FULL_WAV_HEADER_LEN) return WavError::ReadHeaderInvalidHeaderLength;
    if (std::memcmp(buf, "RIFF", 4) != 0) return WavError::ReadHeaderChunkIdNotRIFF;
    if (std::memcmp(buf + 8, "WAVE", 4) != 0) return WavError::ReadHeaderFormatNotWAVE;
    if (std::memcmp(buf + 12, "fmt ", 4) != 0) return WavError::ReadHeaderSubChunk1IdNotFmt;
    if (rd32(buf + 16) != 16) return WavError::ReadHeaderInvalidPcmHeaderLength;
    if (rd16(buf + 20) != 1) return WavError::ReadHeaderAudioFormatNotPcm;
    WavHeader h;
    h.data_with_header_size = rd32(buf + 4);
    h.num_channels = rd16(buf + 22);
    h.sample_rate = rd32(buf + 24);
    h.byte_rate = rd32(buf + 28);
    h.block_align = rd16(buf + 32);
    h.bits_per_sample = rd16(buf + 34);
    if (std::memcmp(buf + 36, "data", 4) == 0) {
        h.data_size = rd32(buf + 40);
        h.data_start_position = FULL_WAV_HEADER_LEN;
    } else if (std::memcmp(buf + 36, "LIST", 4) == 0) {
        h.data_size = rd32(buf + 40);
        h.data_start_position = FULL_WAV_HEADER_LEN + 4;
    } else {
        return WavError::ReadHeaderMissingDataSection;
    }
    *out = h;
    return WavError::Ok;
}

struct Lc3FileOptions {
    bool reference_wav_header = true;      // header written up front with zero sizes and block_align 4
    bool reference_drop_last_frame = true;  // stop at in_cursor + nbytes >= len
    int frames_per_launch = 1024;           // frames of every channel handed to the GPU per call
};

enum class FileStatus { Ok = 0, Io, Wav, Codec, Args };
struct FileResult {
    FileStatus status = FileStatus::Ok;
    WavError wav = WavError::Ok;
    int codec = 0;       // lc3gpu error code
    size_t frames = 0;   // frames (all channels) processed
    std::string message;
};

FileResult encode_wav_to_lc3(const std::string &wav_file, const std::string &lc3_file, int fs_hz, int bits_per_sample,
                             int num_channels, int frame_us, int num_bytes_per_channel, const Lc3FileOptions &opt);
FileResult decode_lc3_to_wav(const std::string &lc3_file, const std::string &wav_file, int fs_hz, int bits_per_sample,
                             int num_channels, int frame_us, int num_bytes_per_channel, const Lc3FileOptions &opt);
// examples/compare.rs:6-36: first differing (frame, byte) of two .lc3 files read in chunks of `chunk` bytes;
// returns 0 when equal (up to the shorter common run of equal-length reads), 1 on a difference, <0 on I/O errors
int compare_files(const std::string &left, const std::string &right, size_t chunk, size_t *frame_index, size_t *byte_index,
                  int *left_byte, int *right_byte);

}  // namespace lc3files
