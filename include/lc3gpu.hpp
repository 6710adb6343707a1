// lc3gpu.hpp -- header-only C++ facade over the C ABI (lc3gpu.h) with the reference's type and method names
// (ninjasource/lc3-codec v0.2.0: src/encoder/lc3_encoder.rs:117-209, src/decoder/lc3_decoder.rs:181-244,
// src/common/config.rs:1-15).  The reference lends caller-allocated working buffers to the codec object; the GPU
// engine owns device memory instead, calc_working_buffer_lengths() is kept for API parity.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "lc3gpu.h"

namespace lc3gpu {

enum class SamplingFrequency : int { Hz8000 = 8000, Hz16000 = 16000, Hz24000 = 24000, Hz32000 = 32000, Hz44100 = 44100, Hz48000 = 48000 };
enum class FrameDuration : int { SevenPointFiveMs = 7500, TenMs = 10000 };

struct Error : std::runtime_error {
    int code;
    explicit Error(int c, const char *what) : std::runtime_error(std::string(what) + ": " + lc3gpu_strerror(c)), code(c) {}
};
// Lc3DecoderError::Only16BitsPerAudioSampleSupported (lc3_decoder.rs:36-42)
struct Only16BitsPerAudioSampleSupported : Error { using Error::Error; };

class Lc3Encoder {
public:
    // (integer_len, scaler_len, complex_len), lc3_encoder.rs:194-209
    static std::tuple<size_t, size_t, size_t> calc_working_buffer_lengths(size_t num_channels, FrameDuration d, SamplingFrequency f) {
        int64_t o[3];
        int rc = lc3gpu_encoder_working_buffer_lengths((int)num_channels, (int)d, (int)f, o);
        if (rc) throw Error(rc, "calc_working_buffer_lengths");
        return {(size_t)o[0], (size_t)o[1], (size_t)o[2]};
    }
    // spec_flags: LC3GPU_SPEC_* corrections of the reference's deviations from the specification (0 = the reference's behaviour)
    Lc3Encoder(size_t num_channels, FrameDuration d, SamplingFrequency f, int spec_flags = 0) {
        int rc = lc3gpu_encoder_create_spec(&h_, (int)num_channels, (int)d, (int)f, spec_flags);
        if (rc) throw Error(rc, "Lc3Encoder::new");
    }
    // one handle for streams of different configurations (one launch per kernel for the whole mixed batch)
    explicit Lc3Encoder(const std::vector<lc3gpu_stream_desc> &streams, int spec_flags = 0) {
        int rc = lc3gpu_encoder_create_mixed_spec(&h_, (int)streams.size(), streams.data(), spec_flags);
        if (rc) throw Error(rc, "Lc3Encoder::mixed");
    }
    ~Lc3Encoder() { lc3gpu_encoder_destroy(h_); }
    Lc3Encoder(const Lc3Encoder &) = delete;
    Lc3Encoder &operator=(const Lc3Encoder &) = delete;
    // encode_frame(channel_index, samples_in, buf_out): buf_out.size() selects the bitrate (lc3_encoder.rs:65)
    void encode_frame(size_t channel_index, const std::vector<int16_t> &samples_in, std::vector<uint8_t> &buf_out) {
        int rc = lc3gpu_encode_frame(h_, (int)channel_index, samples_in.data(), (int)samples_in.size(), buf_out.data(), (int)buf_out.size());
        if (rc) throw Error(rc, "encode_frame");  // the reference panics here; Err() is impossible (empty enum)
    }
    // batch: device pointers, stream-major, asynchronous on `hip_stream`
    void encode(const int16_t *d_pcm, uint8_t *d_out, int nbytes, int n_frames, void *hip_stream = nullptr,
                int layout = LC3GPU_LAYOUT_PLANAR) {
        int rc = lc3gpu_encode_layout(h_, layout, d_pcm, d_out, nbytes, n_frames, hip_stream);
        if (rc) throw Error(rc, "encode");
    }
    // mixed handle: ragged buffers in descriptor order (lc3gpu.h)
    void encode_mixed(const int16_t *d_pcm, uint8_t *d_out, int n_frames, void *hip_stream = nullptr) {
        int rc = lc3gpu_encode_mixed(h_, d_pcm, d_out, n_frames, hip_stream);
        if (rc) throw Error(rc, "encode_mixed");
    }
    uint64_t pair_timeouts() {
        uint64_t v = 0;
        int rc = lc3gpu_encoder_pair_timeouts(h_, &v);
        if (rc) throw Error(rc, "pair_timeouts");
        return v;
    }
    lc3gpu_encoder *handle() { return h_; }

private:
    lc3gpu_encoder *h_ = nullptr;
};

class Lc3Decoder {
public:
    // (scaler_len, complex_len), lc3_decoder.rs:236-244
    static std::tuple<size_t, size_t> calc_working_buffer_lengths(size_t num_channels, FrameDuration d, SamplingFrequency f) {
        int64_t o[2];
        int rc = lc3gpu_decoder_working_buffer_lengths((int)num_channels, (int)d, (int)f, o);
        if (rc) throw Error(rc, "calc_working_buffer_lengths");
        return {(size_t)o[0], (size_t)o[1]};
    }
    Lc3Decoder(size_t num_channels, FrameDuration d, SamplingFrequency f) {
        int rc = lc3gpu_decoder_create(&h_, (int)num_channels, (int)d, (int)f);
        if (rc) throw Error(rc, "Lc3Decoder::new");
    }
    explicit Lc3Decoder(const std::vector<lc3gpu_stream_desc> &streams) {
        int rc = lc3gpu_decoder_create_mixed(&h_, (int)streams.size(), streams.data());
        if (rc) throw Error(rc, "Lc3Decoder::mixed");
    }
    ~Lc3Decoder() { lc3gpu_decoder_destroy(h_); }
    Lc3Decoder(const Lc3Decoder &) = delete;
    Lc3Decoder &operator=(const Lc3Decoder &) = delete;
    // decode_frame(num_bits_per_audio_sample, channel_index, buf_in, samples_out); corrupt frames are concealed
    void decode_frame(size_t num_bits_per_audio_sample, size_t channel_index, const std::vector<uint8_t> &buf_in, std::vector<int16_t> &samples_out) {
        int rc = lc3gpu_decode_frame(h_, (int)num_bits_per_audio_sample, (int)channel_index, buf_in.data(), (int)buf_in.size(), samples_out.data(), (int)samples_out.size());
        if (rc == LC3GPU_EBITS) throw Only16BitsPerAudioSampleSupported(rc, "decode_frame");
        if (rc) throw Error(rc, "decode_frame");
    }
    void decode(const uint8_t *d_in, int16_t *d_pcm, int nbytes, int n_frames, void *hip_stream = nullptr, const uint8_t *d_bad_frame = nullptr,
                int layout = LC3GPU_LAYOUT_PLANAR) {
        int rc = lc3gpu_decode_layout(h_, layout, d_in, d_bad_frame, d_pcm, nbytes, n_frames, hip_stream);
        if (rc) throw Error(rc, "decode");
    }
    void decode_mixed(const uint8_t *d_in, int16_t *d_pcm, int n_frames, void *hip_stream = nullptr, const uint8_t *d_bad_frame = nullptr) {
        int rc = lc3gpu_decode_mixed(h_, d_in, d_bad_frame, d_pcm, n_frames, hip_stream);
        if (rc) throw Error(rc, "decode_mixed");
    }
    uint64_t plc_events() {
        uint64_t v = 0;
        int rc = lc3gpu_decoder_plc_events(h_, &v);
        if (rc) throw Error(rc, "plc_events");
        return v;
    }
    uint64_t pair_timeouts() {
        uint64_t v = 0;
        int rc = lc3gpu_decoder_pair_timeouts(h_, &v);
        if (rc) throw Error(rc, "pair_timeouts");
        return v;
    }
    lc3gpu_decoder *handle() { return h_; }

private:
    lc3gpu_decoder *h_ = nullptr;
};

// ---- the reference's no_std / no-alloc API shape (lc3_encoder.rs:37-40,212-303, lc3_decoder.rs:56-60,247-310): the
// number of channels is a compile-time constant instead of a constructor argument (`Lc3Encoder::<NUM_CH>::new(duration,
// frequency, ..)`, `Lc3Encoder::<NUM_CH>::calc_working_buffer_lengths(duration, frequency)`; default 2 as in the
// reference).  Same engine underneath; the caller-lent working buffers of the reference have no counterpart here.
template <size_t NUM_CHANNELS = 2>
class Lc3EncoderStatic : public Lc3Encoder {
public:
    static constexpr size_t num_channels = NUM_CHANNELS;
    static std::tuple<size_t, size_t, size_t> calc_working_buffer_lengths(FrameDuration d, SamplingFrequency f) {
        return Lc3Encoder::calc_working_buffer_lengths(NUM_CHANNELS, d, f);
    }
    Lc3EncoderStatic(FrameDuration d, SamplingFrequency f) : Lc3Encoder(NUM_CHANNELS, d, f) {}
};
template <size_t NUM_CHANNELS = 2>
class Lc3DecoderStatic : public Lc3Decoder {
public:
    static constexpr size_t num_channels = NUM_CHANNELS;
    static std::tuple<size_t, size_t> calc_working_buffer_lengths(FrameDuration d, SamplingFrequency f) {
        return Lc3Decoder::calc_working_buffer_lengths(NUM_CHANNELS, d, f);
    }
    Lc3DecoderStatic(FrameDuration d, SamplingFrequency f) : Lc3Decoder(NUM_CHANNELS, d, f) {}
};

}  // namespace lc3gpu
