// File drivers on top of the C ABI: see lc3_files.hpp.  Built by __graft_entry__.build() into
// lc3-codec_amd/lib/lc3gpu-tool (hipcc: the drivers own the device staging buffers).
#include "lc3_files.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

namespace lc3files {
namespace {

bool read_all(const std::string &name, std::vector<uint8_t> &out) {
    FILE *f = std::fopen(name.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const size_t got = out.empty() ? 0 : std::fread(out.data(), 1, out.size(), f);
    std::fclose(f);
    return got == out.size();
}

struct DeviceBuf {
    void *p = nullptr;
    ~DeviceBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess; }
};

FileResult fail(FileStatus st, const std::string &msg, int codec = 0, WavError w = WavError::Ok) {
    FileResult r;
    r.status = st;
    r.message = msg;
    r.codec = codec;
    r.wav = w;
    return r;
}

}  // namespace

FileResult encode_wav_to_lc3(const std::string &wav_file, const std::string &lc3_file, int fs_hz, int bits_per_sample,
                             int num_channels, int frame_us, int nbytes, const Lc3FileOptions &opt) {
    if (bits_per_sample != 16 || num_channels <= 0 || nbytes <= 0) return fail(FileStatus::Args, "16-bit PCM, >= 1 channel");
    std::vector<uint8_t> in;
    if (!read_all(wav_file, in)) return fail(FileStatus::Io, "cannot read " + wav_file);
    WavHeader hdr;
    const WavError we = read_header(in.data(), in.size(), &hdr);
    if (we != WavError::Ok) return fail(FileStatus::Wav, "bad WAV header", 0, we);
    int cfg[7];
    int rc = lc3gpu_config(frame_us, fs_hz, cfg);
    if (rc) return fail(FileStatus::Codec, "unsupported configuration", rc);
    const int nf = cfg[5];
    const size_t C = (size_t)num_channels, bytes_per_frame = (size_t)nf * C * 2;
    const size_t start = hdr.data_start_position;
    // for cursor in (start..len).step_by(bytes_per_frame): a last partial frame is zero padded (examples/encode.rs:72-95)
    const size_t total_frames = in.size() > start ? (in.size() - start + bytes_per_frame - 1) / bytes_per_frame : 0;
    FILE *out = std::fopen(lc3_file.c_str(), "wb");
    if (!out) return fail(FileStatus::Io, "cannot create " + lc3_file);
    lc3gpu_encoder *enc = nullptr;
    rc = lc3gpu_encoder_create(&enc, num_channels, frame_us, fs_hz);
    if (rc) { std::fclose(out); return fail(FileStatus::Codec, "encoder create failed", rc); }
    const size_t T = (size_t)std::max(1, opt.frames_per_launch);
    // The WAV sample order (int16[frame][sample][channel]) and the .lc3 frame order (uint8[frame][channel][nbytes]) ARE the
    // interleaved layout of the batch call: what examples/encode.rs:95-115 does per frame on the host (de-interleave, one
    // encode_frame per channel, frames appended channel after channel) happens inside the kernels' loads and stores.
    std::vector<int16_t> inter(T * (size_t)nf * C);
    std::vector<uint8_t> bytes(T * C * (size_t)nbytes);
    DeviceBuf d_pcm, d_out;
    FileResult res;
    if (!d_pcm.alloc(inter.size() * 2) || !d_out.alloc(bytes.size())) {
        res = fail(FileStatus::Codec, "device allocation failed", LC3GPU_EHIP);
    } else {
        for (size_t f0 = 0; f0 < total_frames && res.status == FileStatus::Ok; f0 += T) {
            const size_t tn = std::min(T, total_frames - f0);
            // little-endian samples as they lie in the file; the last partial frame is zero padded (examples/encode.rs:72-95)
            const size_t cursor = start + f0 * bytes_per_frame, n_samples = tn * (size_t)nf * C;
            for (size_t i = 0; i < n_samples; i++) {
                const size_t bpos = cursor + 2 * i;
                inter[i] = bpos + 1 < in.size() ? (int16_t)rd16(in.data() + bpos) : (int16_t)0;
            }
            if (hipMemcpy(d_pcm.p, inter.data(), n_samples * 2, hipMemcpyHostToDevice) != hipSuccess) {
                res = fail(FileStatus::Codec, "upload failed", LC3GPU_EHIP);
                break;
            }
            rc = lc3gpu_encode_layout(enc, LC3GPU_LAYOUT_INTERLEAVED, (const int16_t *)d_pcm.p, (uint8_t *)d_out.p, nbytes, (int)tn, nullptr);
            if (rc) { res = fail(FileStatus::Codec, "encode failed", rc); break; }
            if (hipMemcpy(bytes.data(), d_out.p, tn * C * (size_t)nbytes, hipMemcpyDeviceToHost) != hipSuccess) {
                res = fail(FileStatus::Codec, "download failed", LC3GPU_EHIP);
                break;
            }
            if (std::fwrite(bytes.data(), 1, tn * C * (size_t)nbytes, out) != tn * C * (size_t)nbytes) {
                res = fail(FileStatus::Io, "write failed");
                break;
            }
            res.frames += tn;
        }
    }
    (void)lc3gpu_encoder_destroy(enc);
    std::fclose(out);
    return res;
}

FileResult decode_lc3_to_wav(const std::string &lc3_file, const std::string &wav_file, int fs_hz, int bits_per_sample,
                             int num_channels, int frame_us, int nbytes, const Lc3FileOptions &opt) {
    if (num_channels <= 0 || nbytes <= 0) return fail(FileStatus::Args, ">= 1 channel");
    if (bits_per_sample != 16) return fail(FileStatus::Codec, "Only16BitsPerAudioSampleSupported", LC3GPU_EBITS);
    std::vector<uint8_t> in;
    if (!read_all(lc3_file, in)) return fail(FileStatus::Io, "cannot read " + lc3_file);
    int cfg[7];
    int rc = lc3gpu_config(frame_us, fs_hz, cfg);
    if (rc) return fail(FileStatus::Codec, "unsupported configuration", rc);
    const int nf = cfg[5], fs = cfg[1];
    const size_t C = (size_t)num_channels;
    // channel frames the reference loop decodes: unit u while (u + 1) * nbytes < len (`to_index >= len -> return`,
    // examples/decode.rs:96-104); a frame is written only when all of its channels were decoded
    size_t units;
    if (opt.reference_drop_last_frame) units = in.empty() ? 0 : (in.size() + (size_t)nbytes - 1) / (size_t)nbytes - 1;
    else units = in.size() / (size_t)nbytes;
    const size_t total_frames = units / C;
    FILE *out = std::fopen(wav_file.c_str(), "wb");
    if (!out) return fail(FileStatus::Io, "cannot create " + wav_file);
    WavHeader h;
    h.num_channels = C;
    h.sample_rate = (size_t)fs;
    h.byte_rate = (size_t)fs * C * 16 / 8;
    h.bits_per_sample = 16;
    h.data_start_position = FULL_WAV_HEADER_LEN;
    if (opt.reference_wav_header) {  // examples/decode.rs:71-80: sizes as they are before the first frame
        h.block_align = 4;
        h.data_size = 0;
        h.data_with_header_size = FULL_WAV_HEADER_LEN - RIFF_HEADER_ONLY_LEN;
    } else {
        h.block_align = C * 2;
        h.data_size = total_frames * (size_t)nf * C * 2;
        h.data_with_header_size = h.data_size + FULL_WAV_HEADER_LEN - RIFF_HEADER_ONLY_LEN;
    }
    uint8_t hb[256];
    size_t hl = 0;
    (void)write_header(h, hb, sizeof(hb), &hl);
    if (std::fwrite(hb, 1, hl, out) != hl) { std::fclose(out); return fail(FileStatus::Io, "write failed"); }
    lc3gpu_decoder *dec = nullptr;
    rc = lc3gpu_decoder_create(&dec, num_channels, frame_us, fs_hz);
    if (rc) { std::fclose(out); return fail(FileStatus::Codec, "decoder create failed", rc); }
    const size_t T = (size_t)std::max(1, opt.frames_per_launch);
    // file order in, WAV order out: the interleaved layout of the batch call (examples/decode.rs:86-117 on the host)
    std::vector<int16_t> pcm(T * (size_t)nf * C);
    std::vector<uint8_t> frame_out(T * (size_t)nf * C * 2);
    DeviceBuf d_in, d_pcm;
    FileResult res;
    if (!d_in.alloc(T * C * (size_t)nbytes) || !d_pcm.alloc(pcm.size() * 2)) {
        res = fail(FileStatus::Codec, "device allocation failed", LC3GPU_EHIP);
    } else {
        for (size_t f0 = 0; f0 < total_frames && res.status == FileStatus::Ok; f0 += T) {
            const size_t tn = std::min(T, total_frames - f0);
            if (hipMemcpy(d_in.p, in.data() + f0 * C * (size_t)nbytes, tn * C * (size_t)nbytes, hipMemcpyHostToDevice) != hipSuccess) {
                res = fail(FileStatus::Codec, "upload failed", LC3GPU_EHIP);
                break;
            }
            rc = lc3gpu_decode_layout(dec, LC3GPU_LAYOUT_INTERLEAVED, (const uint8_t *)d_in.p, nullptr, (int16_t *)d_pcm.p, nbytes, (int)tn, nullptr);
            if (rc) { res = fail(FileStatus::Codec, "decode failed", rc); break; }
            const size_t n_samples = tn * (size_t)nf * C;
            if (hipMemcpy(pcm.data(), d_pcm.p, n_samples * 2, hipMemcpyDeviceToHost) != hipSuccess) {
                res = fail(FileStatus::Codec, "download failed", LC3GPU_EHIP);
                break;
            }
            for (size_t i = 0; i < n_samples; i++) wr16(frame_out.data() + 2 * i, (uint16_t)pcm[i]);  // little endian
            if (std::fwrite(frame_out.data(), 1, n_samples * 2, out) != n_samples * 2) {
                res = fail(FileStatus::Io, "write failed");
                break;
            }
            res.frames += tn;
        }
    }
    (void)lc3gpu_decoder_destroy(dec);
    std::fclose(out);
    return res;
}

int compare_files(const std::string &left, const std::string &right, size_t chunk, size_t *frame_index, size_t *byte_index,
                  int *left_byte, int *right_byte) {
    FILE *fl = std::fopen(left.c_str(), "rb"), *fr = std::fopen(right.c_str(), "rb");
    if (!fl || !fr) {
        if (fl) std::fclose(fl);
        if (fr) std::fclose(fr);
        return -1;
    }
    std::vector<uint8_t> bl(chunk), br(chunk);
    size_t frame = 0;
    int result = 0;
    for (;;) {
        frame++;
        const size_t nl = std::fread(bl.data(), 1, chunk, fl), nr = std::fread(br.data(), 1, chunk, fr);
        if (nl != nr || nl == 0) break;  // examples/compare.rs:24-27
        for (size_t i = 0; i < nl; i++)
            if (bl[i] != br[i]) {
                *frame_index = frame;
                *byte_index = i;
                *left_byte = bl[i];
                *right_byte = br[i];
                result = 1;
                break;
            }
        if (result) break;
    }
    std::fclose(fl);
    std::fclose(fr);
    return result;
}

}  // namespace lc3files
