// lc3gpu-tool: command-line front end of the file drivers (lc3_files.hpp).
//   lc3gpu-tool encode  <in.wav> <out.lc3> <fs_hz> <channels> <frame_us> <bytes_per_channel> [--frames-per-launch N]
//   lc3gpu-tool decode  <in.lc3> <out.wav> <fs_hz> <channels> <frame_us> <bytes_per_channel> [--fix-header] [--keep-last-frame]
//   lc3gpu-tool compare <left.lc3> <right.lc3> [chunk_bytes=150]
//   lc3gpu-tool wavinfo <file.wav>
//   lc3gpu-tool frame-timing [fs_hz=48000] [frame_us=10000] [bytes=150] [repeats=200]
//   lc3gpu-tool buffer-lengths <channels 1|2> <fs_hz> <frame_us>
//   lc3gpu-tool throughput [channels=16384] [frames_per_submit=4] [submits=200] [bytes=150]
// The positional arguments are the reference drivers' function parameters (examples/encode.rs:36-44,
// examples/decode.rs:36-44); the sampling frequency is NOT taken from the WAV header there either.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/lc3gpu.hpp"
#include "lc3_files.hpp"

// The headline arrangement from C++, no Python anywhere: what replaces the reference's two caller loops (examples/encode.rs:97-115,
// examples/decode.rs:93-112) for many channels is three calls -- lc3gpu_pipeline_create / _submit / _wait.  Synthetic PCM (a tone with
// two harmonics and a little noise per channel, 64 frames resident and walked through), device buffers from hipMalloc, two byte buffers
// alternating; prints encode + decode frames per second and checks the round trip on the first channel against lc3gpu_encode_frame /
// lc3gpu_decode_frame of the same frames (the single-frame calls of the reference's API shape).
static int throughput(int channels, int T, int submits, int nbytes) {
    using clk = std::chrono::steady_clock;
    const int nf = 480, R = 64 / T * T;
    std::vector<int16_t> pcm((size_t)channels * R * nf);
    unsigned lcg = 12345u;
    for (int c = 0; c < channels; c++) {
        const double f0 = 90.0 + 3.1 * (c % 97), amp = 2000.0 + 37.0 * (c % 211);
        for (int n = 0; n < R * nf; n++) {
            lcg = lcg * 1664525u + 1013904223u;
            const double t = n / 48000.0, x = amp * (std::sin(6.283185307179586 * f0 * t) + 0.5 * std::sin(12.566370614359172 * f0 * t + 0.7) +
                                                      0.33 * std::sin(18.84955592153876 * f0 * t + 1.9)) + ((int)(lcg >> 20) - 2048) * 0.05;
            pcm[((size_t)c * R) * nf + n] = (int16_t)std::lrint(std::max(-32768.0, std::min(32767.0, x)));
        }
    }
    // rotations: int16[R / T][channels][T][nf]
    const int n_rot = R / T;
    std::vector<int16_t> rot(pcm.size());
    for (int k = 0; k < n_rot; k++)
        for (int c = 0; c < channels; c++)
            std::memcpy(&rot[(((size_t)k * channels + c) * T) * nf], &pcm[((size_t)c * R + (size_t)k * T) * nf], sizeof(int16_t) * (size_t)T * nf);
    int16_t *d_pcm = nullptr, *d_out = nullptr;
    uint8_t *d_bytes[2] = {nullptr, nullptr};
    const size_t step_pcm = (size_t)channels * T * nf, step_bytes = (size_t)channels * T * nbytes;
    if (hipMalloc((void **)&d_pcm, sizeof(int16_t) * rot.size()) != hipSuccess || hipMalloc((void **)&d_out, sizeof(int16_t) * step_pcm) != hipSuccess ||
        hipMalloc((void **)&d_bytes[0], step_bytes) != hipSuccess || hipMalloc((void **)&d_bytes[1], step_bytes) != hipSuccess) {
        std::fprintf(stderr, "no device memory\n");
        return 1;
    }
    (void)hipMemcpy(d_pcm, rot.data(), sizeof(int16_t) * rot.size(), hipMemcpyHostToDevice);
    lc3gpu_pipeline *pl = nullptr;
    int rc = lc3gpu_pipeline_create(&pl, channels, 10000, 48000, 0);
    if (rc) { std::fprintf(stderr, "lc3gpu_pipeline_create: %s\n", lc3gpu_strerror(rc)); return 1; }
    // check: the first submission's channel 0 against the single-frame calls
    rc = lc3gpu_pipeline_submit(pl, d_pcm, d_bytes[0], d_out, nbytes, T);
    if (rc == 0) rc = lc3gpu_pipeline_wait(pl);
    if (rc) { std::fprintf(stderr, "submit: %s\n", lc3gpu_strerror(rc)); return 1; }
    std::vector<uint8_t> got_b((size_t)T * nbytes), want_b((size_t)nbytes);
    std::vector<int16_t> got_p((size_t)T * nf), want_p((size_t)nf);
    (void)hipMemcpy(got_b.data(), d_bytes[0], got_b.size(), hipMemcpyDeviceToHost);
    (void)hipMemcpy(got_p.data(), d_out, sizeof(int16_t) * got_p.size(), hipMemcpyDeviceToHost);
    {
        lc3gpu_encoder *e1 = nullptr;
        lc3gpu_decoder *d1 = nullptr;
        if (lc3gpu_encoder_create(&e1, 1, 10000, 48000) || lc3gpu_decoder_create(&d1, 1, 10000, 48000)) return 1;
        for (int t = 0; t < T; t++) {
            if (lc3gpu_encode_frame(e1, 0, &pcm[(size_t)t * nf], nf, want_b.data(), nbytes) ||
                lc3gpu_decode_frame(d1, 16, 0, want_b.data(), nbytes, want_p.data(), nf))
                return 1;
            if (std::memcmp(want_b.data(), &got_b[(size_t)t * nbytes], (size_t)nbytes) || std::memcmp(want_p.data(), &got_p[(size_t)t * nf], sizeof(int16_t) * nf)) {
                std::fprintf(stderr, "pipeline output differs from encode_frame / decode_frame at frame %d\n", t);
                return 3;
            }
        }
        lc3gpu_encoder_destroy(e1);
        lc3gpu_decoder_destroy(d1);
    }
    for (int k = 1; k < 1 + 8; k++) (void)lc3gpu_pipeline_submit(pl, d_pcm + (size_t)(k % n_rot) * step_pcm, d_bytes[k & 1], d_out, nbytes, T);
    (void)lc3gpu_pipeline_wait(pl);
    const auto t0 = clk::now();
    for (int k = 9; k < 9 + submits; k++) {
        rc = lc3gpu_pipeline_submit(pl, d_pcm + (size_t)(k % n_rot) * step_pcm, d_bytes[k & 1], d_out, nbytes, T);
        if (rc) { std::fprintf(stderr, "submit: %s\n", lc3gpu_strerror(rc)); return 1; }
    }
    rc = lc3gpu_pipeline_wait(pl);
    const double sec = std::chrono::duration<double>(clk::now() - t0).count();
    std::printf("%d channels x %d frames per submission, %d submissions: %.3f ms per submission, %.2f M frames/s encode + decode (lc3gpu_pipeline_submit, %d groups)\n",
                channels, T, submits, sec / submits * 1e3, (double)channels * T * submits / sec / 1e6, lc3gpu_pipeline_groups(pl));
    lc3gpu_pipeline_destroy(pl);
    (void)hipFree(d_pcm);
    (void)hipFree(d_out);
    (void)hipFree(d_bytes[0]);
    (void)hipFree(d_bytes[1]);
    return rc ? 1 : 0;
}

// Single-frame timing harness in the shape of the reference's embedded demo (examples/arm/src/main.rs:39-112): one
// channel, the no_std API shape (channel count as a template argument, Lc3EncoderStatic / Lc3DecoderStatic of
// lc3gpu.hpp), one encode_frame and one decode_frame call timed on the host clock -- here repeated, since a call is a
// PCIe round trip plus three or two kernel launches and its time varies.  The input is a synthetic 200 Hz tone with
// two harmonics, not the demo's array.
static int frame_timing(int fs_hz, int frame_us, int nbytes, int repeats) {
    using namespace lc3gpu;
    using clk = std::chrono::steady_clock;
    try {
        const FrameDuration d = (FrameDuration)frame_us;
        const SamplingFrequency f = (SamplingFrequency)fs_hz;
        int cfg[7];
        if (lc3gpu_config(frame_us, fs_hz, cfg)) { std::fprintf(stderr, "unsupported configuration\n"); return 2; }
        const int nf = cfg[5];
        const auto el = Lc3EncoderStatic<1>::calc_working_buffer_lengths(d, f);
        const auto dl = Lc3DecoderStatic<1>::calc_working_buffer_lengths(d, f);
        std::printf("working buffer lengths (reference units): encoder (%zu, %zu, %zu), decoder (%zu, %zu)\n", std::get<0>(el),
                    std::get<1>(el), std::get<2>(el), std::get<0>(dl), std::get<1>(dl));
        Lc3EncoderStatic<1> enc(d, f);
        Lc3DecoderStatic<1> dec(d, f);
        std::vector<int16_t> pcm((size_t)nf), out((size_t)nf);
        std::vector<uint8_t> buf((size_t)nbytes);
        std::vector<double> te, td;
        long t = 0;
        for (int r = 0; r < repeats; r++) {
            for (int n = 0; n < nf; n++, t++) {
                const double x = 2.0 * 3.14159265358979323846 * 200.0 * (double)t / (double)fs_hz;
                pcm[(size_t)n] = (int16_t)std::lrint(6000.0 * std::sin(x) + 3000.0 * std::sin(2.0 * x) + 1500.0 * std::sin(3.0 * x));
            }
            const auto t0 = clk::now();
            enc.encode_frame(0, pcm, buf);
            const auto t1 = clk::now();
            dec.decode_frame(16, 0, buf, out);
            const auto t2 = clk::now();
            te.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            td.push_back(std::chrono::duration<double, std::micro>(t2 - t1).count());
        }
        std::sort(te.begin(), te.end());
        std::sort(td.begin(), td.end());
        std::printf("Encoded in %.0f microseconds (median of %d; min %.0f, p99 %.0f)\n", te[te.size() / 2], repeats, te.front(),
                    te[(te.size() * 99) / 100]);
        std::printf("Decoded in %.0f microseconds (median of %d; min %.0f, p99 %.0f)\n", td[td.size() / 2], repeats, td.front(),
                    td[(td.size() * 99) / 100]);
        return 0;
    } catch (const Error &e) {
        std::fprintf(stderr, "frame-timing failed: %s\n", e.what());
        return 1;
    }
}

static int usage() {
    std::fprintf(stderr,
                 "usage: lc3gpu-tool encode <in.wav> <out.lc3> <fs_hz> <channels> <frame_us> <bytes_per_channel> [--frames-per-launch N]\n"
                 "       lc3gpu-tool decode <in.lc3> <out.wav> <fs_hz> <channels> <frame_us> <bytes_per_channel> [--fix-header] [--keep-last-frame]\n"
                 "       lc3gpu-tool compare <left> <right> [chunk_bytes]\n"
                 "       lc3gpu-tool wavinfo <file.wav>\n"
                 "       lc3gpu-tool frame-timing [fs_hz] [frame_us] [bytes] [repeats]\n"
                 "       lc3gpu-tool buffer-lengths <channels 1|2> <fs_hz> <frame_us>\n"
                 "       lc3gpu-tool throughput [channels] [frames_per_submit] [submits] [bytes]\n");
    return 2;
}

int main(int argc, char **argv) {
    if (argc < 2) return usage();
    const std::string cmd = argv[1];
    if (cmd == "compare") {
        if (argc < 4) return usage();
        size_t frame = 0, byte = 0;
        int l = 0, r = 0;
        const size_t chunk = argc > 4 ? (size_t)std::atoi(argv[4]) : 150;
        const int rc = lc3files::compare_files(argv[2], argv[3], chunk, &frame, &byte, &l, &r);
        if (rc < 0) { std::fprintf(stderr, "cannot open inputs\n"); return 3; }
        if (rc == 1) std::printf("Diff at frame %zu byte index %zu: left: %d right: %d\n", frame, byte, l, r);
        else std::printf("Completed comparing: no difference\n");
        return rc;
    }
    if (cmd == "throughput")
        return throughput(argc > 2 ? std::max(4, std::atoi(argv[2])) : 16384, argc > 3 ? std::max(1, std::min(64, std::atoi(argv[3]))) : 4,
                          argc > 4 ? std::max(1, std::atoi(argv[4])) : 200, argc > 5 ? std::atoi(argv[5]) : 150);
    if (cmd == "frame-timing")
        return frame_timing(argc > 2 ? std::atoi(argv[2]) : 48000, argc > 3 ? std::atoi(argv[3]) : 10000,
                            argc > 4 ? std::atoi(argv[4]) : 150, argc > 5 ? std::max(1, std::atoi(argv[5])) : 200);
    if (cmd == "buffer-lengths") {  // calc_working_buffer_lengths of the static-channel API shape; no device needed
        if (argc < 5) return usage();
        const int ch = std::atoi(argv[2]);
        const lc3gpu::SamplingFrequency f = (lc3gpu::SamplingFrequency)std::atoi(argv[3]);
        const lc3gpu::FrameDuration d = (lc3gpu::FrameDuration)std::atoi(argv[4]);
        try {
            const auto el = ch == 1 ? lc3gpu::Lc3EncoderStatic<1>::calc_working_buffer_lengths(d, f)
                                    : lc3gpu::Lc3EncoderStatic<>::calc_working_buffer_lengths(d, f);
            const auto dl = ch == 1 ? lc3gpu::Lc3DecoderStatic<1>::calc_working_buffer_lengths(d, f)
                                    : lc3gpu::Lc3DecoderStatic<>::calc_working_buffer_lengths(d, f);
            std::printf("%zu %zu %zu %zu %zu\n", std::get<0>(el), std::get<1>(el), std::get<2>(el), std::get<0>(dl), std::get<1>(dl));
        } catch (const lc3gpu::Error &e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 1;
        }
        return 0;
    }
    if (cmd == "wavinfo") {  // header fields as read by lc3files::read_header (src/common/wav.rs:69-127)
        if (argc < 3) return usage();
        FILE *f = std::fopen(argv[2], "rb");
        if (!f) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 3; }
        uint8_t buf[64];
        const size_t n = std::fread(buf, 1, sizeof(buf), f);
        std::fclose(f);
        lc3files::WavHeader h;
        const lc3files::WavError e = lc3files::read_header(buf, n, &h);
        if (e != lc3files::WavError::Ok) { std::printf("error %d\n", (int)e); return 1; }
        std::printf("num_channels %zu sample_rate %zu byte_rate %zu block_align %zu bits_per_sample %zu data_size %zu "
                    "data_start_position %zu data_with_header_size %zu\n",
                    h.num_channels, h.sample_rate, h.byte_rate, h.block_align, h.bits_per_sample, h.data_size,
                    h.data_start_position, h.data_with_header_size);
        return 0;
    }
    if ((cmd != "encode" && cmd != "decode") || argc < 8) return usage();
    lc3files::Lc3FileOptions opt;
    for (int i = 8; i < argc; i++) {
        if (!std::strcmp(argv[i], "--fix-header")) opt.reference_wav_header = false;
        else if (!std::strcmp(argv[i], "--keep-last-frame")) opt.reference_drop_last_frame = false;
        else if (!std::strcmp(argv[i], "--frames-per-launch") && i + 1 < argc) opt.frames_per_launch = std::atoi(argv[++i]);
        else return usage();
    }
    const int fs = std::atoi(argv[4]), ch = std::atoi(argv[5]), us = std::atoi(argv[6]), nb = std::atoi(argv[7]);
    const lc3files::FileResult res = cmd == "encode" ? lc3files::encode_wav_to_lc3(argv[2], argv[3], fs, 16, ch, us, nb, opt)
                                                     : lc3files::decode_lc3_to_wav(argv[2], argv[3], fs, 16, ch, us, nb, opt);
    if (res.status != lc3files::FileStatus::Ok) {
        std::fprintf(stderr, "%s failed: %s (status %d, wav %d, codec %d: %s)\n", cmd.c_str(), res.message.c_str(), (int)res.status,
                     (int)res.wav, res.codec, lc3gpu_strerror(res.codec));
        return 1;
    }
    std::printf("%s: %zu frames x %d channels\n", cmd.c_str(), res.frames, ch);
    return 0;
}
