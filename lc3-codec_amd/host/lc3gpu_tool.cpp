// lc3gpu-tool: command-line front end of the file drivers (lc3_files.hpp).
//   lc3gpu-tool encode  <in.wav> <out.lc3> <fs_hz> <channels> <frame_us> <bytes_per_channel> [--frames-per-launch N]
//   lc3gpu-tool decode  <in.lc3> <out.wav> <fs_hz> <channels> <frame_us> <bytes_per_channel> [--fix-header] [--keep-last-frame]
//   lc3gpu-tool compare <left.lc3> <right.lc3> [chunk_bytes=150]
//   lc3gpu-tool wavinfo <file.wav>
//   lc3gpu-tool frame-timing [fs_hz=48000] [frame_us=10000] [bytes=150] [repeats=200]
//   lc3gpu-tool buffer-lengths <channels 1|2> <fs_hz> <frame_us>
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

#include "../../include/lc3gpu.hpp"
#include "lc3_files.hpp"

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
                 "       lc3gpu-tool buffer-lengths <channels 1|2> <fs_hz> <frame_us>\n");
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
