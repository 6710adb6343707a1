// lc3gpu-tool: command-line front end of the file drivers (lc3_files.hpp).
//   lc3gpu-tool encode  <in.wav> <out.lc3> <fs_hz> <channels> <frame_us> <bytes_per_channel> [--frames-per-launch N]
//   lc3gpu-tool decode  <in.lc3> <out.wav> <fs_hz> <channels> <frame_us> <bytes_per_channel> [--fix-header] [--keep-last-frame]
//   lc3gpu-tool compare <left.lc3> <right.lc3> [chunk_bytes=150]
//   lc3gpu-tool wavinfo <file.wav>
// The positional arguments are the reference drivers' function parameters (examples/encode.rs:36-44,
// examples/decode.rs:36-44); the sampling frequency is NOT taken from the WAV header there either.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "lc3_files.hpp"

static int usage() {
    std::fprintf(stderr,
                 "usage: lc3gpu-tool encode <in.wav> <out.lc3> <fs_hz> <channels> <frame_us> <bytes_per_channel> [--frames-per-launch N]\n"
                 "       lc3gpu-tool decode <in.lc3> <out.wav> <fs_hz> <channels> <frame_us> <bytes_per_channel> [--fix-header] [--keep-last-frame]\n"
                 "       lc3gpu-tool compare <left> <right> [chunk_bytes]\n");
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
