// Host-side plan construction for the LC3 codec kernels: configuration constants
// (reference common/config.rs:42-100), the mixed-radix FFT plan (common/kissfft.rs:47-76),
// the f64->f32 twiddle tables (kissfft.rs:19-27, dct_iv.rs:30-35) and the leaf gather order of
// the kf_work recursion (kissfft.rs:86-131).  Pure C++ (no HIP calls) so that the C ABI layer
// and the CPU wave-emulator used by the tests build the very same plan.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

struct lc3_host_plan {
    std::vector<lc3_cpx> fft_tw, dct_tw;
    std::vector<uint16_t> perm;
};

// returns 0 on success, -1 on an unsupported configuration
static inline int lc3_make_config(lc3_cfg &c, int frame_us, int fs_hz) {
    static const int fs_tab[6] = {8000, 16000, 24000, 32000, 44100, 48000};
    static const int ind_tab[6] = {0, 1, 2, 3, 4, 4};
    static const int nf75[6] = {60, 120, 180, 240, 360, 360};
    static const int nf10[6] = {80, 160, 240, 320, 480, 480};
    int k = -1;
    for (int i = 0; i < 6; i++)
        if (fs_tab[i] == fs_hz) k = i;
    if (k < 0 || (frame_us != 7500 && frame_us != 10000)) return -1;
    std::memset(&c, 0, sizeof(c));
    c.fs = fs_hz;
    c.fs_ind = ind_tab[k];
    c.n_ms_10 = frame_us == 10000;
    if (!c.n_ms_10) {
        c.nf = nf75[k];
        c.ne = c.nf == 360 ? 300 : c.nf;
        c.nb = fs_hz == 8000 ? 60 : 64;
        c.z = 7 * c.nf / 30;
    } else {
        c.nf = nf10[k];
        c.ne = c.nf == 480 ? 400 : c.nf;
        c.nb = 64;
        c.z = 3 * c.nf / 8;
    }
    // encoder LTPF constants (encoder/long_term_post_filter.rs:93-127)
    c.len12 = c.n_ms_10 ? 128 : 96;
    c.len6 = c.n_ms_10 ? 64 : 48;
    c.delay12 = c.n_ms_10 ? 24 : 44;
    float rf = 1.0f;
    switch (fs_hz) {
    case 8000: c.p_up = 24; rf = 0.5f; break;
    case 16000: c.p_up = 12; break;
    case 24000: c.p_up = 8; break;
    case 32000: c.p_up = 6; break;
    default: c.p_up = 4; break;
    }
    c.hist = 240 / c.p_up;
    c.inv_p = (65536 + c.p_up - 1) / c.p_up;
    for (int x = 0; x < 15 * 128; x++)
        if (((x * c.inv_p) >> 16) != x / c.p_up) return -1;
    c.resamp_lim = 120 / c.p_up;
    c.resamp_nt = (2 * c.resamp_lim + 1 + 3) & ~3;
    c.resamp_stride = c.resamp_nt + 4;  // row pitch: keeps the p rows on different LDS banks
    c.resamp_scale = (float)c.p_up * rf;
    // decoder LTPF constants (decoder/long_term_post_filter.rs:104-134)
    switch (fs_hz) {
    case 8000: c.l_den = 4; break;
    case 16000: c.l_den = 4; break;
    case 24000: c.l_den = 6; break;
    case 32000: c.l_den = 8; break;
    case 44100: c.l_den = 11; break;
    default: c.l_den = 12; break;
    }
    c.l_num = c.l_den - 2;
    c.num_mem_blocks = c.n_ms_10 ? 2 : 3;
    c.norm = c.n_ms_10 ? c.nf / 4 : c.nf / 3;
    c.s25 = fs_hz == 44100 ? 48000 / 400 : fs_hz / 400;
    return 0;
}

// fills c.nfft / n_stages / radix / m / fstride and the three tables (host copies)
static inline int lc3_make_plan(lc3_cfg &c, lc3_host_plan &pl) {
    const int nfft = c.nf / 2;
    c.nfft = nfft;
    {
        int n = nfft, p = 4, i = 0;
        const float floor_sqrt = std::floor(std::sqrt((float)n));
        int fstride = 1;
        for (;;) {
            while ((n % p) != 0) {
                if (p == 4) p = 2;
                else if (p == 2) p = 3;
                else p += 2;
                if ((float)p > floor_sqrt) p = n;
            }
            n /= p;
            if (i >= 6 || (p != 2 && p != 3 && p != 4 && p != 5)) return -1;
            c.radix[i] = p;
            c.m[i] = n;
            c.fstride[i] = fstride;
            fstride *= p;
            i++;
            if (n <= 1) break;
        }
        c.n_stages = i;
        for (int s2 = 0; s2 < i; s2++) {
            c.inv_m[s2] = (65536 + c.m[s2] - 1) / c.m[s2];
            for (int u = 0; u < nfft; u++)
                if (((u * c.inv_m[s2]) >> 16) != u / c.m[s2]) return -1;
        }
    }
    pl.fft_tw.resize((size_t)nfft);
    pl.dct_tw.resize((size_t)nfft);
    pl.perm.resize((size_t)nfft);
    const double PI = 3.14159265358979323846264338327950288;
    for (int i = 0; i < nfft; i++) {
        const double phase = -2.0 * PI * (double)i / (double)nfft;
        pl.fft_tw[(size_t)i].r = (float)std::cos(phase);
        pl.fft_tw[(size_t)i].i = (float)std::sin(phase);
        const double t = -PI * (double)(8 * i + 1) / (8.0 * ((double)nfft) * 2.0);
        pl.dct_tw[(size_t)i].r = (float)std::cos(t);
        pl.dct_tw[(size_t)i].i = (float)std::sin(t);
    }
    // output slot o = sum q_s * m_s  <-  input index sum q_s * fstride_s
    for (int o = 0; o < nfft; o++) {
        int rem = o, idx = 0;
        for (int s = 0; s < c.n_stages; s++) {
            const int q = rem / c.m[s];
            rem -= q * c.m[s];
            idx += q * c.fstride[s];
        }
        pl.perm[(size_t)o] = (uint16_t)idx;
    }
    return 0;
}
