// LC3 batched decoder for MI355X -- spectrum reconstruction D4-D8 as kernels of its own: lc3_recon_kernel (ONE WAVE PER FRAME) and
// lc3_tns_kernel (ONE LANE PER FRAME), lc3gpu.hip.  OPT-IN (LC3GPU_RECON=wave), not the default: measured against the reconstruction
// on the parser's lane (lc3_reconstruct_frame, the default of full batches) it loses, 0.383 vs 0.339 ms per 65 536 frames (DESIGN
// section 6); it is kept selectable and tested beside the other two forms.
//
// What DecoderChannel::decode does between the range decoder and the inverse transform (reference decoder/lc3_decoder.rs:99-133):
// residual_spectrum::decode (decoder/residual_spectrum.rs:13-39), noise_filling::apply_noise_filling (decoder/noise_filling.rs:18-56),
// global_gain::apply_global_gain (decoder/global_gain.rs:15-25), temporal_noise_shaping::apply_temporal_noise_shaping
// (decoder/temporal_noise_shaping.rs:24-137) and spectral_noise_shaping::decode (decoder/spectral_noise_shaping.rs:21-235) carries
// nothing from frame to frame.  In this form the lane-per-frame parser (lc3_dev_dec_parse.h) stops after the range
// decoder and the rest is split by the nature of the work:
//   * everything that is parallel over the LINES of a frame runs one wave per frame at full occupancy (lc3_recon_frame_direct): residual
//     bits, noise filling, global gain, scale factors, band gains.  Lane l owns lines 8l .. 8l+7; what the reference carries from line
//     to line becomes a prefix count (the rank of a non-zero line among the non-zero lines picks its residual bit, the rank of a
//     noise-filled line among the filled ones its LCG state; the LCG is affine mod 2^16, f^R comes from a table).  The frame's words go
//     from the plane column straight to registers (side information: one 16-byte unit per lane, single words broadcast with
//     v_readlane; spectrum: two 16-byte loads per lane) and the f32 spectrum straight back; tables are staged once per workgroup;
//   * the TNS synthesis lattice (388 lines x up to 8 dependent stages) is a recursion over the lines: on one lane of a wave per frame it
//     occupies the SIMD 64-fold (measured: 0.60 of 0.73 ms of a reconstruction kernel that did it on lane 0 for the benchmark's frames),
//     so the frames with an active filter get their lattice -- and the band gains of the filtered range, which the reference applies
//     after it -- from a second pass with one LANE per frame (lc3_tns_lane_frame), 64 lattices per instruction.
// The arithmetic per line is that of lc3_reconstruct_frame / lc3_dec_reconstruct_wave, operation for operation.
#pragma once
#include "lc3_dev_dec.h"

// per workgroup (4.4 KB)
struct __attribute__((aligned(16))) lc3_recon_tables {
    float d[16 * 16];               // D[n][col] (decoder/spectral_noise_shaping.rs:60-66)
    float lfcb[32 * 8], hfcb[32 * 8];
    uint32_t jump[512];             // f^R of the noise-filling LCG f(x) = 13849 + 31821 x mod 2^16: A | C << 16
    uint32_t line_band[LC3_MAX_NF / 4];  // band of line k in byte k (255 beyond ne)
};
// per wave
struct __attribute__((aligned(16))) lc3_recon_wave {
    float sc[16 + 64];              // scale factors, band gains
    uint32_t resw[16];              // residual bit mask
};

// the line -> band map of a configuration, four lines to a word (byte j of word i: band of line 4 i + j, 255 beyond ne)
template <class CC>
__device__ __forceinline__ uint32_t lc3_line_band_word(const CC &c, int i) {
    uint32_t w = 0;
    for (int j = 0; j < 4; j++) {
        const int k = 4 * i + j;
        w |= (uint32_t)(k < c.nf ? (int)c.line_band[k] : 255) << (8 * j);
    }
    return w;
}
// all threads of the workgroup; the caller adds the barrier
template <class CC>
__device__ __forceinline__ void lc3_recon_tables_stage(const CC &c, lc3_recon_tables &T, int tid, int nthreads) {
    for (int i = tid; i < 256; i += nthreads) {
        T.d[i] = lc3_f(&LC3T_D_BITS[0][0], i);
        T.lfcb[i] = lc3_f(&LC3T_LFCB_BITS[0][0], i);
        T.hfcb[i] = lc3_f(&LC3T_HFCB_BITS[0][0], i);
    }
    for (int i = tid; i < 512; i += nthreads) {
        uint32_t R = (uint32_t)i, pa = 31821u, pc = 13849u, A = 1u, C = 0u;  // f^(2^b) = pa x + pc; accumulated map A x + C
        for (int b = 0; b < 9; b++) {
            if (R & 1u) {
                A = (pa * A) & 0xFFFFu;
                C = (pa * C + pc) & 0xFFFFu;
            }
            pc = (pa * pc + pc) & 0xFFFFu;
            pa = (pa * pa) & 0xFFFFu;
            R >>= 1;
        }
        T.jump[i] = A | (C << 16);
    }
    for (int i = tid; i < LC3_MAX_NF / 4; i += nthreads) T.line_band[i] = lc3_line_band_word(c, i);
}

// the line range [lo, hi) the TNS filters of a frame of bandwidth bw cover (decoder/temporal_noise_shaping.rs:84-137): one range,
// the second filter starts where the first one stops
template <class CC>
__device__ __forceinline__ void lc3_tns_range(const CC &c, int bw, int &lo, int &hi) {
    lo = c.n_ms_10 ? LC3C_TNSDEC10[bw][0] : LC3C_TNSDEC75[bw][0];
    hi = c.n_ms_10 ? LC3C_TNSDEC10[bw][bw < 3 ? 1 : 3] : LC3C_TNSDEC75[bw][bw < 3 ? 1 : 3];
}

// col: the frame's plane column as the parser left it (side information with AD_NRES and the packed pulse vector AD_Y, integers,
// residual bit mask in the level words); on return the f32 spectrum stands in place of the integers: final, except for the TNS range
// of a frame with an active filter, whose lines are left before the lattice (gain applied, no band gain) with the 64 band gains in the
// level words for lc3_tns_lane_frame.  A frame the parser rejected (AD_OK = 0) is left alone: the synthesis kernel conceals it.
template <class CC>
__device__ __forceinline__ void lc3_recon_frame_direct(const CC &c, const lc3_recon_tables &T, lc3_recon_wave &W, int lane, int32_t *col,
                                                       int nbytes) {
    const int ne = c.ne, nbits = nbytes * 8;
    struct q4 { int32_t v[4]; };
    // side information: 12 units of 16 bytes on lanes 0..11, the residual bit mask (4 units at LC3_PLANE_LEV) on lanes 12..15
    static_assert(LC3_PLANE_X == 48 && LC3_PLANE_LEV % 4 == 0 && SI_WORDS <= LC3_PLANE_X, "side-information units");
    q4 si;
    {
        const int u = lane < 12 ? lane : (lane < 16 ? LC3_PLANE_LEV / 4 + lane - 12 : 0);
        si = __builtin_bit_cast(q4, ((LC3_HBM_CONST(lc3_i4))(LC3_HBM_CONST(int32_t))col)[u]);
    }
    // the lane's eight lines (integers; words at and beyond lastnz are stale)
    const int k0 = 8 * lane, have = k0 < ne;
    q4 xa = {{0, 0, 0, 0}}, xb = {{0, 0, 0, 0}};
    if (have) {
        LC3_HBM_CONST(lc3_i4) x4 = (LC3_HBM_CONST(lc3_i4))((LC3_HBM_CONST(int32_t))col + LC3_PLANE_X);
        xa = __builtin_bit_cast(q4, x4[2 * lane]);
        xb = __builtin_bit_cast(q4, x4[2 * lane + 1]);
    }
#define LC3_RSI(word) lc3_wave_read_i32(si.v[(word) & 3], (word) >> 2, lane)
    if (!LC3_RSI(AD_OK)) return;  // wave-uniform
    const int lastnz = LC3_RSI(SI_LASTNZ), gg_ind = LC3_RSI(SI_GG), bw = LC3_RSI(SI_BW);
    if (lane >= 12 && lane < 16) *(lc3_i4 *)(W.resw + 4 * (lane - 12)) = __builtin_bit_cast(lc3_i4, si);
    // spectral_noise_shaping::decode (:21-73): the pulse vector came de-enumerated from the parser; one scale factor per lane
    {
        const uint32_t yw0 = (uint32_t)LC3_RSI(AD_Y), yw1 = (uint32_t)LC3_RSI(AD_Y + 1), yw2 = (uint32_t)LC3_RSI(AD_Y + 2);
        int y[16];
#pragma unroll
        for (int n = 0; n < 16; n++) y[n] = lc3_pulse_unpack(yw0, yw1, yw2, n);
        const int shape_j = (LC3_RSI(SI_SUB_MSB) << 1) + LC3_RSI(SI_SUB_LSB);
        float y_norm = 0.0f;
#pragma unroll
        for (int n = 0; n < 16; n++) y_norm += (float)y[n] * (float)y[n];
        y_norm = lc3_sqrtf(y_norm);
        float gain;
        const int gi = LC3_RSI(SI_G_IND);
        if (shape_j == 0) gain = lc3_f(LC3T_SNS_VQ_REG_ADJ_GAINS_BITS, gi & 1);
        else if (shape_j == 1) gain = lc3_f(LC3T_SNS_VQ_REG_LF_ADJ_GAINS_BITS, gi & 3);
        else if (shape_j == 2) gain = lc3_f(LC3T_SNS_VQ_NEAR_ADJ_GAINS_BITS, gi & 3);
        else gain = lc3_f(LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS, gi & 7);
        if (y_norm != 0.0f) gain /= y_norm;
        const int ind_lf = LC3_RSI(SI_IND_LF), ind_hf = LC3_RSI(SI_IND_HF);
        const int n = lane & 15;
        float dr[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const lc3_f4 v = ((const lc3_f4 *)T.d)[4 * n + q];
            dr[4 * q] = v.x; dr[4 * q + 1] = v.y; dr[4 * q + 2] = v.z; dr[4 * q + 3] = v.w;
        }
        float factor = 0.0f;
#pragma unroll
        for (int cc = 0; cc < 16; cc++) factor += (float)y[cc] * dr[cc];
        const float st1 = n < 8 ? T.lfcb[(ind_lf & 31) * 8 + n] : T.hfcb[(ind_hf & 31) * 8 + n - 8];
        if (lane < 16) W.sc[n] = st1 + gain * factor;
    }
    LC3_SYNC();
    float g_band = 1.0f;  // lane b: the gain of band b
    {
        lc3_recon_ctx r;
        r.scf = W.sc;
        r.sstride = 1;
        r.mpvq = nullptr;
        r.ifs = nullptr;
        if (lane < c.nb) {
            g_band = lc3_r_band_gain(r, lane, c.nb);
            W.sc[16 + lane] = g_band;
        }
    }
    // global gain :15-25
    float gg;
    {
        const int fs = c.fs_ind + 1, q = nbits / (10 * fs);
        const int gg_off = -(q < 115 ? q : 115) - 105 - (5 * fs);
        gg = LC3_POW10_GG(gg_ind + gg_off);
    }
    int32_t xi[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        xi[j] = k0 + j < lastnz ? xa.v[j] : 0;
        xi[4 + j] = k0 + 4 + j < lastnz ? xb.v[j] : 0;
    }
    float v8[8];
    uint32_t nzmask = 0, absk = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        nzmask |= (uint32_t)(xi[j] != 0) << j;
        absk += (uint32_t)(xi[j] < 0 ? -xi[j] : xi[j]) * (uint32_t)(k0 + j);
    }
    // noise-filling seed :140-145 (a wrapping integer sum: any order) and the zero-frame flag :147-151
    uint32_t lcg = lc3_wave_sum_u32(absk, lane) & 0xFFFFu;
    const int x0 = lc3_wave_read_i32(xi[0], 0, lane), x1 = lc3_wave_read_i32(xi[1], 0, lane);
    const int do_fill = !(lastnz == 2 && x0 == 0 && x1 == 0 && gg_ind == 0);
    // noise filling :18-56: lines with an all-zero neighbourhood (lines at or beyond bw_stop count as zero, none below line 0)
    const int bw_stop = c.n_ms_10 ? LC3C_BWSTOP10[bw] : LC3C_BWSTOP75[bw];
    const int nf_start = c.n_ms_10 ? 24 : 18, nf_width = c.n_ms_10 ? 3 : 2;
    const int lim = bw_stop < ne ? bw_stop : ne;
    const float level = (8.0f - (float)LC3_RSI(SI_NF)) / 16.0f;
#define LC3_BITS_BELOW(n) ((n) <= 0 ? 0u : ((n) >= 8 ? 0xffu : (1u << (n)) - 1u))
    uint32_t fillmask;
    {
        const uint32_t nzw = nzmask & LC3_BITS_BELOW(bw_stop - k0);
        const uint32_t prev = (uint32_t)lc3_wave_shr1_i32((int)nzw, lane), next = (uint32_t)lc3_wave_shl1_i32((int)nzw, lane);
        const uint32_t nz14 = (prev >> 5) | (nzw << 3) | ((next & 7u) << 11);  // bit i <-> line k0 - 3 + i
        uint32_t any;
        if (nf_width == 3) {  // bits j .. j+6
            const uint32_t a = nz14 | (nz14 >> 1), b = a | (a >> 2);
            any = b | (b >> 3);
        } else {              // bits j+1 .. j+5
            const uint32_t n1 = nz14 >> 1, a = n1 | (n1 >> 1), b = a | (a >> 2);
            any = b | (n1 >> 4);
        }
        fillmask = do_fill ? (~any & LC3_BITS_BELOW(lim - k0) & ~LC3_BITS_BELOW(nf_start - k0) & 0xffu) : 0u;
    }
#undef LC3_BITS_BELOW
    {   // the LCG state before the lane's first filled line: f^R(seed), R = filled lines below (< 512)
        const uint32_t R = lc3_wave_exscan_u32((uint32_t)__builtin_popcount(fillmask), lane);
        const uint32_t ac = T.jump[R & 511u];
        lcg = ((ac & 0xFFFFu) * lcg + (ac >> 16)) & 0xFFFFu;
    }
    int rank_nz = (int)lc3_wave_exscan_u32((uint32_t)__builtin_popcount(nzmask), lane);
    // decode_residual_bits (decoder/arithmetic_codec.rs:168-183): one tail bit per non-zero line, at most AD_NRES_MAX; every
    // read_tail_bool bound check is monotone in the bit position, so checking the last position covers all of them
    int n_res = 0;
    if (!LC3_RSI(SI_LSB_MODE)) {
        const int nnz = lc3_wave_read_i32(rank_nz + __builtin_popcount(nzmask), 63, lane), nres_max = LC3_RSI(AD_NRES_MAX);
        n_res = nnz < nres_max ? nnz : nres_max;
        int bad = n_res > 480;  // ResidualBoolDataOverflow (Vec<bool, 480>)
        if (n_res > 0) {
            const int last_byte = (LC3_RSI(AD_TAIL0) + n_res - 1) / 8;
            bad |= (nbytes - LC3_RSI(AD_HEAD) - last_byte + 2 < 0) | (nbytes - last_byte - 1 < 0);
        }
        if (bad) {  // wave-uniform: the frame is concealed, as when the reference returns Err here
            if (lane == 0) col[AD_OK] = 0;
            return;
        }
    }
    // residual bits of the lane's non-zero lines: ranks rank_nz .. rank_nz + 7 lie in at most two mask words (n_res <= 480)
    unsigned long long rwin;
    {
        const int w0 = rank_nz >> 5;
        const uint32_t lo = W.resw[w0 < 15 ? w0 : 15], hi = W.resw[w0 + 1 < 15 ? w0 + 1 : 15];
        rwin = (((unsigned long long)hi << 32) | lo) >> (rank_nz & 31);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float v = (float)xi[j];
        {   // residual_spectrum::decode: the j-th non-zero line takes residual bit j while j < n_res
            const int nz = (int)((nzmask >> j) & 1u), take = nz && rank_nz < n_res;
            const int bit = (int)(rwin & 1ull);
            const float up = v > 0.0f ? 0.3125f : 0.1875f, dn = v > 0.0f ? 0.1875f : 0.3125f;
            const float v_adj = bit ? v + up : v - dn;
            v = take ? v_adj : v;
            rank_nz += nz;
            rwin >>= nz;
        }
        {
            const int fill = (int)((fillmask >> j) & 1u);
            const uint32_t lcg_n = (13849u + LC3_MUL24(lcg, 31821u)) & 0xFFFFu;
            lcg = fill ? lcg_n : lcg;
            v = fill ? (lcg_n < 0x8000u ? level : -level) : v;
        }
        v8[j] = v * gg;
    }
    // TNS :24-137: a frame with an active filter keeps its filter range as it is now (lc3_tns_lane_frame finishes it)
    const int nbands = bw < 3 ? 1 : 2, num_tns = LC3_RSI(SI_NUM_TNS);
    const int ord0 = (0 < nbands && 0 < num_tns) ? LC3_RSI(AD_ORD0) : 0;
    const int ord1 = (1 < nbands && 1 < num_tns) ? LC3_RSI(AD_ORD0 + 1) : 0;
    const int tns = ord0 > 0 || ord1 > 0;  // wave-uniform
    int t_lo, t_hi;
    lc3_tns_range(c, bw, t_lo, t_hi);
    if (tns) ((float *)col)[LC3_PLANE_LEV + lane] = g_band;  // all 64 words, so that the second pass may load them blindly
    // SNS :113-151: the band gain of each line
    if (have) {
        const uint32_t b03 = T.line_band[2 * lane], b47 = T.line_band[2 * lane + 1];
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int b = (int)(((j < 4 ? b03 : b47) >> (8 * (j & 3))) & 0xffu);
            const float shaped = v8[j] * W.sc[16 + (b < 64 ? b : 0)];
            o[j] = (tns && k0 + j >= t_lo && k0 + j < t_hi) ? v8[j] : shaped;
        }
        lc3_f4 o0, o1;
        o0.x = o[0]; o0.y = o[1]; o0.z = o[2]; o0.w = o[3];
        o1.x = o[4]; o1.y = o[5]; o1.z = o[6]; o1.w = o[7];
        lc3_f4 *dst = (lc3_f4 *)(col + LC3_PLANE_X);
        dst[2 * lane] = o0;
        dst[2 * lane + 1] = o1;
    }
    LC3_SYNC();
#undef LC3_RSI
}

// ------------------------------------------------------------------------------------------------------------------
// Second pass, ONE LANE PER FRAME: the TNS synthesis lattice (decoder/temporal_noise_shaping.rs:60-137) and the band gains
// (decoder/spectral_noise_shaping.rs:113-151) of the filter range of the frames lc3_recon_frame_direct left unfinished.  A wave runs
// 64 lattices per instruction; frames without an active filter idle (their lanes are masked), a wave without any returns at once.
//   t = x - rc[ord-1] * st[ord-1];  q = ord-2 .. 0: t -= rc[q] * st[q]; st[q+1] = rc[q] * t + st[q];  x = st[0] = t
// with the order switching from the first filter's to the second's at the second filter's first line, the lattice memory carrying over.
// ------------------------------------------------------------------------------------------------------------------
struct lc3_tns_lane_ctx {
    int32_t *col;              // this lane's plane column (valid lanes)
    float *gains;              // LDS, band b of this lane at gains[b * gstride]
    int gstride;
    const float *sin_tab;      // LDS, the 17 quantised reflection coefficients (lc3_tns_sin_dec_value)
    const uint32_t *line_band; // LDS, band of line k in byte k (lc3_line_band_word)
};
// four lines k0 .. k0+3 (k0 wave-uniform) of this lane's frame: lattice (lc3_tns_lattice4, lc3_dev_dec_parse.h), band gains, store
__device__ __forceinline__ void lc3_tns_group(const lc3_tns_lane_ctx &x, float *xs, int k0, int lo0, int lo1, int hi, int ord0, float (&rq)[8],
                                              const float (&rq1)[8], float (&st)[8], const lc3_f4 &in) {
    const uint32_t bands = x.line_band[k0 >> 2];
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int b = (int)((bands >> (8 * j)) & 0xffu);
        g[j] = x.gains[(b < 64 ? b : 0) * x.gstride];
    }
    float v[4] = {in.x, in.y, in.z, in.w};
    lc3_tns_lattice4(k0, lo0, lo1, hi, ord0, rq, rq1, st, v);
    if (k0 + 3 >= lo0 && k0 < hi) {
        lc3_f4 o;
        o.x = (k0 >= lo0 && k0 < hi) ? v[0] * g[0] : v[0];
        o.y = (k0 + 1 >= lo0 && k0 + 1 < hi) ? v[1] * g[1] : v[1];
        o.z = (k0 + 2 >= lo0 && k0 + 2 < hi) ? v[2] * g[2] : v[2];
        o.w = (k0 + 3 >= lo0 && k0 + 3 < hi) ? v[3] * g[3] : v[3];
        *(lc3_f4 *)(xs + k0) = o;
    }
}
template <class CC>
__device__ __forceinline__ void lc3_tns_lane_frame(const CC &c, const lc3_tns_lane_ctx &x, int valid) {
    struct q4 { int32_t v[4]; };
    // side information words 0..7 (bandwidth, number of filters) and 20..43 (orders, AD_OK, coefficient indices)
    int32_t siw[48];
    static_assert(SI_BW < 8 && SI_NUM_TNS < 8 && AD_ORD0 >= 20 && AD_RCI + 16 <= 44 && AD_OK >= 20, "side-information words of the TNS pass");
    {   // (every load of this function is unconditional -- x.col is a readable column on every lane: a load inside a branch is
        // waited for at the end of the branch, one memory round trip each)
        const lc3_i4 *s4 = (const lc3_i4 *)x.col;
#pragma unroll
        for (int u = 0; u < 11; u++) {
            if (u < 2 || u >= 5) {
                const q4 w = __builtin_bit_cast(q4, s4[u]);
#pragma unroll
                for (int j = 0; j < 4; j++) siw[4 * u + j] = w.v[j];
            }
        }
    }
    const int bw = siw[SI_BW] & 7, bwc = bw < 5 ? bw : 4;
    const int nbands = bwc < 3 ? 1 : 2, num_tns = siw[SI_NUM_TNS];
    const int ord0 = (0 < nbands && 0 < num_tns) ? siw[AD_ORD0] : 0;
    const int ord1 = (1 < nbands && 1 < num_tns) ? siw[AD_ORD0 + 1] : 0;
    const int active = valid && siw[AD_OK] && (ord0 > 0 || ord1 > 0);
    if (!LC3_WAVE_ANY(active)) return;
    // the frame's 64 band gains: 16 units of 16 bytes, to LDS band-major (conflict-free reads in the line loop)
    {
        const lc3_i4 *g4 = (const lc3_i4 *)(x.col + LC3_PLANE_LEV);
        q4 g[16];
#pragma unroll
        for (int u = 0; u < 16; u++) g[u] = __builtin_bit_cast(q4, g4[u]);  // (stale words on lanes without a filter: never used)
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int j = 0; j < 4; j++) x.gains[(4 * u + j) * x.gstride] = __builtin_bit_cast(float, g[u].v[j]);
    }
    // Reflection coefficients of the two filters, zero beyond a filter's order.  The lattice below runs all eight stages without
    // per-line selects: a stage with rc = 0 subtracts rc * st = +-0 from t, which leaves t as it is (t is never -0: the inputs are
    // converted integers, +-level and products of those with positive gains, and a difference of equal values is +0), and what such
    // stages write into the lattice memory beyond the order is discarded where it could be read: at the switch to the second filter the
    // memory beyond the first filter's order is reset to the zeros the reference still has there.
    float rq0[8], rq1[8], st[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int r0 = siw[AD_RCI + q], r1 = siw[AD_RCI + 8 + q];
        const float s0 = x.sin_tab[(unsigned)r0 < 17u ? r0 : 0], s1 = x.sin_tab[(unsigned)r1 < 17u ? r1 : 0];  // sin(step * (ri - 8)); SURVEY A12
        rq0[q] = (r0 != 0 && q < ord0) ? s0 : 0.0f;
        rq1[q] = (r1 != 0 && q < ord1) ? s1 : 0.0f;
        st[q] = 0.0f;
    }
    float *xs = (float *)(x.col + LC3_PLANE_X);
    // this lane's filter ranges: [lo0, lo1) first filter, [lo1, hi) second one (lo1 = hi: there is none); nothing on a lane without a filter
    int lo0 = c.n_ms_10 ? LC3C_TNSDEC10[bwc][0] : LC3C_TNSDEC75[bwc][0];
    int hi = c.n_ms_10 ? LC3C_TNSDEC10[bwc][nbands == 2 ? 3 : 1] : LC3C_TNSDEC75[bwc][nbands == 2 ? 3 : 1];
    int lo1 = nbands == 2 ? (c.n_ms_10 ? LC3C_TNSDEC10[bwc][2] : LC3C_TNSDEC75[bwc][2]) : hi;
    if (!active) lo0 = lo1 = hi = 0;
    const int k_first = (c.n_ms_10 ? 12 : 9) & ~3;                                      // every first filter starts at line 12 / 9
    const int k_end = LC3_LANEWAVE_MAX(hi);                                             // wave-uniform
    float rq[8];
#pragma unroll
    for (int q = 0; q < 8; q++) rq[q] = rq0[q];
    // Sixteen lines per half step, two buffers: the loads of one half run while the other half is worked on (a buffer that is refilled
    // in place needs no register moves: a move of a loaded value would wait for it).  Beyond the lane's range: any readable address.
    lc3_f4 A[4], B[4];
#define LC3_TNS_LOAD(buf, base)                                                                                 \
    _Pragma("unroll") for (int a = 0; a < 4; a++) buf[a] = *(const lc3_f4 *)(xs + ((base) + 4 * a < hi ? (base) + 4 * a : k_first))
#define LC3_TNS_WORK(buf, base)                                                                              \
    _Pragma("unroll") for (int a = 0; a < 4; a++) {                                                          \
        if ((base) + 4 * a < k_end) lc3_tns_group(x, xs, (base) + 4 * a, lo0, lo1, hi, ord0, rq, rq1, st, buf[a]); \
    }
    LC3_TNS_LOAD(A, k_first);
    for (int k0 = k_first; k0 < k_end; k0 += 32) {
        LC3_TNS_LOAD(B, k0 + 16);
        LC3_TNS_WORK(A, k0);
        LC3_TNS_LOAD(A, k0 + 32);
        LC3_TNS_WORK(B, k0 + 16);
    }
#undef LC3_TNS_LOAD
#undef LC3_TNS_WORK
}
