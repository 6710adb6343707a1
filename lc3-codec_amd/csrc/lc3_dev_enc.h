// LC3 batched encoder for MI355X -- device-side stages (one wavefront per stream).
// Mirrors EncoderChannel::encode (reference encoder/lc3_encoder.rs:63-112) stage by stage.
// See lc3_dev_common.h for the execution model and the bit-exactness contract.
#pragma once
#include "lc3_dev_common.h"
#include "lc3_dev_enc_pack.h"
#include "lc3_dev_enc_vq.h"

// ------------------------------------------------------------------------------------------
// Persistent per-stream encoder state as it lives in HBM between launches: the two LTPF sample rings
// and the MDCT history stay in HBM for good (the stages that use them stage them through scratch LDS, once per frame);
// the 16 scalar words ride along in the wave's LDS working set during a launch.
// ------------------------------------------------------------------------------------------
struct lc3_enc_scalars {
    // attack detector (encoder/attack_detector.rs:17-21)
    float att_energy_last, att_max_energy_last;
    int att_pos_last, att_ds_tm1, att_ds_tm2;
    // LTPF (encoder/long_term_post_filter.rs:32-41)
    int t_prev;
    float mem_pitch;
    int mem_ltpf_active;
    float mem_nc, mem_mem_nc, h50_m1, h50_m2;
    // quantiser (encoder/spectral_quantization.rs:56-61); nbits_spec_old stays 0 unless LC3_SPEC_NBITS_SPEC_OLD is set
    int reset_offset_old;
    float nbits_offset_old;
    int nbits_est_old;
    int nbits_spec_old;
    // the LTPF sample rings of the state blob are circular: logical element i lives at (head + i) mod length, a frame overwrites
    // only the oldest len12 / len6 slots instead of shifting the whole ring through HBM
    int ring12_head, ring6_head, pad_[2];
};
struct lc3_enc_state {
    float x12[384];          // LTPF 12.8 kHz ring (encoder/long_term_post_filter.rs:114,227); 10 ms uses all 384
    float x6[180];           // LTPF 6.4 kHz ring, 178 used (:116,228)
    int16_t hist[304];       // MDCT time-buffer history t[0 .. nf-z) (encoder/modified_dct.rs:126-138) = the last nf - z
                             // samples of the previous frame
    lc3_enc_scalars sc;
};
static_assert(sizeof(lc3_enc_scalars) == 80 && sizeof(lc3_enc_state) % 16 == 0 && offsetof(lc3_enc_state, sc) % 16 == 0,
              "encoder state blob: 16-byte units");

// LDS working set of one encoder wave: 8 KB (room for five workgroups of four streams per CU; the kernels run four,
// bound by registers).  Buffers are reused as their
// contents die: `t` holds the MDCT time buffer until the LTPF resampler has read it, then the 6.4 kHz ring (LTPF) and
// finally the quantised spectrum and the residual bits; fa/fb are the FFT buffers and every later stage's scratch.
struct __attribute__((aligned(16))) lc3_enc_lds {
    lc3_enc_scalars st;      // scalar part of the stream state
    float spec[LC3_MAX_NF];  // MDCT output -> SNS -> TNS spectrum (mdct_out / spec_lines)
    lc3_cpx fa[LC3_MAX_NF / 2];  // FFT input; afterwards scratch (must directly follow spec: symbol list spans both)
    lc3_cpx fb[LC3_MAX_NF / 2];  // FFT work buffer; afterwards scratch
    int16_t t[2 * LC3_MAX_NF];   // MDCT time buffer (ModDiscreteCosTrans::freq); later x6 | xq + residual bits
    float sm[32];            // small scratch (per-stage)
    int ism[64];
    int spec_flags;          // LC3_SPEC_* of the launch (0 = the reference's behaviour)
    int pad_;
    unsigned long long prof_last;  // diagnostic build: time of the previous stage stamp
#ifdef LC3_PROFILE
    unsigned long long prof_acc[32];  // diagnostic build: per-wave stage totals, flushed once per launch
#endif
};
#define LC3_ENC_DBG_EB 1472      // [64] band energies
#define LC3_ENC_DBG_ATTACK 1536  // [5] attack detector state: energy_last, max_energy_last, attack_pos_last, downsampled t-1, t-2
#define LC3_ENC_DBG_FLOATS 1600
#define LC3_EB(L) ((float *)(L).fa + 512)                    // [64] band energies: MDCT stage -> bandwidth, SNS
#define LC3_SCF(L) ((float *)(L).fa + 144)                   // [16] SNS target scale factors (lc3_enc_sns_front)
#define LC3_XQ(L) ((L).t)                                    // int16[512] quantised spectrum, zero from ne on (from the quantiser on)
#define LC3_RESW(L) ((uint32_t *)((uint8_t *)(L).t + 1024))  // uint32[13] residual bits, bit j of word j / 32 (behind the 512 padded entries of xq)

LC3_LDS_DECL(lc3_enc_lds, lc3_enc_wg)
static_assert(offsetof(lc3_enc_lds, fa) % 16 == 0 && offsetof(lc3_enc_lds, spec) % 16 == 0 && offsetof(lc3_enc_lds, t) % 16 == 0,
              "128-bit LDS reads need aligned buffers");
#ifndef LC3_PROFILE
static_assert(sizeof(lc3_enc_lds) <= 8192, "encoder working set: 8 KB per stream (the kernels run 16 waves per CU, register-bound)");
#endif

struct lc3_tns_res { int nbits_tns, lpc_weighting, num_tns_filters; int rc_order[2]; };  // rc_i / rc_q live in LDS
struct lc3_ltpf_res { int pitch_index, pitch_present, ltpf_active, nbits_ltpf; };
struct lc3_quant_res { int gg_ind, nbits_spec, nbits_lsb, nbits_trunc, lsb_mode, rate_flag, lastnz_trunc; float gg; };

// order global-memory traffic of the wave's lanes among themselves (state rings written by some lanes, read by others)
#ifndef LC3_HBM_FENCE
#define LC3_HBM_FENCE() LC3_SYNC()
#endif

__device__ __forceinline__ void lc3_enc_state_init(lc3_enc_lds &L, int lane, lc3_enc_state *g, int valid) {
    // fresh channel (Lc3Encoder::new: zeroed working buffers; attack_detector.rs:31-43;
    // long_term_post_filter.rs:74-90 t_prev = K_MIN)
    int *w = (int *)&L.st;
    if (lane < (int)(sizeof(lc3_enc_scalars) / 4)) w[lane] = 0;
    if (valid) {  // rings and history in HBM
        int *gw = (int *)g;
        for (int i = lane; i < (int)(offsetof(lc3_enc_state, sc) / 4); i += LC3_WAVE) gw[i] = 0;
    }
    LC3_SYNC();
    if (lane == 0) {
        L.st.att_pos_last = -1;
        L.st.t_prev = 17;
    }
    LC3_SYNC();
    LC3_HBM_FENCE();
}
__device__ __forceinline__ void lc3_enc_state_load(lc3_enc_lds &L, int lane, const lc3_enc_state *g) {
    lc3_wave_copy_in16(&L.st, &g->sc, (int)(sizeof(lc3_enc_scalars) / 16), lane);
    LC3_SYNC();
}
// scalars back to HBM; the MDCT history of the next launch = samples [z, nf) of the last frame of this one
template <class CC>
__device__ __forceinline__ void lc3_enc_state_store(const CC &c, lc3_enc_lds &L, int lane, lc3_enc_state *g,
                                                    const int16_t *last_frame, int stride = 1) {
    LC3_SYNC();
    lc3_wave_copy_out16(&g->sc, &L.st, (int)(sizeof(lc3_enc_scalars) / 16), lane);
    if (last_frame) {
        uint32_t *dst = (uint32_t *)g->hist;
        const int nw = (c.nf - c.z) / 2;  // <= 150 words
        uint32_t v[3];
        if (stride == 1) {
            LC3_HBM_CONST(uint32_t) src = (LC3_HBM_CONST(uint32_t))(last_frame + c.z);
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int i = lane + LC3_WAVE * u;
                v[u] = i < nw ? src[i] : 0u;
            }
        } else {  // interleaved input: sample j of the frame sits `stride` elements after sample j - 1
            LC3_HBM_CONST(uint16_t) src = (LC3_HBM_CONST(uint16_t))last_frame;
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int i = lane + LC3_WAVE * u;
                v[u] = i < nw ? (uint32_t)src[(size_t)(c.z + 2 * i) * (size_t)stride] |
                                    ((uint32_t)src[(size_t)(c.z + 2 * i + 1) * (size_t)stride] << 16)
                              : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int i = lane + LC3_WAVE * u;
            if (i < nw) dst[i] = v[u];
        }
    }
}

// ------------------------------------------------------------------------------------------
// E1-E6: MDCT analysis (encoder/modified_dct.rs:108-177)
// ------------------------------------------------------------------------------------------
// hist: the nf - z samples before this frame (previous frame's tail in the PCM input, or the state blob's copy for the
// first frame of a launch), 4-byte aligned; nullptr = silence (fresh stream)
// stride / hstride: distance in elements between consecutive samples of the frame / of the history (1 = planar; the channel
// count for interleaved PCM; the state blob's copy of the history is always planar)
LC3_CFG_TEMPLATE __device__ __noinline__ int lc3_enc_mdct(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, const int16_t *pcm,
                                         const int16_t *hist, int stride, int hstride) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    const int nf = c.nf, z = c.z, h = nf / 2, mid = 3 * h;
    const uint32_t *w = lc3_window_bits(c);
    const uint16_t *ifs = lc3_band_index(c);
    // update_time_buffer :126-138: t[0..nf-z) <- history, t[nf-z..2nf-z) <- new frame, tail stays 0.
    // The frame is fetched from HBM as 32-bit words (two samples per lane per load, coalesced).
    float wv[4][4];
    {
        LC3_HBM_CONST(uint32_t) p32 = (LC3_HBM_CONST(uint32_t))pcm;
        LC3_HBM_CONST(uint32_t) h32 = (LC3_HBM_CONST(uint32_t))hist;
        uint32_t v[4], hv[3];  // nf / 2 <= 240 and (nf - z) / 2 <= 150 words, all in flight together
#pragma unroll
        for (int r = 0; r < 4; r++) {  // the window coefficients of this lane's fold outputs (h <= 240): same batch of loads
            const int k = lane + LC3_WAVE * r, in = k < h;
            wv[r][0] = in ? lc3_f(w, mid - 1 - k) : 0.0f;
            wv[r][1] = in ? lc3_f(w, mid + k) : 0.0f;
            wv[r][2] = in ? lc3_f(w, k) : 0.0f;
            wv[r][3] = in ? lc3_f(w, nf - 1 - k) : 0.0f;
        }
        if (stride == 1) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = lane + LC3_WAVE * u;
                v[u] = i < nf / 2 ? p32[i] : 0u;
            }
        } else {  // interleaved PCM (examples/encode.rs:95-102 de-interleaves on the host): two strided 16-bit loads per word
            LC3_HBM_CONST(uint16_t) p16 = (LC3_HBM_CONST(uint16_t))pcm;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = lane + LC3_WAVE * u;
                v[u] = i < nf / 2 ? (uint32_t)p16[(size_t)(2 * i) * (size_t)stride] | ((uint32_t)p16[(size_t)(2 * i + 1) * (size_t)stride] << 16) : 0u;
            }
        }
        if (hstride == 1) {
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int i = lane + LC3_WAVE * u;
                hv[u] = (hist && i < (nf - z) / 2) ? h32[i] : 0u;
            }
        } else {
            LC3_HBM_CONST(uint16_t) h16 = (LC3_HBM_CONST(uint16_t))hist;
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int i = lane + LC3_WAVE * u;
                hv[u] = (hist && i < (nf - z) / 2)
                            ? (uint32_t)h16[(size_t)(2 * i) * (size_t)hstride] | ((uint32_t)h16[(size_t)(2 * i + 1) * (size_t)hstride] << 16)
                            : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int i = lane + LC3_WAVE * u;
            if (i < (nf - z) / 2) {
                L.t[2 * i] = (int16_t)(hv[u] & 0xffffu);
                L.t[2 * i + 1] = (int16_t)(hv[u] >> 16);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = lane + LC3_WAVE * u;
            if (i < nf / 2) {
                L.t[nf - z + 2 * i] = (int16_t)(v[u] & 0xffffu);
                L.t[nf - z + 2 * i + 1] = (int16_t)(v[u] >> 16);
            }
        }
    }
    for (int i = 2 * nf - z + lane; i < 2 * nf; i += LC3_WAVE) L.t[i] = 0;
    LC3_SYNC();
    // apply_mdct :73-105 (window / fold); the window coefficients were requested with the frame (wv)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int k = lane + LC3_WAVE * r;
        if (k < h) {
            L.spec[k] = -((float)L.t[mid - 1 - k] * wv[r][0]) - ((float)L.t[mid + k] * wv[r][1]);
            L.spec[h + k] = ((float)L.t[k] * wv[r][2]) - ((float)L.t[nf - 1 - k] * wv[r][3]);
        }
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 22);
    lc3_dct4_wave(c, lane, L.spec, L.fa, L.fb);
    LC3_STAMP(L, lane, 23);
    {
        const float gain = 1.0f / lc3_sqrtf(2.0f * (float)nf);
        for (int k = lane; k < nf; k += LC3_WAVE) L.spec[k] *= gain;
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 20);
    // apply_energy_estimation :140-152: E_b = sum over the band's lines of (X_k * X_k / width), the division inside the
    // sum.  The terms are independent: one line per lane (seven rounds instead of up to 25 divisions in a row
    // on the lanes of the widest bands); then one lane per band adds its terms in the reference's order.
    {
        float *e = (float *)L.fa;  // fa is free after the transform; the band energies live at fa + 512 floats
        for (int k = lane; k < c.ne; k += LC3_WAVE) {
            const float x = L.spec[k];
            e[k] = x * x / LC3_LINE_WIDTH(c, k);
        }
        LC3_SYNC();
        LC3_STAMP(L, lane, 21);
        for (int b = lane; b < c.nb; b += LC3_WAVE) {
            const int from = ifs[b], to = ifs[b + 1];
            float acc = 0.0f;
            for (int k = from; k < to; k += 4) {  // four terms per LDS round trip
                float x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) x[u] = k + u < to ? e[k + u] : 0.0f;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (k + u < to) acc += x[u];
            }
            LC3_EB(L)[b] = acc;
        }
    }
    LC3_SYNC();
    // is_near_nyquist :154-177
    int nn = 0;
    if (c.fs <= 32000) {
        if (lane == 0) {
            int nn_idx = c.n_ms_10 ? c.nb - 2 : c.nb - 4;
            float lower = 0.0f, upper = 0.0f;
            for (int b = 0; b < c.nb; b++) {
                if (b < nn_idx) lower += LC3_EB(L)[b];
                else upper += LC3_EB(L)[b];
            }
            nn = upper > 30.0f * lower;
        }
        nn = lc3_wave_bcast0_i32(nn, lane);  // lane 0's result in a scalar register: no LDS round trip
    }
    return nn;
}

// ------------------------------------------------------------------------------------------
// E7: bandwidth detector (encoder/bandwidth_detector.rs:64-127), lane 0
// ------------------------------------------------------------------------------------------
LC3_CFG_TEMPLATE __device__ __noinline__ int lc3_enc_bandwidth(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, int *nbits_bw) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    *nbits_bw = LC3C_NBITS_BW[c.fs_ind];
    if (c.fs_ind == 0) return 0;  // :66-71 (the reference cannot construct an 8 kHz encoder)
    int result0 = 0;
    if (lane == 0) {
        const int fsi = c.fs_ind;
        int bw = 0;
        for (int k = fsi - 1; k >= 0; k--) {
            int start = c.n_ms_10 ? LC3C_BW_START10[fsi - 1][k] : LC3C_BW_START75[fsi - 1][k];
            int stop = c.n_ms_10 ? LC3C_BW_STOP10[fsi - 1][k] : LC3C_BW_STOP75[fsi - 1][k];
            float width = (float)(stop + 1 - start), quiet = 0.0f;
            for (int n = start; n <= stop; n++) quiet += LC3_EB(L)[n] / width;
            if (quiet >= (float)LC3C_BW_TQ[k]) {
                bw = k + 1;
                break;
            }
        }
        int result = bw;
        if (fsi != bw) {
            float cutoff_max = 0.0f;
            int l_bw = c.n_ms_10 ? LC3C_BW_L10[bw] : LC3C_BW_L75[bw];
            int start_bw = c.n_ms_10 ? LC3C_BW_START10[fsi - 1][bw] : LC3C_BW_START75[fsi - 1][bw];
            for (int n = start_bw + 1 - l_bw; n < start_bw; n++) {
                float cutoff = LC3_EB(L)[n - l_bw] / LC3_EB(L)[n];  // raw ratio, no dB
                if (L.spec_flags & LC3_SPEC_BW_CUTOFF_DB) cutoff = 10.0f * lc3_log10f(1.1920929e-7f + cutoff);
                cutoff_max = lc3_maxf(cutoff, cutoff_max);
            }
            result = cutoff_max > (float)LC3C_BW_TC[bw] ? bw : fsi;
        }
        result0 = result;
    }
    return lc3_wave_bcast0_i32(result0, lane);
}

// ------------------------------------------------------------------------------------------
// E8: attack detector (encoder/attack_detector.rs:45-128)
// ------------------------------------------------------------------------------------------
LC3_CFG_TEMPLATE __device__ __noinline__ int lc3_enc_attack(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, int nbytes) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    const int num_ds = c.n_ms_10 ? 160 : 120, num_blocks = c.n_ms_10 ? 4 : 3, limit = c.n_ms_10 ? 2 : 1;
    int active;
    if (c.fs < 32000) active = 0;
    else if (!c.n_ms_10)
        active = (c.fs == 32000 && nbytes >= 61 && nbytes < 150) || (c.fs >= 44100 && nbytes >= 75 && nbytes < 150);
    else active = (c.fs == 32000 && nbytes > 80) || (c.fs >= 41000 && nbytes >= 100);
    if (!active) {
        if (lane == 0) {
            L.st.att_energy_last = 0.0f;
            L.st.att_max_energy_last = 0.0f;
            L.st.att_pos_last = -1;
        }
        LC3_SYNC();
        return 0;
    }
    const int block_len = c.nf / num_ds;
    const int16_t *x_s = L.t + (c.nf - c.z);  // the current frame inside the time buffer
    int *ds = (int *)L.fa;                    // 160 ints
    float *hp = (float *)L.fa + 160;          // 160 floats: the high-passed samples squared (the band energies sit at fa + 512: LC3_EB)
    float *en = L.sm;                         // 4 block energies
    for (int n = lane; n < num_ds; n += LC3_WAVE) {
        int acc = 0;
        for (int j = 0; j < block_len; j++) acc += (int)x_s[block_len * n + j];
        ds[n] = acc;
    }
    LC3_SYNC();
    for (int n = lane; n < num_ds; n += LC3_WAVE) {  // filter :118-128
        float x0 = (float)ds[n];
        float x1 = n >= 1 ? (float)ds[n - 1] : (float)L.st.att_ds_tm1;
        float x2 = n >= 2 ? (float)ds[n - 2] : (n == 1 ? (float)L.st.att_ds_tm1 : (float)L.st.att_ds_tm2);
        const float v = 0.375f * x0 - 0.5f * x1 + 0.125f * x2;
        hp[n] = v * v;  // only the squares are needed; the block lanes below just add them
    }
    LC3_SYNC();
    if (lane < num_blocks) {  // block energies, 40 terms each in order
        en[lane] = lc3_sum_seq(hp + 40 * lane, 40, 0.0f);
    }
    LC3_SYNC();
    int attack0 = 0;
    if (lane == 0) {
        int attack_position = -1;
        float e_last = L.st.att_energy_last, m_last = L.st.att_max_energy_last;
        for (int n = 0; n < num_blocks; n++) {
            float energy = en[n];
            float max_energy = lc3_maxf(0.25f * m_last, e_last);
            if (energy > 8.5f * max_energy) attack_position = n;
            e_last = energy;
            m_last = max_energy;
        }
        attack0 = attack_position >= 0 || L.st.att_pos_last >= limit;
        L.st.att_energy_last = e_last;
        L.st.att_max_energy_last = m_last;
        L.st.att_pos_last = attack_position;
        L.st.att_ds_tm1 = ds[num_ds - 1];
        L.st.att_ds_tm2 = ds[num_ds - 2];
    }
    LC3_SYNC();
    return lc3_wave_bcast0_i32(attack0, lane);
}

// ------------------------------------------------------------------------------------------
// E9/E10: spectral noise shaping (encoder/spectral_noise_shaping.rs:203-648)
// ------------------------------------------------------------------------------------------
// E9 front half (encoder/spectral_noise_shaping.rs:75-161, 203-233): band energies -> 16 target scale factors, left in
// LDS at S[144..160) (sSCF) for the caller to hand to the vector quantiser (lc3_dev_enc_vq.h).
// scratch map inside L.fa (floats): sE[64] smoothed/log energies, sP[64] padded, sDS[16], sSCF[16]
LC3_CFG_TEMPLATE __device__ __noinline__ void lc3_enc_sns_front(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, int attack) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    float *S = (float *)L.fa;
    float *sE = S, *sP = S + 64, *sDS = S + 128, *sSCF = S + 144;
    const int diff = 64 - c.nb;

    // padding :75-90
    if (diff > 0) {
        if (lane == 0) {
            for (int b = 0; b < diff; b++) {
                sP[2 * b] = LC3_EB(L)[b];
                sP[2 * b + 1] = LC3_EB(L)[b];
            }
            for (int b = 0; b < c.nb && 2 * diff + b < 64; b++) sP[2 * diff + b] = LC3_EB(L)[diff + b];
        }
    } else {
        sP[lane] = LC3_EB(L)[lane];
    }
    LC3_SYNC();
    // smoothing :92-98, pre-emphasis :214-219 -- one lane per band
    {
        const int b = lane;
        float v;
        if (b == 0) v = 0.75f * sP[0] + 0.25f * sP[1];
        else if (b == 63) v = 0.25f * sP[62] + 0.75f * sP[63];
        else v = 0.25f * sP[b - 1] + 0.5f * sP[b] + 0.25f * sP[b + 1];
        v *= LC3_POW10_TILT(c.fs_ind, b);  // 10^(b * (g_tilt / 630)), the exponent formed as (float)b * ((float)g_tilt / 630.0f)
        sE[b] = v;
    }
    LC3_SYNC();
    // noise floor :221-228 -- 64-term sequential sum on lane 0
    if (lane == 0) {
        float total = lc3_sum_seq(sE, 64, 0.0f);
        total = (total / 64.0f) * lc3_powi(10.0f, -4);
        L.sm[0] = lc3_maxf(lc3_powi(2.0f, -32), total);
    }
    LC3_SYNC();
    {
        float v = lc3_maxf(sE[lane], L.sm[0]);
        sE[lane] = lc3_log2f(1.1920929e-7f + v) / 2.0f;  // :230-233
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 26);
    // band energy grouping :100-124 -- one lane per scale
    if (lane < 16) {
        const float W[6] = {1.0f / 12.0f, 2.0f / 12.0f, 3.0f / 12.0f, 3.0f / 12.0f, 2.0f / 12.0f, 1.0f / 12.0f};
        float d;
        if (lane == 0) {
            d = W[0] * sE[0];
            for (int k = 1; k < 6; k++) d += W[k] * sE[k - 1];
        } else if (lane == 15) {
            d = W[5] * sE[63];
            for (int k = 0; k < 5; k++) d += W[k] * sE[60 + k - 1];
        } else {
            d = 0.0f;
            for (int k = 0; k < 6; k++) d += W[k] * sE[4 * lane - 1 + k];
        }
        sDS[lane] = d;
    }
    LC3_SYNC();
    // mean removal :126-132, attack handling :134-161, and the whole vector quantiser on lane 0
    if (lane == 0) {
        float total = 0.0f;
        for (int n = 0; n < 16; n++) total += sDS[n];
        float avg = total / 16.0f;
        for (int n = 0; n < 16; n++) sDS[n] = 0.85f * (sDS[n] - avg);
        if (attack) {
            sSCF[0] = (sDS[0] + sDS[1] + sDS[2]) / 3.0f;
            sSCF[1] = (sDS[0] + sDS[1] + sDS[2] + sDS[3]) / 4.0f;
            for (int n = 2; n < 14; n++) {
                float wt = 0.0f;
                for (int k = n - 2; k < n + 3; k++) wt += sDS[k];
                sSCF[n] = wt / 5.0f;
            }
            sSCF[14] = (sDS[12] + sDS[13] + sDS[14] + sDS[15]) / 4.0f;
            sSCF[15] = (sDS[13] + sDS[14] + sDS[15]) / 3.0f;
            total = 0.0f;
            for (int n = 0; n < 16; n++) total += sSCF[n];
            avg = total / 16.0f;
            const float att = c.n_ms_10 ? 0.5f : 0.3f;
            for (int n = 0; n < 16; n++) sSCF[n] = att * (sSCF[n] - avg);
        } else {
            for (int n = 0; n < 16; n++) sSCF[n] = sDS[n];
        }
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 27);
}

// ------------------------------------------------------------------------------------------
// E11: temporal noise shaping (encoder/temporal_noise_shaping.rs:40-349), in three parts:
//   lc3_enc_tns_acf    the 54 partial autocorrelations of a frame and their quotients (needs the shaped spectrum in LDS)      -> a slot
//   lc3_enc_tns_lev    Levinson-Durbin and the LPC -> reflection conversion of up to LC3_TNS_CHUNK frames' filters side by side   slots -> slots
//   lc3_enc_tns_apply  coefficient quantisation, bit budget and the MA lattice over the spectrum of one frame                     <- its slot
// The analysis of a frame does not depend on the frames before it, and the recursions are ~350 instructions on TWO lanes (one per
// filter) -- 12 % of the back half's instructions (profiles/r05_knockout_back.txt).  The back half therefore analyses the frames of a
// launch in chunks (lc3_encode_back_stream): autocorrelations of every frame of the chunk first, ONE pass of the recursions for all of
// them (eight lanes for four frames), then frame by frame everything that carries state.  The price is that a frame's spectrum is
// picked up twice.  Gathering the recursions over the workgroup's four STREAMS instead was tried first and lost to its two workgroup
// barriers per frame (profiles/r05_tns_gather_ab.txt); frames of one stream need none.
// rc_i[16] -> L.ism[16..32), rc_q[16] -> L.sm[16..32)
// ------------------------------------------------------------------------------------------
#define LC3_TNS_CHUNK 4
struct lc3_tns_slot {
    float q[54];   // [2][9][3] partial autocorrelations, each divided by its sub-block's energy
    float es[6];   // [2][3] sub-block energies
    int p_bw, near_nyquist, on[2];
    float rc[16];  // reflection coefficients as the recursions leave them (before quantisation)
};
static_assert(sizeof(lc3_tns_slot) * LC3_TNS_CHUNK <= sizeof(((lc3_enc_lds *)0)->fb), "the slots live in fb, which the back half does not use otherwise");
#define LC3_TNS_SLOT(L, u) (((lc3_tns_slot *)(L).fb)[(u)])

LC3_CFG_TEMPLATE __device__ __noinline__ void lc3_enc_tns_acf(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, int p_bw, int near_nyquist,
                                              int slot) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    const lc3_tns_params &tp = c.n_ms_10 ? LC3C_TNS10[p_bw] : LC3C_TNS75[p_bw];
    lc3_tns_slot &T = LC3_TNS_SLOT(L, slot);
    float *sAC = T.q, *sES = T.es;
    float *x = L.spec;
    const int ne = c.ne;
    // compute_normalized_autocorrelation :80-115 -- one lane per (filter, lag, sub-block) partial sum, every sum in the
    // reference's order.  The sub-block energy e_s (:88-93) is the lag-0 sum of its sub-block -- the same products added in the
    // same order -- so the lag-0 lanes supply it.  A lane's sum has 50..70 terms: blocks of eight, the last one masked.
    // Which lane takes which sum is chosen for the LDS banks (every operand is a 4-byte read at its own address): the nine lags of a
    // sub-block on nine neighbouring lanes (consecutive addresses), three sub-blocks per half-wave, and the two sub-blocks of the upper
    // filter that start a multiple of 32 lines apart (160 and 320 at 48 kHz) in different half-waves.  Modelled for the full-band
    // layout: 48 instead of 72 LDS cycles per eight terms (32 if nothing collided).
    LC3_ENC_REPEAT(16) {
    if ((lane & 31) < 27) {
        const int j = (lane & 31) / 9, k = (lane & 31) - 9 * j;
        const int blk = lane < 32 ? (j == 2 ? 3 : j) : (j == 0 ? 2 : j + 3);  // f * 3 + s: {0, 1, 3} | {2, 4, 5}
        const int f = blk >= 3, s = blk - 3 * f;
        float ac = 0.0f;
        if (f < tp.num) {
            const int start = tp.sub_start[f][s], stop = tp.sub_stop[f][s], k_from = start + k;
            if (k_from < ne && k_from < stop) {
                const float *pa = x + start, *pb = x + k_from;  // reads run at most seven lines past `stop` (<= 400 + 7 < nf)
                const int n = stop - k_from;
                for (int i = 0; i < n; i += 8) {
                    float xa[8], xb[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        xa[u] = pa[i + u];
                        xb[u] = pb[i + u];
                    }
                    if (i + 8 <= n) {
#pragma unroll
                        for (int u = 0; u < 8; u++) ac += xa[u] * xb[u];
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const float p = xa[u] * xb[u];
                            ac = i + u < n ? ac + p : ac;
                        }
                    }
                }
            }
        }
        sAC[f * 27 + k * 3 + s] = ac;
        if (k == 0) sES[f * 3 + s] = ac;
    }
    if (lane == 0) {
        T.p_bw = p_bw;
        T.near_nyquist = near_nyquist;
    }
    LC3_SYNC();
    // the 54 quotients ac_s(k) / e_s (:97-104) on the lanes that hold the partial sums: the Levinson lanes below only add
    // three of them per lag (0 + q0 + q1 + q2 in the reference's order)
    if (lane < 54) {
        const int f = lane / 27, r = lane - 27 * f, s3 = r - 3 * (r / 3);
        sAC[lane] = sAC[lane] / sES[f * 3 + s3];
    }
    LC3_SYNC();
    }
    LC3_STAMP(L, lane, 28);
}

// Levinson-Durbin :204-232 and the LPC -> reflection conversion :234-265 for the filters of `n` frames (slots 0 .. n-1): lane 2 u + f runs
// filter f of slot u, the order-8 recursions fully unrolled on register arrays.  What a lane needs is in its slot; its verdict (the
// prediction gain passes :219) and the coefficients go back there.
LC3_CFG_TEMPLATE __device__ __noinline__ void lc3_enc_tns_lev(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, int nbits, int n) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    const int lpc_weighting = c.n_ms_10 ? (nbits < 480) : (nbits < 360);
    LC3_ENC_REPEAT(32)
    if (lane < 2 * n) {
        const int f = lane & 1;
        lc3_tns_slot &T = LC3_TNS_SLOT(L, lane >> 1);
        const float *sAC = T.q, *sES = T.es;
        float *rc_q = T.rc;
        const int num_g = (c.n_ms_10 ? LC3C_TNS10[T.p_bw] : LC3C_TNS75[T.p_bw]).num;
        int on = 0;  // the filter's prediction gain passes :219
        if (f < num_g) {
        float r[9];
        const float e_prod = (1.0f * sES[f * 3] * sES[f * 3 + 1]) * sES[f * 3 + 2];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const float r0 = k == 0 ? 3.0f : 0.0f;
            const float rk = ((0.0f + sAC[f * 27 + k * 3]) + sAC[f * 27 + k * 3 + 1]) + sAC[f * 27 + k * 3 + 2];
            r[k] = (e_prod == 0.0f ? r0 : rk) * LC3C_TNS_LAGW[k];
        }
        float a[9], al[9];
#pragma unroll
        for (int n_ = 0; n_ < 9; n_++) a[n_] = 0.0f;
        float e = r[0];
        a[0] = 1.0f;
#pragma unroll
        for (int k = 1; k < 9; k++) {
#pragma unroll
            for (int n_ = 0; n_ < 9; n_++) al[n_] = a[n_];
            float rc = 0.0f;
#pragma unroll
            for (int n_ = 0; n_ < k; n_++) rc -= al[n_] * r[k - n_];
            if (e != 0.0f) rc /= e;
            a[0] = 1.0f;
#pragma unroll
            for (int n_ = 1; n_ < k; n_++) a[n_] = al[n_] + rc * al[k - n_];
            a[k] = rc;
            e *= 1.0f - rc * rc;
        }
        const float pred_gain = e == 0.0f ? r[0] : r[0] / e;
        if (pred_gain > 1.5f && !T.near_nyquist) {
            on = 1;
            float gamma = 1.0f;
            if (lpc_weighting > 0 && pred_gain < 2.0f)
                gamma -= (1.0f - 0.85f) * (2.0f - pred_gain) / (2.0f - 1.5f);
#pragma unroll
            for (int k = 0; k < 9; k++) a[k] *= lc3_powi(gamma, k);
#pragma unroll
            for (int k = 8; k >= 1; k--) {
                const float rck = a[k];
                rc_q[f * 8 + k - 1] = rck;
                const float ee = 1.0f - rck * rck;
#pragma unroll
                for (int n_ = 1; n_ < k; n_++) al[n_] = (a[n_] - rck * a[k - n_]) / ee;
#pragma unroll
                for (int n_ = 1; n_ < k; n_++) a[n_] = al[n_];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) rc_q[f * 8 + k] = 0.0f;
        }
        } else {  // a filter this bandwidth does not have
#pragma unroll
            for (int k = 0; k < 8; k++) rc_q[f * 8 + k] = 0.0f;
        }
        T.on[f] = on;
    }
    LC3_SYNC();
}

LC3_CFG_TEMPLATE __device__ __noinline__ lc3_tns_res lc3_enc_tns_apply(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, int p_bw, int nbits,
                                                        int slot) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    const lc3_tns_params &tp = c.n_ms_10 ? LC3C_TNS10[p_bw] : LC3C_TNS75[p_bw];
    // at 10 ms / bandwidth index 2 the reference filters lines 12..200 although its sub-blocks (and the decoder) run to
    // 240; LC3_SPEC_TNS_SSWB_STOP restores 240
    const int sswb_stop = (L.spec_flags & LC3_SPEC_TNS_SSWB_STOP) && c.n_ms_10 && p_bw == 2;
    const lc3_tns_slot &T = LC3_TNS_SLOT(L, slot);
    int *rc_i = L.ism + 16;
    float *rc_q = L.sm + 16;
    float *x = L.spec;
    lc3_tns_res res;
    res.num_tns_filters = tp.num;
    res.lpc_weighting = c.n_ms_10 ? (nbits < 480) : (nbits < 360);
    if (lane < 16) rc_q[lane] = T.rc[lane];
    const int tns_on = lane < 2 ? T.on[lane] : 0;  // lanes 0 / 1: the frame's two filters
    LC3_SYNC();
    LC3_STAMP(L, lane, 29);
    // apply_quantization :267-292 -- one lane per coefficient; the orders (:275-291: the last index that is not 8, per filter) and
    // the bit budget (calc_bit_budget :294-311: integer sums, any order is exact) are read off the same 16 lanes by ballot and
    // wave sums instead of a walk on lane 0 with one table fetch per coefficient
    if (lc3_wave_ballot(tns_on, lane) == 0ull) {
        // no filter of the frame is active (nine frames in ten): every coefficient is 0.0 -> index 8, order 0, and each
        // filter costs ceil(2048 / 2048) = 1 bit (:294-311)
        if (lane < 16) {
            rc_i[lane] = 8;
            rc_q[lane] = 0.0f;
        }
        res.rc_order[0] = 0;
        res.rc_order[1] = 0;
        res.nbits_tns = tp.num;
        LC3_SYNC();
        LC3_STAMP(L, lane, 30);
        return res;
    }
    int ri_lane = 8;
    if (lane < 16) {
        const float step = (float)3.14159265358979323846 / 17.0f;  // PI as f32 / 17.0 :268
        float rq_lane = 0.0f;
        if ((lane >> 3) < tp.num) {
            const float q = lc3_asinf(rc_q[lane]) / step;
            ri_lane = (q >= 0.0f ? lc3_f2i8(q + 0.5f) : lc3_f2i8(-(-q + 0.5f))) + 8;
            rq_lane = LC3_TNS_SIN_ENC(ri_lane);  // sin(step * ((float)ri - 8.0f))
        }
        rc_i[lane] = ri_lane;
        rc_q[lane] = rq_lane;
    }
    {
        const unsigned long long live = lc3_wave_ballot(lane < 16 && ri_lane != 8, lane);
        const uint32_t m0 = (uint32_t)(live & 0xffull), m1 = (uint32_t)((live >> 8) & 0xffull);
        const int o0 = (0 < tp.num && m0 != 0u) ? 32 - __builtin_clz(m0) : 0;  // 1 + index of the last coefficient that is not 8
        const int o1 = (1 < tp.num && m1 != 0u) ? 32 - __builtin_clz(m1) : 0;
        const int ord_f = lane < 8 ? o0 : o1;
        uint32_t cb = 0;
        if (lane < 16 && (lane & 7) < ord_f) {
            const int ric = ri_lane < 0 ? 0 : (ri_lane > 16 ? 16 : ri_lane);
            cb = LC3T_AC_TNS_COEF_BITS[lane & 7][ric];
        }
        const uint32_t cb0 = lc3_wave_sum_u32(lane < 8 ? cb : 0u, lane), cb1 = lc3_wave_sum_u32(lane >= 8 ? cb : 0u, lane);
        int nbits_tns = 0;
        if (0 < tp.num) {
            const int ob = o0 != 0 ? LC3T_AC_TNS_ORDER_BITS[res.lpc_weighting][o0 - 1] : 0;
            nbits_tns += (int)lc3_ceilf((2048.0f + (float)ob + (float)(int)cb0) / 2048.0f);
        }
        if (1 < tp.num) {
            const int ob = o1 != 0 ? LC3T_AC_TNS_ORDER_BITS[res.lpc_weighting][o1 - 1] : 0;
            nbits_tns += (int)lc3_ceilf((2048.0f + (float)ob + (float)(int)cb1) / 2048.0f);
        }
        res.rc_order[0] = o0;
        res.rc_order[1] = o1;
        res.nbits_tns = nbits_tns;
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 30);
    // apply_filtering :313-340.  The reference walks the samples serially through an order-8 MA lattice
    //   B_{k+1}[n] = rc_k * F_k[n] + B_k[n-1],   F_{k+1}[n] = F_k[n] + rc_k * B_k[n-1],   F_0 = B_0 = x,
    // where st[k] holds B_k[n-1].  A stage only needs the previous stage at n and n-1, so each stage is computed
    // for all samples at once (one lane per sample, identical f32 operations per element => bit-exact), 8 stages
    // instead of ~200 x 8 dependent steps.  The lattice state is shared across the two filters exactly as in the
    // reference (st[k] = B_k at the last sample of the previous filter, untouched for k >= its order).
    // Here the stages run on registers: lane l holds samples l, l + 64, ... of the filter's range, B_k[n-1] comes from the
    // neighbouring lane (one DPP move; lane 0 takes the previous round's lane 63, or st[k] at the first sample).
    {
        float stv[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // the lattice state, the same on every lane
        for (int f = 0; f < tp.num; f++) {
            const int order = res.rc_order[f];
            if (order == 0) continue;
            const int start = tp.start[f], len = (sswb_stop && f == 0 ? 240 : tp.stop[f]) - tp.start[f];  // len <= 256
            const int lastl = (len - 1) & 63, lastj = (len - 1) >> 6;
            float fv[4], bv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = lane + LC3_WAVE * j;
                fv[j] = bv[j] = i < len ? x[start + i] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k < order) {
                    const float rc = rc_q[f * 8 + k];
                    // B_k at the filter's last sample (before this stage replaces it) -> next st[k]
                    const float b_last = lc3_wave_read_f32(lastj == 0 ? bv[0] : (lastj == 1 ? bv[1] : (lastj == 2 ? bv[2] : bv[3])), lastl, lane);
                    float carry = stv[k];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int i = lane + LC3_WAVE * j;
                        const float from_left = lc3_wave_shr1_f32(bv[j], lane);
                        const float bprev = lane > 0 ? from_left : carry;
                        carry = lc3_wave_read_f32(bv[j], 63, lane);  // B_k[64 j + 63] for the next round's lane 0
                        const float nb = rc * fv[j] + bprev;         // st_tmp = rcq * t + st
                        const float nfv = fv[j] + rc * bprev;        // t += rcq * st
                        bv[j] = i < len ? nb : bv[j];
                        fv[j] = i < len ? nfv : fv[j];
                    }
                    stv[k] = b_last;
                }
            }
            LC3_SYNC();
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = lane + LC3_WAVE * j;
                if (i < len) x[start + i] = fv[j];
            }
            LC3_SYNC();
        }
    }
    LC3_SYNC();
    return res;
}

// ------------------------------------------------------------------------------------------
// E12-E16: long-term post-filter analysis (encoder/long_term_post_filter.rs:139-469)
// ------------------------------------------------------------------------------------------
#define LC3_NMEM 232
#define LC3_KMIN 17
#define LC3_KMAX 114

// Wave-parallel form of the reference's first-maximum scan  `best = a[first]; idx = first; for i: if (a[i] > best) { best = a[i]; idx = i; }`
// over elements held in registers: lane l owns elements l (v0) and l + 64 (v1); in0 / in1 mark the ones inside the scan.  Without
// NaNs the scan returns the lowest index that holds the maximum (-0.0 and +0.0 compare equal): the floats are mapped to integers
// of the same order, the maximum is an integer wave reduction and a ballot finds its first holder.  `any_nan` (wave-uniform) tells
// the caller that a NaN is inside the scan, where the sequential form must be used instead.  Returns -1 for an empty scan.
__device__ __forceinline__ int lc3_float_order_key(float v) {
    const uint32_t b = lc3_bits(v + 0.0f);  // -0.0 + 0.0 = +0.0
    return (int)(b ^ ((uint32_t)((int32_t)b >> 31) & 0x7fffffffu));
}
// the same scan over elements with explicit indices: lane holds (v0, index i0) and (v1, index i1); returns the lowest index that
// holds the maximum, -1 for an empty scan
__device__ __forceinline__ int lc3_wave_argmax_first2(float v0, int i0, int in0, float v1, int i1, int in1, int lane, int *any_nan) {
    const int lowest = -2147483647 - 1, big = 0x3fffffff;
    const int k0 = in0 ? lc3_float_order_key(v0) : lowest, k1 = in1 ? lc3_float_order_key(v1) : lowest;
    *any_nan = lc3_wave_ballot((in0 && v0 != v0) || (in1 && v1 != v1), lane) != 0ull;
    const int m = lc3_wave_max_i32(k0 > k1 ? k0 : k1, lane);
    const int c0 = in0 && k0 == m ? i0 : big, c1 = in1 && k1 == m ? i1 : big;
    const int best = -lc3_wave_max_i32(-(c0 < c1 ? c0 : c1), lane);
    return best == big ? -1 : best;
}
__device__ __forceinline__ int lc3_wave_argmax_first(float v0, int in0, float v1, int in1, int lane, int *any_nan) {
    const int lowest = -2147483647 - 1;
    const int k0 = in0 ? lc3_float_order_key(v0) : lowest, k1 = in1 ? lc3_float_order_key(v1) : lowest;
    *any_nan = lc3_wave_ballot((in0 && v0 != v0) || (in1 && v1 != v1), lane) != 0ull;
    const int m = lc3_wave_max_i32(k0 > k1 ? k0 : k1, lane);
    const unsigned long long b0 = lc3_wave_ballot(in0 && k0 == m, lane), b1 = lc3_wave_ballot(in1 && k1 == m, lane);
    return b0 ? __builtin_ctzll(b0) : (b1 ? 64 + __builtin_ctzll(b1) : -1);
}
__device__ __forceinline__ float lc3_ltpf_interp(const float *r, int rel, int d) {  // :457-469
    float acc = 0.0f;
    for (int m = -4; m <= 4; m++) {
        int n = 4 * m - d;
        if (n > -16 && n < 16) acc += r[rel + m] * lc3_f(LC3T_TAB_LTPF_INTERP_R_BITS, n + 15);
    }
    return acc;
}
__device__ __forceinline__ float lc3_ltpf_dot(const float *x12, int n, int d) {  // :412-424
    float acc = 0.0f;
    for (int k = -2; k <= 2; k++) {
        int h = 4 * k - d;
        if (h > -8 && h < 8) acc += x12[LC3_NMEM + n - k] * lc3_f(LC3T_TAB_LTPF_INTERP_X12K8_BITS, h + 7);
    }
    return acc;
}

// Two outputs of the polyphase resampler (encoder/long_term_post_filter.rs:152-166) that share their polyphase row: n and n + d with 15 d a multiple of p (d = p / gcd(15, p)) start OFF =
// 15 d / p samples apart and use the same taps, so one tap fetch and one window of OFF + 8 samples in registers feed both --
// out0 = sum_j xa[j] * h[j], out1 = sum_j xa[j + OFF] * h[j], each sum in tap order.  nt is a multiple of 4; xa[0 .. nt + OFF) is read.
template <int OFF>
__device__ __forceinline__ void lc3_resample_same_row(const float *xa_, const float *h, int nt, float &out0, float &out1) {
    const float *xa = LC3_LDS_BASE(xa_);
    float win[OFF + 8];
#pragma unroll
    for (int u = 0; u < OFF + 4; u++) win[u] = xa[u];
#pragma unroll
    for (int u = OFF + 4; u < OFF + 8; u++) win[u] = 0.0f;
    float acc0 = 0.0f, acc1 = 0.0f;
    for (int j = 0; j < nt; j += 4) {
        const lc3_f4 a = *(const lc3_f4 *)(h + j);
        if (j + 4 < nt) {
#pragma unroll
            for (int u = 0; u < 4; u++) win[OFF + 4 + u] = xa[j + OFF + 4 + u];
        }
        const float ta[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            acc0 += win[u] * ta[u];
            acc1 += win[u + OFF] * ta[u];
        }
#pragma unroll
        for (int u = 0; u < OFF + 4; u++) win[u] = win[u + 4];
    }
    out0 = acc0;
    out1 = acc1;
}

// g: the stream's state blob in HBM, owner of the two sample rings (x12 at 12.8 kHz, x6 at 6.4 kHz); `store` = 0 for
// the shadow waves of a partial workgroup.  Staging: x12 lives in fa for the duration of the stage, the resampler's
// polyphase table and later the correlation scratch in fb, x6 in `t` once the resampler has consumed the time buffer.
LC3_CFG_TEMPLATE __device__ __forceinline__ lc3_ltpf_res lc3_enc_ltpf(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, int near_nyquist,
                                                    int nbits, lc3_enc_state *g, int store, int ltpf_phase) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    const int len12 = c.len12, len6 = c.len6, p = c.p_up;
    const int x12_len = len12 + c.delay12 + LC3_NMEM;
    // x12 starts 64 floats into fa: the float copy of the resampler's input window (hist + nf <= 540 floats) begins in
    // spec -- dead once the mid column has left -- and runs over into fa's first 64 floats
    float *x12 = (float *)L.fa + 64, *x6 = (float *)L.t;
    float *W = (float *)L.spec;
    float *S = (float *)L.fb;
    float *r6 = S, *rw6 = S + 100, *r12 = S + 200;  // 98 | 98 | 17 floats (the activation stage reuses S for 3 x 128 products)
    int t_nbits = nbits;
    if (!c.n_ms_10) {
        double v = (double)nbits * 10.0 / 7.5;  // .round(): half away from zero (:143)
        t_nbits = (int)(v + 0.5);
    }
    const int gain_ltpf_on = t_nbits < 560 + c.fs_ind * 80;
    lc3_ltpf_res res;

    // shift_out_old_samples :217-229 happens while the ring is fetched: element i of the staged ring = element i + len12
    // of the stored one.  The polyphase table rides along.
    // The 6.4 kHz ring: LC3_KMAX history samples + the frame's len6 new ones.  (At 7.5 ms the reference's 178-element buffer has
    // 16 more elements, which are shifted around and overwritten before any use.)
    const int r6_len = LC3_KMAX + len6;
    const int keep12 = x12_len - len12, keep6 = LC3_KMAX;  // <= 276, 114
    // ring heads: where the oldest samples (the slots this frame overwrites) start, and where the ring starts after the shift
    const int old12 = L.st.ring12_head, old6 = L.st.ring6_head;
    const int head12 = old12 + len12 >= x12_len ? old12 + len12 - x12_len : old12 + len12;
    const int head6 = old6 + len6 >= r6_len ? old6 + len6 - r6_len : old6 + len6;
    LC3_HBM_FENCE();
    // the 6.4 kHz ring is requested together with the 12.8 kHz one (one trip to HBM instead of two); it waits in registers until the
    // resampler has consumed the time buffer whose place it takes
    float v6[3];
    {
        LC3_HBM_CONST(float) g12 = (LC3_HBM_CONST(float))g->x12;
        LC3_HBM_CONST(float) g6 = (LC3_HBM_CONST(float))g->x6;
        float v[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int i = lane + LC3_WAVE * j;
            int ph = head12 + i;  // the ring after this frame's shift starts at the advanced head
            ph -= ph >= x12_len ? x12_len : 0;
            v[j] = i < keep12 ? g12[ph] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = lane + LC3_WAVE * j;
            int ph = head6 + i;
            ph -= ph >= r6_len ? r6_len : 0;
            v6[j] = i < keep6 ? g6[ph] : 0.0f;
        }
        const int p_rows = p * c.resamp_stride;
#ifdef LC3_RESAMP_POLY_IN_LDS
        if (p_rows > 336)  // 8 kHz only (24 rows): larger than the workgroup's staged copy, fetched per frame as before
#endif
            for (int i = lane; i < p_rows; i += LC3_WAVE) S[i] = c.resamp_poly[i];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int i = lane + LC3_WAVE * j;
            if (i < keep12) x12[i] = v[j];
        }
        // x_s_extended as f32, converted once (every sample is a tap operand of several outputs)
        const int16_t *xs16 = L.t + (c.nf - c.z) - c.hist;
        // (the zero-padded taps of a row may reach up to three samples past the window: those operands must be finite)
        for (int i = lane; i < c.hist + c.nf + 4; i += LC3_WAVE) W[i] = i < c.hist + c.nf ? (float)xs16[i] : 0.0f;
        LC3_SYNC();
    }
    // resampling :152-166 -- one lane per 12.8 kHz output, taps accumulated in the reference's order (k ascending).
    // The low-pass is applied in polyphase form: the configuration's table (lc3_resamp_poly_value: one zero-padded
    // row of taps per phase) is staged in LDS, so a lane streams its row with 128-bit reads next to the samples.
    // x_s_extended[i] == time buffer t[(nf - z) - hist + i] == W[i]
    {
        const int nt = c.resamp_nt, lim = c.resamp_lim;
        float *o12 = x12 + c.delay12 + LC3_NMEM;
        // Lane l computes two outputs d = p / gcd(15, p) apart: they use the same polyphase row and start OFF = 15 d / p samples apart
        // (lc3_resample_same_row).  Blocks of 2 d outputs: lane l takes n0 = 2 d (l / d) + l % d and n0 + d; len12 is a multiple of 2 d.
        const int half = len12 / 2, live = lane < half, lp = live ? lane : half - 1;
        const int ds = LC3_UNIFORM_I32(p == 4 ? 2 : (p == 6 ? 1 : (p == 12 ? 2 : 3))), d = 1 << ds;  // p = 4, 6, 8, 12, 24 -> d = 4, 2, 8, 4, 8
#ifdef LC3_RESAMP_MAP_A
        const int n0 = ((lp >> ds) << (ds + 1)) + (lp & (d - 1)), n1 = n0 + d;
#else
        // (which block a lane takes decides the LDS bank pattern of the sample reads: neighbouring lanes on neighbouring BLOCKS, 2 d outputs
        // = 7.5 d samples apart, spread over the banks; neighbouring lanes inside one block collide three deep)
        const int n_blk = half >> ds, blk = lp % n_blk, n0 = (blk << (ds + 1)) + lp / n_blk, n1 = n0 + d;
#endif
        const int q0 = (15 * n0 * c.inv_p) >> 16;  // 15 n / p without integer divisions
        const float *xa = W + c.hist + q0 - 2 * lim;  // tap j <-> k = j - lim
#ifdef LC3_RESAMP_POLY_IN_LDS
        const float *poly = p * c.resamp_stride > 336 ? S : LC3_RESAMP_POLY(c);  // staged once per workgroup
#else
        const float *poly = S;
#endif
        const float *ha = poly + (15 * n0 - q0 * p) * c.resamp_stride;
        float acc0 = 0.0f, acc1 = 0.0f;
        if (p == 4 || p == 8) lc3_resample_same_row<15>(xa, ha, nt, acc0, acc1);  // 48 / 44.1 kHz, 24 kHz
        else lc3_resample_same_row<5>(xa, ha, nt, acc0, acc1);                   // 32, 16, 8 kHz
        LC3_SYNC();
        if (live) {
            o12[n0] = acc0 * c.resamp_scale;
            o12[n1] = acc1 * c.resamp_scale;
        }
        // the time buffer is consumed: stage the 6.4 kHz ring in its place (shifted by len6 on the way in)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = lane + LC3_WAVE * j;
            if (i < keep6) x6[i] = v6[j];
        }
    }
    // the high-pass memories h[-1], h[-2] as every lane needs them below (read before lane 0 moves them on)
    const float hp_m1 = L.st.h50_m1, hp_m2 = L.st.h50_m2;
    LC3_SYNC();
    LC3_STAMP(L, lane, 12);
    // 50 Hz high-pass :168-177.  The reference's loop does two things per sample: the recursion
    //   h[n] = x[n] - a1 * h[n-1] - a2 * h[n-2]
    // and the output y[n] = b0 * h[n] + b1 * h[n-1] + b2 * h[n-2].  Only the recursion is serial (and runs across frames):
    // lane 0 walks it, four operations per sample, and leaves h in place of x; the three-tap output is then formed one
    // sample per lane from the stored h -- the same f32 operations in the same order as in the single loop.
    // The recursion has the same length for every stream: the four streams of the workgroup run it together on four lanes of
    // one wave (LC3_SERIAL_BEGIN: `L` is the lane's stream inside the block), a quarter of the instructions per stream.
    LC3_SERIAL_BEGIN(lc3_enc_lds, L, lane, ltpf_phase, 1)
        float *o12 = (float *)L.fa + 64 + c.delay12 + LC3_NMEM;  // this stream's x12 + delay12 + NMEM
        float m1 = L.st.h50_m1, m2 = L.st.h50_m2;
        #pragma unroll 1
        for (int n0 = 0; n0 < len12; n0 += 8) {  // len12 is a multiple of 8; eight samples per LDS round trip
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = o12[n0 + u];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float h50 = x[u] - -1.9652933726226904f * m1 - 0.9658854605688177f * m2;
                x[u] = h50;
                m2 = m1;
                m1 = h50;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) o12[n0 + u] = x[u];
        }
        L.st.h50_m1 = m1;
        L.st.h50_m2 = m2;
    LC3_SERIAL_END
    LC3_SYNC();
    {
        float *o12 = x12 + c.delay12 + LC3_NMEM;
        float h0[2], h1[2], h2[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int n = lane + LC3_WAVE * q;
            h0[q] = h1[q] = h2[q] = 0.0f;
            if (n < len12) {
                h0[q] = o12[n];
                h1[q] = n >= 1 ? o12[n - 1] : hp_m1;
                h2[q] = n >= 2 ? o12[n - 2] : (n == 1 ? hp_m1 : hp_m2);
            }
        }
        LC3_SYNC();
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int n = lane + LC3_WAVE * q;
            if (n < len12) o12[n] = 0.9827947082978771f * h0[q] + -1.965589416595754f * h1[q] + 0.9827947082978771f * h2[q];
        }
    }
    LC3_SYNC();
    // The frame's new samples go into the oldest slots of the rings in the state blob as soon as they are final -- here for the 12.8 kHz
    // ring, after the decimation below for the 6.4 kHz one --, not at the end of the stage: a store issued right before the next frame's
    // first stage call is waited for at that call, with the whole latency of a trip to HBM exposed; issued here it has the rest of the
    // stage to complete in.
    if (store) {
        float *g12 = g->x12;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = lane + LC3_WAVE * j;
            int ph = old12 + i;
            ph -= ph >= x12_len ? x12_len : 0;
            if (i < len12) g12[ph] = x12[keep12 + i];
        }
    }
    LC3_STAMP(L, lane, 13);
    // pitch_detection :232-290
    for (int n = lane; n < len6; n += LC3_WAVE) {
        const float *s = x12 + LC3_NMEM - 3 + 2 * n;
        x6[LC3_KMAX + n] = 0.1236796411180537f * s[0] + 0.2353512128364889f * s[1] + 0.2819382920909148f * s[2] +
                           0.2353512128364889f * s[3] + 0.1236796411180537f * s[4];
    }
    LC3_SYNC();
    if (store) {
        float *g6 = g->x6;
        int ph = old6 + lane;
        ph -= ph >= r6_len ? r6_len : 0;
        if (lane < len6) g6[ph] = x6[keep6 + lane];
    }
    int lag_t1, lag_t2;
    {   // 98 lags, len6-term sums in order.  Lane l runs the ADJACENT lags 2l and 2l + 1 side by side: their second operands are
        // the same samples one position apart (x6[97 - 2l + n] and x6[96 - 2l + n]), so nine LDS values serve eight terms of both
        // sums, and the first operand (the same samples for every lag) comes as 64-bit broadcasts.
        const int NL = LC3_KMAX + 1 - LC3_KMIN;  // 98
        const int lp = lane < NL / 2 ? lane : NL / 2 - 1, live = lane < NL / 2;
        const int k0 = 2 * lp, k1 = 2 * lp + 1;
        // The two sums of a lane advance together as one packed pair: (acc1, acc0) += a[n] * (pb[n], pb[n + 1]) -- the same two f32
        // multiplications and two additions per term as two scalar chains.  The operand pairs (pb[n], pb[n + 1]) are wanted at BOTH
        // parities of n; pb is 8-byte aligned (x6 is, k0 is even), so the even ones are aligned 64-bit reads, and the odd ones are read as
        // aligned pairs too, from a copy of the ring one sample on (x6s[i] = x6[i + 1], in `spec`, dead since the mid column left):
        // sixteen lanes at a time, two banks each, every bank once.  (Read at odd offsets of x6 itself they came as 4-byte pairs whose
        // addresses all have one parity: two deep in bank conflicts, and twice -- the compiler fetched every sample once per alignment.)
        typedef float lc3_v2 __attribute__((vector_size(8)));
        float *x6s = (float *)L.spec;
        for (int i = lane; i < LC3_KMAX + len6 - 1; i += LC3_WAVE) x6s[i] = x6[i + 1];
        LC3_SYNC();
        const int pbo = LC3_KMAX - LC3_KMIN - 1 - k0;  // pb[n] = x6[pbo + n] = operand of lag k1, pb[n + 1] of lag k0
        const float *pa = LC3_LDS_BASE_ALIGNED(x6 + LC3_KMAX, 8);
        const lc3_v2 *pe = (const lc3_v2 *)LC3_LDS_BASE_ALIGNED(x6 + pbo, 8), *po = (const lc3_v2 *)LC3_LDS_BASE_ALIGNED(x6s + pbo, 8);
        float acc0, acc1;
        {   // the next eight operands of each array are requested before the current eight products are added; two register
            // blocks take turns (len6 is 64 or 48: a whole number of double blocks)
            lc3_v2 acc = {0.0f, 0.0f};
            float a[8], an[8];
            lc3_v2 be[4], bo[4], ben[4], bon[4];  // be[j] = (pb[n + 2j], pb[n + 2j + 1]), bo[j] = (pb[n + 2j + 1], pb[n + 2j + 2])
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = pa[u];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                be[j] = pe[j];
                bo[j] = po[j];
            }
#pragma unroll  // (a configuration view knows len6: straight-line code, no block is moved between registers)
            for (int n = 0; n < len6; n += 16) {
#pragma unroll
                for (int u = 0; u < 8; u++) an[u] = pa[n + 8 + u];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    ben[j] = pe[(n + 8) / 2 + j];
                    bon[j] = po[(n + 8) / 2 + j];
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc = acc + a[2 * j] * be[j];
                    acc = acc + a[2 * j + 1] * bo[j];
                }
                if (n + 16 < len6) {
#pragma unroll
                    for (int u = 0; u < 8; u++) a[u] = pa[n + 16 + u];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        be[j] = pe[(n + 16) / 2 + j];
                        bo[j] = po[(n + 16) / 2 + j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc = acc + an[2 * j] * ben[j];
                    acc = acc + an[2 * j + 1] * bon[j];
                }
            }
            acc1 = acc[0];
            acc0 = acc[1];
        }
        if (live) {
            r6[k0] = acc0;
            r6[k1] = acc1;
        }
        // index_of_max_value :427-443 twice: over the weighted correlations of all lags (T1) and over the plain ones of the
        // lags around the previous frame's (T2).  The values are still in registers: wave-parallel first-maximum scans
        // (lc3_wave_argmax_first2) instead of 98 + 9 compare steps on one lane.
        const float w0 = (1.0f - 0.5f * (float)k0 / (float)(LC3_KMAX - LC3_KMIN)) * acc0;
        const float w1 = (1.0f - 0.5f * (float)k1 / (float)(LC3_KMAX - LC3_KMIN)) * acc1;
        const int t_prev = L.st.t_prev;
        const int k_from = (t_prev - 4 > LC3_KMIN ? t_prev - 4 : LC3_KMIN) - LC3_KMIN;
        const int k_to = (t_prev + 4 < LC3_KMAX ? t_prev + 4 : LC3_KMAX) - LC3_KMIN + 1;
        int nan1, nan2;
        int i1 = lc3_wave_argmax_first2(w0, k0, live, w1, k1, live, lane, &nan1);
        int i2 = lc3_wave_argmax_first2(acc0, k0, live && k0 >= k_from && k0 < k_to, acc1, k1, live && k1 >= k_from && k1 < k_to, lane, &nan2);
        i2 = i2 < 0 ? 0 : i2 - k_from;
        if (nan1 | nan2) {  // a NaN inside a scan (never with finite PCM): the reference's sequential scans, on lane 0
            LC3_SYNC();
            if (lane == 0) {
                int idx = 0;
                float mx = (1.0f - 0.5f * 0.0f / (float)(LC3_KMAX - LC3_KMIN)) * r6[0];
                for (int i = 0; i < NL; i++) {
                    const float v = (1.0f - 0.5f * (float)i / (float)(LC3_KMAX - LC3_KMIN)) * r6[i];
                    if (v > mx) { mx = v; idx = i; }
                }
                L.ism[0] = idx;
                idx = 0;
                if (k_to > k_from) {
                    mx = r6[k_from];
                    lc3_argmax_seq(r6 + k_from, k_to - k_from, mx, idx);
                }
                L.ism[1] = idx;
            }
            LC3_SYNC();
            i1 = L.ism[0];
            i2 = L.ism[1];
            LC3_SYNC();
        }
        lag_t1 = i1 + LC3_KMIN;
        lag_t2 = i2 + k_from + LC3_KMIN;
    }
    LC3_SYNC();
    // compute_normalized_value :445-455 for lag 0, lag_t1, lag_t2: the squares once, one sample per lane, then three lanes
    // add 64 of them each in the reference's order
    float *sq = S + 224;  // 178 floats
    for (int n = lane; n < LC3_KMAX + len6; n += LC3_WAVE) {
        const float v = x6[n];
        sq[n] = v * v;
    }
    // The three sums are serial and equally long for every stream: the four streams of the workgroup run them on twelve lanes of one wave
    // (LC3_SERIAL_BEGIN: inside, `L` is the lane's stream; what the block needs travels through the stream's LDS)
    if (lane == 0) {
        L.ism[8] = lag_t1;
        L.ism[9] = lag_t2;
    }
    LC3_SERIAL_BEGIN(lc3_enc_lds, L, lane, ltpf_phase + 1, 3)
        const float *sq_ = (const float *)L.fb + 224;
        const int lag = sub == 0 ? 0 : L.ism[8 + sub - 1];
        L.sm[8 + sub] = lc3_sum_seq(LC3_LDS_BASE(sq_ + (LC3_KMAX - lag)), len6, 0.0f);
    LC3_SERIAL_END
    int t_current, pitch_present;
    {   // the decision is then the same scalar code on every lane
        const float nv0 = L.sm[8], nv1 = L.sm[9], nv2 = L.sm[10];
        float normcorr1 = lc3_maxf(0.0f, r6[lag_t1 - LC3_KMIN] / lc3_sqrtf(nv0 * nv1));
        float normcorr2 = lag_t1 == lag_t2 ? normcorr1 : lc3_maxf(0.0f, r6[lag_t2 - LC3_KMIN] / lc3_sqrtf(nv0 * nv2));
        if (normcorr2 > 0.85f * normcorr1) {
            t_current = lag_t2;
            pitch_present = normcorr2 > 0.6f;
        } else {
            t_current = lag_t1;
            pitch_present = normcorr1 > 0.6f;
        }
    }
    LC3_STAMP(L, lane, 14);
    // pitch_lag_parameter :292-363
    const int k_min = 2 * t_current - 4 > 32 ? 2 * t_current - 4 : 32;
    const int k_max = 2 * t_current + 4 < 228 ? 2 * t_current + 4 : 228;
    // <= 17 correlations of len12 terms each, a lane per lag: the workgroup's 4 x 17 lanes on one wave and four lanes of the next
    if (lane == 0) {
        L.ism[10] = k_min - 4;
        L.ism[11] = (k_max + 4) - (k_min - 4) + 1;  // <= 17
    }
    LC3_SERIAL_WIDE_BEGIN(lc3_enc_lds, L, lane, ltpf_phase + 2, 17)
        if (sub < L.ism[11]) {
            const float *x12_ = (const float *)L.fa + 64;
            const int k = L.ism[10] + sub;
            ((float *)L.fb)[200 + sub] = lc3_dot_seq(x12_ + LC3_NMEM, LC3_LDS_BASE(x12_ + LC3_NMEM - k), len12, 0.0f);  // len12 is a multiple of 8
        }
    LC3_SERIAL_END
    LC3_STAMP(L, lane, 17);
    // Integer lag = first maximum above zero of the <= 9 in-range correlations (:303-313), then the fractional part = first
    // maximum above zero of up to seven interpolated values (:315-345, interpolate :457-469).  Both scans run wave-parallel
    // (lc3_wave_argmax_first; lane j holds correlation j, then candidate j whose nine taps it adds up in the reference's order)
    // instead of 17 + 7 x 9 steps on one lane.  A NaN in a scan (never with finite PCM) takes the sequential form below.
    int pitch_int, pitch_fr, pitch_index_u;
    {
        const int nk = (k_max + 4) - (k_min - 4) + 1;
        const int k = k_min - 4 + lane, in_k = lane < nk && k >= k_min && k <= k_max;
        const float v = lane < nk ? r12[lane] : 0.0f;
        int nan_a, nan_b;
        const int ia = lc3_wave_argmax_first(v, in_k && v > 0.0f, 0.0f, 0, lane, &nan_a);
        nan_a = lc3_wave_ballot(in_k && v != v, lane) != 0ull;
        pitch_int = ia < 0 ? k_min : k_min - 4 + ia;
        const int rel = pitch_int - (k_min - 4);
        // candidate j: d = d0 + j * dstep, j < nd
        int d0 = 0, dstep = 1, nd = 0;
        if (pitch_int == 32) { d0 = 0; nd = 4; }
        else if (pitch_int < 127 && pitch_int > 32) { d0 = -3; nd = 7; }
        else if (pitch_int >= 127 && pitch_int < 157) { d0 = -2; dstep = 2; nd = 3; }
        const int d = d0 + lane * dstep;
        float acc = 0.0f;
        if (lane < nd) {
#pragma unroll
            for (int m = -4; m <= 4; m++) {  // nine independent table reads (index clamped, the tap skipped where the reference skips it)
                const int n = 4 * m - d, in = n > -16 && n < 16;
                const float t = LC3_LTPF_INTERP_R(in ? n + 15 : 15), r = r12[rel + m];
                acc = in ? acc + r * t : acc;
            }
        }
        const int ib = lc3_wave_argmax_first(acc, lane < nd && acc > 0.0f, 0.0f, 0, lane, &nan_b);
        nan_b = lc3_wave_ballot(lane < nd && acc != acc, lane) != 0ull;
        pitch_fr = ib < 0 ? 0 : d0 + ib * dstep;
        if (nan_a | nan_b) {
            LC3_SYNC();
            if (lane == 0) {
                float max_corr = 0.0f;
                int pitch_int = k_min, pitch_fr = 0;
                for (int k = k_min - 4; k <= k_max + 4; k++) {
                    float v = r12[k - (k_min - 4)];
                    if (v > max_corr && k >= k_min && k <= k_max) {
                        max_corr = v;
                        pitch_int = k;
                    }
                }
                const int rel = pitch_int - (k_min - 4);
                if (pitch_int == 32) {
                    float mx = 0.0f;
                    for (int d = 0; d <= 3; d++) {
                        float v = lc3_ltpf_interp(r12, rel, d);
                        if (v > mx) { mx = v; pitch_fr = d; }
                    }
                } else if (pitch_int < 127 && pitch_int > 32) {
                    float mx = 0.0f;
                    for (int d = -3; d <= 3; d++) {
                        float v = lc3_ltpf_interp(r12, rel, d);
                        if (v > mx) { mx = v; pitch_fr = d; }
                    }
                } else if (pitch_int >= 127 && pitch_int < 157) {
                    float mx = 0.0f;
                    for (int d = -2; d <= 2; d += 2) {
                        float v = lc3_ltpf_interp(r12, rel, d);
                        if (v > mx) { mx = v; pitch_fr = d; }
                    }
                }
                L.ism[4] = pitch_int;
                L.ism[5] = pitch_fr;
            }

            LC3_SYNC();
            pitch_int = L.ism[4];
            pitch_fr = L.ism[5];
            LC3_SYNC();
        }
        if (pitch_fr < 0) {
            pitch_int -= 1;
            pitch_fr += 4;
        }
        int pitch_index;
        if (pitch_int < 127) pitch_index = 4 * pitch_int + pitch_fr - 128;
        else if (pitch_int < 157) pitch_index = 2 * pitch_int + pitch_fr / 2 - 126;
        else pitch_index = pitch_int + 283;
        pitch_index_u = pitch_index;
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 15);
    // activation_bit :365-409 -- interpolated signals in parallel, the three 128-term sums on three lanes.
    // The normalised correlation `nc` of a frame has two uses: the activation decision of this frame and of the next two (through
    // mem_nc / mem_mem_nc), and that decision is `false` whatever nc is while the bit rate keeps the filter off (gain_ltpf_on, :146 --
    // every frame of the headline configuration).  The bit rate is the launch's, so in such a launch only the nc of its LAST TWO frames can
    // ever matter (they are what the state blob keeps for a later launch at another rate): the caller marks the other frames (bit 8 of
    // ltpf_phase) and they skip the 3 x 128 products and sums.  Workgroup-uniform: every wave of the workgroup is at the same frame of the
    // same launch.
    const int skip_nc = ((ltpf_phase >> 8) & 1) && !gain_ltpf_on;
    if (!skip_nc) {
        // the products of the three sums are formed here, one sample per lane (a product is the same f32 operation wherever
        // it runs); the three lanes below only add them up in the reference's order
        for (int n = lane; n < len12; n += LC3_WAVE) {
            const float a = lc3_ltpf_dot(x12, n, 0);
            const float b = lc3_ltpf_dot(x12, n - pitch_int, pitch_fr);
            S[n] = a * b;
            S[128 + n] = a * a;
            S[256 + n] = b * b;
        }
        // sub 0: sum dA*dB, 1: sum dA*dA, 2: sum dB*dB -- twelve lanes of one wave for the workgroup's four streams
        LC3_SERIAL_BEGIN(lc3_enc_lds, L, lane, ltpf_phase + 3, 3)
            L.sm[12 + sub] = lc3_sum_seq((const float *)L.fb + 128 * sub, len12, 0.0f);
        LC3_SERIAL_END
    }
    int ltpf_active = 0;
    {
        const float num = skip_nc ? 0.0f : L.sm[12], nd = skip_nc ? 0.0f : L.sm[13], sh = skip_nc ? 0.0f : L.sm[14];
        const float den = lc3_sqrtf(nd * sh);
        float nc = den > 0.0f ? num / den : 0.0f;  // (a skipped frame: 0, never looked at)
        const float pitch = (float)pitch_int + (float)pitch_fr / 4.0f;
        if (gain_ltpf_on && !near_nyquist) {
            const int ma = L.st.mem_ltpf_active;
            ltpf_active = (!ma && (c.n_ms_10 || L.st.mem_mem_nc > 0.94f) && L.st.mem_nc > 0.94f && nc > 0.94f) ||
                          (ma && nc > 0.9f) ||
                          (ma && lc3_absf(pitch - L.st.mem_pitch) < 2.0f && (nc - L.st.mem_nc) > -0.1f && nc > 0.84f);
        }
        int pitch_index = pitch_index_u;
        if (!pitch_present) {  // :184-214
            pitch_index = 0;
            nc = 0.0f;
        }
        const float mem_nc_old = L.st.mem_nc;
        LC3_SYNC();  // every lane has read the memories before lane 0 moves them on
        if (lane == 0) {
            L.st.t_prev = t_current;
            L.st.mem_mem_nc = mem_nc_old;
            L.st.mem_pitch = pitch_present ? pitch : 0.0f;
            L.st.mem_ltpf_active = pitch_present ? ltpf_active : 0;
            L.st.mem_nc = pitch_present ? nc : 0.0f;
        }
        res.pitch_index = pitch_index;
    }
    LC3_SYNC();
    res.pitch_present = pitch_present;
    res.ltpf_active = ltpf_active;
    res.nbits_ltpf = res.pitch_present ? 11 : 1;
    if (lane == 0) {
        L.st.ring12_head = head12;
        L.st.ring6_head = head6;
    }
    LC3_SYNC();
    return res;
}
// the same as a function of its own (a kernel that carries several configuration views: lc3_enc_front_mixed_kernel)
LC3_CFG_TEMPLATE __device__ __noinline__ lc3_ltpf_res lc3_enc_ltpf_call(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, int near_nyquist,
                                                                   int nbits, lc3_enc_state *g, int store, int ltpf_phase) {
    return lc3_enc_ltpf(LC3_CFG_PASS, LC3_LDS_PASS lane, near_nyquist, nbits, g, store, ltpf_phase);
}

// ------------------------------------------------------------------------------------------
// E17: spectral quantisation (encoder/spectral_quantization.rs:75-395)
// ------------------------------------------------------------------------------------------
struct lc3_bitcons { int rate_flag, lastnz, nbits_lsb, lastnz_trunc, nbits_est, nbits_trunc, mode_flag, lsb_mode; float gg; };

// quantize_spectrum :230-263 + compute_bit_consumption :265-348, lane-parallel: lane l owns lines 8l .. 8l+7, i.e. the
// four tuples 4l .. 4l+3, from the division to the bit estimate (the quantised values go to LDS only for the later stages).
LC3_CFG_TEMPLATE __device__ __noinline__ lc3_bitcons lc3_quantize_spectrum(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, int nbits,
                                                            int gg_off, int gg_ind, int nbits_spec) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    const int ne = c.ne;
    const float gg = LC3_POW10_GG(gg_ind + gg_off);  // 10^(((float)gg_ind + (float)gg_off) / 28): the sum of two small integers is exact in f32
    const int rate_flag = nbits > (160 + c.fs_ind * 160) ? 512 : 0;
    const int mode_flag = nbits >= (480 + c.fs_ind * 160);
    lc3_bitcons bc;
    const int ntup_all = ne / 2, k0 = 4 * lane;
    int q8[8];
    {   // eight lines per lane (ne <= 400 < 512: lines at and above ne quantise to 0, the tuple scan below relies on it)
        const lc3_divisor dv = lc3_divisor_make(gg);
        const lc3_f4 xa = *(const lc3_f4 *)(L.spec + 8 * lane), xb = *(const lc3_f4 *)(L.spec + 8 * lane + 4);  // past spec: fa (selected away)
        const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float xu = 8 * lane + u < ne ? x[u] : 0.0f;
            // `(x / gg + (x >= 0 ? 0.375 : -0.375)) as i16`, a - 0.375 == a + (-0.375).  The offset takes x's sign BIT: the one operand
            // that treats differently, x = -0.0, gives -0.0 - 0.375 -> 0 where the reference has -0.0 + 0.375 -> 0.  The quotient is no
            // NaN (x is finite: an MDCT of int16 samples, scaled by finite gains; gg = 10^(n / 28) is positive and finite), so the cast's NaN
            // rule has nothing to act on.
            q8[u] = lc3_f2i16_no_nan(lc3_div_by(xu, dv) + lc3_with_sign_of(0.375f, xu));
        }
        const lc3_i4 w = {(q8[0] & 0xffff) | (int)((uint32_t)q8[1] << 16), (q8[2] & 0xffff) | (int)((uint32_t)q8[3] << 16),
                          (q8[4] & 0xffff) | (int)((uint32_t)q8[5] << 16), (q8[6] & 0xffff) | (int)((uint32_t)q8[7] << 16)};  // int16 x 8
        *(lc3_i4 *)(LC3_XQ(L) + 8 * lane) = w;  // int16[512]: the whole padded array
    }
    // compute_bit_consumption :265-348.  Everything here is integer arithmetic, so any evaluation
    // order is exact.  The context of tuple k only depends on the (a, b, level) class tt of tuples k-1 and k-2:
    //   c_k = (c_{k-1} & 15) * 16 + tt_{k-1}  ==  16 * tt_{k-2} + tt_{k-1}          (tt <= 15)
    // so all contexts are known after one pass over the quantised pairs.  The running
    // bit estimate a tuple sees is (sum over lower lanes) + (running sum inside the lane).
    {
        uint32_t loc[4];   // a | b << 8 | n_esc << 16 | flags for the lane's tuples
        int tt[4];         // (a, b, level) class of the lane's tuples: the context of a tuple is 16 * tt(k-2) + tt(k-1)
        int hi_nz = -1;
#pragma unroll
        for (int j = 0; j < 4; j++) {  // tuples at and above ne / 2 are all-zero: class 1, never counted (k < ntup below)
            const int q0 = q8[2 * j], q1 = q8[2 * j + 1];
            const unsigned a = (unsigned)(q0 < 0 ? -q0 : q0), b = (unsigned)(q1 < 0 ? -q1 : q1);
            const unsigned m = a > b ? a : b;
            const int n_esc = m >= 4 ? (32 - __builtin_clz(m)) - 2 : 0;
            const unsigned af = a >> n_esc, bf = b >> n_esc;
            const int lev = n_esc < 3 ? n_esc : 3;
            tt[j] = lev <= 1 ? 1 + (int)((af + bf) << lev) : 12 + lev;  // (a + b) * (lev + 1) for lev = 0, 1
            const int nz = (q0 | q1) != 0;
            hi_nz = nz ? k0 + j : hi_nz;
            loc[j] = af | (bf << 8) | ((uint32_t)n_esc << 16) | ((uint32_t)nz << 24) | ((uint32_t)(a == 1) << 25) |
                     ((uint32_t)(b == 1) << 26) | ((uint32_t)(a != 0) << 27) | ((uint32_t)(b != 0) << 28);
        }
        const int hi_all = lc3_wave_max_i32(hi_nz + 1, lane);  // 1 + index of the last non-zero tuple
        const int lastnz = hi_all < 1 ? 2 : 2 * hi_all;  // `while lastnz > 2 && last pair == 0` (:270-273)
        const int ntup = lastnz / 2;
        // the two classes before the lane's first tuple come from the lane below (one DPP move each; 0 below lane 0)
        const int p2 = lc3_wave_shr1_i32(tt[2], lane), p3 = lc3_wave_shr1_i32(tt[3], lane);
        uint32_t est4[4], run = 0, lsb_sum = 0;
        int tctx[4], esc_max = 0;
        // main symbols first: four independent lookup -> bits chains per lane, straight-line so that they overlap; the
        // lookups of tuples at and above ntup are made too (valid indices) and their result dropped
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = k0 + j;
            const int live = k < ntup;
            const uint32_t v = loc[j];
            const unsigned af = v & 0xff, bf = (v >> 8) & 0xff;
            const int n_esc = (int)((v >> 16) & 0xff);
            const int cctx = 16 * (j == 0 ? p2 : (j == 1 ? p3 : tt[j - 2])) + (j == 0 ? p3 : tt[j - 1]);
            const int t = cctx + rate_flag + ((2 * k) > ne / 2 ? 256 : 0);
            tctx[j] = t;
            const int levf = n_esc < 3 ? n_esc : 3;
            const int pki = LC3_SPEC_LOOKUP(t + levf * 1024);
            uint32_t est = LC3_SPEC_BITS(pki, af + 4 * bf);
            est += ((v >> 27) & 1u) * 2048u + ((v >> 28) & 1u) * 2048u;
            if (live && n_esc > 0) {
                if (mode_flag) lsb_sum += 2 + ((v >> 25) & 1) + ((v >> 26) & 1);
                esc_max = n_esc > esc_max ? n_esc : esc_max;
            }
            est4[j] = live ? est : 0u;
        }
        // escape symbols (magnitudes >= 4 only): one per dropped bit plane, plane i coded with the table of level min(i, 3):
        //   sum_{i < n_esc} bits(level min(i, 3)) + 2 sign... = planes 0, 1, 2 one by one + (n_esc - 3) times the plane-3 cost
        // (integer sums: any order is exact).  A plane no tuple of the wave reaches is skipped (wave-uniform).
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (!LC3_WAVE_ANY(esc_max > i)) break;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n_esc = (int)((loc[j] >> 16) & 0xff);
                const int pki = LC3_SPEC_LOOKUP(tctx[j] + i * 1024);
                uint32_t e = LC3_SPEC_BITS(pki, 16) + ((i == 0 && mode_flag) ? 0u : 2u * 2048u);
                if (i == 3) e *= (uint32_t)(n_esc > 3 ? n_esc - 3 : 0);
                est4[j] += (k0 + j < ntup && n_esc > i) ? e : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) run += est4[j];
        const uint32_t base = lc3_wave_exscan_u32(run, lane);
        const uint32_t est_total = (uint32_t)lc3_wave_read_i32((int)(base + run), LC3_WAVE - 1, lane);
        const uint32_t lsb_total = lc3_wave_sum_u32(lsb_sum, lane);
        // lastnz_trunc / nbits_trunc: the last non-zero tuple whose running estimate still fits nbits_spec (:327-330; the
        // reference rounds the estimate to f32 before the division, which matters above 2^24)
        // `ceil(acc as f32 / 2048) <= nbits_spec` as an integer comparison: below 2^24 the conversion is exact and the two say the same;
        // from 2^24 on the quotient is >= 8192 > nbits_spec (<= 3200) and acc > 2048 * nbits_spec (< 2^23): both false
        int cand_k = -1;
        uint32_t cand_est = 0, acc = base;
        const uint32_t fits = nbits_spec < 0 ? 0u : 2048u * (uint32_t)nbits_spec;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc += est4[j];
            const int ok = k0 + j < ntup && (loc[j] & (1u << 24)) && acc <= fits && nbits_spec >= 0;
            LC3_GUARD_ASSERT((acc <= fits && nbits_spec >= 0) == ((int)lc3_ceilf((float)acc / 2048.0f) <= nbits_spec));
            cand_k = ok ? k0 + j : cand_k;
            cand_est = ok ? acc : cand_est;
        }
        // the owner of the highest qualifying tuple (lane best_k / 4) holds its running estimate; every quantity below is
        // wave-uniform already (reductions end in scalar registers), so nothing goes through LDS
        const int best_k = lc3_wave_max_i32(cand_k, lane);
        const uint32_t best_est = best_k >= 0 ? (uint32_t)lc3_wave_read_i32((int)cand_est, best_k >> 2, lane) : 0u;
        bc.lastnz = lastnz;
        bc.lastnz_trunc = best_k < 0 ? 2 : 2 * best_k + 2;
        bc.nbits_est = (int)lc3_ceilf((float)est_total / 2048.0f) + (int)lsb_total;
        bc.nbits_trunc = (int)lc3_ceilf((float)best_est / 2048.0f);
        bc.nbits_lsb = (int)lsb_total;
    }
    (void)ntup_all;
    bc.rate_flag = rate_flag;
    bc.mode_flag = mode_flag;
    LC3_SYNC();
    for (int n = bc.lastnz_trunc + lane; n < bc.lastnz; n += LC3_WAVE) LC3_XQ(L)[n] = 0;  // truncation :249-252
    LC3_SYNC();
    bc.lsb_mode = bc.mode_flag && bc.nbits_est > nbits_spec;
    bc.gg = gg;
    return bc;
}

LC3_CFG_TEMPLATE __device__ __noinline__ lc3_quant_res lc3_enc_quant(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, int nbits, int nbits_bw,
                                                      int nbits_tns, int nbits_ltpf) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    const int ne = c.ne, ne4 = ne / 4;
    lc3_quant_res res;
    // calc_bit_budget :122-134: ceil(log2(ne/2)) + {3,4,5}
    int nbits_ari = 0;
    while ((1 << nbits_ari) < ne / 2) nbits_ari++;
    nbits_ari += nbits <= 1280 ? 3 : (nbits <= 2560 ? 4 : 5);
    const int nbits_spec = nbits - (nbits_bw + nbits_tns + nbits_ltpf + 38 + 8 + 3 + nbits_ari);
    // get_global_gain_estimation_parameter :156-172
    float nbits_offset;
    if (L.st.reset_offset_old) nbits_offset = 0.0f;
    else {
        float prev = L.st.nbits_offset_old + (float)L.st.nbits_spec_old - (float)L.st.nbits_est_old;  // nbits_spec_old = 0 by default (A1)
        nbits_offset = 0.8f * L.st.nbits_offset_old + 0.2f * lc3_minf(40.0f, lc3_maxf(-40.0f, prev));
    }
    const int nbits_spec_adj = lc3_f2u16((float)nbits_spec + nbits_offset + 0.5f);
    int gg_off;
    {
        int q = (int)(int16_t)nbits / (10 * (c.fs_ind + 1));
        gg_off = -(q < 115 ? q : 115) - 105 - 5 * (c.fs_ind + 1);
    }
    // compute_spectral_energy :390-395 -- one lane per 4-line group (two groups per lane, kept in registers together
    // with the two products of the energy that the gain search needs)
    float e14[2], e28[2], amax = 0.0f;
    // (`x * 28.0 / 20.0`: the division by the constant through lc3_div_by -- the hardware's division sequence without the scaling and
    // fix-up steps that only act on operands these values never are; four divisions per lane and frame)
    const lc3_divisor by20 = lc3_divisor_make(20.0f);
    LC3_ENC_REPEAT(1)
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int n = lane + LC3_WAVE * q;
        const lc3_f4 x = *(const lc3_f4 *)(L.spec + 4 * (n < ne4 ? n : 0));
        const float total = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        const float ei = n < ne4 ? 10.0f * lc3_log10f(1.1920929e-7f + total) : 0.0f;
        // f32::max of magnitudes: with no negative zero among the operands the hardware maximum (IEEE mode: a NaN operand yields the
        // other one) returns what Rust's NaN-ignoring max returns, in one instruction instead of two compares and two selects each
        const float m = __builtin_fmaxf(__builtin_fmaxf(lc3_absf(x.x), lc3_absf(x.y)), __builtin_fmaxf(lc3_absf(x.z), lc3_absf(x.w)));
        amax = __builtin_fmaxf(amax, m);  // (a lane past the last group repeats group 0)
        e14[q] = lc3_div_by(ei * 28.0f, by20);
        e28[q] = lc3_div_by(2.0f * ei * 28.0f, by20);
    }
    // global_gain_limitation's max |x| :214-217 (the maximum is order-independent; non-negative floats order like
    // their bit patterns, so the wave maximum is an integer reduction)
    const float x_f_max = lc3_from_bits((uint32_t)lc3_wave_max_i32((int)lc3_bits(amax), lane));
    LC3_SYNC();
    LC3_STAMP(L, lane, 9);
    // global_gain_estimation :174-209 -- 8-step bisection on gg_ind.  Per step the reference walks the group energies
    // from the top: a group below the gain adds a constant only once a group at/above the gain has been seen
    // ("is_zero"), the others add a linear term; tmp is a sequential f32 sum of ~100 terms and the ONLY use of tmp is
    // the comparison `tmp > thr`.  Every term is >= 0 (3.78, 0, >= 9.8, > 70), so the sequential f32 sum of n <= 100
    // terms lies within (n - 1) * 2^-24 < 6e-6 (relative) of the exact sum, and so does a tree sum over the wave: when
    // the tree sum is further than LC3_BISECT_GUARD (1e-4, relative) from the threshold the comparison is decided without
    // the chain of dependent additions; otherwise (about one step in 10^4) the step repeats the sum in the reference's
    // order.  LC3_SPEC_TEST_SEQ_SUMS forces that path (tests).
    int gg_min = 0, reset_offset = 0, gg_ind_u = 0;
    LC3_ENC_REPEAT(2) {
        int fac = 256, gg_ind = 255;
        float *tv = (float *)L.fa;  // the terms of one step in the reference's order (sequential path only)
        const float thr = lc3_div_by((float)nbits_spec_adj * 1.4f * 28.0f, by20);
        const int seq_always = (L.spec_flags & LC3_SPEC_TEST_SEQ_SUMS) != 0;
        for (int level = 0; level < 8; level++) {
            fac >>= 1;
            gg_ind -= fac;
            const float g = (float)gg_ind + (float)gg_off;
            // highest group at/above the gain: two ballots (groups 0..63, 64..127)
            const unsigned long long m0 = lc3_wave_ballot(lane < ne4 && !(e14[0] < g), lane);
            const unsigned long long m1 = lc3_wave_ballot(lane + LC3_WAVE < ne4 && !(e14[1] < g), lane);
            const int h = m1 ? 64 + (63 - __builtin_clzll(m1)) : (m0 ? 63 - __builtin_clzll(m0) : -1);
            float term[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int n = lane + LC3_WAVE * q;
                float t;
                if (e14[q] < g) t = n < h ? 2.7f * 28.0f / 20.0f : 0.0f;
                else if (g < (e14[q] - 43.0f * 28.0f / 20.0f)) t = e28[q] - 2.0f * g - 36.0f * 28.0f / 20.0f;
                else t = e14[q] - g + 7.0f * 28.0f / 20.0f;
                term[q] = n < ne4 ? t : 0.0f;
            }
            const float tsum = lc3_wave_sum_f32_any(term[0] + term[1], lane);
            int over = -1;
            if (!seq_always) {
                if (tsum * (1.0f - LC3_BISECT_GUARD) > thr) over = 1;
                else if (tsum * (1.0f + LC3_BISECT_GUARD) <= thr) over = 0;
            }
            if (over < 0 || LC3_GUARD_SELFCHECK) {  // too close to call from a reordered sum: groups from the top down, one sequential f32 sum
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int n = lane + LC3_WAVE * q;
                    if (n < ne4) tv[ne4 - 1 - n] = term[q];
                }
                LC3_SYNC();
                const float tmp = lc3_sum_seq(tv, ne4, 0.0f);  // every lane the same sum (broadcast reads)
                LC3_GUARD_ASSERT(over < 0 || over == (int)(tmp > thr));
                over = tmp > thr;
                LC3_SYNC();
            }
            // `fac >>= 1; gg_ind -= fac; if tmp > thr && hi >= 0 { gg_ind += fac }`
            if (over && h >= 0) gg_ind += fac;
        }
        // global_gain_limitation :212-228 (every input is wave-uniform: the same scalar code on every lane).
        // gg_min = ceil(28 log10(x_f_max / 32767.625)) - gg_off is only ever COMPARED with the gain index -- here, and once more after the
        // gain adjustment has moved the index down by at most one (:381-383) --, and it costs a division, a log10 and a ceiling on the
        // vector unit (~75 instructions per frame, on a value that is the same on every lane).  An integer bound from the exponent of
        // x_f_max decides both comparisons for every frame that is not within a factor of ~2.5 of clipping: for 2^(n-1) <= x < 2^n,
        //   28 log10(x / 32767.625) < 28 n log10(2) - 28 log10(32767.625) = 8.42884 n - 126.4324,
        // the routine's value is within 1e-4 of that real number and the ceiling of anything below a - 126.42 is at most ceil(a) - 126:
        // B = ceil(8.42884 n) - 125 >= the reference's ceiling, with a whole unit to spare.  When gg_ind + gg_off - 1 >= B both
        // comparisons are false whatever gg_min is; otherwise (and for zero, subnormal or non-finite x_f_max) the reference's expression
        // is evaluated.  The CPU emulator evaluates it beside EVERY bound and stops if the bound is below it.
        gg_min = 0;
        int gg_min_needed = 1;
        {
            const uint32_t xb = lc3_bits(x_f_max);
            if (xb >= 0x00800000u && xb < 0x7f800000u) {
                const int n = (int)(xb >> 23) - 126;  // x_f_max < 2^n
                const int a = n >= 0 ? (8632 * n + 1023) >> 10 : -((8631 * -n) >> 10);  // >= ceil(8.42884 n): 8631 / 1024 < 8.42884 < 8632 / 1024
                const int bound = a - 125;
                LC3_GUARD_ASSERT(bound >= lc3_f2i16(lc3_ceilf(28.0f * lc3_log10f(x_f_max / (32768.0f - 0.375f)))));
                gg_min_needed = !(gg_ind + gg_off - 1 >= bound);
                gg_min = -100000;  // (stands for "below every index it is compared with")
            }
        }
        if (LC3_UNIFORM_I32(gg_min_needed)) {
            gg_min = 0;
            if (x_f_max > 0.0f) gg_min = lc3_f2i16(lc3_ceilf(28.0f * lc3_log10f(x_f_max / (32768.0f - 0.375f)))) - gg_off;
        }
        if (gg_ind < gg_min || x_f_max == 0.0f) {
            reset_offset = 1;
            gg_ind = gg_min;
        }
        gg_ind_u = gg_ind;
    }
    int gg_ind = gg_ind_u;
    LC3_SYNC();
    LC3_STAMP(L, lane, 10);
    lc3_bitcons bc = lc3_quantize_spectrum(LC3_CFG_PASS, LC3_LDS_PASS lane, nbits, gg_off, gg_ind, nbits_spec);
    LC3_ENC_REPEAT_MORE(4) bc = lc3_quantize_spectrum(LC3_CFG_PASS, LC3_LDS_PASS lane, nbits, gg_off, gg_ind, nbits_spec);
    LC3_STAMP(L, lane, 11);
    // save state after the FIRST pass :97-100
    if (lane == 0) {
        L.st.nbits_offset_old = nbits_offset;
        L.st.nbits_est_old = bc.nbits_est;
        if (L.spec_flags & LC3_SPEC_NBITS_SPEC_OLD) L.st.nbits_spec_old = nbits_spec;
        L.st.reset_offset_old = reset_offset;
    }
    // global_gain_adjustment :350-388 (wave-uniform scalar code)
    {
        const int t1 = LC3C_GGA_T1[c.fs_ind], t2 = LC3C_GGA_T2[c.fs_ind], t3 = LC3C_GGA_T3[c.fs_ind], est = bc.nbits_est, origin = gg_ind;
        float delta;
        if (est < t1) delta = ((float)est + 48.0f) / 16.0f;
        else if (est < t2) {
            float tmp1 = (float)t1 / 16.0f + 3.0f, tmp2 = (float)t2 / 48.0f;
            delta = ((float)est - (float)t1) * (tmp2 - tmp1) / ((float)t2 - (float)t1) + tmp1;
        } else if (est < t3) delta = (float)est / 48.0f;
        else delta = (float)t3 / 48.0f;
        delta = lc3_floorf(delta + 0.5f);
        const float delta2 = delta + 2.0f;
        if ((gg_ind < 255 && est > nbits_spec) || (gg_ind > 0 && (float)est < ((float)nbits_spec - delta2))) {
            if ((float)est < ((float)nbits_spec - delta2)) gg_ind -= 1;
            else if (gg_ind == 254 || (float)est < ((float)nbits_spec + delta)) gg_ind += 1;
            else gg_ind += 2;
            if (gg_ind < gg_min) gg_ind = gg_min;
        }
        if (origin != gg_ind)
            LC3_ENC_REPEAT(8) bc = lc3_quantize_spectrum(LC3_CFG_PASS, LC3_LDS_PASS lane, nbits, gg_off, gg_ind, nbits_spec);
    }
    res.gg_ind = gg_ind;
    res.nbits_spec = nbits_spec;
    res.nbits_lsb = bc.nbits_lsb;
    res.lsb_mode = bc.lsb_mode;
    res.nbits_trunc = bc.nbits_trunc;
    res.rate_flag = bc.rate_flag;
    res.lastnz_trunc = bc.lastnz_trunc;
    res.gg = bc.gg;
    return res;
}

// ------------------------------------------------------------------------------------------
// E18 residual bits (encoder/residual_spectrum.rs:33-62), E19 noise level
// (encoder/noise_level_estimation.rs:21-55).  Returns n_res via L.ism[0], noise factor via L.ism[1].
// ------------------------------------------------------------------------------------------
// returns n_res | noise factor << 16
LC3_CFG_TEMPLATE __device__ __noinline__ int lc3_enc_residual_noise(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane,
                                                    const lc3_quant_res q, int bw_ind) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    // Both loops of the reference walk the spectrum in order and act on a SUBSET of the lines (non-zero lines for
    // the residual bits, lines with an all-zero neighbourhood for the noise level); the position of a line inside
    // its subset is a prefix count.  Lane l owns lines 8l .. 8l+7 (one 16-byte unit of quantised values, two of spectrum);
    // membership is kept as 8-bit masks, the neighbours' bits come over from the adjacent lanes:
    //  - residual bit j (j < nbits_residual_max) is set by the owner of the j-th non-zero line,
    //  - the noise contributions |x| / gg are added up per lane and over the wave (see below).
    const int ne = c.ne;
    float *compact = (float *)L.fa;          // relevant |x| / gg values, in line order (<= 376; sequential path only)
    const int bw_stop = c.n_ms_10 ? LC3C_BWSTOP10[bw_ind] : LC3C_BWSTOP75[bw_ind];
    const int nf_start = c.n_ms_10 ? 24 : 18, nf_width = c.n_ms_10 ? 3 : 2;
    const int nf_stop = ne < bw_stop ? ne : bw_stop;
    int mx = q.nbits_spec - q.nbits_trunc + 4;  // nbits_residual_max (encoder/residual_spectrum.rs:42-43)
    if (mx < 0) mx = 0;
    const int k0 = 8 * lane;
    int xv[8];
    float sv[8];
    {
        const lc3_i4 w = *(const lc3_i4 *)(LC3_XQ(L) + k0);  // zero from ne on (lc3_quantize_spectrum)
        const lc3_f4 sa = *(const lc3_f4 *)(L.spec + k0), sb = *(const lc3_f4 *)(L.spec + k0 + 4);  // past ne: never selected
        int wi[4];
        __builtin_memcpy(wi, &w, 16);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            xv[2 * i] = (int)(int16_t)(wi[i] & 0xffff);
            xv[2 * i + 1] = wi[i] >> 16;
        }
        sv[0] = sa.x; sv[1] = sa.y; sv[2] = sa.z; sv[3] = sa.w; sv[4] = sb.x; sv[5] = sb.y; sv[6] = sb.z; sv[7] = sb.w;
    }
    // bit j of a mask <-> line k0 + j.  lc3_bits_below(n) = the low clamp(n, 0, 8) bits
#define LC3_BITS_BELOW(n) ((n) <= 0 ? 0u : ((n) >= 8 ? 0xffu : (1u << (n)) - 1u))
    uint32_t nzmask = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) nzmask |= (uint32_t)(xv[j] != 0) << j;
    // the noise window only sees lines below bw_stop (the reference clips it there) and none below line 0
    const uint32_t nzw = nzmask & LC3_BITS_BELOW(bw_stop - k0);
    const uint32_t prev = (uint32_t)lc3_wave_shr1_i32((int)nzw, lane), next = (uint32_t)lc3_wave_shl1_i32((int)nzw, lane);
    const uint32_t nz14 = (prev >> 5) | (nzw << 3) | ((next & 7u) << 11);  // bit i <-> line k0 - 3 + i
    // encoder/noise_level_estimation.rs:35-42: x_q[k - w .. min(bw_stop, k + w + 1)) all zero, k in [nf_start, nf_stop)
    uint32_t any;
    if (nf_width == 3) {  // bits j .. j+6
        const uint32_t a = nz14 | (nz14 >> 1), b = a | (a >> 2);
        any = b | (b >> 3);
    } else {              // bits j+1 .. j+5
        const uint32_t n1 = nz14 >> 1, a = n1 | (n1 >> 1), b = a | (a >> 2);
        any = b | (n1 >> 4);
    }
    const uint32_t relmask = ~any & LC3_BITS_BELOW(nf_stop - k0) & ~LC3_BITS_BELOW(nf_start - k0) & 0xffu;
#undef LC3_BITS_BELOW
    if (lane < 13) LC3_RESW(L)[lane] = 0u;
    const uint32_t cnt_nz = (uint32_t)__builtin_popcount(nzmask), cnt_rel = (uint32_t)__builtin_popcount(relmask);
    const int rank0 = (int)lc3_wave_exscan_u32(cnt_nz, lane);
    const int tot_nz = (int)lc3_wave_sum_u32(cnt_nz, lane), tot_rel = (int)lc3_wave_sum_u32(cnt_rel, lane);
    LC3_SYNC();  // the zeroed words before the bits
    float part = 0.0f;  // the lane's share of the noise-level sum (tree order: see below)
    {
        const lc3_divisor dv = lc3_divisor_make(q.gg);  // (quotients below 2^-100 may differ in their last bit: they cannot move a mean)
        uint32_t local = 0;  // the lane's residual bits: bit p <-> its p-th non-zero line
        int p = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int nz = (int)((nzmask >> j) & 1u);
            const int bit = nz && rank0 + p < mx && sv[j] >= (float)xv[j] * q.gg;  // :50-55
            local |= (uint32_t)bit << p;
            p += nz;
            const float term = lc3_div_by(lc3_absf(sv[j]), dv);
            part += ((relmask >> j) & 1u) ? term : 0.0f;
        }
        // into the 13-word mask in place (LDS atomic or): at most two words per lane
        const int sh = rank0 & 31;
        const uint32_t lo = local << sh, hi = sh > 24 ? local >> (32 - sh) : 0u;
        if (lo) __atomic_fetch_or(&LC3_RESW(L)[rank0 >> 5], lo, __ATOMIC_RELAXED);
        if (hi) __atomic_fetch_or(&LC3_RESW(L)[(rank0 >> 5) + 1], hi, __ATOMIC_RELAXED);
    }
    // noise level :44-55: nfac = min(7, trunc(8 - 16 * mean + 0.5)) only depends on which side of the boundaries
    // 1, 2 .. 7 the value v = 8.5 - 16 * mean falls.  The terms |x| / gg are >= 0, so the reference's sequential f32 sum
    // of n <= 376 terms is within 375 * 2^-24 (relative) of the exact sum, a tree sum within 16 * 2^-24: the two means
    // differ by < 2.4e-5 * mean, the two v by < 3.9e-4 * mean + a few ulps of 8.5.  Further than `tol` from a boundary
    // the tree sum decides; otherwise the sum is repeated in the reference's order.
    int nfac = 0, decided = 0;
    {
        const float tsum = lc3_wave_sum_f32_any(part, lane);
        const float mean = tot_rel > 0 ? tsum / (float)tot_rel : 0.0f;
        const float v = 8.5f - 16.0f * mean, tol = 1.0e-5f + 6.0e-4f * mean;
        if (!(L.spec_flags & LC3_SPEC_TEST_SEQ_SUMS)) {
            if (v < 1.0f) decided = (1.0f - v) > tol;  // includes every mean > 0.47 and NaN-free large sums
            else if (v >= 8.0f) decided = 1;
            else {
                const float fl = lc3_floorf(v), d_lo = v - fl, d_hi = 1.0f - d_lo;
                decided = d_lo > tol && (fl >= 7.0f || d_hi > tol);
                nfac = lc3_f2i32(fl);
            }
            if (v >= 8.0f) nfac = 7;
        }
    }
    const int nfac_tree = nfac;
    if (!decided || LC3_GUARD_SELFCHECK) {  // wave-uniform: the same terms in line order, one sequential f32 sum
        int rank_rel = (int)lc3_wave_exscan_u32(cnt_rel, lane);
        const lc3_divisor dv = lc3_divisor_make(q.gg);
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (relmask & (1u << j)) compact[rank_rel++] = lc3_div_by(lc3_absf(sv[j]), dv);
        LC3_SYNC();
        float sum0 = 0.0f;
        if (lane == 0) sum0 = lc3_sum_seq(compact, tot_rel, 0.0f);
        const float sum = lc3_wave_bcast0_f32(sum0, lane);
        const float level = tot_rel > 0 ? sum / (float)tot_rel : 0.0f;
        const float diff = 8.0f - 16.0f * level;
        nfac = 0;
        if (diff >= 0.0f) {
            const int v = lc3_f2i32(diff + 0.5f);
            nfac = v < 7 ? v : 7;
        }
        LC3_GUARD_ASSERT(!decided || nfac == nfac_tree);
    }
    (void)nfac_tree;
    LC3_SYNC();
    return (nfac << 16) | (tot_nz < mx ? tot_nz : mx);
}

// ------------------------------------------------------------------------------------------
// E20, the part that does not depend on the range coder's state: which symbols spectral_data (encoder/bitstream_encoding.rs:246-326)
// will code, with which model interval, and the bits that follow each of them.  Lane l owns the pairs 4l .. 4l+3 (as in
// lc3_quantize_spectrum): classes and contexts in parallel, positions by prefix sum, one LC3_SYM_WORD per symbol into the
// packer plane.  The lane-per-frame packer (lc3_dev_enc_pack.h) then only runs the coder: half the instructions per symbol
// on its serial path.  Small launches only (LC3_LAUNCH_PREP_SYMBOLS, lc3_dev_enc_pack.h); more than LC3_SYM_CAP symbols (never
// seen; the bit budget allows it in theory): EP_NSYM = -1 and the packer derives the symbols itself.
// ------------------------------------------------------------------------------------------
// q8: the lane's eight quantised values (pairs 4 lane .. 4 lane + 3; zero from lastnz_trunc on and beyond ne)
__device__ __forceinline__ void lc3_enc_symbols_core(int ne, int lane, const int (&q8)[8], int lastnz_trunc, int lsb_mode, int rate_flag,
                                                     int32_t *plane, int st) {
    const int k0 = 4 * lane, ntup = lastnz_trunc / 2;
    unsigned a4[4], b4[4];
    int ne4[4], tt[4], cnt = 0, lsbs = 0, esc_max = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q0 = q8[2 * j], q1 = q8[2 * j + 1], live = k0 + j < ntup;
        const unsigned a = (unsigned)(q0 < 0 ? -q0 : q0), b = (unsigned)(q1 < 0 ? -q1 : q1);
        const unsigned m = a > b ? a : b;
        const int n_esc = m >= 4 ? (32 - __builtin_clz(m)) - 2 : 0;
        const int lev = n_esc < 3 ? n_esc : 3;
        tt[j] = lev <= 1 ? 1 + (int)(((a >> n_esc) + (b >> n_esc)) << lev) : 12 + lev;
        a4[j] = a;
        b4[j] = b;
        ne4[j] = live ? n_esc : -1;  // -1: not coded
        cnt += live ? n_esc + 1 : 0;
        esc_max = (live && n_esc > esc_max) ? n_esc : esc_max;
        // the LSB list (:298-312): in lsb_mode an escaped pair leaves its two LSBs, and the sign of a value whose upper part is zero
        if (live && lsb_mode && n_esc > 0) lsbs += 2 + (int)((a >> 1) == 0u && q0 != 0) + (int)((b >> 1) == 0u && q1 != 0);
    }
    const int p2 = lc3_wave_shr1_i32(tt[2], lane), p3 = lc3_wave_shr1_i32(tt[3], lane);
    const uint32_t base = lc3_wave_exscan_u32((uint32_t)cnt, lane);
    const int total = lc3_wave_read_i32((int)base + cnt, LC3_WAVE - 1, lane);
    const int nlsbs = (int)lc3_wave_sum_u32((uint32_t)lsbs, lane);
    const int fits = total <= LC3_SYM_CAP;
    if (lane == 0) {
        plane[EP_NSYM * st] = fits ? total : -1;
        plane[EP_NLSBS * st] = nlsbs;
    }
    if (!fits) return;
    int pos[4], tctx[4];
    {
        int p = (int)base;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = k0 + j;
            pos[j] = p;
            p += ne4[j] + 1;  // (a pair that is not coded has ne4 = -1)
            const int cctx = 16 * (j == 0 ? p2 : (j == 1 ? p3 : tt[j - 2])) + (j == 0 ? p3 : tt[j - 1]);
            tctx[j] = cctx + rate_flag + ((2 * k) > ne / 2 ? 256 : 0);
        }
    }
    // main symbols: the magnitudes above the escaped bit planes, then the signs of the values that are non-zero there
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n_esc = ne4[j];
        if (n_esc >= 0) {
            const int lv = n_esc < 3 ? n_esc : 3;
            const int pki = LC3_SPEC_LOOKUP(tctx[j] + lv * 1024);
            const int sym = (int)((a4[j] >> n_esc) + 4u * (b4[j] >> n_esc));
            const int lsb_here = lsb_mode && n_esc > 0;
            const unsigned a_l = lsb_here ? a4[j] >> 1 : a4[j], b_l = lsb_here ? b4[j] >> 1 : b4[j];
            const int s0 = q8[2 * j] <= 0, s1 = q8[2 * j + 1] <= 0;
            const int nb = (int)(a_l > 0u) + (int)(b_l > 0u);
            const int bits = a_l > 0u ? (s0 | (s1 << 1)) : s1;
            plane[(EP_SYM + pos[j] + n_esc) * st] =
                (int32_t)LC3_SYM_WORD((int)LC3T_AC_SPEC_CUMFREQ[pki][sym], (int)LC3T_AC_SPEC_FREQ[pki][sym], nb, bits);
        }
    }
    // escape symbols, bit plane i of the pairs that have one (a wave-uniform trip count), coded with the table of level min(i, 3);
    // each is followed by that plane's two bits unless the plane is the LSB plane of an lsb_mode frame (those go to the LSB list)
    for (int i = 0; LC3_WAVE_ANY(esc_max > i); i++) {
        const int lv = i < 3 ? i : 3;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (ne4[j] > i) {
                const int pki = LC3_SPEC_LOOKUP(tctx[j] + lv * 1024);
                const int nb = (lsb_mode && i == 0) ? 0 : 2;
                const int bits = (int)((a4[j] >> i) & 1u) | ((int)((b4[j] >> i) & 1u) << 1);
                plane[(EP_SYM + pos[j] + i) * st] =
                    (int32_t)LC3_SYM_WORD((int)LC3T_AC_SPEC_CUMFREQ[pki][16], (int)LC3T_AC_SPEC_FREQ[pki][16], nb, bits);
            }
        }
    }
}
// the eight values of a lane from four packed words (x_q[2k] | x_q[2k+1] << 16)
__device__ __forceinline__ void lc3_enc_symbols_unpack(const int (&wi)[4], int (&q8)[8]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        q8[2 * i] = (int)(int16_t)(wi[i] & 0xffff);
        q8[2 * i + 1] = wi[i] >> 16;
    }
}
// ... inside the analysis kernel's back half, from the quantised spectrum in LDS
LC3_CFG_TEMPLATE __device__ __noinline__ void lc3_enc_symbols(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_enc_lds) int lane, const lc3_quant_res q,
                                              int32_t *plane, int st) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_enc_lds, lc3_enc_wg);
    const lc3_i4 w = *(const lc3_i4 *)(LC3_XQ(L) + 8 * lane);  // zero from lastnz_trunc on
    int wi[4], q8[8];
    __builtin_memcpy(wi, &w, 16);
    lc3_enc_symbols_unpack(wi, q8);
    lc3_enc_symbols_core(c.ne, lane, q8, q.lastnz_trunc, q.lsb_mode, q.rate_flag, plane, st);
}
// ... as a stage of its own between the back half and the packer, one WAVE per frame, from the frame's plane column (frame-major
// planes: the lane's four words are one 16-byte unit of the column; EP_XQ and EP_WORDS are multiples of four words)
__device__ __forceinline__ void lc3_enc_symbols_frame(int ne, int lane, int32_t *plane) {
    static_assert(EP_XQ % 4 == 0 && EP_WORDS % 4 == 0 && LC3_PLANE_STRIDE == 1, "16-byte units of the packer plane");
    const int lastnz_trunc = LC3_UNIFORM_I32(plane[EP_LASTNZ_TRUNC]), lsb_mode = LC3_UNIFORM_I32(plane[EP_LSB_MODE]),
              rate_flag = LC3_UNIFORM_I32(plane[EP_RATE_FLAG]);
    int wi[4] = {0, 0, 0, 0}, q8[8];
    if (4 * lane < 200) {
        const lc3_i4 w = *(const lc3_i4 *)(plane + EP_XQ + 4 * lane);
        __builtin_memcpy(wi, &w, 16);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) wi[i] = 4 * lane + i < ne / 2 ? wi[i] : 0;  // words beyond the configuration's pairs are stale
    lc3_enc_symbols_unpack(wi, q8);
    lc3_enc_symbols_core(ne, lane, q8, lastnz_trunc, lsb_mode, rate_flag, plane, 1);
}

// ------------------------------------------------------------------------------------------
// EncoderChannel::encode (encoder/lc3_encoder.rs:63-112): one frame of one stream on one wave.
// pcm: nf samples in HBM (4-byte aligned); plane/plane_stride: this frame's column of the packer planes
// (lc3_dev_enc_pack.h); nbytes selects the bitrate.  dbg (optional): float[3*480] stage dumps.
// ------------------------------------------------------------------------------------------
// The analysis of one frame runs in two wave-per-stream halves with the lane-per-frame SNS vector quantiser
// (lc3_dev_enc_vq.h) between them:
//   front: E1-E9a, E12-E16  MDCT, band energies, bandwidth, attack, SNS target scale factors, LTPF analysis
//          -> "mid" plane column (MP_*: spectrum, targets, flags) + the packer-plane words it already knows
//   back : E9b, E11, E17-E19  spectral shaping with the quantised gains, TNS, quantiser, residual bits, noise level
//          -> the rest of the packer plane column
// hist: see lc3_enc_mdct; g: the stream's state blob (LTPF rings); mid/plane == nullptr marks a shadow wave that
// stores nothing.  dbg (optional): float[LC3_ENC_DBG_FLOATS] stage dumps (include/lc3gpu.h, lc3gpu_encode_frame_debug).
LC3_CFG_TEMPLATE __device__ __forceinline__ void lc3_encode_front_wave(LC3_CFG_PARAM, lc3_enc_lds &L, int lane, const int16_t *pcm,
                                                      const int16_t *hist, lc3_enc_state *g, float *mid, int32_t *plane,
                                                      int plane_stride, int nbytes, float *dbg, int stride = 1, int hstride = 1,
                                                      int phase = 0, int outline_ltpf = 0) {
    LC3_CFG_BIND;
    const int nbits = nbytes * 8;
    LC3_STAMP(L, lane, 0);
    const int near_nyquist = LC3_KO(LC3_ENC_KO, 1) ? 0 : lc3_enc_mdct(LC3_CFG_PASS, LC3_LDS_PASS lane, pcm, hist, stride, hstride);
    LC3_STAMP(L, lane, 1);
    if (dbg) for (int i = lane; i < c.nf; i += LC3_WAVE) dbg[i] = L.spec[i];
    if (dbg && lane < c.nb) dbg[LC3_ENC_DBG_EB + lane] = LC3_EB(L)[lane];  // the band energies (modified_dct.rs:140-152)
    int nbits_bw = 3;
    const int bw_ind = LC3_KO(LC3_ENC_KO, 2) ? 4 : lc3_enc_bandwidth(LC3_CFG_PASS, LC3_LDS_PASS lane, &nbits_bw);
    LC3_STAMP(L, lane, 24);
    const int attack = LC3_KO(LC3_ENC_KO, 4) ? 0 : lc3_enc_attack(LC3_CFG_PASS, LC3_LDS_PASS lane, nbytes);
    LC3_STAMP(L, lane, 2);
    if (!LC3_KO(LC3_ENC_KO, 8)) lc3_enc_sns_front(LC3_CFG_PASS, LC3_LDS_PASS lane, attack);
    if (mid) {  // targets and spectrum leave LDS before the LTPF stage reuses fa/fb
        if (lane < 16) mid[MP_SCF + lane] = LC3_SCF(L)[lane];
        lc3_wave_copy_out16(mid + MP_SPEC, L.spec, c.ne / 4, lane);  // (the ne lines the later stages use: ne is a multiple of four)
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 3);
    lc3_ltpf_res pf = {0, 0, 0, 1};
    if (LC3_KO(LC3_ENC_KO, 16)) {
    } else if (outline_ltpf) pf = lc3_enc_ltpf_call(LC3_CFG_PASS, LC3_LDS_PASS lane, near_nyquist, nbits, g, plane != nullptr, phase);
    else pf = lc3_enc_ltpf(LC3_CFG_PASS, LC3_LDS_PASS lane, near_nyquist, nbits, g, plane != nullptr, phase);
    LC3_STAMP(L, lane, 5);
    if (dbg && lane == 0) {
        float *d = dbg + 1440;
        d[0] = (float)bw_ind; d[1] = (float)attack; d[9] = (float)pf.pitch_index; d[10] = (float)pf.pitch_present;
        d[11] = (float)pf.ltpf_active; d[21] = (float)near_nyquist;
        // the attack detector's state after this frame (attack_detector.rs:17-21)
        float *a = dbg + LC3_ENC_DBG_ATTACK;
        a[0] = L.st.att_energy_last; a[1] = L.st.att_max_energy_last; a[2] = (float)L.st.att_pos_last;
        a[3] = (float)L.st.att_ds_tm1; a[4] = (float)L.st.att_ds_tm2;
    }
    if (plane && mid && lane == 0) {
        const int st = plane_stride;
        int32_t *mf = (int32_t *)mid + MP_FLAGS;
        mf[MPF_BW] = bw_ind;
        mf[MPF_NBITS_BW] = nbits_bw;
        mf[MPF_NEAR_NYQUIST] = near_nyquist;
        mf[MPF_NBITS_LTPF] = pf.nbits_ltpf;
        plane[EP_BW * st] = bw_ind;
        plane[EP_NBITS_BW * st] = nbits_bw;
        plane[EP_PITCH_PRESENT * st] = pf.pitch_present;
        plane[EP_LTPF_ACTIVE * st] = pf.ltpf_active;
        plane[EP_PITCH_INDEX * st] = pf.pitch_index;
    }
    LC3_SYNC();
}

// A frame's mid-plane words in flight: the back half issues the loads of frame t + 1 before it works on frame t and
// keeps them in registers (two 16-byte units of spectrum per lane, one flag word on lanes 0..3)
struct lc3_mid_fetch {
    lc3_i4 u[2];
    uint32_t bands[2];  // the band of each of the lane's eight lines, one byte per line (c.line_band)
    float g;            // band gain of band `lane` (the vector quantiser's output)
    int32_t flag;
};
template <class CC>
__device__ __forceinline__ void lc3_mid_issue(const CC &c, int lane, const float *mid, lc3_mid_fetch &m) {
    LC3_HBM_CONST(lc3_i4) s4 = (LC3_HBM_CONST(lc3_i4))(mid + MP_SPEC);
    const int n4 = c.ne / 4;  // <= 100: the lines below ne (the front half stores no others)
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int i = lane + LC3_WAVE * j;
        if (i < n4) {
            m.u[j] = s4[i];
            m.bands[j] = ((LC3_HBM_CONST(uint32_t))c.line_band)[i];
        }
    }
    m.g = lane < c.nb ? ((LC3_HBM_CONST(float))mid)[MP_G + lane] : 0.0f;
    m.flag = lane < 4 ? ((LC3_HBM_CONST(int32_t))mid)[MP_FLAGS + lane] : 0;
}

// The back half of a frame in three parts (lc3_encode_back_stream below strings them together):
//   lc3_encode_back_pickup   the frame's mid-column words (fetched by lc3_mid_issue) -> LDS: flags, band gains, the spectrum shaped by its gains
//   lc3_encode_back_analyse  TNS filter (from the frame's slot), quantiser, residual bits, noise level -- everything that carries state
//   lc3_encode_back_store    the packer column: the longest stretch of a frame without a stage call
struct lc3_back_res {
    lc3_tns_res tns;
    lc3_quant_res spec;
    int n_res, noise_factor;
};
// m: this frame's mid-plane words, fetched by lc3_mid_issue
LC3_CFG_TEMPLATE __device__ __forceinline__ void lc3_encode_back_pickup(LC3_CFG_PARAM, lc3_enc_lds &L, int lane, const lc3_mid_fetch &m) {
    LC3_CFG_BIND;
    LC3_STAMP(L, lane, 0);
    // pick up the frame: flags and band gains -> LDS, then the spectrum -> LDS (16-byte units), each line scaled by its band's
    // gain on the way in (E9 back half: spectral shaping :264-268 with the gains the vector quantiser stage produced)
    float *gs = (float *)L.fa;  // [64] band gains
    if (lane < 4) L.ism[lane] = m.flag;
    if (lane < c.nb) gs[lane] = m.g;
    LC3_SYNC();
    LC3_ENC_REPEAT(128) {
        const int n4 = c.ne / 4;  // (L.spec from ne on is not written: nothing below reads a line there without selecting it away)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = lane + LC3_WAVE * j;
            if (i < n4) {
                const lc3_f4 xin = __builtin_bit_cast(lc3_f4, m.u[j]);
                float x[4] = {xin.x, xin.y, xin.z, xin.w};
                float gk[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t b = (m.bands[j] >> (8 * q)) & 0xffu;
                    gk[q] = gs[b < 64u ? b : 0u];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t b = (m.bands[j] >> (8 * q)) & 0xffu;
                    x[q] = b < 64u ? x[q] * gk[q] : x[q];
                }
                lc3_f4 o;
                o.x = x[0]; o.y = x[1]; o.z = x[2]; o.w = x[3];
                ((lc3_f4 *)L.spec)[i] = o;
            }
        }
    }
    LC3_SYNC();
}
// slot: where lc3_enc_tns_lev left this frame's TNS analysis
LC3_CFG_TEMPLATE __device__ __forceinline__ lc3_back_res lc3_encode_back_analyse(LC3_CFG_PARAM, lc3_enc_lds &L, int lane, int nbytes, int slot,
                                                                float *dbg) {
    LC3_CFG_BIND;
    const int nbits = nbytes * 8;
    const int bw_ind = L.ism[MPF_BW], nbits_bw = L.ism[MPF_NBITS_BW];
    const int nbits_ltpf = L.ism[MPF_NBITS_LTPF];
    LC3_STAMP(L, lane, 18);
    LC3_STAMP(L, lane, 19);
    if (dbg) for (int i = lane; i < c.nf; i += LC3_WAVE) dbg[480 + i] = L.spec[i];
    lc3_tns_res tns = {};
    if (LC3_KO(LC3_ENC_KO, 256)) tns.nbits_tns = 2, tns.num_tns_filters = 2;
    else tns = lc3_enc_tns_apply(LC3_CFG_PASS, LC3_LDS_PASS lane, bw_ind, nbits, slot);
    LC3_STAMP(L, lane, 4);
    if (dbg) for (int i = lane; i < c.nf; i += LC3_WAVE) dbg[960 + i] = L.spec[i];
    lc3_quant_res spec = {};
    if (LC3_KO(LC3_ENC_KO, 512)) spec.lastnz_trunc = 2, spec.gg = 1.0f;
    else spec = lc3_enc_quant(LC3_CFG_PASS, LC3_LDS_PASS lane, nbits, nbits_bw, tns.nbits_tns, nbits_ltpf);
    LC3_STAMP(L, lane, 6);
    int rn = LC3_KO(LC3_ENC_KO, 1024) ? 0 : lc3_enc_residual_noise(LC3_CFG_PASS, LC3_LDS_PASS lane, spec, bw_ind);
    LC3_ENC_REPEAT_MORE(64) rn = lc3_enc_residual_noise(LC3_CFG_PASS, LC3_LDS_PASS lane, spec, bw_ind);
    LC3_STAMP(L, lane, 7);
    lc3_back_res out;
    out.tns = tns;
    out.spec = spec;
    out.n_res = rn & 0xffff;
    out.noise_factor = rn >> 16;
    return out;
}
LC3_CFG_TEMPLATE __device__ __forceinline__ void lc3_encode_back_store(LC3_CFG_PARAM, lc3_enc_lds &L, int lane, const lc3_back_res &r,
                                                      int32_t *plane, int plane_stride, int store, float *dbg) {
    LC3_CFG_BIND;
    const lc3_tns_res &tns = r.tns;
    const lc3_quant_res &spec = r.spec;
    const int n_res = r.n_res, noise_factor = r.noise_factor;
    if (dbg && lane == 0) {
        float *d = dbg + 1440;
        const int st = plane_stride;
        d[2] = (float)plane[EP_IND_LF * st]; d[3] = (float)plane[EP_IND_HF * st];
        d[4] = (float)plane[EP_SHAPE_J * st]; d[5] = (float)plane[EP_GIND * st];
        d[6] = (float)tns.rc_order[0]; d[7] = (float)tns.rc_order[1]; d[8] = (float)tns.nbits_tns;
        d[12] = (float)spec.gg_ind; d[13] = (float)spec.lastnz_trunc;
        d[14] = (float)spec.nbits_lsb; d[15] = (float)spec.lsb_mode; d[16] = (float)n_res; d[17] = (float)noise_factor;
        d[18] = spec.gg; d[19] = (float)spec.nbits_spec; d[20] = (float)spec.nbits_trunc;
        const uint32_t joint = (uint32_t)plane[EP_JOINT * st];  // (up to 25 bits: as two exact halves)
        d[22] = (float)plane[EP_LS_INDA * st]; d[23] = (float)(joint & 0xffffu); d[24] = (float)(joint >> 16);
    }
    // E20/E21 run as a separate lane-per-frame stage (lc3_dev_enc_pack.h): leave this frame's plane column in HBM
    if (store) {
        const int st = plane_stride;
        if (lane == 0) {
            plane[EP_LASTNZ_TRUNC * st] = spec.lastnz_trunc;
            plane[EP_LSB_MODE * st] = spec.lsb_mode;
            plane[EP_GG_IND * st] = spec.gg_ind;
            plane[EP_NUM_TNS * st] = tns.num_tns_filters;
            plane[EP_ORD0 * st] = tns.rc_order[0];
            plane[EP_ORD1 * st] = tns.rc_order[1];
            plane[EP_LPC_W * st] = tns.lpc_weighting;
            plane[EP_NOISE * st] = noise_factor;
            plane[EP_RATE_FLAG * st] = spec.rate_flag;
            plane[EP_N_RES * st] = n_res;
        }
        if (lane < 16) plane[(EP_RCI + lane) * st] = L.ism[16 + lane];
        if (lane < 13) plane[(EP_RES + lane) * st] = (int32_t)LC3_RESW(L)[lane];  // residual bits as a bit mask
        // the quantised pairs x_q[2k] | x_q[2k+1] << 16 are the words of the int16 array as they lie in LDS: 16-byte units (frame-major
        // planes; up to three words beyond ne / 2 go along, inside the column's 200 and zero)
        if (LC3_PLANE_STRIDE == 1) lc3_wave_copy_out16(plane + EP_XQ, LC3_XQ(L), (c.ne / 2 + 3) / 4, lane);
        else
            for (int k = lane; k < c.ne / 2; k += LC3_WAVE)
                plane[(EP_XQ + k) * st] = (int32_t)(((uint32_t)(uint16_t)LC3_XQ(L)[2 * k]) | ((uint32_t)(uint16_t)LC3_XQ(L)[2 * k + 1] << 16));
        if (L.spec_flags & LC3_LAUNCH_PREP_SYMBOLS) lc3_enc_symbols(LC3_CFG_PASS, LC3_LDS_PASS lane, spec, plane, st);
        else if (lane == 0) plane[EP_NSYM * st] = -1;
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 8);
}
// The frames of one stream in one launch (the body of lc3_enc_back_kernel's frame loop; the CPU emulator of the tests runs the same).
// Frames are analysed in chunks of up to LC3_TNS_CHUNK: first the TNS autocorrelations of every frame of the chunk (pick-up,
// lc3_enc_tns_acf), then ONE pass of the Levinson recursions for all of them (lc3_enc_tns_lev), then frame by frame -- picked up a
// second time -- everything that carries state from frame to frame.  A chunk of one frame picks it up once.  The mid column of the NEXT
// visit is requested before the current one is worked on.
// mid / planes: the launch's columns; fbase: the stream's first frame; store = 0: a shadow wave that stores nothing
LC3_CFG_TEMPLATE __device__ __forceinline__ void lc3_encode_back_stream(LC3_CFG_PARAM, lc3_enc_lds &L, int lane, const float *mid, int32_t *planes,
                                                       size_t fbase, int n_frames, int nbytes, int store, float *dbg) {
    LC3_CFG_BIND;
    lc3_mid_fetch cur, nxt;
    if (n_frames > 0) lc3_mid_issue(c, lane, mid + fbase * (size_t)MP_WORDS, cur);
    nxt = cur;
    for (int t0 = 0; t0 < n_frames; t0 += LC3_TNS_CHUNK) {
        const int nc = n_frames - t0 < LC3_TNS_CHUNK ? n_frames - t0 : LC3_TNS_CHUNK;
        if (nc > 1) {
            for (int u = 0; u < nc; u++) {
                const int tn = u + 1 < nc ? t0 + u + 1 : t0;  // next visit: the chunk's next frame, then its first frame again
                lc3_mid_issue(c, lane, mid + (fbase + (size_t)tn) * (size_t)MP_WORDS, nxt);
                lc3_encode_back_pickup(LC3_CFG_PASS, L, lane, cur);
                lc3_enc_tns_acf(LC3_CFG_PASS, LC3_LDS_PASS lane, L.ism[MPF_BW], L.ism[MPF_NEAR_NYQUIST], u);
                cur = nxt;
            }
            lc3_enc_tns_lev(LC3_CFG_PASS, LC3_LDS_PASS lane, nbytes * 8, nc);
        }
        for (int u = 0; u < nc; u++) {
            const int t = t0 + u;
            if (t + 1 < n_frames) lc3_mid_issue(c, lane, mid + (fbase + (size_t)t + 1) * (size_t)MP_WORDS, nxt);  // (the next chunk starts at t0 + nc)
            lc3_encode_back_pickup(LC3_CFG_PASS, L, lane, cur);
            if (nc == 1) {
                lc3_enc_tns_acf(LC3_CFG_PASS, LC3_LDS_PASS lane, L.ism[MPF_BW], L.ism[MPF_NEAR_NYQUIST], 0);
                lc3_enc_tns_lev(LC3_CFG_PASS, LC3_LDS_PASS lane, nbytes * 8, 1);
            }
            const lc3_back_res r = lc3_encode_back_analyse(LC3_CFG_PASS, L, lane, nbytes, u, dbg);
            lc3_encode_back_store(LC3_CFG_PASS, L, lane, r, LC3_PLANE_COL(planes, fbase + (size_t)t, EP_WORDS), LC3_PLANE_STRIDE, store, dbg);
            cur = nxt;
        }
    }
}
