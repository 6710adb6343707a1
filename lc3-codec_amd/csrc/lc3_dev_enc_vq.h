// LC3 batched encoder for MI355X -- SNS vector quantiser, ONE LANE PER FRAME.
//
// sns_quant_stage1 / stage2, add_unit_pulse, normalize_candidate, mvpq_enum and the scale-factor synthesis and
// interpolation of the reference (encoder/spectral_noise_shaping.rs:163-201, 254-257, 285-648) take 16 target scale
// factors and return codebook / shape / gain indices and 64 band gains.  They carry no state from frame to frame and
// are almost entirely serial (greedy pulse search: ~2 k dependent operations), so -- like the range coder -- they run
// with every lane working on its own frame, everything in registers, between the two wave-per-stream halves of the
// analysis (lc3_dev_enc.h: front = MDCT .. SNS targets + LTPF, back = shaping, TNS, quantiser).  The arithmetic of
// each frame is the reference's, operation for operation (the wave kernel ran the same code on lane 0 before).
#pragma once
#include "lc3_dev_common.h"
#include "lc3_dev_enc_pack.h"

// "mid" plane: what the front half leaves per frame for this stage and the back half (f32 / int32 words)
enum {
    MP_SCF = 0,             // [16] f32 target scale factors (after mean removal / attack smoothing)
    MP_G = 16,              // [64] f32 band gains exp2(-scf_interpolated), written by this stage
    MP_FLAGS = 80,          // bw_ind, nbits_bw, near_nyquist, nbits_ltpf, 4 spare
    MP_SPEC = 88,           // [nf <= 480] f32 MDCT spectrum before shaping
    MP_WORDS = 568
};
enum { MPF_BW = 0, MPF_NBITS_BW, MPF_NEAR_NYQUIST, MPF_NBITS_LTPF };

struct lc3_vq_ctx {
    const float *mid;        // this frame's mid-plane column (reads MP_SCF)
    float *gains;            // = mid + MP_G
    int32_t *plane;          // packer plane column (writes EP_IND_LF ... EP_JOINT)
    int stride;
    const uint32_t *mpvq;    // MPVQ_OFFSETS[16][11]
    int nb;
    int spec_flags;          // LC3_SPEC_SNS_LAST_GAIN: also try the last gain of every shape (the same for every frame of a launch)
};

// add_unit_pulse :285-316 (corr_xy / energy_y written through on every probe: SURVEY A2).  Magnitudes and pulse
// counts are register arrays (N = 16 or 10 candidates, fully unrolled).
template <int N>
__device__ __forceinline__ void lc3_add_unit_pulse_r(const float (&abs_x)[16], int (&cand)[16], int k, int k_max,
                                                     float &corr_xy, float &energy_y) {
    float corr_last = corr_xy, en_last = energy_y;
    for (int it = k; it < k_max; it++) {
        int n_best = 0;
        corr_xy = corr_last + abs_x[0];
        float best_corr_sq = corr_xy * corr_xy;
        float best_en = en_last + 2.0f * (float)cand[0] + 1.0f;
        float best_abs = abs_x[0];
        int best_cand = cand[0];
#pragma unroll
        for (int n_c = 1; n_c < N; n_c++) {
            corr_xy = corr_last + abs_x[n_c];
            energy_y = en_last + 2.0f * (float)cand[n_c] + 1.0f;
            if (corr_xy * corr_xy * best_en > best_corr_sq * energy_y) {
                n_best = n_c;
                best_corr_sq = corr_xy * corr_xy;
                best_en = energy_y;
                best_abs = abs_x[n_c];
                best_cand = cand[n_c];
            }
        }
        corr_last += best_abs;
        en_last += 2.0f * (float)best_cand + 1.0f;
#pragma unroll
        for (int n = 0; n < N; n++) cand[n] += n == n_best;
    }
}
// interpolated scale factor of band slot b (:163-183) from the 16 quantised ones (static indices: callers unroll over b)
__device__ __forceinline__ float lc3_vq_interp(const float (&q)[16], int b) {
    if (b < 2) return q[0];
    if (b >= 62) return q[15] + ((b == 62 ? 0.125f : 0.375f) * (q[15] - q[14]));
    const int n = (b - 2) >> 2, r = (b - 2) & 3;
    const float in0 = q[n], d = q[n + 1] - q[n];
    const float w = r == 0 ? 0.125f : (r == 1 ? 0.375f : (r == 2 ? 0.625f : 0.875f));
    return in0 + (w * d);
}

// mvpq_enum :584-629 on a register vector (positions [base, base + DIM) of y)
template <int DIM>
__device__ __forceinline__ void lc3_vq_enum(const lc3_vq_ctx &v, const int (&y)[16], int base, uint32_t &index,
                                            int &lead_sign_ind) {
    int next_sign_ind = (-2147483647 - 1);
    int k_val_acc = 0, n = 0;
    uint32_t tmp_h_row = v.mpvq[0];
    index = 0;
#pragma unroll
    for (int pos = DIM - 1; pos >= 0; pos--) {
        const int tmp_val = (int)(int8_t)y[base + pos];
        if (next_sign_ind >= 0 && tmp_val != 0) index = 2 * index + (uint32_t)next_sign_ind;
        if (tmp_val < 0) next_sign_ind = 1;
        else if (tmp_val > 0) next_sign_ind = 0;
        index += tmp_h_row;
        k_val_acc += tmp_val < 0 ? -tmp_val : tmp_val;
        if (pos != 0) n += 1;
        tmp_h_row = k_val_acc >= 11 ? v.mpvq[(n + 1) * 11 + k_val_acc % 11] : v.mpvq[n * 11 + k_val_acc];
    }
    lead_sign_ind = next_sign_ind;
}

// distortion of one (shape, gain) pair :472-521 and the running minimum in the reference's order
#define LC3_VQ_TRY(J, I, GAINS)                                              \
    {                                                                        \
        const float g_ = lc3_f(GAINS, I);                                    \
        float d_ = 0.0f;                                                     \
        _Pragma("unroll") for (int n_ = 0; n_ < 16; n_++) {                  \
            const float df_ = t2[n_] - g_ * xq[n_];                          \
            d_ += df_ * df_;                                                 \
        }                                                                    \
        if (d_ < d_min) {                                                    \
            shape_j = J;                                                     \
            gind = I;                                                        \
            d_min = d_;                                                      \
            g_sel = g_;                                                      \
            _Pragma("unroll") for (int n_ = 0; n_ < 16; n_++) xq_sel[n_] = xq[n_]; \
        }                                                                    \
    }

// normalize_candidate :632-648 into xq (n_max leading entries, zero behind)
#define LC3_VQ_NORMALIZE(Y, NMAX)                                                          \
    {                                                                                      \
        float norm_ = 0.0f;                                                                \
        _Pragma("unroll") for (int n_ = 0; n_ < NMAX; n_++)                                \
            if (Y[n_] != 0) norm_ += (float)Y[n_] * (float)Y[n_];                          \
        norm_ = lc3_sqrtf(norm_);                                                          \
        _Pragma("unroll") for (int n_ = 0; n_ < 16; n_++) {                                \
            float v_ = (float)Y[n_];                                                       \
            if (Y[n_] != 0) v_ /= norm_;                                                   \
            xq[n_] = n_ < NMAX ? v_ : 0.0f;                                                \
        }                                                                                  \
    }

__device__ __forceinline__ void lc3_sns_vq_frame(const lc3_vq_ctx &v) {
    float s[16];
    {   // the 16 targets as four 128-bit loads (the column and MP_SCF are 16-byte aligned)
        const lc3_f4 *s4 = (const lc3_f4 *)(v.mid + MP_SCF);
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const lc3_f4 w = s4[n];
            s[4 * n] = w.x; s[4 * n + 1] = w.y; s[4 * n + 2] = w.z; s[4 * n + 3] = w.w;
        }
    }
    // stage 1 :318-361: nearest LF / HF codebook entries (first minimum)
    int ind_lf = 0, ind_hf = 0;
    {
        float lf_min = __builtin_inff(), hf_min = __builtin_inff();
        for (int i = 0; i < 32; i++) {
            float dl = 0.0f, dh = 0.0f;
#pragma unroll
            for (int n = 0; n < 8; n++) {
                const float a = s[n] - lc3_f(&LC3T_LFCB_BITS[i][0], n), b = s[8 + n] - lc3_f(&LC3T_HFCB_BITS[i][0], n);
                dl += a * a;
                dh += b * b;
            }
            if (dl < lf_min) { ind_lf = i; lf_min = dl; }
            if (dh < hf_min) { ind_hf = i; hf_min = dh; }
        }
    }
    float st1[16], t2[16], ax[16];
#pragma unroll
    for (int n = 0; n < 8; n++) {
        st1[n] = lc3_f(&LC3T_LFCB_BITS[ind_lf][0], n);
        st1[8 + n] = lc3_f(&LC3T_HFCB_BITS[ind_hf][0], n);
    }
    // stage 2 target: t2rot = r1 * D, row-by-row accumulation order (:378-384)
    uint32_t neg = 0;
    {
        float r1[16];
#pragma unroll
        for (int n = 0; n < 16; n++) r1[n] = s[n] - st1[n];
#pragma unroll
        for (int col = 0; col < 16; col++) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; i++) acc += r1[i] * lc3_f(&LC3T_D_BITS[i][0], col);
            t2[col] = acc;
            ax[col] = lc3_absf(acc);
            if (acc < 0.0f) neg |= 1u << col;
        }
    }
    // pulse search :285-316, :386-470 (corr_xy / energy_y written through on every probe: SURVEY A2)
    int y0[16], y1[16], y2[16], y3[16];
    {
        int cand[16];
        int k = 0;
        float abs_sum = 0.0f, corr_xy = 0.0f, energy_y = 0.0f;
#pragma unroll
        for (int n = 0; n < 16; n++) abs_sum += ax[n];
        const float proj = (6.0f - 1.0f) / abs_sum;
#pragma unroll
        for (int n = 0; n < 16; n++) {
            const int q = lc3_f2i32(lc3_floorf(ax[n] * proj));
            cand[n] = q;
            if (q != 0) {
                k += q;
                corr_xy += (float)q * ax[n];
                energy_y += (float)q * (float)q;
            }
        }
        lc3_add_unit_pulse_r<16>(ax, cand, k, 6, corr_xy, energy_y);
#pragma unroll
        for (int n = 0; n < 16; n++) y3[n] = (neg >> n) & 1u ? -cand[n] : cand[n];
        lc3_add_unit_pulse_r<16>(ax, cand, 6, 8, corr_xy, energy_y);
#pragma unroll
        for (int n = 0; n < 16; n++) y2[n] = (neg >> n) & 1u ? -cand[n] : cand[n];
        int ks = 8;
#pragma unroll
        for (int n = 10; n < 16; n++) {
            if (cand[n] != 0) {
                ks -= cand[n];
                corr_xy -= (float)cand[n] * ax[n];
                energy_y -= (float)cand[n] * (float)cand[n];
            }
            cand[n] = 0;
        }
        lc3_add_unit_pulse_r<10>(ax, cand, ks, 10, corr_xy, energy_y);
        float max_abs = 0.0f;
        int n_best = 0;
#pragma unroll
        for (int n = 10; n < 16; n++) {
            if (ax[n] > max_abs) {
                max_abs = ax[n];
                n_best = n;
            }
        }
#pragma unroll
        for (int n = 0; n < 16; n++) {
            y1[n] = n < 10 ? ((neg >> n) & 1u ? -cand[n] : cand[n]) : 0;
            y0[n] = y1[n];
            if (n >= 10 && n == n_best) y0[n] = (neg >> n) & 1u ? -1 : 1;
        }
        // no positive magnitude in 10..15: the reference's pulse lands on line 0 and takes that line's sign
        if (n_best == 0) y0[0] = neg & 1u ? -1 : 1;
    }
    // shape / gain search :472-521; the last gain of every shape is never tried
    int shape_j = 0, gind = 0;
    float g_sel = 0.0f, d_min = __builtin_inff();
    float xq[16], xq_sel[16];
#pragma unroll
    for (int n = 0; n < 16; n++) xq_sel[n] = 0.0f;
    LC3_VQ_NORMALIZE(y0, 16)
#pragma unroll
    for (int n = 0; n < 16; n++) xq_sel[n] = xq[n];  // what a search that never improves on +inf is left with
    const int last_gain = (v.spec_flags & LC3_SPEC_SNS_LAST_GAIN) != 0;  // off: the reference's search
    LC3_VQ_TRY(0, 0, LC3T_SNS_VQ_REG_ADJ_GAINS_BITS)
    if (last_gain) LC3_VQ_TRY(0, 1, LC3T_SNS_VQ_REG_ADJ_GAINS_BITS)
    LC3_VQ_NORMALIZE(y1, 10)
    LC3_VQ_TRY(1, 0, LC3T_SNS_VQ_REG_LF_ADJ_GAINS_BITS)
    LC3_VQ_TRY(1, 1, LC3T_SNS_VQ_REG_LF_ADJ_GAINS_BITS)
    LC3_VQ_TRY(1, 2, LC3T_SNS_VQ_REG_LF_ADJ_GAINS_BITS)
    if (last_gain) LC3_VQ_TRY(1, 3, LC3T_SNS_VQ_REG_LF_ADJ_GAINS_BITS)
    LC3_VQ_NORMALIZE(y2, 16)
    LC3_VQ_TRY(2, 0, LC3T_SNS_VQ_NEAR_ADJ_GAINS_BITS)
    LC3_VQ_TRY(2, 1, LC3T_SNS_VQ_NEAR_ADJ_GAINS_BITS)
    LC3_VQ_TRY(2, 2, LC3T_SNS_VQ_NEAR_ADJ_GAINS_BITS)
    if (last_gain) LC3_VQ_TRY(2, 3, LC3T_SNS_VQ_NEAR_ADJ_GAINS_BITS)
    LC3_VQ_NORMALIZE(y3, 16)
    LC3_VQ_TRY(3, 0, LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS)
    LC3_VQ_TRY(3, 1, LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS)
    LC3_VQ_TRY(3, 2, LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS)
    LC3_VQ_TRY(3, 3, LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS)
    LC3_VQ_TRY(3, 4, LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS)
    LC3_VQ_TRY(3, 5, LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS)
    LC3_VQ_TRY(3, 6, LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS)
    if (last_gain) LC3_VQ_TRY(3, 7, LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS)
    // mvpq_enum :584-629 of the selected shape
    uint32_t idxa = 0, idxb = 0, joint;
    int ls_inda = 0, ls_indb = 0;
    const int lsb_gain = gind & 1;
    if (shape_j == 0) {
        lc3_vq_enum<10>(v, y0, 0, idxa, ls_inda);
        lc3_vq_enum<6>(v, y0, 10, idxb, ls_indb);
        joint = (2u * idxb + (uint32_t)ls_indb + 2u) * 2390004u + idxa;
    } else if (shape_j == 1) {
        lc3_vq_enum<10>(v, y1, 0, idxa, ls_inda);
        joint = (uint32_t)lsb_gain * 2390004u + idxa;
    } else if (shape_j == 2) {
        lc3_vq_enum<16>(v, y2, 0, idxa, ls_inda);
        joint = idxa;
    } else {
        lc3_vq_enum<16>(v, y3, 0, idxa, ls_inda);
        joint = 15158272u + (uint32_t)lsb_gain + (2u * idxa);
    }
    const int st = v.stride;
    v.plane[EP_IND_LF * st] = ind_lf;
    v.plane[EP_IND_HF * st] = ind_hf;
    v.plane[EP_SHAPE_J * st] = shape_j;
    v.plane[EP_GIND * st] = gind;
    v.plane[EP_LS_INDA * st] = ls_inda;
    v.plane[EP_JOINT * st] = (int32_t)joint;
    (void)ls_indb;
    // synthesis :552-559, interpolation :163-201, gains g = exp2(-scf) :254-257
    float q[16];
#pragma unroll
    for (int n = 0; n < 16; n++) {
        float factor = 0.0f;
#pragma unroll
        for (int col = 0; col < 16; col++) factor += xq_sel[col] * lc3_f(&LC3T_D_BITS[n][0], col);
        q[n] = st1[n] + g_sel * factor;
    }
    const int diff = 64 - v.nb;
    if (diff == 0) {
        lc3_f4 *g4 = (lc3_f4 *)v.gains;  // 64 gains as sixteen 128-bit stores
#pragma unroll
        for (int b = 0; b < 64; b += 4) {
            lc3_f4 o;
            o.x = lc3_exp2f(-lc3_vq_interp(q, b));
            o.y = lc3_exp2f(-lc3_vq_interp(q, b + 1));
            o.z = lc3_exp2f(-lc3_vq_interp(q, b + 2));
            o.w = lc3_exp2f(-lc3_vq_interp(q, b + 3));
            g4[b / 4] = o;
        }
    } else {  // :185-201 (encoder form); only 8 kHz has nb < 64 and the reference cannot encode 8 kHz
        float f[64];
#pragma unroll
        for (int b = 0; b < 64; b++) f[b] = lc3_vq_interp(q, b);
        for (int b = 0; b < 64; b++) {
            float val = 0.0f;
#pragma unroll
            for (int u = 0; u < 64; u++) {  // static indices only: f lives in registers
                if (b < diff && u == 2 * b) val = (f[u] + f[u + 1 < 64 ? u + 1 : 63]) / 2.0f;
                if (b >= diff && u == diff + 1) val = f[u];
            }
            v.gains[b] = lc3_exp2f(-val);
        }
    }
}
