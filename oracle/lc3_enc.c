/* ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See lc3_oracle.h.
 * Restates reference src/encoder/ (all stage modules) stage by stage; every f32 expression is
 * evaluated in the reference's order (compile with -ffp-contract=off). */
#include "lc3_oracle.h"
#include "lc3_math.h"
#include <math.h>
#include <string.h>

/* path statistic for the measurement notes (tools/quantiser_paths.py): [0] frames through the quantiser, [1] frames whose gain
 * adjustment changes the gain (spectral_quantization.rs:103-107: a SECOND quantise + bit-count pass), [2] frames with at least one
 * active TNS filter, [3] lsb_mode frames.  Off unless a tool switches it on; atomic adds (encode threads share the counters) */
long lc3o_enc_path_count[4];
int lc3o_enc_path_counting = 0;
#define LC3O_ENC_COUNT(i) do { if (lc3o_enc_path_counting) __sync_fetch_and_add(&lc3o_enc_path_count[i], 1L); } while (0)

#define LC3_TABLE_QUAL static const
#include "../tables/lc3_tables.h"
#define TF(name) ((const float *)(const void *)LC3T_##name##_BITS)

static const float *mdct_window(const lc3o_config *c) {
    if (c->n_ms_10) {
        switch (c->fs_ind) {
        case 0: return TF(W_N80_10MS);
        case 1: return TF(W_N160_10MS);
        case 2: return TF(W_N240_10MS);
        case 3: return TF(W_N320_10MS);
        default: return TF(W_N480_10MS);
        }
    }
    switch (c->fs_ind) {
    case 0: return TF(W_N60_7P5MS);
    case 1: return TF(W_N120_7P5MS);
    case 2: return TF(W_N180_7P5MS);
    case 3: return TF(W_N240_7P5MS);
    default: return TF(W_N360_7P5MS);
    }
}
static const uint16_t *band_index(const lc3o_config *c) {
    if (c->n_ms_10) {
        switch (c->fs_ind) {
        case 0: return LC3T_I_8000_10MS;
        case 1: return LC3T_I_16000_10MS;
        case 2: return LC3T_I_24000_10MS;
        case 3: return LC3T_I_32000_10MS;
        default: return LC3T_I_48000_10MS;
        }
    }
    switch (c->fs_ind) {
    case 0: return LC3T_I_8000_7P5MS;
    case 1: return LC3T_I_16000_7P5MS;
    case 2: return LC3T_I_24000_7P5MS;
    case 3: return LC3T_I_32000_7P5MS;
    default: return LC3T_I_48000_7P5MS;
    }
}
/* exported for the decoder translation unit */
const float *lc3o_mdct_window(const lc3o_config *c) { return mdct_window(c); }
const uint16_t *lc3o_band_index(const lc3o_config *c) { return band_index(c); }

/* ================================================================= MDCT (encoder/modified_dct.rs) */
/* modified_dct.rs:108-177 ; returns near_nyquist flag */
int lc3o_enc_mdct_run(lc3o_encoder *e, const int16_t *x_s, float *out, float *energy_bands) {
    const lc3o_config *c = &e->cfg;
    int nf = c->nf, z = c->z, h = nf / 2, mid = 3 * h, k, b;
    const float *w = mdct_window(c);
    const uint16_t *ifs = band_index(c);
    int16_t *t = e->tbuf;
    float gain;

    /* update_time_buffer :126-138 */
    memmove(t, t + nf, sizeof(int16_t) * (size_t)(nf - z));
    memcpy(t + (nf - z), x_s, sizeof(int16_t) * (size_t)nf);

    /* apply_mdct :73-105 */
    for (k = 0; k < h; k++)
        out[k] = -((float)t[mid - 1 - k] * w[mid - 1 - k]) - ((float)t[mid + k] * w[mid + k]);
    for (k = 0; k < h; k++)
        out[h + k] = ((float)t[k] * w[k]) - ((float)t[nf - 1 - k] * w[nf - 1 - k]);
    lc3o_dct4_run(&e->dct, out);
    gain = 1.0f / sqrtf(2.0f * (float)nf);
    for (k = 0; k < nf; k++) out[k] *= gain;

    /* apply_energy_estimation :140-152 */
    for (b = 0; b < c->nb; b++) {
        int from = ifs[b], to = ifs[b + 1];
        float width = (float)(to - from), acc = 0.0f;
        for (k = from; k < to; k++) acc += out[k] * out[k] / width;
        energy_bands[b] = acc;
    }

    /* is_near_nyquist :154-177 */
    if (c->fs <= 32000) {
        int nn_idx = c->n_ms_10 ? c->nb - 2 : c->nb - 4;
        float lower = 0.0f, upper = 0.0f;
        for (b = 0; b < c->nb; b++) {
            if (b < nn_idx) lower += energy_bands[b];
            else upper += energy_bands[b];
        }
        return upper > 30.0f * lower;
    }
    return 0;
}

/* ================================================================= bandwidth detector */
/* encoder/bandwidth_detector.rs:5-18,64-127.  fs_ind == 0 returns early (:66-71); the reference
 * constructor panics for 8 kHz (:36-37), so that case has no reference behaviour. */
lc3o_bw_result lc3o_enc_bandwidth(const lc3o_config *c, const float *e_b) {
    static const int START10[4][4] = {{53, 0, 0, 0}, {47, 59, 0, 0}, {44, 54, 60, 0}, {41, 51, 57, 61}};
    static const int STOP10[4][4] = {{63, 0, 0, 0}, {56, 63, 0, 0}, {52, 59, 63, 0}, {49, 55, 60, 63}};
    static const int START75[4][4] = {{51, 0, 0, 0}, {45, 58, 0, 0}, {42, 53, 60, 0}, {40, 51, 57, 61}};
    static const int STOP75[4][4] = {{63, 0, 0, 0}, {55, 63, 0, 0}, {51, 58, 63, 0}, {48, 55, 60, 63}};
    static const int NBITS_BW[5] = {0, 1, 2, 2, 3};
    static const int TQ[4] = {20, 10, 10, 10};
    static const int TC[4] = {15, 23, 20, 20};
    static const int L10[4] = {4, 4, 3, 1};
    static const int L75[4] = {4, 4, 3, 2};
    lc3o_bw_result r;
    int fsi = c->fs_ind, k, n, bw = 0;
    const int *start, *stop, *l;
    r.nbits_bandwidth = NBITS_BW[fsi];
    r.bandwidth_ind = 0;
    if (fsi == 0) return r;
    start = c->n_ms_10 ? START10[fsi - 1] : START75[fsi - 1];
    stop = c->n_ms_10 ? STOP10[fsi - 1] : STOP75[fsi - 1];
    l = c->n_ms_10 ? L10 : L75;
    for (k = fsi - 1; k >= 0; k--) {
        float width = (float)(stop[k] + 1 - start[k]), quiet = 0.0f;
        for (n = start[k]; n <= stop[k]; n++) quiet += e_b[n] / width;
        if (quiet >= (float)TQ[k]) {
            bw = k + 1;
            break;
        }
    }
    if (fsi == bw) {
        r.bandwidth_ind = bw;
        return r;
    } else {
        float cutoff_max = 0.0f;
        int l_bw = l[bw];
        int from = start[bw] + 1 - l_bw, to = start[bw];
        for (n = from; n < to; n++) {
            float cutoff = e_b[n - l_bw] / e_b[n];
            if (c->spec_flags & LC3O_SPEC_BW_CUTOFF_DB) cutoff = 10.0f * lc3m_log10f(1.1920929e-7f + cutoff);
            cutoff_max = lc3m_maxf(cutoff, cutoff_max);
        }
        r.bandwidth_ind = cutoff_max > (float)TC[bw] ? bw : fsi;
        return r;
    }
}

/* ================================================================= attack detector */
/* encoder/attack_detector.rs:45-128 */
int lc3o_enc_attack(const lc3o_config *c, lc3o_attack_state *st, const int16_t *x_s, int nbytes) {
    int num_ds = c->n_ms_10 ? 160 : 120, num_blocks = c->n_ms_10 ? 4 : 3, limit = c->n_ms_10 ? 2 : 1;
    int active, n, j, block_len, attack_position = -1, detected;
    int32_t ds[160];
    float hp[160], tm1, tm2;
    if (c->fs < 32000) active = 0;
    else if (!c->n_ms_10)
        active = (c->fs == 32000 && nbytes >= 61 && nbytes < 150) || (c->fs >= 44100 && nbytes >= 75 && nbytes < 150);
    else active = (c->fs == 32000 && nbytes > 80) || (c->fs >= 41000 && nbytes >= 100);
    if (!active) {
        st->energy_last = 0.0f;
        st->max_energy_last = 0.0f;
        st->attack_pos_last = -1;
        return 0;
    }
    block_len = c->nf / num_ds;
    for (n = 0; n < num_ds; n++) {
        int32_t acc = 0;
        for (j = 0; j < block_len; j++) acc += (int32_t)x_s[block_len * n + j];
        ds[n] = acc;
    }
    tm1 = (float)st->ds_tm1;
    tm2 = (float)st->ds_tm2;
    hp[0] = 0.375f * (float)ds[0] - 0.5f * tm1 + 0.125f * tm2;
    hp[1] = 0.375f * (float)ds[1] - 0.5f * (float)ds[0] + 0.125f * tm1;
    for (n = 2; n < num_ds; n++) hp[n] = 0.375f * (float)ds[n] - 0.5f * (float)ds[n - 1] + 0.125f * (float)ds[n - 2];
    st->ds_tm1 = ds[num_ds - 1];
    st->ds_tm2 = ds[num_ds - 2];
    for (n = 0; n < num_blocks; n++) {
        float energy = 0.0f, max_energy;
        for (j = 40 * n; j < 40 * n + 40; j++) energy += hp[j] * hp[j];
        max_energy = lc3m_maxf(0.25f * st->max_energy_last, st->energy_last);
        if (energy > 8.5f * max_energy) attack_position = n;
        st->energy_last = energy;
        st->max_energy_last = max_energy;
    }
    detected = attack_position >= 0 || st->attack_pos_last >= limit;
    st->attack_pos_last = attack_position;
    return detected;
}

/* ================================================================= SNS (encoder/spectral_noise_shaping.rs) */
/* :285-316 -- corr_xy / energy_y are written through on every probe */
static void add_unit_pulse(const float *abs_x, int n_max, int k, int k_max, int32_t *cand, float *corr_xy,
                           float *energy_y) {
    float corr_last = *corr_xy, en_last = *energy_y;
    int it, n_c;
    for (it = k; it < k_max; it++) {
        int n_best = 0;
        float best_corr_sq, best_en;
        *corr_xy = corr_last + abs_x[0];
        best_corr_sq = *corr_xy * *corr_xy;
        best_en = en_last + 2.0f * (float)cand[0] + 1.0f;
        for (n_c = 1; n_c < n_max; n_c++) {
            *corr_xy = corr_last + abs_x[n_c];
            *energy_y = en_last + 2.0f * (float)cand[n_c] + 1.0f;
            if (*corr_xy * *corr_xy * best_en > best_corr_sq * *energy_y) {
                n_best = n_c;
                best_corr_sq = *corr_xy * *corr_xy;
                best_en = *energy_y;
            }
        }
        corr_last += abs_x[n_best];
        en_last += 2.0f * (float)cand[n_best] + 1.0f;
        cand[n_best] += 1;
    }
}

/* :632-648 */
static void normalize_candidate(const int32_t *y, float *xq, int n_max) {
    float norm = 0.0f;
    int n;
    for (n = 0; n < n_max; n++)
        if (y[n] != 0) norm += (float)y[n] * (float)y[n];
    norm = sqrtf(norm);
    for (n = 0; n < n_max; n++) {
        xq[n] = (float)y[n];
        if (y[n] != 0) xq[n] /= norm;
    }
    for (n = n_max; n < 16; n++) xq[n] = 0.0f;
}

/* :584-629 */
static void mvpq_enum(uint32_t *index, int32_t *lead_sign_ind, int dim_in, const int32_t *vec_in) {
    int32_t next_sign_ind = INT32_MIN;
    int k_val_acc = 0, n = 0, pos;
    uint32_t tmp_h_row = LC3T_MPVQ_OFFSETS[0][0];
    *index = 0;
    for (pos = dim_in - 1; pos >= 0; pos--) {
        int tmp_val = (int8_t)vec_in[pos];
        /* enc_push_sign :616-629 */
        if ((next_sign_ind & INT32_MIN) == 0 && tmp_val != 0) *index = 2 * *index + (uint32_t)next_sign_ind;
        if (tmp_val < 0) next_sign_ind = 1;
        else if (tmp_val > 0) next_sign_ind = 0;
        *index += tmp_h_row;
        k_val_acc += tmp_val < 0 ? -tmp_val : tmp_val;
        if (pos != 0) n += 1;
        tmp_h_row = k_val_acc >= 11 ? LC3T_MPVQ_OFFSETS[n + 1][k_val_acc % 11] : LC3T_MPVQ_OFFSETS[n][k_val_acc];
    }
    *lead_sign_ind = next_sign_ind;
}

/* run_quant :569-581 = sns_quant_stage1 :318-361 + sns_quant_stage2 :363-567 */
void lc3o_enc_sns_quant(const float *scf, float *scfq, lc3o_sns_result *res) { lc3o_enc_sns_quant_spec(scf, scfq, res, 0); }
void lc3o_enc_sns_quant_spec(const float *scf, float *scfq, lc3o_sns_result *res, int spec_flags) {
    const float *LFCB = TF(LFCB), *HFCB = TF(HFCB), *D = TF(D);
    float st1[16], r1[16], t2rot[16], abs_x[16];
    int32_t y0[16], y1[16], y2[16], y3[16];
    float xq[4][16];
    float dmse_lf_min = INFINITY, dmse_hf_min = INFINITY;
    int ind_lf = 0, ind_hf = 0, i, n, col, j;

    /* stage 1 */
    for (i = 0; i < 32; i++) {
        float dmse_lf = 0.0f, dmse_hf = 0.0f;
        for (n = 0; n < 8; n++) {
            dmse_lf += (scf[n] - LFCB[i * 8 + n]) * (scf[n] - LFCB[i * 8 + n]);
            dmse_hf += (scf[8 + n] - HFCB[i * 8 + n]) * (scf[8 + n] - HFCB[i * 8 + n]);
        }
        if (dmse_lf < dmse_lf_min) { ind_lf = i; dmse_lf_min = dmse_lf; }
        if (dmse_hf < dmse_hf_min) { ind_hf = i; dmse_hf_min = dmse_hf; }
    }
    for (n = 0; n < 8; n++) {
        st1[n] = LFCB[ind_lf * 8 + n];
        st1[8 + n] = HFCB[ind_hf * 8 + n];
    }
    for (n = 0; n < 16; n++) r1[n] = scf[n] - st1[n];

    /* stage 2: t2rot = r1 * D, accumulated row by row (:378-384) */
    for (n = 0; n < 16; n++) t2rot[n] = 0.0f;
    for (i = 0; i < 16; i++)
        for (col = 0; col < 16; col++) t2rot[col] += r1[i] * D[i * 16 + col];

    {
        int k = 0, k_max = 6, n_max = 16;
        float abs_sum = 0.0f, proj, corr_xy = 0.0f, energy_y = 0.0f;
        int32_t ks;
        for (n = 0; n < 16; n++) {
            abs_x[n] = fabsf(t2rot[n]);
            abs_sum += abs_x[n];
        }
        proj = ((float)k_max - 1.0f) / abs_sum;
        for (n = 0; n < 16; n++) {
            y3[n] = lc3m_f32_to_i32(floorf(abs_x[n] * proj));
            if (y3[n] != 0) {
                k += y3[n]; /* `as usize` of a positive i32 */
                corr_xy += (float)y3[n] * abs_x[n];
                energy_y += (float)y3[n] * (float)y3[n];
            }
        }
        /* shape 3: K = 6 over 16 */
        add_unit_pulse(abs_x, n_max, k, k_max, y3, &corr_xy, &energy_y);
        /* shape 2: K = 8 over 16 */
        memcpy(y2, y3, sizeof(y2));
        add_unit_pulse(abs_x, 16, 6, 8, y2, &corr_xy, &energy_y);
        /* shape 1: K = 10 over 10 */
        for (n = 0; n < 10; n++) y1[n] = y2[n];
        for (n = 10; n < 16; n++) y1[n] = 0;
        ks = 8;
        for (n = 10; n < 16; n++) {
            if (y2[n] != 0) {
                ks -= y2[n];
                corr_xy -= (float)y2[n] * abs_x[n];
                energy_y -= (float)y2[n] * (float)y2[n];
            }
        }
        add_unit_pulse(abs_x, 10, (int)ks, 10, y1, &corr_xy, &energy_y);
        /* shape 0: one pulse in set B */
        {
            float max_abs = 0.0f;
            int n_best = 0;
            for (n = 0; n < 10; n++) y0[n] = y1[n];
            for (n = 10; n < 16; n++) {
                y0[n] = 0;
                if (abs_x[n] > max_abs) {
                    max_abs = abs_x[n];
                    n_best = n;
                }
            }
            y0[n_best] = 1;
        }
    }
    /* signs :456-482 */
    for (n = 0; n < 10; n++)
        if (t2rot[n] < 0.0f) { y0[n] *= -1; y1[n] *= -1; y2[n] *= -1; y3[n] *= -1; }
    for (n = 10; n < 16; n++)
        if (t2rot[n] < 0.0f) { y0[n] *= -1; y2[n] *= -1; y3[n] *= -1; }

    normalize_candidate(y0, xq[0], 16);
    normalize_candidate(y1, xq[1], 10);
    normalize_candidate(y2, xq[2], 16);
    normalize_candidate(y3, xq[3], 16);

    {
        /* gain search excludes the last gain of every shape */
        static const int g_maxind_ref[4] = {1, 3, 3, 7}, g_maxind_all[4] = {2, 4, 4, 8};
        const int *g_maxind = (spec_flags & LC3O_SPEC_SNS_LAST_GAIN) ? g_maxind_all : g_maxind_ref;
        const float *gains[4];
        int shape_j = 0, gind = 0;
        float g_sel = 0.0f, d_mse_min = INFINITY;
        const float *xq_sel = xq[0];
        uint32_t idxa = 0, idxb = 0, index_joint;
        int32_t ls_inda = 0, ls_indb = 0;
        int lsb_gain;
        gains[0] = TF(SNS_VQ_REG_ADJ_GAINS);
        gains[1] = TF(SNS_VQ_REG_LF_ADJ_GAINS);
        gains[2] = TF(SNS_VQ_NEAR_ADJ_GAINS);
        gains[3] = TF(SNS_VQ_FAR_ADJ_GAINS);
        for (j = 0; j < 4; j++) {
            for (i = 0; i < g_maxind[j]; i++) {
                float g = gains[j][i], d_mse = 0.0f;
                for (n = 0; n < 16; n++) {
                    float diff = t2rot[n] - g * xq[j][n];
                    d_mse += diff * diff;
                }
                if (d_mse < d_mse_min) {
                    shape_j = j;
                    gind = i;
                    d_mse_min = d_mse;
                    g_sel = g;
                    xq_sel = xq[j];
                }
            }
        }
        lsb_gain = gind & 1;
        switch (shape_j) {
        case 0:
            mvpq_enum(&idxa, &ls_inda, 10, y0);
            mvpq_enum(&idxb, &ls_indb, 6, y0 + 10);
            index_joint = (2u * idxb + (uint32_t)ls_indb + 2u) * 2390004u + idxa;
            break;
        case 1:
            mvpq_enum(&idxa, &ls_inda, 10, y1);
            index_joint = (uint32_t)lsb_gain * 2390004u + idxa;
            break;
        case 2:
            mvpq_enum(&idxa, &ls_inda, 16, y2);
            index_joint = idxa;
            break;
        default:
            mvpq_enum(&idxa, &ls_inda, 16, y3);
            index_joint = 15158272u + (uint32_t)lsb_gain + (2u * idxa);
            break;
        }
        /* synthesis :552-559 */
        for (n = 0; n < 16; n++) {
            float factor = 0.0f;
            for (col = 0; col < 16; col++) factor += xq_sel[col] * D[n * 16 + col];
            scfq[n] = st1[n] + g_sel * factor;
        }
        res->ind_lf = ind_lf;
        res->ind_hf = ind_hf;
        res->shape_j = shape_j;
        res->gind = gind;
        res->ls_inda = ls_inda;
        res->ls_indb = ls_indb;
        res->index_joint_j = index_joint;
    }
}

/* SpectralNoiseShaping::run :203-282 */
lc3o_sns_result lc3o_enc_sns(const lc3o_config *c, float *x, const float *e_b, int attack) {
    static const int G_TILT[5] = {14, 18, 22, 26, 30};
    float W[6];
    float padded[64], eb[64], ds[16], scf[16], scfq[16], interp[64];
    const uint16_t *ifs = band_index(c);
    lc3o_sns_result res;
    int b, k, n, diff = 64 - c->nb;
    float exponent, total, noise_floor, avg;

    /* WEIGHTING :59 -- f32 constant divisions */
    W[0] = 1.0f / 12.0f; W[1] = 2.0f / 12.0f; W[2] = 3.0f / 12.0f;
    W[3] = 3.0f / 12.0f; W[4] = 2.0f / 12.0f; W[5] = 1.0f / 12.0f;

    /* apply_padding_for_narrow_band :75-90 */
    if (diff > 0) {
        for (b = 0; b < diff; b++) {
            padded[2 * b] = e_b[b];
            padded[2 * b + 1] = e_b[b];
        }
        /* NOTE: reference indexes output[2*diff + i] for i < num_bands; entries past 63 would panic in
         * Rust (8 kHz / 7.5 ms encode is unreachable in the reference anyway) -- clamp. */
        for (b = 0; b < c->nb && 2 * diff + b < 64; b++) padded[2 * diff + b] = e_b[diff + b];
    } else {
        memcpy(padded, e_b, sizeof(padded));
    }
    /* energy_band_smoothing :92-98 */
    eb[0] = 0.75f * padded[0] + 0.25f * padded[1];
    for (b = 1; b < 63; b++) eb[b] = 0.25f * padded[b - 1] + 0.5f * padded[b] + 0.25f * padded[b + 1];
    eb[63] = 0.25f * padded[62] + 0.75f * padded[63];
    /* pre-emphasis :214-219 */
    exponent = (float)G_TILT[c->fs_ind] / 630.0f;
    for (b = 0; b < 64; b++) eb[b] *= lc3m_powf(10.0f, (float)b * exponent);
    /* noise floor :221-228 */
    total = 0.0f;
    for (b = 0; b < 64; b++) total += eb[b];
    total = (total / 64.0f) * lc3m_powi(10.0f, -4);
    noise_floor = lc3m_maxf(lc3m_powi(2.0f, -32), total);
    for (b = 0; b < 64; b++) eb[b] = lc3m_maxf(eb[b], noise_floor);
    /* logarithm :230-233 */
    for (b = 0; b < 64; b++) eb[b] = lc3m_log2f(1.1920929e-7f + eb[b]) / 2.0f;
    /* downsample :100-124 */
    ds[0] = W[0] * eb[0];
    for (k = 1; k < 6; k++) ds[0] += W[k] * eb[k - 1];
    for (b = 1; b < 15; b++) {
        ds[b] = 0.0f;
        for (k = 0; k < 6; k++) ds[b] += W[k] * eb[4 * b - 1 + k];
    }
    ds[15] = W[5] * eb[63];
    for (k = 0; k < 5; k++) ds[15] += W[k] * eb[60 + k - 1];
    /* mean_removal_and_scaling :126-132 */
    total = 0.0f;
    for (n = 0; n < 16; n++) total += ds[n];
    avg = total / 16.0f;
    for (n = 0; n < 16; n++) ds[n] = 0.85f * (ds[n] - avg);
    /* attack_handling :134-161 */
    if (attack) {
        float att;
        scf[0] = (ds[0] + ds[1] + ds[2]) / 3.0f;
        scf[1] = (ds[0] + ds[1] + ds[2] + ds[3]) / 4.0f;
        for (n = 2; n < 14; n++) {
            float wt = 0.0f;
            for (k = n - 2; k < n + 3; k++) wt += ds[k];
            scf[n] = wt / 5.0f;
        }
        scf[14] = (ds[12] + ds[13] + ds[14] + ds[15]) / 4.0f;
        scf[15] = (ds[13] + ds[14] + ds[15]) / 3.0f;
        total = 0.0f;
        for (n = 0; n < 16; n++) total += scf[n];
        avg = total / 16.0f;
        att = c->n_ms_10 ? 0.5f : 0.3f;
        for (n = 0; n < 16; n++) scf[n] = att * (scf[n] - avg);
    } else {
        memcpy(scf, ds, sizeof(scf));
    }
    lc3o_enc_sns_quant_spec(scf, scfq, &res, c->spec_flags);
    /* apply_scale_factor_interpolation :163-183 */
    interp[0] = scfq[0];
    interp[1] = scfq[0];
    for (n = 0; n < 15; n++) {
        float in0 = scfq[n], d = scfq[n + 1] - scfq[n];
        interp[2 + 4 * n + 0] = in0 + (0.125f * d);
        interp[2 + 4 * n + 1] = in0 + (0.375f * d);
        interp[2 + 4 * n + 2] = in0 + (0.625f * d);
        interp[2 + 4 * n + 3] = in0 + (0.875f * d);
    }
    interp[62] = scfq[15] + (0.125f * (scfq[15] - scfq[14]));
    interp[63] = scfq[15] + (0.375f * (scfq[15] - scfq[14]));
    /* reduce_scale_factors_for_narrow_band :185-201 */
    if (diff > 0) {
        for (b = 0; b < diff; b++) interp[b] = (interp[2 * b] + interp[2 * b + 1]) / 2.0f;
        for (b = diff; b < c->nb; b++) interp[b] = interp[diff + 1];
    }
    for (b = 0; b < 64; b++) interp[b] = lc3m_exp2f(-interp[b]);
    /* spectral shaping :264-268: zip(interpolated[64], band windows) -> nb bands */
    for (b = 0; b < c->nb && b < 64; b++)
        for (k = ifs[b]; k < ifs[b + 1]; k++) x[k] *= interp[b];
    return res;
}

/* ================================================================= TNS (encoder/temporal_noise_shaping.rs) */
typedef struct { int num, start[2], stop[2], sub_start[2][3], sub_stop[2][3]; } tns_params;
static const tns_params TNS10[5] = { /* :119-155 */
    {1, {12, 160}, {80, 0}, {{12, 34, 57}, {0, 0, 0}}, {{34, 57, 80}, {0, 0, 0}}},
    {1, {12, 160}, {160, 0}, {{12, 61, 110}, {0, 0, 0}}, {{61, 110, 160}, {0, 0, 0}}},
    {1, {12, 160}, {200, 0}, {{12, 88, 164}, {0, 0, 0}}, {{88, 164, 240}, {0, 0, 0}}}, /* SURVEY A5 */
    {2, {12, 160}, {160, 320}, {{12, 61, 110}, {160, 213, 266}}, {{61, 110, 160}, {213, 266, 320}}},
    {2, {12, 200}, {200, 400}, {{12, 74, 137}, {200, 266, 333}}, {{74, 137, 200}, {266, 333, 400}}},
};
static const tns_params TNS75[5] = { /* :160-196 */
    {1, {9, 120}, {60, 0}, {{9, 26, 43}, {0, 0, 0}}, {{26, 43, 60}, {0, 0, 0}}},
    {1, {9, 120}, {120, 0}, {{9, 46, 83}, {0, 0, 0}}, {{46, 83, 120}, {0, 0, 0}}},
    {1, {9, 120}, {180, 0}, {{9, 66, 123}, {0, 0, 0}}, {{66, 123, 180}, {0, 0, 0}}},
    {2, {9, 120}, {120, 240}, {{9, 46, 82}, {120, 159, 200}}, {{46, 82, 120}, {159, 200, 240}}},
    {2, {9, 150}, {150, 300}, {{9, 56, 103}, {150, 200, 250}}, {{56, 103, 150}, {200, 250, 300}}},
};

static int tns_to_int(float x) { /* :343-349 */
    if (x >= 0.0f) return lc3m_f32_to_i8(x + 0.5f);
    return lc3m_f32_to_i8(-(-x + 0.5f));
}

lc3o_tns_result lc3o_enc_tns(const lc3o_config *c, float *x_s, int p_bw, int nbits, int near_nyquist) {
    /* lag window :81-84, f32 literals */
    static const float LAGW[9] = {1.0f, 0.9980280260203829f, 0.9921354055113971f, 0.9823915844707989f,
                                  0.9689107911912967f, 0.9518498073692735f, 0.9314049334023056f,
                                  0.9078082299969592f, 0.8813231366694713f};
    tns_params tp_v = c->n_ms_10 ? TNS10[p_bw] : TNS75[p_bw];
    const tns_params *tp = &tp_v;
    const float step = (float)3.14159265358979323846 / 17.0f; /* PI as f32 / 17.0 :268 */
    lc3o_tns_result res;
    int f, k, n, s, ne = c->ne;
    memset(&res, 0, sizeof(res));
    if ((c->spec_flags & LC3O_SPEC_TNS_SSWB_STOP) && c->n_ms_10 && p_bw == 2) tp_v.stop[0] = 240; /* SURVEY A5 corrected */
    res.num_tns_filters = tp->num;
    res.lpc_weighting = c->n_ms_10 ? (nbits < 480) : (nbits < 360);

    for (f = 0; f < tp->num; f++) {
        float r[9], a_mem[2][9], *a = a_mem[0], *a_last = a_mem[1], e, pred_gain;
        /* compute_normalized_autocorrelation :80-115 */
        for (k = 0; k < 9; k++) {
            float r0 = k == 0 ? 3.0f : 0.0f, rk = 0.0f, e_prod = 1.0f;
            for (s = 0; s < 3; s++) {
                int start = tp->sub_start[f][s], stop = tp->sub_stop[f][s], k_from = start + k;
                float es = 0.0f, ac = 0.0f;
                for (n = start; n < stop; n++) es += x_s[n] * x_s[n];
                if (k_from < ne && k_from < stop)
                    for (n = 0; k_from + n < stop; n++) ac += x_s[start + n] * x_s[k_from + n];
                e_prod *= es;
                rk += ac / es;
            }
            r[k] = (e_prod == 0.0f ? r0 : rk) * LAGW[k];
        }
        /* tns_analysis :204-265 */
        memset(a_mem, 0, sizeof(a_mem));
        e = r[0];
        a[0] = 1.0f;
        for (k = 1; k < 9; k++) {
            float rc = 0.0f, *tmp = a_last;
            a_last = a;
            a = tmp;
            for (n = 0; n < k; n++) rc -= a_last[n] * r[k - n];
            if (e != 0.0f) rc /= e;
            a[0] = 1.0f;
            for (n = 1; n < k; n++) a[n] = a_last[n] + rc * a_last[k - n];
            a[k] = rc;
            e *= 1.0f - rc * rc;
        }
        pred_gain = e == 0.0f ? r[0] : r[0] / e;
        if (pred_gain > 1.5f && !near_nyquist) {
            float gamma = 1.0f, *rc = res.rc_q + f * 8, *a_k = a, *a_km1 = a_last;
            if (res.lpc_weighting > 0 && pred_gain < 2.0f)
                gamma -= (1.0f - 0.85f) * (2.0f - pred_gain) / (2.0f - 1.5f);
            for (k = 0; k < 9; k++) a[k] *= lc3m_powi(gamma, k);
            for (k = 8; k >= 1; k--) {
                float ee, *tmp;
                rc[k - 1] = a_k[k];
                ee = 1.0f - rc[k - 1] * rc[k - 1];
                for (n = 1; n < k; n++) {
                    a_km1[n] = a_k[n] - rc[k - 1] * a_k[k - n];
                    a_km1[n] /= ee;
                }
                tmp = a_k;
                a_k = a_km1;
                a_km1 = tmp;
            }
        } else {
            for (k = 0; k < 8; k++) res.rc_q[f * 8 + k] = 0.0f;
        }
    }
    /* apply_quantization :267-292 */
    for (f = 0; f < tp->num; f++) {
        for (k = 0; k < 8; k++) {
            int ri = tns_to_int(lc3m_asinf(res.rc_q[f * 8 + k]) / step) + 8;
            res.rc_i[f * 8 + k] = ri;
            res.rc_q[f * 8 + k] = lc3m_sinf(step * ((float)ri - 8.0f));
        }
        k = 7;
        while (k >= 0 && res.rc_i[f * 8 + k] == 8) k--;
        res.rc_order[f] = k + 1;
    }
    for (f = tp->num; f < 2; f++) {
        for (k = 0; k < 8; k++) {
            res.rc_i[f * 8 + k] = 8;
            res.rc_q[f * 8 + k] = 0.0f;
        }
        res.rc_order[f] = 0;
    }
    /* calc_bit_budget :294-311 */
    for (f = 0; f < tp->num; f++) {
        int order_bits = res.rc_order[f] != 0 ? LC3T_AC_TNS_ORDER_BITS[res.lpc_weighting][res.rc_order[f] - 1] : 0;
        int coef_bits = 0;
        for (k = 0; k < res.rc_order[f]; k++) {
            int ri = res.rc_i[f * 8 + k];
            coef_bits += LC3T_AC_TNS_COEF_BITS[k][ri < 0 ? 0 : (ri > 16 ? 16 : ri)];
        }
        res.nbits_tns += (int)ceilf((2048.0f + (float)order_bits + (float)coef_bits) / 2048.0f);
    }
    if (res.rc_order[0] != 0 || res.rc_order[1] != 0) LC3O_ENC_COUNT(2);
    /* apply_filtering :313-340 -- lattice state shared across both filters */
    {
        float st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (f = 0; f < tp->num; f++) {
            if (res.rc_order[f] != 0) {
                int prev_order = res.rc_order[f] - 1;
                for (n = tp->start[f]; n < tp->stop[f]; n++) {
                    float t = x_s[n], st_save = t;
                    for (k = 0; k < prev_order; k++) {
                        float rcq = res.rc_q[f * 8 + k];
                        float st_tmp = rcq * t + st[k];
                        t += rcq * st[k];
                        st[k] = st_save;
                        st_save = st_tmp;
                    }
                    t += res.rc_q[f * 8 + prev_order] * st[prev_order];
                    st[prev_order] = st_save;
                    x_s[n] = t;
                }
            }
        }
    }
    return res;
}

/* ================================================================= LTPF (encoder/long_term_post_filter.rs) */
#define NMEM 232
#define K_MIN 17
#define K_MAX 114

static void ltpf_fields(const lc3o_config *c, int *len12, int *len6, int *delay, int *p, float *rf, int *xs_len) {
    *len12 = c->n_ms_10 ? 128 : 96;
    *len6 = c->n_ms_10 ? 64 : 48;
    *delay = c->n_ms_10 ? 24 : 44;
    switch (c->fs) { /* :105-113 */
    case 8000: *p = 24; *rf = 0.5f; *xs_len = 10 + c->nf; break;
    case 16000: *p = 12; *rf = 1.0f; *xs_len = 20 + c->nf; break;
    case 24000: *p = 8; *rf = 1.0f; *xs_len = 30 + c->nf; break;
    case 32000: *p = 6; *rf = 1.0f; *xs_len = 40 + c->nf; break;
    default: *p = 4; *rf = 1.0f; *xs_len = 60 + c->nf; break;
    }
}

void lc3o_ltpf_enc_init(const lc3o_config *c, lc3o_ltpf_enc_state *st) {
    (void)c;
    memset(st, 0, sizeof(*st));
    st->t_prev = K_MIN; /* :84 */
}

static float ltpf_norm_value(const lc3o_ltpf_enc_state *st, int lag, int len6) { /* :445-455 */
    float v = 0.0f;
    int n, from = K_MAX - lag;
    for (n = from; n < from + len6; n++) v += st->x6[n] * st->x6[n];
    return v;
}
static int index_of_max(const float *s, int n) { /* :427-443 */
    int i, idx = 0;
    float mx;
    if (n <= 0) return 0;
    mx = s[0];
    for (i = 0; i < n; i++)
        if (s[i] > mx) { idx = i; mx = s[i]; }
    return idx;
}
static float ltpf_interpolate(const float *r, int rel, int d) { /* :457-469 */
    const float *TR = TF(TAB_LTPF_INTERP_R);
    float acc = 0.0f;
    int m;
    for (m = -4; m <= 4; m++) {
        int n = 4 * m - d;
        if (n > -16 && n < 16) acc += r[rel + m] * TR[n + 15];
    }
    return acc;
}
static float ltpf_dot(const lc3o_ltpf_enc_state *st, int n, int d) { /* :412-424 */
    const float *TX = TF(TAB_LTPF_INTERP_X12K8);
    float acc = 0.0f;
    int k;
    for (k = -2; k <= 2; k++) {
        int h = 4 * k - d;
        if (h > -8 && h < 8) acc += st->x12[NMEM + n - k] * TX[h + 7];
    }
    return acc;
}

lc3o_ltpf_result lc3o_enc_ltpf(const lc3o_config *c, lc3o_ltpf_enc_state *st, const int16_t *x_s,
                               int near_nyquist, int nbits) {
    const float *TRS = TF(TAB_RESAMP_FILTER);
    int len12, len6, delay, p, xs_len, n, k, num_hist, x12_len;
    float rf;
    int t_nbits, gain_ltpf_on;
    lc3o_ltpf_result res;
    ltpf_fields(c, &len12, &len6, &delay, &p, &rf, &xs_len);
    x12_len = len12 + delay + NMEM;
    t_nbits = c->n_ms_10 ? nbits : (int)lc3m_f64_to_usize(round((double)nbits * 10.0 / 7.5));
    gain_ltpf_on = t_nbits < 560 + c->fs_ind * 80;

    /* shift_out_old_samples :217-229 */
    num_hist = 240 / p;
    memmove(st->x_s_ext, st->x_s_ext + (xs_len - num_hist), sizeof(int16_t) * (size_t)num_hist);
    memcpy(st->x_s_ext + num_hist, x_s, sizeof(int16_t) * (size_t)c->nf);
    memmove(st->x12, st->x12 + len12, sizeof(float) * (size_t)(x12_len - len12));
    memmove(st->x6, st->x6 + len6, sizeof(float) * (size_t)(64 + K_MAX - len6));

    /* resampling :152-166 */
    {
        float *x12 = st->x12 + delay + NMEM;
        int lim = 120 / p; /* -120 / p truncates toward zero as well */
        for (n = 0; n < len12; n++) {
            float acc = 0.0f;
            for (k = -lim; k <= lim; k++) {
                int index_x_s = (15 * n) / p + k - 120 / p;
                int index_h = p * k - ((15 * n) % p);
                if (index_h > -120 && index_h < 120) acc += (float)st->x_s_ext[240 / p + index_x_s] * TRS[119 + index_h];
            }
            x12[n] = acc * ((float)p * rf);
        }
        /* high-pass 50 Hz :168-177, recursive across frames */
        for (n = 0; n < len12; n++) {
            float h50 = x12[n] - -1.9652933726226904f * st->h50_m1 - 0.9658854605688177f * st->h50_m2;
            x12[n] = 0.9827947082978771f * h50 + -1.965589416595754f * st->h50_m1 + 0.9827947082978771f * st->h50_m2;
            st->h50_m2 = st->h50_m1;
            st->h50_m1 = h50;
        }
    }

    /* pitch_detection :232-290 */
    {
        float r6[K_MAX + 1 - K_MIN], rw6[K_MAX + 1 - K_MIN];
        int lag_t1, lag_t2, k_from, k_to, t_current, pitch_present;
        float nv0, nv1, normcorr1, normcorr2;
        for (n = 0; n < len6; n++) {
            const float *s = st->x12 + NMEM - 3 + 2 * n;
            st->x6[K_MAX + n] = 0.1236796411180537f * s[0] + 0.2353512128364889f * s[1] + 0.2819382920909148f * s[2] +
                                0.2353512128364889f * s[3] + 0.1236796411180537f * s[4];
        }
        for (k = 0; k < K_MAX + 1 - K_MIN; k++) {
            int from_k = K_MAX - K_MIN - k;
            float acc = 0.0f, weight;
            for (n = 0; n < len6; n++) acc += st->x6[K_MAX + n] * st->x6[from_k + n];
            r6[k] = acc;
            weight = 1.0f - 0.5f * (float)k / (float)(K_MAX - K_MIN);
            rw6[k] = weight * acc;
        }
        lag_t1 = index_of_max(rw6, K_MAX + 1 - K_MIN) + K_MIN;
        k_from = (st->t_prev - 4 > K_MIN ? st->t_prev - 4 : K_MIN) - K_MIN;
        k_to = (st->t_prev + 4 < K_MAX ? st->t_prev + 4 : K_MAX) - K_MIN + 1;
        lag_t2 = index_of_max(r6 + k_from, k_to - k_from) + k_from + K_MIN;
        nv0 = ltpf_norm_value(st, 0, len6);
        nv1 = ltpf_norm_value(st, lag_t1, len6);
        normcorr1 = lc3m_maxf(0.0f, r6[lag_t1 - K_MIN] / sqrtf(nv0 * nv1));
        if (lag_t1 == lag_t2) normcorr2 = normcorr1;
        else {
            float nv2 = ltpf_norm_value(st, lag_t2, len6);
            normcorr2 = lc3m_maxf(0.0f, r6[lag_t2 - K_MIN] / sqrtf(nv0 * nv2));
        }
        if (normcorr2 > 0.85f * normcorr1) {
            t_current = lag_t2;
            pitch_present = normcorr2 > 0.6f;
        } else {
            t_current = lag_t1;
            pitch_present = normcorr1 > 0.6f;
        }

        /* pitch_lag_parameter :292-363 */
        {
            int k_min = 2 * t_current - 4 > 32 ? 2 * t_current - 4 : 32;
            int k_max = 2 * t_current + 4 < 228 ? 2 * t_current + 4 : 228;
            float r12[228 + 4 + 1];
            float max_corr = 0.0f;
            int pitch_int = k_min, pitch_fr = 0, rel, d, pitch_index;
            for (k = k_min - 4; k <= k_max + 4; k++) {
                float acc = 0.0f;
                for (n = 0; n < len12; n++) acc += st->x12[NMEM + n] * st->x12[NMEM - k + n];
                r12[k - (k_min - 4)] = acc;
                if (acc > max_corr && k >= k_min && k <= k_max) {
                    max_corr = acc;
                    pitch_int = k;
                }
            }
            rel = pitch_int - (k_min - 4);
            if (pitch_int == 32) {
                float mx = 0.0f;
                for (d = 0; d <= 3; d++) {
                    float v = ltpf_interpolate(r12, rel, d);
                    if (v > mx) { mx = v; pitch_fr = d; }
                }
            } else if (pitch_int < 127 && pitch_int > 32) {
                float mx = 0.0f;
                for (d = -3; d <= 3; d++) {
                    float v = ltpf_interpolate(r12, rel, d);
                    if (v > mx) { mx = v; pitch_fr = d; }
                }
            } else if (pitch_int >= 127 && pitch_int < 157) {
                float mx = 0.0f;
                for (d = -2; d <= 2; d += 2) {
                    float v = ltpf_interpolate(r12, rel, d);
                    if (v > mx) { mx = v; pitch_fr = d; }
                }
            }
            if (pitch_fr < 0) {
                pitch_int -= 1;
                pitch_fr += 4;
            }
            if (pitch_int < 127) pitch_index = 4 * pitch_int + pitch_fr - 128;
            else if (pitch_int >= 127 && pitch_int < 157) pitch_index = 2 * pitch_int + pitch_fr / 2 - 126;
            else pitch_index = pitch_int + 283;

            /* activation_bit :365-409 */
            {
                float num = 0.0f, nd = 0.0f, sh = 0.0f, den, nc, pitch;
                int ltpf_active;
                for (n = 0; n < len12; n++) {
                    float a = ltpf_dot(st, n, 0);
                    float b = ltpf_dot(st, n - pitch_int, pitch_fr);
                    num += a * b;
                    nd += a * a;
                    sh += b * b;
                }
                den = sqrtf(nd * sh);
                nc = den > 0.0f ? num / den : 0.0f;
                pitch = (float)pitch_int + (float)pitch_fr / 4.0f;
                if (gain_ltpf_on && !near_nyquist) {
                    ltpf_active = (!st->mem_ltpf_active && (c->n_ms_10 || st->mem_mem_nc > 0.94f) &&
                                   st->mem_nc > 0.94f && nc > 0.94f) ||
                                  (st->mem_ltpf_active && nc > 0.9f) ||
                                  (st->mem_ltpf_active && fabsf(pitch - st->mem_pitch) < 2.0f &&
                                   (nc - st->mem_nc) > -0.1f && nc > 0.84f);
                } else {
                    ltpf_active = 0;
                }
                /* run() tail :184-214 */
                res.nbits_ltpf = pitch_present ? 11 : 1;
                if (!pitch_present) {
                    pitch_index = 0;
                    nc = 0.0f;
                }
                st->t_prev = t_current;
                st->mem_mem_nc = st->mem_nc;
                if (pitch_present) {
                    st->mem_pitch = pitch;
                    st->mem_ltpf_active = ltpf_active;
                    st->mem_nc = nc;
                } else {
                    st->mem_pitch = 0.0f;
                    st->mem_ltpf_active = 0;
                    st->mem_nc = 0.0f;
                }
                res.ltpf_active = ltpf_active;
                res.pitch_index = pitch_index;
                res.pitch_present = pitch_present;
            }
        }
    }
    return res;
}

/* ================================================================= spectral quantization */
typedef struct { int rate_flag, lastnz, nbits_lsb, lastnz_trunc, nbits_est, nbits_trunc, mode_flag; } bit_consumption;

/* encoder/spectral_quantization.rs:265-348 */
static bit_consumption compute_bit_consumption(const lc3o_config *c, const int16_t *x_q, int nbits, int nbits_spec) {
    bit_consumption bc;
    int ne = c->ne, lastnz = ne, n, cctx = 0;
    uint32_t est = 0, trunc = 0;
    int nbits_lsb = 0, lastnz_trunc = 2;
    bc.rate_flag = nbits > (160 + c->fs_ind * 160) ? 512 : 0;
    bc.mode_flag = nbits >= (480 + c->fs_ind * 160);
    while (lastnz > 2 && x_q[lastnz - 1] == 0 && x_q[lastnz - 2] == 0) lastnz -= 2;
    for (n = 0; n < lastnz; n += 2) {
        int t = cctx + bc.rate_flag, lev = 0, pki, sym;
        unsigned a = (unsigned)(x_q[n] < 0 ? -(int)x_q[n] : x_q[n]), b = (unsigned)(x_q[n + 1] < 0 ? -(int)x_q[n + 1] : x_q[n + 1]);
        unsigned a_lsb = a, b_lsb = b;
        if (n > ne / 2) t += 256;
        while ((a > b ? a : b) >= 4) {
            pki = LC3T_AC_SPEC_LOOKUP[t + lev * 1024];
            est += LC3T_AC_SPEC_BITS[pki][16];
            if (lev == 0 && bc.mode_flag) nbits_lsb += 2;
            else est += 2 * 2048;
            a >>= 1;
            b >>= 1;
            lev = lev + 1 < 3 ? lev + 1 : 3;
        }
        pki = LC3T_AC_SPEC_LOOKUP[t + lev * 1024];
        sym = (int)(a + 4 * b);
        est += LC3T_AC_SPEC_BITS[pki][sym];
        if (a_lsb > 0) est += 2048;
        if (b_lsb > 0) est += 2048;
        if (lev > 0 && bc.mode_flag) {
            a_lsb >>= 1;
            b_lsb >>= 1;
            if (a_lsb == 0 && x_q[n] != 0) nbits_lsb += 1;
            if (b_lsb == 0 && x_q[n + 1] != 0) nbits_lsb += 1;
        }
        if ((x_q[n] != 0 || x_q[n + 1] != 0) && (int)lc3m_f64_to_usize(ceilf((float)est / 2048.0f)) <= nbits_spec) {
            lastnz_trunc = n + 2;
            trunc = est;
        }
        t = lev <= 1 ? 1 + (int)(a + b) * (lev + 1) : 12 + lev;
        cctx = (cctx & 15) * 16 + t;
    }
    bc.lastnz = lastnz;
    bc.lastnz_trunc = lastnz_trunc;
    bc.nbits_est = (int)lc3m_f64_to_usize(ceilf((float)est / 2048.0f)) + nbits_lsb;
    bc.nbits_trunc = (int)lc3m_f64_to_usize(ceilf((float)trunc / 2048.0f));
    bc.nbits_lsb = nbits_lsb;
    return bc;
}

/* :230-263 */
static bit_consumption quantize_spectrum(const lc3o_config *c, const float *x_f, int16_t *x_q, int nbits, int gg_off,
                                         int gg_ind, int nbits_spec, float *gg_out, int *lsb_mode) {
    float gg = lc3m_powf(10.0f, ((float)gg_ind + (float)gg_off) / 28.0f);
    bit_consumption bc;
    int n;
    for (n = 0; n < c->ne; n++)
        x_q[n] = x_f[n] >= 0.0f ? lc3m_f32_to_i16(x_f[n] / gg + 0.375f) : lc3m_f32_to_i16(x_f[n] / gg - 0.375f);
    bc = compute_bit_consumption(c, x_q, nbits, nbits_spec);
    for (n = bc.lastnz_trunc; n < bc.lastnz; n++) x_q[n] = 0;
    *lsb_mode = bc.mode_flag && bc.nbits_est > nbits_spec;
    *gg_out = gg;
    return bc;
}

/* SpectralQuantization::run :75-120 */
lc3o_quant_result lc3o_enc_quant(const lc3o_config *c, lc3o_quant_state *st, const float *x_f, int16_t *x_q,
                                 int nbits, int nbits_bw, int nbits_tns, int nbits_ltpf) {
    lc3o_quant_result res;
    int ne = c->ne, n, it;
    int nbits_ari, nbits_spec, nbits_spec_adj, gg_off, gg_ind, gg_min, reset_offset, lsb_mode;
    float nbits_offset, e[LC3O_MAX_NE / 4], x_f_max, gg;
    bit_consumption bc;

    /* calc_bit_budget :122-134 */
    nbits_ari = (int)lc3m_f64_to_usize(ceilf(lc3m_log2f((float)ne / 2.0f)));
    nbits_ari += nbits <= 1280 ? 3 : (nbits <= 2560 ? 4 : 5);
    nbits_spec = nbits - (nbits_bw + nbits_tns + nbits_ltpf + 38 + 8 + 3 + nbits_ari);

    /* get_global_gain_estimation_parameter :156-172 (nbits_spec_old is never updated) */
    if (st->reset_offset_old) nbits_offset = 0.0f;
    else {
        float prev = st->nbits_offset_old + (float)st->nbits_spec_old - (float)st->nbits_est_old;
        nbits_offset = 0.8f * st->nbits_offset_old + 0.2f * lc3m_minf(40.0f, lc3m_maxf(-40.0f, prev));
    }
    nbits_spec_adj = lc3m_f32_to_u16((float)nbits_spec + nbits_offset + 0.5f);
    {
        int q = (int16_t)nbits / (10 * (c->fs_ind + 1));
        gg_off = -(q < 115 ? q : 115) - 105 - 5 * (c->fs_ind + 1);
    }
    /* compute_spectral_energy :390-395 */
    for (n = 0; n < ne / 4; n++) {
        const float *x = x_f + 4 * n;
        float total = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
        e[n] = 10.0f * lc3m_log10f(1.1920929e-7f + total);
    }
    /* global_gain_estimation :174-209 */
    {
        int fac = 256;
        gg_ind = 255;
        for (it = 0; it < 8; it++) {
            float tmp = 0.0f, g;
            int is_zero = 1;
            fac >>= 1;
            gg_ind -= fac;
            g = (float)gg_ind + (float)gg_off;
            for (n = ne / 4 - 1; n >= 0; n--) {
                float ei = e[n];
                if (ei * 28.0f / 20.0f < g) {
                    if (!is_zero) tmp += 2.7f * 28.0f / 20.0f;
                } else {
                    if (g < (ei * 28.0f / 20.0f - 43.0f * 28.0f / 20.0f))
                        tmp += 2.0f * ei * 28.0f / 20.0f - 2.0f * g - 36.0f * 28.0f / 20.0f;
                    else
                        tmp += ei * 28.0f / 20.0f - g + 7.0f * 28.0f / 20.0f;
                    is_zero = 0;
                }
            }
            if ((tmp > (float)nbits_spec_adj * 1.4f * 28.0f / 20.0f) && !is_zero) gg_ind += fac;
        }
    }
    /* global_gain_limitation :212-228 */
    x_f_max = 0.0f;
    for (n = 0; n < ne; n++) x_f_max = lc3m_maxf(x_f_max, fabsf(x_f[n]));
    if (x_f_max > 0.0f)
        gg_min = (int)lc3m_f32_to_i16(ceilf(28.0f * lc3m_log10f(x_f_max / (32768.0f - 0.375f)))) - gg_off;
    else gg_min = 0;
    if (gg_ind < gg_min || x_f_max == 0.0f) {
        reset_offset = 1;
        gg_ind = gg_min;
    } else reset_offset = 0;

    bc = quantize_spectrum(c, x_f, x_q, nbits, gg_off, gg_ind, nbits_spec, &gg, &lsb_mode);

    /* save state :97-100 (after the FIRST pass) */
    st->nbits_offset_old = nbits_offset;
    st->nbits_est_old = bc.nbits_est;
    if (c->spec_flags & LC3O_SPEC_NBITS_SPEC_OLD) st->nbits_spec_old = nbits_spec;
    st->reset_offset_old = reset_offset;

    /* global_gain_adjustment :350-388 */
    {
        static const int T1[5] = {80, 230, 380, 530, 680};
        static const int T2[5] = {500, 1025, 1550, 2075, 2600};
        static const int T3[5] = {850, 1700, 2550, 3400, 4250};
        int t1 = T1[c->fs_ind], t2 = T2[c->fs_ind], t3 = T3[c->fs_ind], est = bc.nbits_est, origin = gg_ind;
        float delta, delta2;
        if (est < t1) delta = ((float)est + 48.0f) / 16.0f;
        else if (est < t2) {
            float tmp1 = (float)t1 / 16.0f + 3.0f, tmp2 = (float)t2 / 48.0f;
            delta = ((float)est - (float)t1) * (tmp2 - tmp1) / ((float)t2 - (float)t1) + tmp1;
        } else if (est < t3) delta = (float)est / 48.0f;
        else delta = (float)t3 / 48.0f;
        delta = floorf(delta + 0.5f);
        delta2 = delta + 2.0f;
        if ((gg_ind < 255 && est > nbits_spec) || (gg_ind > 0 && (float)est < ((float)nbits_spec - delta2))) {
            if ((float)est < ((float)nbits_spec - delta2)) gg_ind -= 1;
            else if (gg_ind == 254 || (float)est < ((float)nbits_spec + delta)) gg_ind += 1;
            else gg_ind += 2;
            if (gg_ind < gg_min) gg_ind = gg_min;
        }
        LC3O_ENC_COUNT(0);
        if (origin != gg_ind) {
            LC3O_ENC_COUNT(1);
            bc = quantize_spectrum(c, x_f, x_q, nbits, gg_off, gg_ind, nbits_spec, &gg, &lsb_mode);
        }
        if (lsb_mode) LC3O_ENC_COUNT(3);
    }
    res.gg_ind = gg_ind;
    res.nbits_spec = nbits_spec;
    res.nbits_lsb = bc.nbits_lsb;
    res.lsb_mode = lsb_mode;
    res.nbits_trunc = bc.nbits_trunc;
    res.rate_flag = bc.rate_flag;
    res.lastnz_trunc = bc.lastnz_trunc;
    res.gg = gg;
    return res;
}

/* ================================================================= residual bits (encoder/residual_spectrum.rs:33-62) */
int lc3o_enc_residual(int nbits_spec, int nbits_trunc, int ne, float gg, const float *x_f, const int16_t *x_q,
                      uint8_t *bits_out) {
    int mx = nbits_spec - nbits_trunc + 4, n = 0, k;
    if (mx < 0) mx = 0;
    if (mx > 0) {
        for (k = 0; k < ne; k++) {
            if (n >= mx) break;
            if (x_q[k] != 0) bits_out[n++] = x_f[k] >= (float)x_q[k] * gg;
        }
    }
    return n;
}

/* ================================================================= noise level (encoder/noise_level_estimation.rs:21-55) */
int lc3o_enc_noise_factor(const lc3o_config *c, const float *x_f, const int16_t *x_q, int bw_ind, float gg) {
    static const int BW10[5] = {80, 160, 240, 320, 400};
    static const int BW75[5] = {60, 120, 180, 240, 300};
    int bw_stop = c->n_ms_10 ? BW10[bw_ind] : BW75[bw_ind];
    int nf_start = c->n_ms_10 ? 24 : 18, nf_width = c->n_ms_10 ? 3 : 2;
    int nf_stop = c->ne < bw_stop ? c->ne : bw_stop, k, j, count = 0;
    float sum = 0.0f, level, diff;
    for (k = nf_start; k < nf_stop; k++) {
        int from = k - nf_width, to = bw_stop < k + nf_width + 1 ? bw_stop : k + nf_width + 1, rel = 1;
        for (j = from; j < to; j++)
            if (x_q[j] != 0) { rel = 0; break; }
        if (rel) {
            sum += fabsf(x_f[k]) / gg;
            count++;
        }
    }
    level = count > 0 ? sum / (float)count : 0.0f;
    diff = 8.0f - 16.0f * level;
    if (diff >= 0.0f) {
        int v = lc3m_f32_to_i32(diff + 0.5f);
        return v < 7 ? v : 7;
    }
    return 0;
}

/* ================================================================= bitstream (encoder/bitstream_encoding.rs, buffer_writer.rs) */
typedef struct {
    uint8_t *buf;
    int nbytes, nbits;
    int bp, bp_side, mask_side; /* BufferWriter, buffer_writer.rs:5-19 */
    uint32_t low, range;        /* ArithmeticEncoderState :27-34 */
    int cache, carry, carry_count;
} bitwriter;

static void bw_bool_backward(bitwriter *w, int bit) { /* buffer_writer.rs:27-40 */
    if (w->bp_side >= 0 && w->bp_side < w->nbytes) {
        if (!bit) w->buf[w->bp_side] &= (uint8_t)~w->mask_side;
        else w->buf[w->bp_side] |= (uint8_t)w->mask_side;
    }
    if (w->mask_side == 0x80) {
        w->mask_side = 1;
        w->bp_side -= 1;
    } else w->mask_side <<= 1;
}
static void bw_uint_backward(bitwriter *w, uint32_t val, int nbits) { /* :19-25 */
    int i;
    for (i = 0; i < nbits; i++) {
        bw_bool_backward(w, (int)(val & 1u));
        val >>= 1;
    }
}
static void bw_byte_forward(bitwriter *w, int val) { /* :55-58 */
    if (w->bp >= 0 && w->bp < w->nbytes) w->buf[w->bp] = (uint8_t)val;
    w->bp += 1;
}
static void bw_uint_forward(bitwriter *w, unsigned val, int nbits) { /* :42-53 */
    unsigned mask = 0x80;
    int i;
    for (i = 0; i < nbits; i++) {
        if (w->bp >= 0 && w->bp < w->nbytes) {
            if (((val & 0xff) & mask) == 0) w->buf[w->bp] &= (uint8_t)~mask;
            else w->buf[w->bp] |= (uint8_t)mask;
        }
        mask >>= 1;
    }
}
static int ilog2_u32(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }
static int bw_nbits_side_written(const bitwriter *w) { /* buffer_writer.rs:60-66 */
    return w->nbits - (8 * w->bp_side + 8 - ilog2_u32((uint32_t)w->mask_side));
}
static void ac_shift(bitwriter *w) { /* bitstream_encoding.rs:397-415 */
    if (w->low < 0x00ff0000u || w->carry == 1) {
        if (w->cache >= 0) bw_byte_forward(w, (w->cache + w->carry) & 0xff);
        while (w->carry_count > 0) {
            bw_byte_forward(w, (w->carry + 0xff) & 0xff);
            w->carry_count -= 1;
        }
        w->cache = (int)(w->low >> 16);
        w->carry = 0;
    } else w->carry_count += 1;
    w->low <<= 8;
    w->low &= 0x00ffffffu;
}
static void ac_encode(bitwriter *w, int cum_freq, int sym_freq) { /* :417-429 */
    uint32_t r = w->range >> 10;
    w->low += r * (uint32_t)cum_freq;
    if ((w->low >> 24) != 0) w->carry = 1;
    w->low &= 0x00ffffffu;
    w->range = r * (uint32_t)sym_freq;
    while (w->range < 0x10000u) {
        w->range <<= 8;
        ac_shift(w);
    }
}

void lc3o_enc_bitstream(const lc3o_config *c, lc3o_bw_result bw, const lc3o_sns_result *sns,
                        const lc3o_tns_result *tns, lc3o_ltpf_result ltpf, const lc3o_quant_result *spec,
                        const uint8_t *res_bits, int n_res_bits, int noise_factor, const int16_t *x_q,
                        uint8_t *out, int nbytes) {
    bitwriter w;
    int f, k, ne = c->ne;
    uint8_t lsbs[480 * 8];
    int nlsbs = 0;
    /* init :138-144 */
    memset(&w, 0, sizeof(w));
    w.buf = out;
    w.nbytes = nbytes;
    w.nbits = nbytes * 8;
    w.bp = 0;
    w.bp_side = nbytes - 1;
    w.mask_side = 1;
    memset(out, 0, (size_t)nbytes);
    /* side information :92-112 */
    if (bw.nbits_bandwidth > 0) bw_uint_backward(&w, (uint32_t)bw.bandwidth_ind, bw.nbits_bandwidth);
    {
        int nb = 0, half = ne / 2;
        while ((1 << nb) < half) nb++; /* ceil(log2(ne/2)) :154 */
        bw_uint_backward(&w, (uint32_t)((spec->lastnz_trunc >> 1) - 1), nb);
    }
    bw_bool_backward(&w, spec->lsb_mode);
    bw_uint_backward(&w, (uint32_t)spec->gg_ind, 8);
    for (f = 0; f < tns->num_tns_filters; f++) bw_bool_backward(&w, tns->rc_order[f] != 0);
    bw_bool_backward(&w, ltpf.pitch_present);
    bw_uint_backward(&w, (uint32_t)sns->ind_lf, 5);
    bw_uint_backward(&w, (uint32_t)sns->ind_hf, 5);
    { /* encode_scf_vq_2nd_stage :183-205 */
        int submode_msb = (sns->shape_j >> 1) != 0;
        bw_bool_backward(&w, submode_msb);
        bw_uint_backward(&w, (uint32_t)(sns->gind >> LC3T_SNS_GAIN_LSB_BITS[sns->shape_j]),
                         LC3T_SNS_GAIN_MSB_BITS[sns->shape_j]);
        bw_bool_backward(&w, sns->ls_inda != 0);
        if (!submode_msb) {
            bw_uint_backward(&w, sns->index_joint_j, 13);
            bw_uint_backward(&w, sns->index_joint_j >> 13, 12);
        } else {
            bw_uint_backward(&w, sns->index_joint_j, 12);
            bw_uint_backward(&w, sns->index_joint_j >> 12, 12);
        }
    }
    if (ltpf.pitch_present) {
        bw_bool_backward(&w, ltpf.ltpf_active);
        bw_uint_backward(&w, (uint32_t)ltpf.pitch_index, 9);
    }
    bw_uint_backward(&w, (uint32_t)noise_factor, 3);
    /* ac_enc_init :216-222 */
    w.low = 0;
    w.range = 0x00ffffffu;
    w.cache = -1;
    w.carry = 0;
    w.carry_count = 0;
    /* tns_data :224-244 */
    for (f = 0; f < tns->num_tns_filters; f++) {
        if (tns->rc_order[f] > 0) {
            ac_encode(&w, LC3T_AC_TNS_ORDER_CUMFREQ[tns->lpc_weighting][tns->rc_order[f] - 1],
                      LC3T_AC_TNS_ORDER_FREQ[tns->lpc_weighting][tns->rc_order[f] - 1]);
            for (k = 0; k < tns->rc_order[f]; k++) {
                int ri = tns->rc_i[k + 8 * f];
                if (ri < 0) ri = 0;
                if (ri > 16) ri = 16;
                ac_encode(&w, LC3T_AC_TNS_COEF_CUMFREQ[k][ri], LC3T_AC_TNS_COEF_FREQ[k][ri]);
            }
        }
    }
    /* spectral_data :246-326 */
    {
        int cctx = 0;
        for (k = 0; k < spec->lastnz_trunc; k += 2) {
            int t = cctx + spec->rate_flag + (k > ne / 2 ? 256 : 0), lev = 0, pki, sym;
            unsigned a = (unsigned)(x_q[k] < 0 ? -(int)x_q[k] : x_q[k]);
            unsigned b = (unsigned)(x_q[k + 1] < 0 ? -(int)x_q[k + 1] : x_q[k + 1]);
            unsigned a_lsb = a, b_lsb = b;
            int lsb0 = 0, lsb1 = 0;
            while ((a > b ? a : b) >= 4) {
                pki = LC3T_AC_SPEC_LOOKUP[t + (lev < 3 ? lev : 3) * 1024];
                ac_encode(&w, LC3T_AC_SPEC_CUMFREQ[pki][16], LC3T_AC_SPEC_FREQ[pki][16]);
                if (spec->lsb_mode && lev == 0) {
                    lsb0 = (int)(a & 1u);
                    lsb1 = (int)(b & 1u);
                } else {
                    bw_bool_backward(&w, (a & 1u) == 1u);
                    bw_bool_backward(&w, (b & 1u) == 1u);
                }
                a >>= 1;
                b >>= 1;
                lev += 1;
            }
            pki = LC3T_AC_SPEC_LOOKUP[t + (lev < 3 ? lev : 3) * 1024];
            sym = (int)(a + 4 * b);
            ac_encode(&w, LC3T_AC_SPEC_CUMFREQ[pki][sym], LC3T_AC_SPEC_FREQ[pki][sym]);
            if (spec->lsb_mode && lev > 0) {
                a_lsb >>= 1;
                b_lsb >>= 1;
                if (nlsbs < (int)sizeof(lsbs)) lsbs[nlsbs] = (uint8_t)lsb0;
                nlsbs++;
                if (a_lsb == 0 && x_q[k] != 0) {
                    if (nlsbs < (int)sizeof(lsbs)) lsbs[nlsbs] = x_q[k] > 0 ? 0 : 1;
                    nlsbs++;
                }
                if (nlsbs < (int)sizeof(lsbs)) lsbs[nlsbs] = (uint8_t)lsb1;
                nlsbs++;
                if (b_lsb == 0 && x_q[k + 1] != 0) {
                    if (nlsbs < (int)sizeof(lsbs)) lsbs[nlsbs] = x_q[k + 1] > 0 ? 0 : 1;
                    nlsbs++;
                }
            }
            if (a_lsb > 0) bw_bool_backward(&w, x_q[k] <= 0);
            if (b_lsb > 0) bw_bool_backward(&w, x_q[k + 1] <= 0);
            lev = lev < 3 ? lev : 3;
            t = lev <= 1 ? 1 + (int)(a + b) * (lev + 1) : 12 + lev;
            cctx = (cctx & 15) * 16 + t;
        }
    }
    /* residual_data_and_finalization :328-352 */
    {
        int nbits_side = bw_nbits_side_written(&w);
        int nbits_ari = w.bp * 8 + 25 - ilog2_u32(w.range); /* nbits_side_forcast :64-75 */
        int n_enc;
        if (w.carry >= 0) nbits_ari += 8;
        if (w.carry_count > 0) nbits_ari += w.carry_count * 8;
        n_enc = w.nbits - (nbits_side + nbits_ari);
        if (n_enc < 0) n_enc = 0;
        if (!spec->lsb_mode) {
            for (k = 0; k < n_enc && k < n_res_bits; k++) bw_bool_backward(&w, res_bits[k]);
        } else {
            if (n_enc > nlsbs) n_enc = nlsbs;
            for (k = 0; k < n_enc; k++) bw_bool_backward(&w, lsbs[k] == 1);
        }
    }
    /* ac_enc_finish :354-395 */
    {
        int bits = 1;
        uint32_t mask, val, over1, high, over2;
        while ((w.range >> (24 - bits)) == 0) bits++;
        mask = 0x00ffffffu >> bits;
        val = w.low + mask;
        over1 = val >> 24;
        high = w.low + w.range;
        over2 = high >> 24;
        val &= 0x00ffffffu & ~mask;
        if (over1 == over2) {
            if ((val + mask) >= high) {
                bits += 1;
                mask >>= 1;
                val = ((w.low + mask) & 0x00ffffffu) & ~mask;
            }
            if (val < w.low) w.carry = 1;
        }
        w.low = val;
        while (bits > 0) {
            ac_shift(&w);
            bits -= 8;
        }
        bits += 8;
        if (w.carry_count > 0) {
            bw_byte_forward(&w, w.cache & 0xff);
            while (w.carry_count > 1) {
                bw_byte_forward(&w, 0xff);
                w.carry_count -= 1;
            }
            bw_uint_forward(&w, 0xffu >> (8 - bits), bits);
        } else {
            bw_uint_forward(&w, (unsigned)w.cache, bits);
        }
    }
}

/* ================================================================= top level (encoder/lc3_encoder.rs) */
int lc3o_encoder_init(lc3o_encoder *e, int fs_hz, int frame_us) { return lc3o_encoder_init_spec(e, fs_hz, frame_us, 0); }
int lc3o_encoder_init_spec(lc3o_encoder *e, int fs_hz, int frame_us, int spec_flags) {
    memset(e, 0, sizeof(*e));
    if (lc3o_config_new(&e->cfg, fs_hz, frame_us)) return -1;
    e->cfg.spec_flags = spec_flags;
    lc3o_dct4_init(&e->dct, e->cfg.nf);
    e->att.attack_pos_last = -1; /* attack_detector.rs:38 */
    lc3o_ltpf_enc_init(&e->cfg, &e->ltpf);
    return 0;
}

/* EncoderChannel::encode, lc3_encoder.rs:63-112 */
int lc3o_encode_frame(lc3o_encoder *e, const int16_t *x_s, uint8_t *out, int nbytes) {
    const lc3o_config *c = &e->cfg;
    int nbits = nbytes * 8, near_nyquist, attack, n_res, noise_factor;
    lc3o_bw_result bw;
    lc3o_sns_result sns;
    lc3o_tns_result tns;
    lc3o_ltpf_result pf;
    lc3o_quant_result spec;
    e->frame_index += 1;
    near_nyquist = lc3o_enc_mdct_run(e, x_s, e->mdct_out, e->energy_bands);
    bw = lc3o_enc_bandwidth(c, e->energy_bands);
    attack = lc3o_enc_attack(c, &e->att, x_s, nbytes);
    sns = lc3o_enc_sns(c, e->mdct_out, e->energy_bands, attack);
    tns = lc3o_enc_tns(c, e->mdct_out, bw.bandwidth_ind, nbits, near_nyquist);
    pf = lc3o_enc_ltpf(c, &e->ltpf, x_s, near_nyquist, nbits);
    spec = lc3o_enc_quant(c, &e->quant, e->mdct_out, e->x_q, nbits, bw.nbits_bandwidth, tns.nbits_tns, pf.nbits_ltpf);
    n_res = lc3o_enc_residual(spec.nbits_spec, spec.nbits_trunc, c->ne, spec.gg, e->mdct_out, e->x_q, e->res_bits);
    noise_factor = lc3o_enc_noise_factor(c, e->mdct_out, e->x_q, bw.bandwidth_ind, spec.gg);
    lc3o_enc_bitstream(c, bw, &sns, &tns, pf, &spec, e->res_bits, n_res, noise_factor, e->x_q, out, nbytes);
    return 0;
}

/* lc3_encoder.rs:194-209 */
void lc3o_encoder_working_buffer_lengths(int num_channels, int fs_hz, int frame_us, int64_t out[3]) {
    lc3o_config c;
    int len12, len6, delay, p, xs_len;
    float rf;
    out[0] = out[1] = out[2] = 0;
    if (lc3o_config_new(&c, fs_hz, frame_us)) return;
    ltpf_fields(&c, &len12, &len6, &delay, &p, &rf, &xs_len);
    out[0] = (int64_t)(c.nf * 2 + xs_len + c.ne) * num_channels;
    out[1] = (int64_t)((len12 + delay + NMEM) + (64 + K_MAX) + c.nf + c.nb) * num_channels;
    out[2] = (int64_t)(c.nf / 2 * 4) * num_channels;
}
