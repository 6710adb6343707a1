/* ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See lc3_oracle.h.
 * Restates reference src/decoder/ (all stage modules) stage by stage. */
#include "lc3_oracle.h"
#include "lc3_math.h"
#include <math.h>
#include <string.h>

#define LC3_TABLE_QUAL static const
#include "../tables/lc3_tables.h"

/* test-coverage statistic: how often each LTPF transition case (1..5) ran; not thread-safe, informational */
long lc3o_ltpf_trans_count[6];
int lc3o_ltpf_trans_counting = 0; /* tests switch the coverage statistic on (oracle_lib.ltpf_transition_counts); off, decode threads share no counter */
#define TF(name) ((const float *)(const void *)LC3T_##name##_BITS)

const float *lc3o_mdct_window(const lc3o_config *c);
const uint16_t *lc3o_band_index(const lc3o_config *c);

/* ================================================================= BufferReader (decoder/buffer_reader.rs) */
static int read_head_byte(lc3o_reader *r, const uint8_t *buf, int len, uint32_t *v) { /* :42-50 */
    if (r->head_byte_cursor < len) {
        *v = buf[r->head_byte_cursor++];
        return 0;
    }
    return -1;
}
static int read_head_u24(lc3o_reader *r, const uint8_t *buf, int len, uint32_t *v) { /* :52-60 */
    if (r->head_byte_cursor + 2 < len) {
        const uint8_t *p = buf + r->head_byte_cursor;
        *v = ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2];
        r->head_byte_cursor += 3;
        return 0;
    }
    return -1;
}
/* :63-98 */
int lc3o_read_tail_usize(lc3o_reader *r, const uint8_t *buf, int len, int num_bits, uint32_t *val) {
    int byte_index = r->tail_bit_cursor / 8, bit_index = r->tail_bit_cursor % 8;
    int bits_left = 8 - bit_index;
    int add_bytes = (num_bits > bits_left && num_bits < 8) ? 2 : 1;
    int num_bytes = num_bits / 8 + add_bytes, from, i, shift_by;
    uint32_t value = 0;
    if (len - r->head_byte_cursor - byte_index - num_bytes < 0) return -1;
    from = len - byte_index - num_bytes;
    if (num_bytes >= 1 && num_bytes <= 4)
        for (i = 0; i < num_bytes; i++) value = (value << 8) | buf[from + i];
    shift_by = 32 - num_bits - bit_index;
    value <<= shift_by;
    value >>= shift_by + bit_index;
    r->tail_bit_cursor += num_bits;
    *val = value;
    return 0;
}
/* :100-116 */
int lc3o_read_tail_bool(lc3o_reader *r, const uint8_t *buf, int len, int *bit) {
    int byte_index = r->tail_bit_cursor / 8, bit_index = r->tail_bit_cursor % 8, from;
    uint8_t byte;
    if (len - r->head_byte_cursor - byte_index + 2 < 0) return -1;
    from = len - byte_index - 1;
    if (from < 0) return -1; /* the reference would panic on the index; treated as a read error */
    byte = buf[from];
    byte = (uint8_t)(byte << (7 - bit_index));
    byte >>= 7;
    r->tail_bit_cursor += 1;
    *bit = byte == 1;
    return 0;
}

/* ================================================================= side info (decoder/side_info_reader.rs:29-200) */
#define RD(nb, dst) do { if (lc3o_read_tail_usize(r, buf, len, (nb), &(dst))) return -1; } while (0)
#define RDB(dst) do { if (lc3o_read_tail_bool(r, buf, len, &(dst))) return -1; } while (0)
int lc3o_dec_side_info(const uint8_t *buf, int len, lc3o_reader *r, int fs_ind, int ne, lc3o_side_info *si) {
    static const int NBITS_BW[5] = {0, 1, 2, 2, 3};
    uint32_t v;
    int nbits_bw = NBITS_BW[fs_ind], p_bw = 0, b, f, lastnz_bits = 0, half = ne / 2;
    lc3o_sns_vq *q = &si->sns_vq;
    memset(si, 0, sizeof(*si));
    if (nbits_bw > 0) {
        RD(nbits_bw, v);
        if (fs_ind < (int)v) return -2; /* BandwidthIdxOutOfRange */
        p_bw = (int)v;
    }
    while ((1 << lastnz_bits) < half) lastnz_bits++; /* ((ne/2) as f32).log2().ceil() :53 */
    RD(lastnz_bits, v);
    si->lastnz = (int)((v + 1) << 1);
    if (si->lastnz > ne) return -3;
    RDB(b);
    si->lsb_mode = b;
    RD(8, v);
    si->global_gain_index = (int)v;
    si->num_tns_filters = p_bw < 3 ? 1 : 2;
    for (f = 0; f < si->num_tns_filters; f++) {
        RDB(b);
        si->rc_order_ari_input[f] = b;
    }
    RDB(b);
    si->ltpf.pitch_present = b;
    /* read_sns_vq :131-200 */
    RD(5, v);
    q->ind_lf = (int)v;
    RD(5, v);
    q->ind_hf = (int)v;
    RDB(b);
    q->submode_msb = b;
    if (q->submode_msb == 0) RD(1, v);
    else RD(2, v);
    q->g_ind = (int)v;
    RDB(b);
    q->ls_inda = b;
    if (q->submode_msb == 0) {
        uint32_t tmp, idx_bor;
        int32_t s;
        RD(25, tmp);
        if (tmp >= 33460056u) return -4;
        idx_bor = tmp / 2390004u;
        q->idx_a = tmp - idx_bor * 2390004u;
        q->submode_lsb = 0;
        s = (int32_t)idx_bor - 2;
        if (s < 0) q->submode_lsb = 1;
        s = s + q->submode_lsb * 2;
        if (q->submode_lsb != 0) {
            q->g_ind = (q->g_ind << 1) + s;
            q->idx_b = 0;
            q->ls_indb = 0;
        } else {
            q->idx_b = (uint32_t)s >> 1;
            q->ls_indb = s & 1;
        }
    } else {
        uint32_t tmp;
        q->ls_indb = 0;
        q->idx_b = 0;
        q->submode_lsb = 0;
        RD(24, tmp);
        if (tmp >= 16708096u) return -5;
        if (tmp >= 15158272u) {
            tmp -= 15158272u;
            q->submode_lsb = 1;
            q->g_ind = (q->g_ind << 1) + (int)(tmp & 1u);
            q->idx_a = tmp >> 1;
        } else q->idx_a = tmp;
    }
    /* read_long_term_post_filter_info :106-129 */
    if (si->ltpf.pitch_present) {
        RDB(b);
        si->ltpf.is_active = b;
        RD(9, v);
        si->ltpf.pitch_index = (int)v;
    } else {
        si->ltpf.is_active = 0;
        si->ltpf.pitch_index = 0;
    }
    RD(3, v);
    si->noise_factor = (int)v;
    si->bandwidth = p_bw;
    return 0;
}

/* ================================================================= arithmetic decoder (decoder/arithmetic_codec.rs) */
typedef struct { uint32_t low, range; } ac_state;

/* ac_decode :67-97 */
static int ac_decode(const uint8_t *buf, int len, lc3o_reader *r, ac_state *st, const int16_t *cum, const int16_t *freq,
                     int nsym, int *sym) {
    uint32_t tmp = st->range >> 10, limit = tmp << 10, b;
    int val = nsym - 1;
    if (st->low >= limit) return -1;
    while (st->low < tmp * (uint32_t)cum[val]) val--;
    st->low -= tmp * (uint32_t)cum[val];
    st->range = tmp * (uint32_t)freq[val];
    while (st->range < 0x10000u) {
        st->low <<= 8;
        st->low &= 0x00ffffffu;
        if (read_head_byte(r, buf, len, &b)) return -1;
        st->low += b;
        st->range <<= 8;
    }
    *sym = val;
    return 0;
}
static int ilog2_u32(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }

/* read_res_bit :339-383 */
static int read_res_bit(int32_t *x, lc3o_reader *r, const uint8_t *buf, int len, int idx, int *nbits_res, int *cont) {
    int bit;
    if (*nbits_res == 0) { *cont = 0; return 0; }
    if (lc3o_read_tail_bool(r, buf, len, &bit)) return -1;
    *nbits_res -= 1;
    if (bit) {
        if (x[idx] > 0) x[idx] += 1;
        else if (x[idx] < 0) x[idx] -= 1;
        else {
            if (*nbits_res == 0) { *cont = 0; return 0; }
            if (lc3o_read_tail_bool(r, buf, len, &bit)) return -1;
            *nbits_res -= 1;
            x[idx] = bit ? -1 : 1;
        }
    }
    *cont = 1;
    return 0;
}

/* decode :109-158 */
int lc3o_dec_arith(const uint8_t *buf, int len, lc3o_reader *r, int fs_ind, int ne, const lc3o_side_info *si,
                   int n_ms_10, int32_t *x, lc3o_arith_data *ad) {
    int nbits = len * 8, f, k, sym;
    ac_state st;
    int32_t save_lev[LC3O_MAX_NE];
    memset(ad, 0, sizeof(*ad));
    memset(save_lev, 0, sizeof(save_lev));
    /* ac_dec_init :57-65 */
    if (read_head_u24(r, buf, len, &st.low)) return -1;
    st.range = 0x00ffffffu;
    /* decode_tns_data :304-337 */
    {
        int wt = nbits < (n_ms_10 ? 480 : 360);
        ad->rc_order[0] = si->rc_order_ari_input[0];
        ad->rc_order[1] = si->rc_order_ari_input[1];
        for (f = 0; f < si->num_tns_filters; f++) {
            if (ad->rc_order[f] > 0) {
                if (ac_decode(buf, len, r, &st, LC3T_AC_TNS_ORDER_CUMFREQ[wt], LC3T_AC_TNS_ORDER_FREQ[wt], 8, &sym))
                    return -2;
                ad->rc_order[f] = sym + 1;
                for (k = 0; k < ad->rc_order[f]; k++) {
                    if (ac_decode(buf, len, r, &st, LC3T_AC_TNS_COEF_CUMFREQ[k], LC3T_AC_TNS_COEF_FREQ[k], 17, &sym))
                        return -3;
                    ad->rc_i[f * 8 + k] = sym;
                }
            }
        }
    }
    /* decode_spectral_data :211-302 */
    {
        int rate_flag = nbits > (160 + fs_ind * 160) ? 512 : 0, cctx = 0, tup;
        for (tup = 0; tup < si->lastnz / 2; tup++) {
            int t = cctx + rate_flag + ((tup * 2) > (ne / 2) ? 256 : 0), lev = 0, a, b, bit;
            int32_t xk = 0, xk1 = 0;
            sym = 0;
            while (lev < 14) {
                int pki = LC3T_AC_SPEC_LOOKUP[t + (lev < 3 ? lev : 3) * 1024];
                if (ac_decode(buf, len, r, &st, LC3T_AC_SPEC_CUMFREQ[pki], LC3T_AC_SPEC_FREQ[pki], 17, &sym)) return -4;
                if (sym < 16) break;
                if (!si->lsb_mode || lev > 0) {
                    if (lc3o_read_tail_bool(r, buf, len, &bit)) return -5;
                    xk += (int32_t)((uint32_t)bit << lev);
                    if (lc3o_read_tail_bool(r, buf, len, &bit)) return -5;
                    xk1 += (int32_t)((uint32_t)bit << lev);
                }
                lev += 1;
            }
            if (si->lsb_mode) save_lev[tup] = lev; /* indexed by TUPLE here, by LINE when read back (:184-195) */
            a = sym & 3;
            b = sym >> 2;
            xk += (int32_t)((uint32_t)a << lev);
            xk1 += (int32_t)((uint32_t)b << lev);
            if (xk > 0) {
                if (lc3o_read_tail_bool(r, buf, len, &bit)) return -5;
                if (bit) xk = -xk;
            }
            if (xk1 > 0) {
                if (lc3o_read_tail_bool(r, buf, len, &bit)) return -5;
                if (bit) xk1 = -xk1;
            }
            x[2 * tup] = xk;
            x[2 * tup + 1] = xk1;
            lev = lev < 3 ? lev : 3;
            t = lev <= 1 ? 1 + (a + b) * (lev + 1) : 12 + lev;
            cctx = (cctx & 15) * 16 + t;
        }
    }
    for (k = si->lastnz; k < LC3O_MAX_NE; k++) x[k] = 0; /* :131-133 */
    /* decode_residual_bits :160-208 */
    {
        int nbits_side = r->tail_bit_cursor - 8;
        int nbits_ari = (r->head_byte_cursor + 1 - 3) * 8 + 25 - ilog2_u32(st.range); /* :385-405 */
        int nres, bit, cont;
        if (nbits < nbits_side + nbits_ari) return -6; /* NegativeResidualNumBits */
        nres = nbits - nbits_side - nbits_ari;
        if (!si->lsb_mode) {
            for (k = 0; k < ne; k++) {
                if (x[k] != 0) {
                    if (ad->n_residual_bits == nres) break;
                    if (lc3o_read_tail_bool(r, buf, len, &bit)) return -7;
                    if (ad->n_residual_bits >= 480) return -8;
                    ad->residual_bits[ad->n_residual_bits++] = (uint8_t)bit;
                }
            }
        } else {
            for (k = 0; k < si->lastnz; k += 2) {
                if (save_lev[k] > 0) {
                    if (read_res_bit(x, r, buf, len, k, &nres, &cont)) return -7;
                    if (!cont) break;
                    if (read_res_bit(x, r, buf, len, k + 1, &nres, &cont)) return -7;
                    if (!cont) break;
                }
            }
        }
    }
    /* noise filling seed :140-145 (wrapping i32 sum) */
    {
        uint32_t seed = 0;
        for (k = 0; k < ne; k++) {
            uint32_t a = (uint32_t)(x[k] < 0 ? -(int64_t)x[k] : x[k]);
            seed += a * (uint32_t)k;
        }
        ad->noise_filling_seed = (int)(seed & 0xFFFFu);
    }
    ad->is_zero_frame = si->lastnz == 2 && x[0] == 0 && x[1] == 0 && si->global_gain_index == 0;
    ad->frame_num_bits = nbits;
    return 0;
}

/* ================================================================= residual (decoder/residual_spectrum.rs:13-39) */
void lc3o_dec_residual(int lsb_mode, const uint8_t *bits, int nbits, float *spec, int ne) {
    int k, n = 0;
    if (lsb_mode) return;
    for (k = 0; k < ne; k++) {
        if (spec[k] != 0.0f) {
            if (n >= nbits) break;
            if (bits[n]) {
                if (spec[k] > 0.0f) spec[k] += 0.3125f;
                else spec[k] += 0.1875f;
            } else {
                if (spec[k] > 0.0f) spec[k] -= 0.1875f;
                else spec[k] -= 0.3125f;
            }
            n++;
        }
    }
}

/* ================================================================= noise filling (decoder/noise_filling.rs:18-56) */
void lc3o_dec_noise_filling(int is_zero_frame, int seed, int bandwidth, int n_ms_10, int noise_factor,
                            const int32_t *x_int, float *spec, int ne) {
    static const int BW75[5] = {60, 120, 180, 240, 300};
    static const int BW10[5] = {80, 160, 240, 320, 400};
    int bw_stop, nf_start, nf_width, k, j, nf = seed, lim;
    float level;
    if (is_zero_frame) return;
    bw_stop = n_ms_10 ? BW10[bandwidth] : BW75[bandwidth];
    nf_start = n_ms_10 ? 24 : 18;
    nf_width = n_ms_10 ? 3 : 2;
    level = (8.0f - (float)noise_factor) / 16.0f;
    lim = bw_stop < ne ? bw_stop : ne; /* iter over spec_lines_float.take(bw_stop) */
    for (k = nf_start; k < lim; k++) {
        int from = k - nf_width, to = (bw_stop - 1) < (k + nf_width) ? (bw_stop - 1) : (k + nf_width), all0 = 1;
        for (j = from; j <= to; j++)
            if (x_int[j] != 0) { all0 = 0; break; }
        if (all0) {
            nf = (13849 + nf * 31821) & 0xFFFF;
            spec[k] = nf < 0x8000 ? level : -level;
        }
    }
}

/* ================================================================= global gain (decoder/global_gain.rs:15-25) */
void lc3o_dec_global_gain(int frame_num_bits, int fs_ind, int gg_ind, float *spec, int ne) {
    int fs = fs_ind + 1, q = frame_num_bits / (10 * fs), k;
    int gg_off = -(q < 115 ? q : 115) - 105 - (5 * fs);
    float exponent = ((float)gg_ind + (float)gg_off) / 28.0f;
    float gg = lc3m_powf(10.0f, exponent);
    for (k = 0; k < ne; k++) spec[k] *= gg;
}

/* ================================================================= TNS (decoder/temporal_noise_shaping.rs:24-137) */
void lc3o_dec_tns(int n_ms_10, int bandwidth, int num_tns_filters, const int *rc_order, const int *rc_i,
                  float *spec) {
    static const int B10[5][4] = {{12, 80, 0, 0}, {12, 160, 0, 0}, {12, 240, 0, 0}, {12, 160, 160, 320}, {12, 200, 200, 400}};
    static const int B75[5][4] = {{9, 60, 0, 0}, {9, 120, 0, 0}, {9, 180, 0, 0}, {9, 120, 120, 240}, {9, 150, 150, 300}};
    const int *bnd = n_ms_10 ? B10[bandwidth] : B75[bandwidth];
    int nbands = bandwidth < 3 ? 1 : 2;
    const float step = (float)(3.14159265358979323846 / 17.0); /* (PI / 17.0) as f32 :41 */
    float rc_quant[16], st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int f, k, n;
    for (k = 0; k < 16; k++) {
        rc_quant[k] = 0.0f;
        if (rc_i[k] != 0) rc_quant[k] = lc3m_sinf(step * (float)(rc_i[k] - 8)); /* SURVEY A12 */
    }
    for (f = 0; f < nbands && f < num_tns_filters; f++) {
        int order = rc_order[f], off = f * 8;
        if (order > 0) {
            for (n = bnd[2 * f]; n < bnd[2 * f + 1]; n++) {
                float t = spec[n] - rc_quant[order - 1 + off] * st[order - 1];
                for (k = order - 2; k >= 0; k--) {
                    float rc = rc_quant[k + off];
                    t -= rc * st[k];
                    st[k + 1] = rc * t + st[k];
                }
                spec[n] = t;
                st[0] = t;
            }
        }
    }
}

/* ================================================================= SNS (decoder/spectral_noise_shaping.rs) */
/* mpvq_deenum :155-199 + helpers :201-235 */
void lc3o_mpvq_deenum(int dim_in, int k_val_in, int ls_ind, uint32_t mpvq_ind, int32_t *vec_out) {
    int leading_sign = ls_ind == 0 ? 1 : -1, k_max_local = k_val_in, pos, k_acc;
    uint32_t ind = mpvq_ind;
    for (pos = 0; pos < dim_in; pos++) vec_out[pos] = 0;
    for (pos = 0; pos < dim_in; pos++) {
        const uint32_t *h_row = LC3T_MPVQ_OFFSETS[dim_in - 1 - pos];
        int k_delta;
        if (ind != 0) {
            uint32_t ul_diff = 0;
            int wrap;
            k_acc = k_max_local;
            wrap = ind < h_row[k_acc];
            if (!wrap) ul_diff = ind - h_row[k_acc];
            while (wrap) {
                k_acc -= 1;
                wrap = ind < h_row[k_acc];
                if (!wrap) ul_diff = ind - h_row[k_acc];
            }
            ind = ul_diff;
            k_delta = k_max_local - k_acc;
        } else {
            vec_out[pos] = leading_sign < 0 ? -k_max_local : k_max_local;
            break;
        }
        if (k_delta != 0) { /* setval_update_sign :201-216 */
            vec_out[pos] = leading_sign < 0 ? -k_delta : k_delta;
            leading_sign = (ind & 1u) ? -1 : 1;
            ind >>= 1;
            k_max_local -= k_delta;
        }
    }
}

/* decode :21-151 */
void lc3o_dec_sns(const lc3o_config *c, const lc3o_sns_vq *sns, float *spec) {
    const float *LFCB = TF(LFCB), *HFCB = TF(HFCB), *D = TF(D);
    const uint16_t *ifs = lc3o_band_index(c);
    float st1[16], scf[16], sfi[64], y_norm, gain, g;
    int32_t y[16], z[16];
    int shape_j = (sns->submode_msb << 1) + sns->submode_lsb, n, col, b, k, nb = c->nb, n2;
    for (n = 0; n < 8; n++) {
        st1[n] = LFCB[sns->ind_lf * 8 + n];
        st1[8 + n] = HFCB[sns->ind_hf * 8 + n];
    }
    memset(y, 0, sizeof(y));
    memset(z, 0, sizeof(z));
    switch (shape_j) {
    case 0:
        lc3o_mpvq_deenum(10, 10, sns->ls_inda, sns->idx_a, y);
        lc3o_mpvq_deenum(6, 1, sns->ls_indb, sns->idx_b, z);
        for (n = 0; n < 6; n++) y[10 + n] = z[n];
        break;
    case 1:
        lc3o_mpvq_deenum(10, 10, sns->ls_inda, sns->idx_a, y);
        for (n = 10; n < 16; n++) y[n] = 0;
        break;
    case 2: lc3o_mpvq_deenum(16, 8, sns->ls_inda, sns->idx_a, y); break;
    default: lc3o_mpvq_deenum(16, 6, sns->ls_inda, sns->idx_a, y); break;
    }
    y_norm = 0.0f;
    for (n = 0; n < 16; n++) y_norm += (float)y[n] * (float)y[n];
    y_norm = sqrtf(y_norm);
    switch (shape_j) {
    case 0: gain = TF(SNS_VQ_REG_ADJ_GAINS)[sns->g_ind & 1]; break;
    case 1: gain = TF(SNS_VQ_REG_LF_ADJ_GAINS)[sns->g_ind & 3]; break;
    case 2: gain = TF(SNS_VQ_NEAR_ADJ_GAINS)[sns->g_ind & 3]; break;
    default: gain = TF(SNS_VQ_FAR_ADJ_GAINS)[sns->g_ind & 7]; break;
    }
    if (y_norm != 0.0f) gain /= y_norm;
    for (n = 0; n < 16; n++) {
        float factor = 0.0f;
        for (col = 0; col < 16; col++) factor += (float)y[col] * D[n * 16 + col];
        scf[n] = st1[n] + gain * factor;
    }
    sfi[0] = scf[0];
    sfi[1] = scf[0];
    for (n = 0; n <= 14; n++) {
        float fn = scf[n], d = scf[n + 1] - fn;
        sfi[4 * n + 2] = fn + (1.0f / 8.0f * d);
        sfi[4 * n + 3] = fn + (3.0f / 8.0f * d);
        sfi[4 * n + 4] = fn + (5.0f / 8.0f * d);
        sfi[4 * n + 5] = fn + (7.0f / 8.0f * d);
    }
    sfi[62] = scf[15] + 1.0f / 8.0f * (scf[15] - scf[14]);
    sfi[63] = scf[15] + 3.0f / 8.0f * (scf[15] - scf[14]);
    n2 = 64 - nb;
    if (n2 != 0) { /* :100-111 (decoder form) */
        for (b = 0; b < n2; b++) sfi[b] = (sfi[2 * b] + sfi[2 * b + 1]) / 2.0f;
        for (b = n2; b < nb; b++) sfi[b] = sfi[b + n2];
    }
    for (b = 0; b < nb; b++) {
        g = lc3m_exp2_raw(sfi[b]); /* fast_math::exp2_raw :122 */
        for (k = ifs[b]; k < ifs[b + 1]; k++) spec[k] *= g;
    }
}

/* ================================================================= PLC (decoder/packet_loss_concealment.rs) */
void lc3o_dec_plc_save(lc3o_decoder *d, const float *spec) { /* :49-53 */
    d->plc_num_lost = 0;
    d->plc_alpha = 1.0f;
    memcpy(d->plc_last_good, spec, sizeof(float) * (size_t)d->cfg.ne);
}
lc3o_ltpf_info lc3o_dec_plc_load(lc3o_decoder *d, float *spec) { /* :63-85 */
    lc3o_ltpf_info info = {0, 0, 0};
    int k;
    if (d->plc_num_lost >= 4) d->plc_alpha *= d->plc_num_lost < 8 ? 0.9f : 0.85f;
    d->plc_num_lost += 1;
    for (k = 0; k < d->cfg.ne; k++) {
        d->plc_seed = (16831u + d->plc_seed * 12821u) & 0xFFFFu;
        spec[k] = d->plc_seed < 0x8000u ? d->plc_last_good[k] * d->plc_alpha : d->plc_last_good[k] * -d->plc_alpha;
    }
    return info;
}

/* ================================================================= IMDCT (decoder/modified_dct.rs:76-151) */
void lc3o_dec_imdct(lc3o_decoder *d, const float *spec, float *freq) {
    const lc3o_config *c = &d->cfg;
    const float *w = lc3o_mdct_window(c);
    int nf = c->nf, ne = c->ne, z = c->z, h = nf / 2, n;
    float *t = d->t_hat, gain;
    for (n = 0; n < ne; n++) freq[n] = spec[n];
    for (n = ne; n < nf; n++) freq[n] = 0.0f;
    lc3o_dct4_run(&d->dct, freq);
    /* unfold :97-136: t = [u[h..nf], -rev(u)[0..h], -rev(u)[h..nf], -u[0..h]] */
    for (n = 0; n < h; n++) {
        t[n] = freq[h + n];
        t[h + n] = -freq[nf - 1 - n];
        t[nf + n] = -freq[h - 1 - n];
        t[3 * h + n] = -freq[n];
    }
    gain = 1.0f / sqrtf(2.0f * (float)nf);
    for (n = 0; n < 2 * nf; n++) t[n] *= gain;
    for (n = 0; n < 2 * nf; n++) t[n] *= w[2 * nf - 1 - n];
    /* overlap_add :138-151 */
    for (n = 0; n < nf - z; n++) freq[n] = d->mem_ola_add[n] + t[z + n];
    for (n = 0; n < nf - z; n++) d->mem_ola_add[n] = t[nf + z + n];
    for (n = 0; n < z; n++) freq[nf - z + n] = t[nf + n];
}

/* ================================================================= LTPF (decoder/long_term_post_filter.rs) */
void lc3o_ltpf_dec_init(const lc3o_config *c, lc3o_ltpf_dec_state *st) { /* :61-134 */
    int l_den;
    memset(st, 0, sizeof(*st));
    switch (c->fs) {
    case 8000: l_den = 4; break;
    case 16000: l_den = 4; break;
    case 24000: l_den = 6; break;
    case 32000: l_den = 8; break;
    case 44100: l_den = 11; break; /* SURVEY A9 */
    default: l_den = 12; break;
    }
    st->l_den = l_den;
    st->l_num = l_den - 2;
    st->num_mem_blocks = c->n_ms_10 ? 2 : 3;
    st->norm = c->n_ms_10 ? c->nf / 4 : c->nf / 3;
}

static int wrap_neg(const lc3o_config *c, const lc3o_ltpf_dec_state *st, int idx) { /* :244-250 */
    return idx < 0 ? idx + st->num_mem_blocks * c->nf : idx;
}
static float ltpf_filter(const lc3o_config *c, const lc3o_ltpf_dec_state *st, const float *cn, const float *cd,
                         int start, int pitch_int) { /* compute_filter :380-415 */
    float acc = 0.0f;
    int k, sden;
    for (k = 0; k <= st->l_num; k++) acc += cn[k] * st->x_hat_mem[wrap_neg(c, st, start - k)];
    sden = start - pitch_int + st->l_den / 2;
    for (k = 0; k <= st->l_den; k++) acc -= cd[k] * st->x_hat_ltpf_mem[wrap_neg(c, st, sden - k)];
    return acc;
}

void lc3o_dec_ltpf(const lc3o_config *c, lc3o_ltpf_dec_state *st, const lc3o_ltpf_info *info, int nbits,
                   float *freq) {
    int nf = c->nf, pitch_int = 0, pitch_frac = 0, n, k, blk = st->block_start_index;
    int s25 = c->fs == 44100 ? 48000 / 400 : c->fs / 400, trans;
    int ncn = st->l_num + 1, ncd = st->l_den + 1;
    /* compute_filter_parameters :164-189 (f64) */
    if (info->is_active) {
        int pi = info->pitch_index, p_i;
        double p_fr, pitch, pitch_fs;
        uint64_t p_up;
        if (pi >= 440) { p_i = pi - 283; p_fr = 0.0; }
        else if (pi >= 380) { p_i = pi / 2 - 63; p_fr = (double)(2 * pi - 4 * p_i - 252); }
        else { p_i = pi / 4 + 32; p_fr = (double)(pi + 128 - 4 * p_i); }
        pitch = (double)p_i + p_fr / 4.0;
        pitch_fs = pitch * (8000.0 * ceil((double)c->fs / 8000.0) / 12800.0);
        p_up = lc3m_f64_to_usize((pitch_fs * 4.0) + 0.5);
        pitch_int = (int)(p_up / 4);
        pitch_frac = (int)(p_up - 4 * (uint64_t)pitch_int);
    }
    /* compute_filter_coeffs :192-242 */
    memcpy(st->c_num_mem, st->c_num, sizeof(float) * (size_t)ncn);
    memcpy(st->c_den_mem, st->c_den, sizeof(float) * (size_t)ncd);
    if (!info->is_active) {
        for (k = 0; k < ncn; k++) st->c_num[k] = 0.0f;
        for (k = 0; k < ncd; k++) st->c_den[k] = 0.0f;
    } else {
        int t_nbits = c->n_ms_10 ? nbits : (int)lc3m_f64_to_usize(round((double)nbits * 10.0 / 7.5));
        int sf = c->fs_ind * 80, gain_ind, tn, td;
        float gain;
        const float *tnum, *tden;
        if (t_nbits < 320 + sf) { gain = 0.4f; gain_ind = 0; }
        else if (t_nbits < 400 + sf) { gain = 0.35f; gain_ind = 1; }
        else if (t_nbits < 480 + sf) { gain = 0.3f; gain_ind = 2; }
        else if (t_nbits < 560 + sf) { gain = 0.25f; gain_ind = 3; }
        else { gain = 0.0f; gain_ind = 0; } /* SURVEY A11 */
        switch (c->fs) {
        case 8000: tnum = TF(TAB_LTPF_NUM_8000) + gain_ind * 3; tn = 3; tden = TF(TAB_LTPF_DEN_8000) + pitch_frac * 5; td = 5; break;
        case 16000: tnum = TF(TAB_LTPF_NUM_16000) + gain_ind * 3; tn = 3; tden = TF(TAB_LTPF_DEN_16000) + pitch_frac * 5; td = 5; break;
        case 24000: tnum = TF(TAB_LTPF_NUM_24000) + gain_ind * 5; tn = 5; tden = TF(TAB_LTPF_DEN_24000) + pitch_frac * 7; td = 7; break;
        case 32000: tnum = TF(TAB_LTPF_NUM_32000) + gain_ind * 7; tn = 7; tden = TF(TAB_LTPF_DEN_32000) + pitch_frac * 9; td = 9; break;
        default: tnum = TF(TAB_LTPF_NUM_48000) + gain_ind * 11; tn = 11; tden = TF(TAB_LTPF_DEN_48000) + pitch_frac * 13; td = 13; break;
        }
        /* zip() truncates to the shorter side */
        for (k = 0; k < ncn && k < tn; k++) st->c_num[k] = 0.85f * gain * tnum[k];
        for (k = 0; k < ncd && k < td; k++) st->c_den[k] = gain * tden[k];
    }
    memcpy(st->x_hat_mem + blk, freq, sizeof(float) * (size_t)nf);

    if (!info->is_active && !st->ltpf_active_prev) trans = 1;
    else if (info->is_active && !st->ltpf_active_prev) trans = 2;
    else if (!info->is_active && st->ltpf_active_prev) trans = 3;
    else if (pitch_int == st->p_int_mem && pitch_frac == st->p_fr_mem) trans = 4;
    else trans = 5;
    if (lc3o_ltpf_trans_counting) lc3o_ltpf_trans_count[trans]++; /* test coverage statistic (which LTPF transitions the test inputs reach) */

    switch (trans) {
    case 1:
        memcpy(st->x_hat_ltpf_mem + blk, st->x_hat_mem + blk, sizeof(float) * (size_t)nf);
        break;
    case 2:
        for (n = 0; n < s25; n++) {
            float fo;
            st->x_hat_ltpf_mem[blk + n] = st->x_hat_mem[blk + n];
            fo = ltpf_filter(c, st, st->c_num, st->c_den, blk + n, pitch_int);
            fo *= (float)n / (float)st->norm;
            st->x_hat_ltpf_mem[blk + n] -= fo;
        }
        for (n = s25; n < nf; n++) {
            st->x_hat_ltpf_mem[blk + n] = st->x_hat_mem[blk + n];
            st->x_hat_ltpf_mem[blk + n] -= ltpf_filter(c, st, st->c_num, st->c_den, blk + n, pitch_int);
        }
        break;
    case 3:
    case 5:
        /* deactive_first_2p5ms :417-424 */
        for (n = 0; n < s25; n++) {
            float fo;
            st->x_hat_ltpf_mem[blk + n] = st->x_hat_mem[blk + n];
            fo = ltpf_filter(c, st, st->c_num_mem, st->c_den_mem, blk + n, st->p_int_mem);
            fo *= 1.0f - ((float)n / (float)st->norm);
            st->x_hat_ltpf_mem[blk + n] -= fo;
        }
        if (trans == 3) {
            for (n = s25; n < nf; n++) st->x_hat_ltpf_mem[blk + n] = st->x_hat_mem[blk + n];
        } else {
            /* activate_first_2p5ms_from_mem :345-378 */
            float scratch[12 + LC3O_MAX_NF / 3 + 8];
            int l_num = st->l_num, sl = l_num + st->norm;
            if (blk < l_num) {
                int from = st->num_mem_blocks * nf - l_num;
                memcpy(scratch, st->x_hat_ltpf_mem + from, sizeof(float) * (size_t)l_num);
                memcpy(scratch + l_num, st->x_hat_ltpf_mem, sizeof(float) * (size_t)st->norm);
            } else {
                memcpy(scratch, st->x_hat_ltpf_mem + blk - l_num, sizeof(float) * (size_t)sl);
            }
            for (n = 0; n < s25; n++) {
                float fo = 0.0f;
                int sden;
                st->x_hat_ltpf_mem[blk + n] = scratch[n + l_num];
                for (k = 0; k <= l_num; k++) fo += st->c_num[k] * scratch[l_num + n - k];
                sden = (blk + n) - pitch_int + st->l_den / 2;
                for (k = 0; k <= st->l_den; k++) fo -= st->c_den[k] * st->x_hat_ltpf_mem[wrap_neg(c, st, sden - k)];
                fo *= (float)n / (float)st->norm;
                st->x_hat_ltpf_mem[blk + n] -= fo;
            }
            for (n = s25; n < nf; n++) {
                st->x_hat_ltpf_mem[blk + n] = st->x_hat_mem[blk + n];
                st->x_hat_ltpf_mem[blk + n] -= ltpf_filter(c, st, st->c_num, st->c_den, blk + n, pitch_int);
            }
        }
        break;
    default: /* 4 */
        for (n = 0; n < nf; n++) {
            st->x_hat_ltpf_mem[blk + n] = st->x_hat_mem[blk + n];
            st->x_hat_ltpf_mem[blk + n] -= ltpf_filter(c, st, st->c_num, st->c_den, blk + n, pitch_int);
        }
        break;
    }
    memcpy(freq, st->x_hat_ltpf_mem + blk, sizeof(float) * (size_t)nf);
    st->block_start_index += nf;
    if (st->block_start_index > (st->num_mem_blocks - 1) * nf) st->block_start_index = 0;
    st->ltpf_active_prev = info->is_active;
    st->p_int_mem = pitch_int;
    st->p_fr_mem = pitch_frac;
}

/* ================================================================= output scaling (decoder/output_scaling.rs:13-25) */
void lc3o_dec_output(const float *x, int16_t *out, int n) {
    int i;
    for (i = 0; i < n; i++) {
        int32_t tmp = x[i] > 0.0f ? lc3m_f32_to_i32(x[i] + 0.5f) : lc3m_f32_to_i32(x[i] - 0.5f);
        if (tmp > 32767) tmp = 32767;
        if (tmp < -32768) tmp = -32768;
        out[i] = (int16_t)tmp;
    }
}

/* ================================================================= top level (decoder/lc3_decoder.rs) */
int lc3o_decoder_init(lc3o_decoder *d, int fs_hz, int frame_us) {
    memset(d, 0, sizeof(*d));
    if (lc3o_config_new(&d->cfg, fs_hz, frame_us)) return -1;
    lc3o_dct4_init(&d->dct, d->cfg.nf);
    d->plc_seed = 24607; /* packet_loss_concealment.rs:31 */
    d->plc_alpha = 1.0f;
    lc3o_ltpf_dec_init(&d->cfg, &d->ltpf);
    return 0;
}

/* DecoderChannel::decode :73-154 */
int lc3o_decode_frame(lc3o_decoder *d, int bits_per_sample, const uint8_t *in, int nbytes, int16_t *pcm_out) {
    const lc3o_config *c = &d->cfg;
    int nbits = nbytes * 8, k, ok;
    int32_t x[LC3O_MAX_NE];
    lc3o_reader rd = {0, 0};
    lc3o_side_info si;
    lc3o_arith_data ad;
    lc3o_ltpf_info info;
    if (bits_per_sample != 16) return 1;
    d->frame_index += 1;
    memset(x, 0, sizeof(x));
    ok = lc3o_dec_side_info(in, nbytes, &rd, c->fs_ind, c->ne, &si) == 0;
    if (ok) ok = lc3o_dec_arith(in, nbytes, &rd, c->fs_ind, c->ne, &si, c->n_ms_10, x, &ad) == 0;
    if (ok) {
        for (k = 0; k < c->ne; k++) d->spec_lines[k] = (float)x[k];
        lc3o_dec_residual(si.lsb_mode, ad.residual_bits, ad.n_residual_bits, d->spec_lines, c->ne);
        lc3o_dec_noise_filling(ad.is_zero_frame, ad.noise_filling_seed, si.bandwidth, c->n_ms_10, si.noise_factor, x,
                               d->spec_lines, c->ne);
        lc3o_dec_global_gain(ad.frame_num_bits, c->fs_ind, si.global_gain_index, d->spec_lines, c->ne);
        lc3o_dec_tns(c->n_ms_10, si.bandwidth, si.num_tns_filters, ad.rc_order, ad.rc_i, d->spec_lines);
        lc3o_dec_sns(c, &si.sns_vq, d->spec_lines);
        lc3o_dec_plc_save(d, d->spec_lines);
        info = si.ltpf;
        d->last_frame_was_plc = 0;
    } else {
        info = lc3o_dec_plc_load(d, d->spec_lines);
        d->last_frame_was_plc = 1;
    }
    lc3o_dec_imdct(d, d->spec_lines, d->freq_samples);
    lc3o_dec_ltpf(c, &d->ltpf, &info, nbits, d->freq_samples);
    lc3o_dec_output(d->freq_samples, pcm_out, c->nf);
    return 0;
}

/* lc3_decoder.rs:155-162,236-244 */
void lc3o_decoder_working_buffer_lengths(int num_channels, int fs_hz, int frame_us, int64_t out[2]) {
    lc3o_config c;
    lc3o_ltpf_dec_state st;
    int64_t dct_scaler, ltpf_len, c_num, c_den, scratch;
    out[0] = out[1] = 0;
    if (lc3o_config_new(&c, fs_hz, frame_us)) return;
    lc3o_ltpf_dec_init(&c, &st);
    dct_scaler = c.nf / 2 + (c.nf - c.ne) + (c.nf - c.z) + c.nf * 2 + c.nf;
    c_num = st.l_num + 1;
    c_den = st.l_den + 1;
    scratch = st.l_num + st.norm;
    ltpf_len = c_den * 3 + c_num * 2 + (int64_t)c.nf * st.num_mem_blocks * 2 + scratch;
    out[0] = (c.ne + c.ne + dct_scaler + ltpf_len) * num_channels;
    out[1] = (int64_t)(c.nf / 2 * 4) * num_channels;
}
