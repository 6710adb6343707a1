/* ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Flat-argument shims so that tests/ can drive single stages of the restatement
 * through ctypes without mirroring C structs (one shim per reference KAT). */
#include "lc3_oracle.h"
#include "lc3_math.h"
#include <stdlib.h>
#include <string.h>

void *lc3o_encoder_new(int fs_hz, int frame_us) {
    lc3o_encoder *e = (lc3o_encoder *)malloc(sizeof(*e));
    if (e && lc3o_encoder_init(e, fs_hz, frame_us)) { free(e); return 0; }
    return e;
}
void lc3o_encoder_free(void *e) { free(e); }
void *lc3o_decoder_new(int fs_hz, int frame_us) {
    lc3o_decoder *d = (lc3o_decoder *)malloc(sizeof(*d));
    if (d && lc3o_decoder_init(d, fs_hz, frame_us)) { free(d); return 0; }
    return d;
}
void lc3o_decoder_free(void *d) { free(d); }
int lc3o_decoder_last_plc(void *d) { return ((lc3o_decoder *)d)->last_frame_was_plc; }

void lc3o_kat_config(int fs_hz, int frame_us, int out[7]) {
    lc3o_config c;
    memset(out, 0, sizeof(int) * 7);
    if (lc3o_config_new(&c, fs_hz, frame_us)) return;
    out[0] = c.fs_ind; out[1] = c.fs; out[2] = c.ne; out[3] = c.n_ms_10; out[4] = c.nb; out[5] = c.nf; out[6] = c.z;
}
void lc3o_kat_fft(int nfft, const float *re, const float *im, float *ore, float *oim) {
    lc3o_fft f;
    lc3o_cpx in[240], out[240];
    int i;
    lc3o_fft_init(&f, nfft);
    for (i = 0; i < nfft; i++) { in[i].r = re[i]; in[i].i = im[i]; }
    lc3o_fft_run(&f, in, out);
    for (i = 0; i < nfft; i++) { ore[i] = out[i].r; oim[i] = out[i].i; }
}
void lc3o_kat_dct4(int nf, float *buf) {
    lc3o_dct4 *d = (lc3o_dct4 *)malloc(sizeof(*d));
    lc3o_dct4_init(d, nf);
    lc3o_dct4_run(d, buf);
    free(d);
}
int lc3o_kat_enc_mdct(void *enc, const int16_t *x_s, float *out, float *eb) {
    return lc3o_enc_mdct_run((lc3o_encoder *)enc, x_s, out, eb);
}
void lc3o_kat_bandwidth(int fs_hz, int frame_us, const float *e_b, int out[2]) {
    lc3o_config c; lc3o_bw_result r;
    lc3o_config_new(&c, fs_hz, frame_us);
    r = lc3o_enc_bandwidth(&c, e_b);
    out[0] = r.bandwidth_ind; out[1] = r.nbits_bandwidth;
}
int lc3o_kat_attack(void *enc, const int16_t *x_s, int nbytes, float fout[2], int iout[3]) {
    lc3o_encoder *e = (lc3o_encoder *)enc;
    int r = lc3o_enc_attack(&e->cfg, &e->att, x_s, nbytes);
    fout[0] = e->att.energy_last; fout[1] = e->att.max_energy_last;
    iout[0] = e->att.attack_pos_last; iout[1] = e->att.ds_tm1; iout[2] = e->att.ds_tm2;
    return r;
}
static void sns_out(const lc3o_sns_result *r, int64_t out[7]) {
    out[0] = r->ind_lf; out[1] = r->ind_hf; out[2] = r->shape_j; out[3] = r->gind;
    out[4] = r->ls_inda; out[5] = r->ls_indb; out[6] = r->index_joint_j;
}
void lc3o_kat_sns(int fs_hz, int frame_us, float *x, const float *e_b, int attack, int64_t out[7]) {
    lc3o_config c; lc3o_sns_result r;
    lc3o_config_new(&c, fs_hz, frame_us);
    r = lc3o_enc_sns(&c, x, e_b, attack);
    sns_out(&r, out);
}
void lc3o_kat_sns_quant(const float *scf, float *scfq, int64_t out[7]) {
    lc3o_sns_result r;
    memset(&r, 0, sizeof(r));
    lc3o_enc_sns_quant(scf, scfq, &r);
    sns_out(&r, out);
}
void lc3o_kat_tns(int fs_hz, int frame_us, float *x, int p_bw, int nbits, int nn, int iout[21], float rc_q[16]) {
    lc3o_config c; lc3o_tns_result r;
    int i;
    lc3o_config_new(&c, fs_hz, frame_us);
    r = lc3o_enc_tns(&c, x, p_bw, nbits, nn);
    iout[0] = r.nbits_tns; iout[1] = r.lpc_weighting; iout[2] = r.num_tns_filters;
    iout[3] = r.rc_order[0]; iout[4] = r.rc_order[1];
    for (i = 0; i < 16; i++) { iout[5 + i] = r.rc_i[i]; rc_q[i] = r.rc_q[i]; }
}
void lc3o_kat_ltpf_enc(void *enc, const int16_t *x_s, int nn, int nbits, int out[4]) {
    lc3o_encoder *e = (lc3o_encoder *)enc;
    lc3o_ltpf_result r = lc3o_enc_ltpf(&e->cfg, &e->ltpf, x_s, nn, nbits);
    out[0] = r.pitch_index; out[1] = r.pitch_present; out[2] = r.ltpf_active; out[3] = r.nbits_ltpf;
}
void lc3o_kat_quant(void *enc, const float *x_f, int16_t *x_q, int nbits, int nbw, int ntns, int nltpf, int iout[7],
                    float *gg) {
    lc3o_encoder *e = (lc3o_encoder *)enc;
    lc3o_quant_result r = lc3o_enc_quant(&e->cfg, &e->quant, x_f, x_q, nbits, nbw, ntns, nltpf);
    iout[0] = r.gg_ind; iout[1] = r.nbits_spec; iout[2] = r.nbits_lsb; iout[3] = r.nbits_trunc;
    iout[4] = r.lsb_mode; iout[5] = r.rate_flag; iout[6] = r.lastnz_trunc;
    *gg = r.gg;
}
int lc3o_kat_noise_factor(int fs_hz, int frame_us, const float *x_f, const int16_t *x_q, int bw, float gg) {
    lc3o_config c;
    lc3o_config_new(&c, fs_hz, frame_us);
    return lc3o_enc_noise_factor(&c, x_f, x_q, bw, gg);
}
/* bitstream_encoding_run: all arguments flat */
void lc3o_kat_bitstream(int fs_hz, int frame_us, int bw_ind, int nbits_bw, int lastnz_trunc, int lsb_mode, int gg_ind,
                        int num_tns, const int *rc_order, const int *rc_i, int lpc_weighting, int pitch_present,
                        int ltpf_active, int pitch_index, const int64_t *sns /* lf,hf,shape,gind,ls_inda,joint */,
                        int noise_factor, int rate_flag, int nbits_lsb, const int16_t *x_q, const uint8_t *res_bits,
                        int n_res, uint8_t *out, int nbytes) {
    lc3o_config c; lc3o_bw_result bw; lc3o_sns_result s; lc3o_tns_result t; lc3o_ltpf_result l; lc3o_quant_result q;
    int i;
    lc3o_config_new(&c, fs_hz, frame_us);
    memset(&s, 0, sizeof(s)); memset(&t, 0, sizeof(t)); memset(&q, 0, sizeof(q));
    bw.bandwidth_ind = bw_ind; bw.nbits_bandwidth = nbits_bw;
    s.ind_lf = (int)sns[0]; s.ind_hf = (int)sns[1]; s.shape_j = (int)sns[2]; s.gind = (int)sns[3];
    s.ls_inda = (int)sns[4]; s.index_joint_j = (uint32_t)sns[5];
    t.num_tns_filters = num_tns; t.lpc_weighting = lpc_weighting;
    t.rc_order[0] = rc_order[0]; t.rc_order[1] = rc_order[1];
    for (i = 0; i < 16; i++) t.rc_i[i] = rc_i[i];
    l.pitch_present = pitch_present; l.ltpf_active = ltpf_active; l.pitch_index = pitch_index; l.nbits_ltpf = 0;
    q.gg_ind = gg_ind; q.lsb_mode = lsb_mode; q.rate_flag = rate_flag; q.lastnz_trunc = lastnz_trunc;
    q.nbits_lsb = nbits_lsb;
    lc3o_enc_bitstream(&c, bw, &s, &t, l, &q, res_bits, n_res, noise_factor, x_q, out, nbytes);
}
/* buffer_writer KATs are covered through lc3o_enc_bitstream; reader KATs: */
int lc3o_kat_read_tail_usize(const uint8_t *buf, int len, int head, int tail, int nbits, uint32_t *val, int *tail_out) {
    lc3o_reader r = {head, tail};
    int rc = lc3o_read_tail_usize(&r, buf, len, nbits, val);
    *tail_out = r.tail_bit_cursor;
    return rc;
}
int lc3o_kat_read_tail_bool(const uint8_t *buf, int len, int head, int tail, int *bit) {
    lc3o_reader r = {head, tail};
    return lc3o_read_tail_bool(&r, buf, len, bit);
}
static void si_out(const lc3o_side_info *si, int64_t o[20]) {
    o[0] = si->bandwidth; o[1] = si->lastnz; o[2] = si->lsb_mode; o[3] = si->global_gain_index;
    o[4] = si->num_tns_filters; o[5] = si->rc_order_ari_input[0]; o[6] = si->rc_order_ari_input[1];
    o[7] = si->sns_vq.ind_lf; o[8] = si->sns_vq.ind_hf; o[9] = si->sns_vq.ls_inda; o[10] = si->sns_vq.ls_indb;
    o[11] = si->sns_vq.idx_a; o[12] = si->sns_vq.idx_b; o[13] = si->sns_vq.submode_lsb;
    o[14] = si->sns_vq.submode_msb; o[15] = si->sns_vq.g_ind; o[16] = si->ltpf.pitch_present;
    o[17] = si->ltpf.is_active; o[18] = si->ltpf.pitch_index; o[19] = si->noise_factor;
}
static void si_in(lc3o_side_info *si, const int64_t o[20]) {
    memset(si, 0, sizeof(*si));
    si->bandwidth = (int)o[0]; si->lastnz = (int)o[1]; si->lsb_mode = (int)o[2]; si->global_gain_index = (int)o[3];
    si->num_tns_filters = (int)o[4]; si->rc_order_ari_input[0] = (int)o[5]; si->rc_order_ari_input[1] = (int)o[6];
    si->sns_vq.ind_lf = (int)o[7]; si->sns_vq.ind_hf = (int)o[8]; si->sns_vq.ls_inda = (int)o[9];
    si->sns_vq.ls_indb = (int)o[10]; si->sns_vq.idx_a = (uint32_t)o[11]; si->sns_vq.idx_b = (uint32_t)o[12];
    si->sns_vq.submode_lsb = (int)o[13]; si->sns_vq.submode_msb = (int)o[14]; si->sns_vq.g_ind = (int)o[15];
    si->ltpf.pitch_present = (int)o[16]; si->ltpf.is_active = (int)o[17]; si->ltpf.pitch_index = (int)o[18];
    si->noise_factor = (int)o[19];
}
int lc3o_kat_side_info(const uint8_t *buf, int len, int fs_ind, int ne, int64_t out[20], int *tail) {
    lc3o_reader r = {0, 0};
    lc3o_side_info si;
    int rc = lc3o_dec_side_info(buf, len, &r, fs_ind, ne, &si);
    si_out(&si, out);
    *tail = r.tail_bit_cursor;
    return rc;
}
int lc3o_kat_arith(const uint8_t *buf, int len, int head, int tail, int fs_ind, int ne, const int64_t si_flat[20],
                   int n_ms_10, int32_t *x, int iout[22], uint8_t *res_bits) {
    lc3o_reader r = {head, tail};
    lc3o_side_info si;
    lc3o_arith_data ad;
    int rc, i;
    si_in(&si, si_flat);
    rc = lc3o_dec_arith(buf, len, &r, fs_ind, ne, &si, n_ms_10, x, &ad);
    iout[0] = ad.rc_order[0]; iout[1] = ad.rc_order[1];
    for (i = 0; i < 16; i++) iout[2 + i] = ad.rc_i[i];
    iout[18] = ad.n_residual_bits; iout[19] = ad.noise_filling_seed; iout[20] = ad.is_zero_frame;
    iout[21] = ad.frame_num_bits;
    memcpy(res_bits, ad.residual_bits, 480);
    return rc;
}
void lc3o_kat_dec_sns(int fs_hz, int frame_us, const int64_t si_flat[20], float *spec) {
    lc3o_config c; lc3o_side_info si;
    lc3o_config_new(&c, fs_hz, frame_us);
    si_in(&si, si_flat);
    lc3o_dec_sns(&c, &si.sns_vq, spec);
}
void lc3o_kat_plc(int ne, const float *save, int n_loads, float *out) {
    lc3o_decoder *d = (lc3o_decoder *)lc3o_decoder_new(48000, 10000);
    int i;
    d->cfg.ne = ne;
    lc3o_dec_plc_save(d, save);
    for (i = 0; i < n_loads; i++) lc3o_dec_plc_load(d, out);
    free(d);
}
void lc3o_kat_imdct(void *dec, const float *spec, float *freq) { lc3o_dec_imdct((lc3o_decoder *)dec, spec, freq); }
void lc3o_kat_dec_ltpf(void *dec, int is_active, int pitch_present, int pitch_index, int nbits, float *freq) {
    lc3o_decoder *d = (lc3o_decoder *)dec;
    lc3o_ltpf_info info = {pitch_present, is_active, pitch_index};
    lc3o_dec_ltpf(&d->cfg, &d->ltpf, &info, nbits, freq);
}
float lc3o_kat_powf(float x, float y) { return lc3m_powf(x, y); }

/* vectorised float routines for the device-math test (tests/test_gpu_parity.py::test_device_math_on_the_device):
 * which = 2 log2f, 3 log10f, 4 exp2f, 5 asinf, 6 exp2_raw, 7 powf(10, x), 8 sinf */
void lc3o_kat_math(int which, const float *x, int n, float *out) {
    int i;
    for (i = 0; i < n; i++) {
        switch (which) {
        case 2: out[i] = lc3m_log2f(x[i]); break;
        case 3: out[i] = lc3m_log10f(x[i]); break;
        case 4: out[i] = lc3m_exp2f(x[i]); break;
        case 5: out[i] = lc3m_asinf(x[i]); break;
        case 6: out[i] = lc3m_exp2_raw(x[i]); break;
        case 7: out[i] = lc3m_powf(10.0f, x[i]); break;
        case 8: out[i] = lc3m_sinf(x[i]); break;
        default: out[i] = 0.0f;
        }
    }
}
