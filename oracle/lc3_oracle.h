/* ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the per-frame DSP hot path of ninjasource/lc3-codec
 * v0.2.0 (Lc3Encoder::encode_frame / Lc3Decoder::decode_frame and every stage
 * below them).  The reference is Rust and cannot be built
 * here (no cargo/rustc, no crates), so this restatement is the parity oracle and
 * the timed CPU baseline ("port").  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (lc3-codec_amd/) never does.
 *
 * Pinning: validated against every hot-path known-answer vector in the
 * reference's own in-file tests (tests/golden/ref_kats.json, tests/test_oracle_kats.py).
 * Third-party float math (libm crate, fast-math) is restated in lc3_math.c; the
 * run-time flavour of log2f/exp2f/log10f/asinf is unpinned (see lc3_math.h).
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef LC3_ORACLE_H_
#define LC3_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC3O_MAX_NF 480
#define LC3O_MAX_NE 400

typedef struct { float r, i; } lc3o_cpx; /* common/complex.rs:4-8 */

/* common/config.rs:18-100 */
typedef struct {
    int fs_ind, fs, ne, n_ms_10 /* 1 = TenMs, 0 = SevenPointFiveMs */, nb, nf, z;
    int spec_flags; /* LC3O_SPEC_*: 0 = every deviation of the reference from the LC3 specification reproduced (the default) */
} lc3o_config;
/* Opt-in corrections of the reference's deviations, one bit each; the product mirrors them as LC3GPU_SPEC_*
 * (include/lc3gpu.h).  With a bit set there is no reference behaviour to compare with: the oracle then only pins the GPU. */
#define LC3O_SPEC_8KHZ_ENCODE 1     /* A6: an 8 kHz encoder can be built (the bandwidth detector returns at once, :66-71) */
#define LC3O_SPEC_TNS_SSWB_STOP 2   /* A5: 10 ms, bandwidth index 2: the TNS filter stops at line 240, not 200 */
#define LC3O_SPEC_BW_CUTOFF_DB 4    /* A7: cut-off test on 10 log10(eps + E[n-L] / E[n]) as the note at bandwidth_detector.rs:106-107 says the specification asks */
#define LC3O_SPEC_SNS_LAST_GAIN 8   /* A3: the SNS gain search also tries the last gain of every shape */
#define LC3O_SPEC_NBITS_SPEC_OLD 16 /* A1: nbits_spec_old follows nbits_spec */
int lc3o_config_new(lc3o_config *c, int fs_hz, int frame_us);

/* common/kissfft.rs + common/dct_iv.rs */
typedef struct {
    int nfft;
    int factors[64];
    lc3o_cpx tw[LC3O_MAX_NF / 2];
} lc3o_fft;
typedef struct {
    int nf;
    lc3o_fft fft;
    lc3o_cpx tw[LC3O_MAX_NF / 2];
    lc3o_cpx in[LC3O_MAX_NF / 2], out[LC3O_MAX_NF / 2];
} lc3o_dct4;
void lc3o_fft_init(lc3o_fft *f, int nfft);
void lc3o_fft_run(const lc3o_fft *f, const lc3o_cpx *fin, lc3o_cpx *fout);
void lc3o_dct4_init(lc3o_dct4 *d, int nf);
void lc3o_dct4_run(lc3o_dct4 *d, float *buf);

/* ---------------------------------------------------------------- encoder results */
typedef struct { int bandwidth_ind, nbits_bandwidth; } lc3o_bw_result;
typedef struct { int ind_lf, ind_hf, shape_j, gind, ls_inda, ls_indb; uint32_t index_joint_j; } lc3o_sns_result;
typedef struct {
    int nbits_tns, lpc_weighting, num_tns_filters;
    int rc_order[2];
    int rc_i[16];
    float rc_q[16];
} lc3o_tns_result;
typedef struct { int pitch_index, pitch_present, ltpf_active, nbits_ltpf; } lc3o_ltpf_result;
typedef struct {
    int gg_ind, nbits_spec, nbits_lsb, nbits_trunc, lsb_mode, rate_flag, lastnz_trunc;
    float gg;
} lc3o_quant_result;

/* ---------------------------------------------------------------- encoder state */
typedef struct { /* encoder/attack_detector.rs:9-22 */
    float energy_last, max_energy_last;
    int attack_pos_last, ds_tm1, ds_tm2;
} lc3o_attack_state;
typedef struct { /* encoder/long_term_post_filter.rs:21-42 */
    int t_prev;
    float mem_pitch;
    int mem_ltpf_active;
    float mem_nc, mem_mem_nc;
    int16_t x_s_ext[60 + LC3O_MAX_NF];
    float x12[128 + 44 + 232];
    float x6[64 + 114];
    float h50_m1, h50_m2;
} lc3o_ltpf_enc_state;
typedef struct { /* encoder/spectral_quantization.rs:50-61 */
    int reset_offset_old;
    float nbits_offset_old;
    int nbits_spec_old, nbits_est_old;
} lc3o_quant_state;

typedef struct {
    lc3o_config cfg;
    lc3o_dct4 dct;
    int16_t tbuf[2 * LC3O_MAX_NF]; /* ModDiscreteCosTrans::freq, encoder/modified_dct.rs:21 */
    lc3o_attack_state att;
    lc3o_ltpf_enc_state ltpf;
    lc3o_quant_state quant;
    /* scratch */
    float mdct_out[LC3O_MAX_NF];
    float energy_bands[64];
    int16_t x_q[LC3O_MAX_NE];
    uint8_t res_bits[LC3O_MAX_NE];
    int frame_index;
} lc3o_encoder;

int lc3o_encoder_init(lc3o_encoder *e, int fs_hz, int frame_us);
int lc3o_encoder_init_spec(lc3o_encoder *e, int fs_hz, int frame_us, int spec_flags);
/* Lc3Encoder::encode_frame, encoder/lc3_encoder.rs:63-112,175-191.  nbytes = buf_out.len() */
int lc3o_encode_frame(lc3o_encoder *e, const int16_t *pcm, uint8_t *out, int nbytes);

/* stage entry points (for the reference's stage KATs) */
int lc3o_enc_mdct_run(lc3o_encoder *e, const int16_t *x_s, float *out, float *energy_bands);
lc3o_bw_result lc3o_enc_bandwidth(const lc3o_config *c, const float *e_b);
int lc3o_enc_attack(const lc3o_config *c, lc3o_attack_state *st, const int16_t *x_s, int nbytes);
lc3o_sns_result lc3o_enc_sns(const lc3o_config *c, float *x, const float *e_b, int attack);
void lc3o_enc_sns_quant(const float *scf, float *scfq, lc3o_sns_result *r);
void lc3o_enc_sns_quant_spec(const float *scf, float *scfq, lc3o_sns_result *r, int spec_flags);
lc3o_tns_result lc3o_enc_tns(const lc3o_config *c, float *x, int p_bw, int nbits, int near_nyquist);
void lc3o_ltpf_enc_init(const lc3o_config *c, lc3o_ltpf_enc_state *st);
lc3o_ltpf_result lc3o_enc_ltpf(const lc3o_config *c, lc3o_ltpf_enc_state *st, const int16_t *x_s,
                               int near_nyquist, int nbits);
lc3o_quant_result lc3o_enc_quant(const lc3o_config *c, lc3o_quant_state *st, const float *x_f, int16_t *x_q,
                                 int nbits, int nbits_bw, int nbits_tns, int nbits_ltpf);
int lc3o_enc_residual(int nbits_spec, int nbits_trunc, int ne, float gg, const float *x_f, const int16_t *x_q,
                      uint8_t *bits_out);
int lc3o_enc_noise_factor(const lc3o_config *c, const float *x_f, const int16_t *x_q, int bw_ind, float gg);
void lc3o_enc_bitstream(const lc3o_config *c, lc3o_bw_result bw, const lc3o_sns_result *sns,
                        const lc3o_tns_result *tns, lc3o_ltpf_result ltpf, const lc3o_quant_result *spec,
                        const uint8_t *res_bits, int n_res_bits, int noise_factor, const int16_t *x_q,
                        uint8_t *out, int nbytes);

/* ---------------------------------------------------------------- decoder */
typedef struct { int pitch_present, is_active, pitch_index; } lc3o_ltpf_info;
typedef struct {
    int ind_lf, ind_hf, ls_inda, ls_indb;
    uint32_t idx_a, idx_b;
    int submode_lsb, submode_msb, g_ind;
} lc3o_sns_vq;
typedef struct { /* decoder/side_info.rs:22-34 */
    int bandwidth, lastnz, lsb_mode, global_gain_index, num_tns_filters;
    int rc_order_ari_input[2];
    lc3o_sns_vq sns_vq;
    lc3o_ltpf_info ltpf;
    int noise_factor;
} lc3o_side_info;
typedef struct { /* decoder/arithmetic_codec.rs:99-107 */
    int rc_order[2];
    int rc_i[16];
    uint8_t residual_bits[480];
    int n_residual_bits;
    int noise_filling_seed, is_zero_frame, frame_num_bits;
} lc3o_arith_data;
typedef struct { int head_byte_cursor, tail_bit_cursor; } lc3o_reader;

typedef struct { /* decoder/long_term_post_filter.rs:12-29 */
    int num_mem_blocks, norm, l_num, l_den;
    int ltpf_active_prev, block_start_index;
    float c_num[12], c_den[14], c_num_mem[12], c_den_mem[14];
    int p_int_mem, p_fr_mem;
    float x_hat_ltpf_mem[3 * LC3O_MAX_NF], x_hat_mem[3 * LC3O_MAX_NF];
} lc3o_ltpf_dec_state;

typedef struct {
    lc3o_config cfg;
    lc3o_dct4 dct;
    float spec_lines[LC3O_MAX_NE];
    float freq_samples[LC3O_MAX_NF];
    /* PLC, decoder/packet_loss_concealment.rs:7-22 */
    float plc_last_good[LC3O_MAX_NE];
    int plc_num_lost;
    float plc_alpha;
    uint32_t plc_seed;
    /* IMDCT, decoder/modified_dct.rs:14-22 */
    float mem_ola_add[LC3O_MAX_NF];
    float t_hat[2 * LC3O_MAX_NF];
    lc3o_ltpf_dec_state ltpf;
    int frame_index;
    int last_frame_was_plc; /* diagnostic only */
} lc3o_decoder;

int lc3o_decoder_init(lc3o_decoder *d, int fs_hz, int frame_us);
/* Lc3Decoder::decode_frame, decoder/lc3_decoder.rs:73-154,217-234.
 * returns 0 = Ok(()), 1 = Err(Only16BitsPerAudioSampleSupported) */
int lc3o_decode_frame(lc3o_decoder *d, int bits_per_sample, const uint8_t *in, int nbytes, int16_t *pcm_out);

/* stage entry points */
int lc3o_read_tail_usize(lc3o_reader *r, const uint8_t *buf, int len, int num_bits, uint32_t *val);
int lc3o_read_tail_bool(lc3o_reader *r, const uint8_t *buf, int len, int *bit);
int lc3o_dec_side_info(const uint8_t *buf, int len, lc3o_reader *r, int fs_ind, int ne, lc3o_side_info *si);
int lc3o_dec_arith(const uint8_t *buf, int len, lc3o_reader *r, int fs_ind, int ne, const lc3o_side_info *si,
                   int n_ms_10, int32_t *x, lc3o_arith_data *ad);
void lc3o_dec_residual(int lsb_mode, const uint8_t *bits, int nbits, float *spec, int ne);
void lc3o_dec_noise_filling(int is_zero_frame, int seed, int bandwidth, int n_ms_10, int noise_factor,
                            const int32_t *x_int, float *spec, int ne);
void lc3o_dec_global_gain(int frame_num_bits, int fs_ind, int gg_ind, float *spec, int ne);
void lc3o_dec_tns(int n_ms_10, int bandwidth, int num_tns_filters, const int *rc_order, const int *rc_i,
                  float *spec);
void lc3o_dec_sns(const lc3o_config *c, const lc3o_sns_vq *sns, float *spec);
void lc3o_mpvq_deenum(int dim_in, int k_val_in, int ls_ind, uint32_t mpvq_ind, int32_t *vec_out);
void lc3o_dec_plc_save(lc3o_decoder *d, const float *spec);
lc3o_ltpf_info lc3o_dec_plc_load(lc3o_decoder *d, float *spec);
void lc3o_dec_imdct(lc3o_decoder *d, const float *spec, float *freq);
void lc3o_ltpf_dec_init(const lc3o_config *c, lc3o_ltpf_dec_state *st);
void lc3o_dec_ltpf(const lc3o_config *c, lc3o_ltpf_dec_state *st, const lc3o_ltpf_info *info, int nbits,
                   float *freq);
void lc3o_dec_output(const float *x, int16_t *out, int n);

/* ---------------------------------------------------------------- batch helpers (cpu_baseline, parity tests)
 * streams are independent codec channels; frames of one stream are consecutive in time.
 * pcm: int16[S][T][nf]   bytes: uint8[S][T][nbytes]  (both stream-major)
 * n_threads <= 1 runs in the caller's thread. */
int lc3o_encode_batch_spec(int fs_hz, int frame_us, int nbytes, int n_streams, int n_frames, const int16_t *pcm, uint8_t *bytes,
                          int n_threads, int spec_flags);
int lc3o_encode_batch(int fs_hz, int frame_us, int nbytes, int n_streams, int n_frames, const int16_t *pcm,
                      uint8_t *bytes, int n_threads);
int lc3o_decode_batch(int fs_hz, int frame_us, int nbytes, int n_streams, int n_frames, const uint8_t *bytes,
                      int16_t *pcm, int n_threads);
/* bench.py's cpu_baseline leg (lc3_batch.c): n_threads threads with persistent codec objects code for `seconds`; nothing is
 * allocated and no thread is created inside the timed region */
int lc3o_timed_run(int fs_hz, int frame_us, int nbytes, int n_frames, const int16_t *pcm, int n_distinct, int n_threads, int roundtrip,
                   double seconds, double *frames_out, double *elapsed_out);
extern int lc3o_ltpf_trans_counting;
/* buffer lengths the reference API reports (lc3_encoder.rs:194-209, lc3_decoder.rs:236-244) */
void lc3o_encoder_working_buffer_lengths(int num_channels, int fs_hz, int frame_us, int64_t out[3]);
void lc3o_decoder_working_buffer_lengths(int num_channels, int fs_hz, int frame_us, int64_t out[2]);

#ifdef __cplusplus
}
#endif
#endif
