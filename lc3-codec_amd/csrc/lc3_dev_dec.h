// LC3 batched decoder for MI355X -- device-side stages (one wavefront per stream).
// Mirrors DecoderChannel::decode (reference decoder/lc3_decoder.rs:73-154) stage by stage.
// See lc3_dev_common.h for the execution model and the bit-exactness contract.
#pragma once
#include "lc3_dev_common.h"

// Persistent per-stream decoder state.  `core` is what a wave keeps resident in LDS while it
// works on the stream; plc_last_good stays in HBM (written once per good frame, read only when concealing).
struct lc3_dec_core {
    float mem_ola[304];            // IMDCT overlap memory, nf - z used (decoder/modified_dct.rs:30,149)
    float x_hat_ltpf_mem[1080];    // LTPF output ring, num_mem_blocks * nf (decoder/long_term_post_filter.rs:127-128)
    float x_tail[12];              // last l_num samples of the previous LTPF input frame: the only part of the
                                   // reference's x_hat_mem ring that is ever read back (:380-387, k <= l_num)
    float c_num[12], c_den[14];    // current LTPF coefficients (:20-27); the *_mem copies are per-frame temporaries
    int ltpf_active_prev, block_start_index, p_int_mem, p_fr_mem;
    int plc_num_lost;
    float plc_alpha;
    uint32_t plc_seed;
    int plc_events;                // counter: frames concealed so far (not in the reference; reporting only)
};
struct lc3_dec_state {
    lc3_dec_core core;
    float plc_last_good[LC3_MAX_NE];  // decoder/packet_loss_concealment.rs:7-22
};
#define LC3_DEC_CORE_WORDS ((int)(sizeof(lc3_dec_core) / 4))

// LDS working set of one decoder wave (~13 KB -> 12 waves per CU)
struct lc3_dec_lds {
    lc3_dec_core st;
    float spec[LC3_MAX_NF];        // spec_lines, then freq_samples
    lc3_cpx fa[LC3_MAX_NF / 2];    // save_lev during parsing | FFT in   | t_hat_mdct[0 .. nf)
    lc3_cpx fb[LC3_MAX_NF / 2];    // integer spectrum xi     | FFT work | t_hat_mdct[nf .. 2nf)  (contiguous with fa)
    alignas(16) uint8_t in[LC3_MAX_NE + 112];  // frame bytes (padded to 512: read as 128 dwords by the parser)
    float sm[192];
    int ism[64];
    unsigned long long prof_last;  // diagnostic build: time of the previous stage stamp
};

__device__ __forceinline__ void lc3_dec_state_init(lc3_dec_lds &L, int lane, lc3_dec_state *g) {
    for (int i = lane; i < LC3_MAX_NE; i += LC3_WAVE) g->plc_last_good[i] = 0.0f;
    int *w = (int *)&L.st;
    for (int i = lane; i < LC3_DEC_CORE_WORDS; i += LC3_WAVE) w[i] = 0;
    LC3_SYNC();
    if (lane == 0) {
        L.st.plc_seed = 24607;  // packet_loss_concealment.rs:31
        L.st.plc_alpha = 1.0f;
    }
    LC3_SYNC();
}
__device__ __forceinline__ void lc3_dec_state_load(lc3_dec_lds &L, int lane, const lc3_dec_state *g) {
    const int *src = (const int *)&g->core;
    int *w = (int *)&L.st;
    for (int i = lane; i < LC3_DEC_CORE_WORDS; i += LC3_WAVE) w[i] = src[i];
    LC3_SYNC();
}
__device__ __forceinline__ void lc3_dec_state_store(lc3_dec_lds &L, int lane, lc3_dec_state *g) {
    int *dst = (int *)&g->core;
    const int *w = (const int *)&L.st;
    LC3_SYNC();
    for (int i = lane; i < LC3_DEC_CORE_WORDS; i += LC3_WAVE) dst[i] = w[i];
}

// ------------------------------------------------------------------------------------------
// Frame parsing runs in "uniform-scalar" style: the whole wave executes the serial bitstream state
// machine with identical (wave-uniform) values, so on the GPU the state lives in SGPRs and the integer
// work issues on the scalar unit instead of a 1-of-64-lanes vector stream.  LDS reads are made uniform
// with LC3_U() (v_readfirstlane), LDS writes go through LC3_UST() (lane 0 stores), and the 17-way
// symbol search of the range decoder is done by 17 lanes at once (lc3_sym_search: ballot + readlane).
// The including translation unit defines LC3_UNIFORM_LEADER / LC3_U / LC3_UST / lc3_sym_search.
// ------------------------------------------------------------------------------------------
// D1: BufferReader (decoder/buffer_reader.rs:11-116)
__device__ __forceinline__ int lc3_rd_tail(lc3_reader &r, int num_bits, uint32_t &val) {  // read_tail_usize :63-98
    const int byte_index = r.tail / 8, bit_index = r.tail % 8;
    const int bits_left = 8 - bit_index;
    const int add_bytes = (num_bits > bits_left && num_bits < 8) ? 2 : 1;
    const int num_bytes = num_bits / 8 + add_bytes;
    if (r.len - r.head - byte_index - num_bytes < 0) return -1;
    const int from = r.len - byte_index - num_bytes;
    uint32_t value = 0;
    if (num_bytes <= 4)
        for (int i = 0; i < num_bytes; i++) value = (value << 8) | (uint32_t)LC3_FRAME_BYTE(r, from + i);
    const int shift_by = 32 - num_bits - bit_index;
    value <<= shift_by;
    value >>= shift_by + bit_index;
    r.tail += num_bits;
    val = value;
    return 0;
}
__device__ __forceinline__ int lc3_rd_bool(lc3_reader &r, int &bit) {  // read_tail_bool :100-116
    const int byte_index = r.tail / 8, bit_index = r.tail % 8;
    if (r.len - r.head - byte_index + 2 < 0) return -1;
    const int from = r.len - byte_index - 1;
    if (from < 0) return -1;  // the reference would panic here; treated as a read error (-> PLC)
    uint32_t byte = (uint32_t)LC3_FRAME_BYTE(r, from);
    byte = (byte << (7 - bit_index)) & 0xffu;
    byte >>= 7;
    r.tail += 1;
    bit = byte == 1;
    return 0;
}

// decoded side information, kept in L.ism[]:
enum {
    SI_BW = 0, SI_LASTNZ, SI_LSB_MODE, SI_GG, SI_NUM_TNS, SI_ORD0, SI_ORD1, SI_IND_LF, SI_IND_HF, SI_LS_A, SI_LS_B,
    SI_IDX_A, SI_IDX_B, SI_SUB_LSB, SI_SUB_MSB, SI_G_IND, SI_PITCH_PRESENT, SI_LTPF_ACTIVE, SI_PITCH_INDEX, SI_NF,
    AD_ORD0, AD_ORD1, AD_NRES, AD_SEED, AD_ZERO, AD_OK, AD_RCI /* 16 entries */, AD_TAIL0 = AD_RCI + 16, AD_NRES_MAX,
    AD_HEAD
};

#define LC3_RD(nb, dst) do { if (lc3_rd_tail(r, (nb), (dst))) return -1; } while (0)
#define LC3_RDB(dst) do { if (lc3_rd_bool(r, (dst))) return -1; } while (0)

// D2: side_info_reader::read (decoder/side_info_reader.rs:29-200)
__device__ __forceinline__ int lc3_dec_side_info(lc3_reader &r, int fs_ind, int ne, int *si, int lane) {
    uint32_t v;
    int b, p_bw = 0, lastnz_bits = 0;
    const int nbits_bw = LC3C_NBITS_BW[fs_ind];
    if (nbits_bw > 0) {
        LC3_RD(nbits_bw, v);
        if (fs_ind < (int)v) return -2;
        p_bw = (int)v;
    }
    while ((1 << lastnz_bits) < ne / 2) lastnz_bits++;
    LC3_RD(lastnz_bits, v);
    const int lastnz = (int)((v + 1) << 1);
    LC3_UST(si[SI_LASTNZ], lastnz);
    if (lastnz > ne) return -3;
    LC3_RDB(b);
    LC3_UST(si[SI_LSB_MODE], b);
    LC3_RD(8, v);
    LC3_UST(si[SI_GG], (int)v);
    const int num_tns = p_bw < 3 ? 1 : 2;
    LC3_UST(si[SI_NUM_TNS], num_tns);
    LC3_UST(si[SI_ORD0], 0);
    LC3_UST(si[SI_ORD1], 0);
    for (int f = 0; f < num_tns; f++) {
        LC3_RDB(b);
        LC3_UST(si[SI_ORD0 + f], b);
    }
    LC3_RDB(b);
    const int pitch_present = b;
    LC3_UST(si[SI_PITCH_PRESENT], b);
    // read_sns_vq :131-200
    LC3_RD(5, v);
    LC3_UST(si[SI_IND_LF], (int)v);
    LC3_RD(5, v);
    LC3_UST(si[SI_IND_HF], (int)v);
    LC3_RDB(b);
    const int submode_msb = b;
    if (submode_msb == 0) LC3_RD(1, v);
    else LC3_RD(2, v);
    int g_ind = (int)v;
    LC3_RDB(b);
    LC3_UST(si[SI_LS_A], b);
    int submode_lsb = 0, ls_indb = 0;
    uint32_t idx_a, idx_b = 0;
    if (submode_msb == 0) {
        uint32_t tmp;
        LC3_RD(25, tmp);
        if (tmp >= 33460056u) return -4;
        const uint32_t idx_bor = tmp / 2390004u;
        idx_a = tmp - idx_bor * 2390004u;
        int s = (int)idx_bor - 2;
        if (s < 0) submode_lsb = 1;
        s = s + submode_lsb * 2;
        if (submode_lsb != 0) g_ind = (g_ind << 1) + s;
        else {
            idx_b = (uint32_t)s >> 1;
            ls_indb = s & 1;
        }
    } else {
        uint32_t tmp;
        LC3_RD(24, tmp);
        if (tmp >= 16708096u) return -5;
        if (tmp >= 15158272u) {
            tmp -= 15158272u;
            submode_lsb = 1;
            g_ind = (g_ind << 1) + (int)(tmp & 1u);
            idx_a = tmp >> 1;
        } else idx_a = tmp;
    }
    LC3_UST(si[SI_LS_B], ls_indb);
    LC3_UST(si[SI_IDX_A], (int)idx_a);
    LC3_UST(si[SI_IDX_B], (int)idx_b);
    LC3_UST(si[SI_SUB_LSB], submode_lsb);
    LC3_UST(si[SI_SUB_MSB], submode_msb);
    LC3_UST(si[SI_G_IND], g_ind);
    // read_long_term_post_filter_info :106-129
    int ltpf_active = 0, pitch_index = 0;
    if (pitch_present) {
        LC3_RDB(b);
        ltpf_active = b;
        LC3_RD(9, v);
        pitch_index = (int)v;
    }
    LC3_UST(si[SI_LTPF_ACTIVE], ltpf_active);
    LC3_UST(si[SI_PITCH_INDEX], pitch_index);
    LC3_RD(3, v);
    LC3_UST(si[SI_NF], (int)v);
    LC3_UST(si[SI_BW], p_bw);
    return 0;
}

// D3: arithmetic decoder (decoder/arithmetic_codec.rs:57-405)
// ac_decode :67-97.  The reference searches the symbol linearly from the top (`while low < tmp * cum[val]`);
// here lane i evaluates symbol i and the highest lane whose test holds is the answer (same integer compare).
__device__ __forceinline__ int lc3_ac_decode(lc3_reader &r, lc3_acdec &st, const int16_t *cum, const int16_t *freq,
                                             int nsym, int lane, int &sym) {
    const uint32_t tmp = st.range >> 10, limit = tmp << 10;
    if (st.low >= limit) return -1;
    uint32_t cval, fval;
    const int val = lc3_sym_search(cum, freq, nsym, st.low, tmp, lane, cval, fval);
    st.low -= tmp * cval;
    st.range = tmp * fval;
    while (st.range < 0x10000u) {
        st.low <<= 8;
        st.low &= 0x00ffffffu;
        if (r.head >= r.len) return -1;  // read_head_byte :42-50
        st.low += (uint32_t)LC3_FRAME_BYTE(r, r.head);
        r.head += 1;
        st.range <<= 8;
    }
    sym = val;
    return 0;
}
// Spectral symbols: the 64 x 17 (cum, freq) model and the 4096-entry context lookup are held in vector registers
// for the duration of the parse (lc3_dec_tabs, filled by the translation unit's lc3_dec_tabs_load) and read with
// v_readlane, so a symbol costs a handful of scalar compares instead of two dependent trips to memory.
__device__ __forceinline__ int lc3_ac_decode_spec(lc3_reader &r, lc3_acdec &st, const lc3_dec_tabs &T, int pki, int &sym) {
    const uint32_t tmp = st.range >> 10, limit = tmp << 10;
    if (st.low >= limit) return -1;
    uint32_t sv;
    const int val = lc3_tab_search(T, pki, st.low, tmp, sv);  // largest j with low >= tmp * cum[j]; sv = cum | freq << 16
    st.low -= tmp * (sv & 0xffffu);
    st.range = tmp * (sv >> 16);
    while (st.range < 0x10000u) {
        st.low <<= 8;
        st.low &= 0x00ffffffu;
        if (r.head >= r.len) return -1;
        st.low += (uint32_t)LC3_FRAME_BYTE(r, r.head);
        r.head += 1;
        st.range <<= 8;
    }
    sym = val;
    return 0;
}
__device__ __forceinline__ int lc3_read_res_bit(int32_t *x, lc3_reader &r, int idx, int &nbits_res, int &cont, int lane) {
    // :339-383
    int bit;
    if (nbits_res == 0) { cont = 0; return 0; }
    if (lc3_rd_bool(r, bit)) return -1;
    nbits_res -= 1;
    if (bit) {
        const int xv = LC3_U(x[idx]);
        if (xv > 0) LC3_UST(x[idx], xv + 1);
        else if (xv < 0) LC3_UST(x[idx], xv - 1);
        else {
            if (nbits_res == 0) { cont = 0; return 0; }
            if (lc3_rd_bool(r, bit)) return -1;
            nbits_res -= 1;
            LC3_UST(x[idx], bit ? -1 : 1);
        }
    }
    cont = 1;
    return 0;
}

// arithmetic_codec::decode :109-158 up to (not including) the non-lsb residual bits and the noise seed, which
// the caller does lane-parallel.  x and save_lev are zero on entry.
__device__ __forceinline__ int lc3_dec_arith(lc3_reader &r, const lc3_dec_tabs &T, int ne, int fs_ind, int n_ms_10, int *si,
                                             int32_t *x, int32_t *save_lev, int lane) {
    const int nbits = r.len * 8;
    const int num_tns = LC3_U(si[SI_NUM_TNS]), lastnz = LC3_U(si[SI_LASTNZ]), lsb_mode = LC3_U(si[SI_LSB_MODE]);
    lc3_acdec st;
    int sym = 0;
    // ac_dec_init :57-65
    if (!(r.head + 2 < r.len)) return -1;
    st.low = ((uint32_t)LC3_FRAME_BYTE(r, r.head) << 16) | ((uint32_t)LC3_FRAME_BYTE(r, r.head + 1) << 8) |
             (uint32_t)LC3_FRAME_BYTE(r, r.head + 2);
    r.head += 3;
    st.range = 0x00ffffffu;
    // decode_tns_data :304-337
    {
        const int wt = nbits < (n_ms_10 ? 480 : 360);
        for (int k = 0; k < 16; k++) LC3_UST(si[AD_RCI + k], 0);
        for (int f = 0; f < 2; f++) {
            int order = LC3_U(si[SI_ORD0 + f]);
            if (f < num_tns && order > 0) {
                if (lc3_ac_decode(r, st, LC3T_AC_TNS_ORDER_CUMFREQ[wt], LC3T_AC_TNS_ORDER_FREQ[wt], 8, lane, sym)) return -2;
                order = sym + 1;
                for (int k = 0; k < order; k++) {
                    int s2;
                    if (lc3_ac_decode(r, st, LC3T_AC_TNS_COEF_CUMFREQ[k], LC3T_AC_TNS_COEF_FREQ[k], 17, lane, s2)) return -3;
                    LC3_UST(si[AD_RCI + f * 8 + k], s2);
                }
            }
            LC3_UST(si[AD_ORD0 + f], order);
        }
    }
    // decode_spectral_data :211-302
    {
        const int rate_flag = nbits > (160 + fs_ind * 160) ? 512 : 0;
        const int ntup = lastnz / 2;
        int cctx = 0;
        for (int tup = 0; tup < ntup; tup++) {
            int t = cctx + rate_flag + ((tup * 2) > (ne / 2) ? 256 : 0), lev = 0, bit;
            int32_t xk = 0, xk1 = 0;
            sym = 0;
            while (lev < 14) {
                const int pki = lc3_tab_lookup(T, t + (lev < 3 ? lev : 3) * 1024);
                if (lc3_ac_decode_spec(r, st, T, pki, sym)) return -4;
                if (sym < 16) break;
                if (!lsb_mode || lev > 0) {
                    if (lc3_rd_bool(r, bit)) return -5;
                    xk += (int32_t)((uint32_t)bit << lev);
                    if (lc3_rd_bool(r, bit)) return -5;
                    xk1 += (int32_t)((uint32_t)bit << lev);
                }
                lev += 1;
            }
            if (lsb_mode) LC3_UST(save_lev[tup], lev);  // written by TUPLE index, read back by LINE index (:184-195)
            const int a = sym & 3, b = sym >> 2;
            xk += (int32_t)((uint32_t)a << lev);
            xk1 += (int32_t)((uint32_t)b << lev);
            if (xk > 0) {
                if (lc3_rd_bool(r, bit)) return -5;
                if (bit) xk = -xk;
            }
            if (xk1 > 0) {
                if (lc3_rd_bool(r, bit)) return -5;
                if (bit) xk1 = -xk1;
            }
            LC3_UST(x[2 * tup], xk);
            LC3_UST(x[2 * tup + 1], xk1);
            lev = lev < 3 ? lev : 3;
            t = lev <= 1 ? 1 + (a + b) * (lev + 1) : 12 + lev;
            cctx = (cctx & 15) * 16 + t;
        }
    }
    // x[lastnz ..] stays 0 (:131-133).  calc_num_residual_bits :385-405
    {
        const int nbits_side = r.tail - 8;
        const int nbits_ari = (r.head + 1 - 3) * 8 + 25 - lc3_ilog2(st.range);
        if (nbits < nbits_side + nbits_ari) return -6;
        int nres = nbits - nbits_side - nbits_ari, cont;
        LC3_UST(si[AD_TAIL0], r.tail);
        LC3_UST(si[AD_NRES_MAX], nres);
        LC3_UST(si[AD_HEAD], r.head);
        if (lsb_mode) {  // decode_residual_bits :184-206, lsb mode refines the integers in place (serial)
            for (int k = 0; k < lastnz; k += 2) {
                if (LC3_U(save_lev[k]) > 0) {
                    if (lc3_read_res_bit(x, r, k, nres, cont, lane)) return -7;
                    if (!cont) break;
                    if (lc3_read_res_bit(x, r, k + 1, nres, cont, lane)) return -7;
                    if (!cont) break;
                }
            }
        }
    }
    return 0;
}

// D8 helper: mpvq_deenum (decoder/spectral_noise_shaping.rs:155-235)
__device__ __forceinline__ void lc3_mpvq_deenum(int dim_in, int k_val_in, int ls_ind, uint32_t mpvq_ind, int *vec_out) {
    int leading_sign = ls_ind == 0 ? 1 : -1, k_max_local = k_val_in;
    uint32_t ind = mpvq_ind;
    for (int pos = 0; pos < dim_in; pos++) vec_out[pos] = 0;
    for (int pos = 0; pos < dim_in; pos++) {
        const uint32_t *h_row = LC3T_MPVQ_OFFSETS[dim_in - 1 - pos];
        int k_delta;
        if (ind != 0) {
            int k_acc = k_max_local;
            uint32_t ul_diff = 0;
            int wrap = ind < h_row[k_acc];
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
        if (k_delta != 0) {
            vec_out[pos] = leading_sign < 0 ? -k_delta : k_delta;
            leading_sign = (ind & 1u) ? -1 : 1;
            ind >>= 1;
            k_max_local -= k_delta;
        }
    }
}

// ------------------------------------------------------------------------------------------
// D10: IMDCT + window + overlap-add (decoder/modified_dct.rs:76-151); spec -> time samples in spec
// ------------------------------------------------------------------------------------------
__device__ __noinline__ void lc3_dec_imdct(const lc3_cfg &c, lc3_dec_lds &L, int lane) {
    const int nf = c.nf, ne = c.ne, z = c.z, h = nf / 2;
    const uint32_t *w = lc3_window_bits(c);
    float *freq = L.spec;
    float *t = (float *)L.fa;  // t_hat_mdct[2*nf] aliases the FFT buffers (free once the DCT-IV has finished)
    for (int n = ne + lane; n < nf; n += LC3_WAVE) freq[n] = 0.0f;
    LC3_SYNC();
    lc3_dct4_wave(c, lane, freq, L.fa, L.fb);
    // unfold :97-136, gain, reversed window :89-91 -- values are staged in registers because t aliases fa/fb
    const float gain = 1.0f / lc3_sqrtf(2.0f * (float)nf);
    for (int n = lane; n < h; n += LC3_WAVE) {
        float v0 = freq[h + n], v1 = -freq[nf - 1 - n], v2 = -freq[h - 1 - n], v3 = -freq[n];
        v0 *= gain; v1 *= gain; v2 *= gain; v3 *= gain;
        t[n] = v0 * lc3_f(w, 2 * nf - 1 - n);
        t[h + n] = v1 * lc3_f(w, 2 * nf - 1 - (h + n));
        t[nf + n] = v2 * lc3_f(w, 2 * nf - 1 - (nf + n));
        t[3 * h + n] = v3 * lc3_f(w, 2 * nf - 1 - (3 * h + n));
    }
    LC3_SYNC();
    // overlap_add :138-151
    for (int n = lane; n < nf - z; n += LC3_WAVE) {
        freq[n] = L.st.mem_ola[n] + t[z + n];
        L.st.mem_ola[n] = t[nf + z + n];
    }
    for (int n = lane; n < z; n += LC3_WAVE) freq[nf - z + n] = t[nf + n];
    LC3_SYNC();
}

// ------------------------------------------------------------------------------------------
// D11: long-term post-filter synthesis (decoder/long_term_post_filter.rs:142-424)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lc3_wrap_neg(const lc3_cfg &c, int idx) {  // :244-250
    return idx < 0 ? idx + c.num_mem_blocks * c.nf : idx;
}
// compute_filter / compute_filter_mem (:380-415).  The reference's x_hat_mem ring is only ever read at
// start - k, k <= l_num: the current input frame (freq_in) or the last l_num samples of the previous one.
__device__ __forceinline__ float lc3_ltpf_filter(const lc3_cfg &c, const lc3_dec_core &st, const float *freq_in,
                                                 const float *cn, const float *cd, int blk, int n, int pitch_int) {
    float acc = 0.0f;
    for (int k = 0; k <= c.l_num; k++) {
        const int j = n - k;
        acc += cn[k] * (j >= 0 ? freq_in[j] : st.x_tail[c.l_num + j]);
    }
    const int sden = blk + n - pitch_int + c.l_den / 2;
    for (int k = 0; k <= c.l_den; k++) acc -= cd[k] * st.x_hat_ltpf_mem[lc3_wrap_neg(c, sden - k)];
    return acc;
}

__device__ __noinline__ void lc3_dec_ltpf(const lc3_cfg &c, lc3_dec_lds &L, int lane, int is_active, int pitch_index,
                                             int nbits) {
    const int nf = c.nf, blk = L.st.block_start_index, s25 = c.s25;
    const int ncn = c.l_num + 1, ncd = c.l_den + 1;
    float *freq = L.spec;
    float *cnm = L.sm + 64, *cdm = L.sm + 80;  // c_num_mem / c_den_mem
    float *scratch = L.sm;                      // activate_first_2p5ms scratch, l_num + norm <= 130 floats
    int pitch_int = 0, pitch_frac = 0;
    // compute_filter_parameters :164-189 (f64)
    if (is_active) {
        const int pi = pitch_index;
        int p_i;
        double p_fr;
        if (pi >= 440) { p_i = pi - 283; p_fr = 0.0; }
        else if (pi >= 380) { p_i = pi / 2 - 63; p_fr = (double)(2 * pi - 4 * p_i - 252); }
        else { p_i = pi / 4 + 32; p_fr = (double)(pi + 128 - 4 * p_i); }
        const double pitch = (double)p_i + p_fr / 4.0;
        const double fs_ceil = (double)((c.fs + 7999) / 8000);  // (fs / 8000.0).ceil()
        const double pitch_fs = pitch * (8000.0 * fs_ceil / 12800.0);
        const long long p_up = (long long)((pitch_fs * 4.0) + 0.5);
        pitch_int = (int)(p_up / 4);
        pitch_frac = (int)(p_up - 4 * (long long)pitch_int);
    }
    const int prev_active = L.st.ltpf_active_prev;
    int trans;
    if (!is_active && !prev_active) trans = 1;
    else if (is_active && !prev_active) trans = 2;
    else if (!is_active && prev_active) trans = 3;
    else if (pitch_int == L.st.p_int_mem && pitch_frac == L.st.p_fr_mem) trans = 4;
    else trans = 5;
    const int p_int_mem = L.st.p_int_mem;
    LC3_SYNC();
    // compute_filter_coeffs :192-242 (lane 0) and the copy of the frame into the input ring (all lanes)
    if (lane == 0) {
        for (int k = 0; k < ncn; k++) cnm[k] = L.st.c_num[k];
        for (int k = 0; k < ncd; k++) cdm[k] = L.st.c_den[k];
        if (!is_active) {
            for (int k = 0; k < ncn; k++) L.st.c_num[k] = 0.0f;
            for (int k = 0; k < ncd; k++) L.st.c_den[k] = 0.0f;
        } else {
            int t_nbits = nbits;
            if (!c.n_ms_10) t_nbits = (int)((double)nbits * 10.0 / 7.5 + 0.5);
            const int sf = c.fs_ind * 80;
            float gain;
            int gain_ind;
            if (t_nbits < 320 + sf) { gain = 0.4f; gain_ind = 0; }
            else if (t_nbits < 400 + sf) { gain = 0.35f; gain_ind = 1; }
            else if (t_nbits < 480 + sf) { gain = 0.3f; gain_ind = 2; }
            else if (t_nbits < 560 + sf) { gain = 0.25f; gain_ind = 3; }
            else { gain = 0.0f; gain_ind = 0; }
            const uint32_t *tnum, *tden;
            int tn, td;
            switch (c.fs) {
            case 8000: tnum = &LC3T_TAB_LTPF_NUM_8000_BITS[gain_ind][0]; tn = 3; tden = &LC3T_TAB_LTPF_DEN_8000_BITS[pitch_frac][0]; td = 5; break;
            case 16000: tnum = &LC3T_TAB_LTPF_NUM_16000_BITS[gain_ind][0]; tn = 3; tden = &LC3T_TAB_LTPF_DEN_16000_BITS[pitch_frac][0]; td = 5; break;
            case 24000: tnum = &LC3T_TAB_LTPF_NUM_24000_BITS[gain_ind][0]; tn = 5; tden = &LC3T_TAB_LTPF_DEN_24000_BITS[pitch_frac][0]; td = 7; break;
            case 32000: tnum = &LC3T_TAB_LTPF_NUM_32000_BITS[gain_ind][0]; tn = 7; tden = &LC3T_TAB_LTPF_DEN_32000_BITS[pitch_frac][0]; td = 9; break;
            default: tnum = &LC3T_TAB_LTPF_NUM_48000_BITS[gain_ind][0]; tn = 11; tden = &LC3T_TAB_LTPF_DEN_48000_BITS[pitch_frac][0]; td = 13; break;
            }
            for (int k = 0; k < ncn && k < tn; k++) L.st.c_num[k] = 0.85f * gain * lc3_f(tnum, k);  // zip truncation: A9
            for (int k = 0; k < ncd && k < td; k++) L.st.c_den[k] = gain * lc3_f(tden, k);
        }
    }
    LC3_SYNC();
    if (trans == 1) {
        // inactive -> inactive: plain copy (lane-parallel)
        for (int n = lane; n < nf; n += LC3_WAVE) L.st.x_hat_ltpf_mem[blk + n] = freq[n];
    } else if (lane == 0) {
        // the IIR recursion over the output ring is serial in n
        lc3_dec_core &st = L.st;
        if (trans == 2) {
            for (int n = 0; n < nf; n++) {
                st.x_hat_ltpf_mem[blk + n] = freq[n];
                float fo = lc3_ltpf_filter(c, st, freq, st.c_num, st.c_den, blk, n, pitch_int);
                if (n < s25) fo *= (float)n / (float)c.norm;
                st.x_hat_ltpf_mem[blk + n] -= fo;
            }
        } else if (trans == 4) {
            for (int n = 0; n < nf; n++) {
                st.x_hat_ltpf_mem[blk + n] = freq[n];
                st.x_hat_ltpf_mem[blk + n] -= lc3_ltpf_filter(c, st, freq, st.c_num, st.c_den, blk, n, pitch_int);
            }
        } else {
            // deactive_first_2p5ms :417-424
            for (int n = 0; n < s25; n++) {
                st.x_hat_ltpf_mem[blk + n] = freq[n];
                float fo = lc3_ltpf_filter(c, st, freq, cnm, cdm, blk, n, p_int_mem);
                fo *= 1.0f - ((float)n / (float)c.norm);
                st.x_hat_ltpf_mem[blk + n] -= fo;
            }
            if (trans == 3) {
                for (int n = s25; n < nf; n++) st.x_hat_ltpf_mem[blk + n] = freq[n];
            } else {
                // activate_first_2p5ms_from_mem :345-378
                const int l_num = c.l_num;
                // cnm/cdm are no longer needed: scratch may overlap them
                if (blk < l_num) {
                    const int from = c.num_mem_blocks * nf - l_num;
                    for (int i = 0; i < l_num; i++) scratch[i] = st.x_hat_ltpf_mem[from + i];
                    for (int i = 0; i < c.norm; i++) scratch[l_num + i] = st.x_hat_ltpf_mem[i];
                } else {
                    for (int i = 0; i < l_num + c.norm; i++) scratch[i] = st.x_hat_ltpf_mem[blk - l_num + i];
                }
                for (int n = 0; n < s25; n++) {
                    float fo = 0.0f;
                    st.x_hat_ltpf_mem[blk + n] = scratch[n + l_num];
                    for (int k = 0; k <= l_num; k++) fo += st.c_num[k] * scratch[l_num + n - k];
                    const int sden = (blk + n) - pitch_int + c.l_den / 2;
                    for (int k = 0; k <= c.l_den; k++) fo -= st.c_den[k] * st.x_hat_ltpf_mem[lc3_wrap_neg(c, sden - k)];
                    fo *= (float)n / (float)c.norm;
                    st.x_hat_ltpf_mem[blk + n] -= fo;
                }
                for (int n = s25; n < nf; n++) {
                    st.x_hat_ltpf_mem[blk + n] = freq[n];
                    st.x_hat_ltpf_mem[blk + n] -= lc3_ltpf_filter(c, st, freq, st.c_num, st.c_den, blk, n, pitch_int);
                }
            }
        }
    }
    LC3_SYNC();
    if (lane < c.l_num) L.st.x_tail[lane] = freq[nf - c.l_num + lane];  // input history for the next frame
    LC3_SYNC();
    if (trans != 1)
        for (int n = lane; n < nf; n += LC3_WAVE) freq[n] = L.st.x_hat_ltpf_mem[blk + n];
    if (lane == 0) {
        int nb = blk + nf;
        if (nb > (c.num_mem_blocks - 1) * nf) nb = 0;
        L.st.block_start_index = nb;
        L.st.ltpf_active_prev = is_active;
        L.st.p_int_mem = pitch_int;
        L.st.p_fr_mem = pitch_frac;
    }
    LC3_SYNC();
}

// ------------------------------------------------------------------------------------------
// D0/D2/D3: read_frame (decoder/lc3_decoder.rs:165-177).  The range decoder is a serial state machine: lane 0.
// Returns 1 when the frame parsed (side info in L.ism, integer spectrum in xi), 0 -> conceal.
// ------------------------------------------------------------------------------------------
__device__ __noinline__ int lc3_dec_read_frame(const lc3_cfg &c, lc3_dec_lds &L, int lane, const uint8_t *in,
                                               int nbytes_in, int force_plc_in) {
    // arguments of a non-inlined device function arrive in vector registers: re-establish wave-uniformity so that
    // the serial parser below is scalarised
    const int nbytes = LC3_U(nbytes_in), force_plc = LC3_U(force_plc_in);
    const int u_ne = LC3_U(c.ne), u_fs_ind = LC3_U(c.fs_ind), u_n_ms_10 = LC3_U(c.n_ms_10);
    int *si = L.ism;
    int32_t *save_lev = (int32_t *)L.fa;  // 400 ints, free until the IMDCT
    int32_t *xi = (int32_t *)L.fb;        // 400 ints
    const int ne = c.ne;
    for (int i = lane; i < nbytes; i += LC3_WAVE) L.in[i] = in[i];
    for (int i = lane; i < LC3_MAX_NE; i += LC3_WAVE) {
        xi[i] = 0;
        save_lev[i] = 0;
    }
    LC3_SYNC();
    if (LC3_UNIFORM_LEADER(lane)) {
        // uniform-scalar region: on the GPU every lane runs the same state machine with identical values
        lc3_reader r;
        r.buf = L.in;
        r.len = nbytes;
        r.head = 0;
        r.tail = 0;
        lc3_dec_tabs T;
        lc3_dec_tabs_load(T, r, lane);
        int rc = force_plc ? -100 : lc3_dec_side_info(r, u_fs_ind, u_ne, si, lane);
        if (rc == 0) rc = lc3_dec_arith(r, T, u_ne, u_fs_ind, u_n_ms_10, si, xi, save_lev, lane);
        LC3_UST(si[AD_OK], rc == 0);
    }
    LC3_SYNC();
    if (!si[AD_OK]) return 0;
    // Lane-parallel epilogue (integer work, order-free): number of non-zero lines, the residual-bit count and its
    // bounds check, the noise-filling seed  sum |x_k| * k  (:140-145, wrapping) and the zero-frame flag.
    uint32_t *part = (uint32_t *)L.sm;  // [0,64) nnz per lane, [64,128) seed partial sums
    {
        uint32_t nnz = 0, seed = 0;
        for (int k = lane; k < ne; k += LC3_WAVE) {
            const int32_t v = xi[k];
            nnz += v != 0;
            seed += (uint32_t)(v < 0 ? -v : v) * (uint32_t)k;
        }
        part[lane] = nnz;
        part[64 + lane] = seed;
    }
    LC3_SYNC();
    uint32_t nnz = 0, seed = 0;
    _Pragma("nounroll") for (int i = 0; i < LC3_WAVE; i++) {
        nnz += part[i];
        seed += part[64 + i];
    }
    const int lsb_mode = si[SI_LSB_MODE], tail0 = si[AD_TAIL0], nres_max = si[AD_NRES_MAX];
    int n_res = 0, ok = 1;
    if (!lsb_mode) {
        // decode_residual_bits :168-183: one tail bit per non-zero line, at most nres_max.  Every read_tail_bool
        // bound check is monotone in the bit position, so checking the last position covers all of them.
        n_res = (int)nnz < nres_max ? (int)nnz : nres_max;
        if (n_res > 480) ok = 0;  // ResidualBoolDataOverflow (Vec<bool, 480>)
        if (n_res > 0) {
            const int last_byte = (tail0 + n_res - 1) / 8;
            if (nbytes - si[AD_HEAD] - last_byte + 2 < 0) ok = 0;
            if (nbytes - last_byte - 1 < 0) ok = 0;
        }
    }
    LC3_SYNC();
    if (lane == 0) {
        si[AD_NRES] = n_res;
        si[AD_SEED] = (int)(seed & 0xFFFFu);
        si[AD_ZERO] = si[SI_LASTNZ] == 2 && xi[0] == 0 && xi[1] == 0 && si[SI_GG] == 0;
        si[AD_OK] = ok;
    }
    LC3_SYNC();
    return ok;
}

// ------------------------------------------------------------------------------------------
// D4-D8: residual refinement, noise filling, global gain, TNS synthesis, SNS (decoder/lc3_decoder.rs:93-131)
// ------------------------------------------------------------------------------------------
__device__ __noinline__ void lc3_dec_spectrum(const lc3_cfg &c, lc3_dec_lds &L, int lane, int nbits) {
    const int ne = c.ne;
    int *si = L.ism;
    const int32_t *xi = (const int32_t *)L.fb;
    for (int k = lane; k < ne; k += LC3_WAVE) L.spec[k] = (float)xi[k];
    LC3_SYNC();
    // residual_spectrum::decode (decoder/residual_spectrum.rs:13-39): the j-th non-zero line takes residual bit j
    // (tail bit AD_TAIL0 + j, read_tail_bool: bit (pos % 8) of byte len - 1 - pos / 8), for j < AD_NRES.
    // noise_filling::apply_noise_filling (decoder/noise_filling.rs:18-56): the j-th line whose +-width neighbourhood
    // is all zero takes state j + 1 of the LCG s <- (13849 + 31821 s) & 0xFFFF.  Both ranks are prefix counts, and
    // the LCG is affine mod 2^16, so each lane owns 7 consecutive lines and jumps straight to its first state.
    {
        uint32_t *part = (uint32_t *)L.sm;  // [0,64) non-zero counts, [64,128) fill counts
        const int k0 = 7 * lane;
        const int n_res = si[SI_LSB_MODE] ? 0 : si[AD_NRES];
        const int do_fill = !si[AD_ZERO];
        const int bw_stop = c.n_ms_10 ? LC3C_BWSTOP10[si[SI_BW]] : LC3C_BWSTOP75[si[SI_BW]];
        const int nf_start = c.n_ms_10 ? 24 : 18, nf_width = c.n_ms_10 ? 3 : 2;
        const int lim = bw_stop < ne ? bw_stop : ne;
        uint32_t nzmask = 0, fillmask = 0;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int k = k0 + j;
            if (k < ne) {
                if (xi[k] != 0) nzmask |= 1u << j;
                if (do_fill && k >= nf_start && k < lim) {
                    const int from = k - nf_width, to = (bw_stop - 1) < (k + nf_width) ? (bw_stop - 1) : (k + nf_width);
                    int all0 = 1;
                    for (int q = from; q <= to; q++)
                        if (xi[q] != 0) all0 = 0;
                    if (all0) fillmask |= 1u << j;
                }
            }
        }
        part[lane] = (uint32_t)__builtin_popcount(nzmask);
        part[64 + lane] = (uint32_t)__builtin_popcount(fillmask);
        LC3_SYNC();
        int rank_nz = 0, rank_fill = 0;
        _Pragma("nounroll") for (int i = 0; i < lane; i++) {
            rank_nz += (int)part[i];
            rank_fill += (int)part[64 + i];
        }
        // LCG state after rank_fill steps: compose the affine map with itself by binary exponentiation
        uint32_t lcg = (uint32_t)si[AD_SEED];
        {
            uint32_t ra = 1, rcst = 0, ba = 31821u, bc = 13849u;
            for (int n = rank_fill; n > 0; n >>= 1) {
                if (n & 1) {
                    rcst = (ba * rcst + bc) & 0xFFFFu;
                    ra = (ba * ra) & 0xFFFFu;
                }
                bc = (ba * bc + bc) & 0xFFFFu;
                ba = (ba * ba) & 0xFFFFu;
            }
            lcg = (ra * lcg + rcst) & 0xFFFFu;
        }
        const float level = (8.0f - (float)si[SI_NF]) / 16.0f;
        const int tail0 = si[AD_TAIL0], nbytes = nbits / 8;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int k = k0 + j;
            if (k < ne) {
                if (nzmask & (1u << j)) {
                    if (rank_nz < n_res) {
                        const int pos = tail0 + rank_nz;
                        const int bit = (L.in[nbytes - 1 - pos / 8] >> (pos % 8)) & 1;
                        float v = L.spec[k];
                        if (bit) v += v > 0.0f ? 0.3125f : 0.1875f;
                        else v -= v > 0.0f ? 0.1875f : 0.3125f;
                        L.spec[k] = v;
                    }
                    rank_nz++;
                }
                if (fillmask & (1u << j)) {
                    lcg = (13849u + lcg * 31821u) & 0xFFFFu;
                    L.spec[k] = lcg < 0x8000u ? level : -level;
                }
            }
        }
    }
    LC3_SYNC();
    // global_gain::apply_global_gain (decoder/global_gain.rs:15-25)
    {
        const int fs = c.fs_ind + 1, q = nbits / (10 * fs);
        const int gg_off = -(q < 115 ? q : 115) - 105 - (5 * fs);
        const float gg = lc3_pow10f(((float)si[SI_GG] + (float)gg_off) / 28.0f);
        for (int k = lane; k < ne; k += LC3_WAVE) L.spec[k] *= gg;
    }
    LC3_SYNC();
    if (lane == 0) {
        // temporal_noise_shaping::apply_temporal_noise_shaping (decoder/temporal_noise_shaping.rs:24-137):
        // all-pole lattice, recursive in n -> serial; state shared across both filters
        const int bw = si[SI_BW];
        const int nbands = bw < 3 ? 1 : 2;
        const float step = (float)(3.14159265358979323846 / 17.0);  // (PI / 17.0) as f32 :41
        float st[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int f = 0; f < nbands && f < si[SI_NUM_TNS]; f++) {
            const int order = si[AD_ORD0 + f];
            if (order > 0) {
                float rq[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int ri = si[AD_RCI + f * 8 + k];
                    rq[k] = ri != 0 ? lc3_sinf_small(step * (float)(ri - 8)) : 0.0f;
                }
                const int lo = c.n_ms_10 ? LC3C_TNSDEC10[bw][2 * f] : LC3C_TNSDEC75[bw][2 * f];
                const int hi = c.n_ms_10 ? LC3C_TNSDEC10[bw][2 * f + 1] : LC3C_TNSDEC75[bw][2 * f + 1];
                for (int n = lo; n < hi; n++) {
                    float t = L.spec[n];
                    // t = x - rc[order-1]*st[order-1]; then k = order-2 .. 0
#pragma unroll
                    for (int k = 7; k >= 0; k--) {
                        if (k == order - 1) t -= rq[k] * st[k];
                        else if (k < order - 1) {
                            t -= rq[k] * st[k];
                            st[k + 1] = rq[k] * t + st[k];
                        }
                    }
                    L.spec[n] = t;
                    st[0] = t;
                }
            }
        }
        // spectral_noise_shaping::decode (decoder/spectral_noise_shaping.rs:21-151): scale factors
        int *y = (int *)(L.sm + 96), *zv = y + 16;
        float *scf = L.sm, *sfi = L.sm + 16;  // 16 + 64
        const int shape_j = (si[SI_SUB_MSB] << 1) + si[SI_SUB_LSB];
        for (int n = 0; n < 16; n++) { y[n] = 0; zv[n] = 0; }
        if (shape_j == 0) {
            lc3_mpvq_deenum(10, 10, si[SI_LS_A], (uint32_t)si[SI_IDX_A], y);
            lc3_mpvq_deenum(6, 1, si[SI_LS_B], (uint32_t)si[SI_IDX_B], zv);
            for (int n = 0; n < 6; n++) y[10 + n] = zv[n];
        } else if (shape_j == 1) {
            lc3_mpvq_deenum(10, 10, si[SI_LS_A], (uint32_t)si[SI_IDX_A], y);
            for (int n = 10; n < 16; n++) y[n] = 0;
        } else if (shape_j == 2) lc3_mpvq_deenum(16, 8, si[SI_LS_A], (uint32_t)si[SI_IDX_A], y);
        else lc3_mpvq_deenum(16, 6, si[SI_LS_A], (uint32_t)si[SI_IDX_A], y);
        float y_norm = 0.0f;
        for (int n = 0; n < 16; n++) y_norm += (float)y[n] * (float)y[n];
        y_norm = lc3_sqrtf(y_norm);
        float gain;
        const int gi = si[SI_G_IND];
        if (shape_j == 0) gain = lc3_f(LC3T_SNS_VQ_REG_ADJ_GAINS_BITS, gi & 1);
        else if (shape_j == 1) gain = lc3_f(LC3T_SNS_VQ_REG_LF_ADJ_GAINS_BITS, gi & 3);
        else if (shape_j == 2) gain = lc3_f(LC3T_SNS_VQ_NEAR_ADJ_GAINS_BITS, gi & 3);
        else gain = lc3_f(LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS, gi & 7);
        if (y_norm != 0.0f) gain /= y_norm;
        for (int n = 0; n < 16; n++) {
            float factor = 0.0f;
            for (int col = 0; col < 16; col++) factor += (float)y[col] * lc3_f(&LC3T_D_BITS[n][0], col);
            const float st1 = n < 8 ? lc3_f(&LC3T_LFCB_BITS[si[SI_IND_LF]][0], n) : lc3_f(&LC3T_HFCB_BITS[si[SI_IND_HF]][0], n - 8);
            scf[n] = st1 + gain * factor;
        }
        sfi[0] = scf[0];
        sfi[1] = scf[0];
        for (int n = 0; n <= 14; n++) {
            const float fn = scf[n], d = scf[n + 1] - fn;
            sfi[4 * n + 2] = fn + (1.0f / 8.0f * d);
            sfi[4 * n + 3] = fn + (3.0f / 8.0f * d);
            sfi[4 * n + 4] = fn + (5.0f / 8.0f * d);
            sfi[4 * n + 5] = fn + (7.0f / 8.0f * d);
        }
        sfi[62] = scf[15] + 1.0f / 8.0f * (scf[15] - scf[14]);
        sfi[63] = scf[15] + 3.0f / 8.0f * (scf[15] - scf[14]);
        const int n2 = 64 - c.nb;
        if (n2 != 0) {  // :100-111 (decoder form)
            for (int b = 0; b < n2; b++) sfi[b] = (sfi[2 * b] + sfi[2 * b + 1]) / 2.0f;
            for (int b = n2; b < c.nb; b++) sfi[b] = sfi[b + n2];
        }
    }
    LC3_SYNC();
    // band gains via fast_math::exp2_raw and spectral shaping -- one lane per band
    if (lane < c.nb) {
        const uint16_t *ifs = lc3_band_index(c);
        const float g = lc3_exp2_raw(L.sm[16 + lane]);
        for (int k = ifs[lane]; k < ifs[lane + 1]; k++) L.spec[k] *= g;
    }
    LC3_SYNC();
}

// ------------------------------------------------------------------------------------------
// D9: packet loss concealment (decoder/packet_loss_concealment.rs:49-85).  plc_last_good lives in HBM.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lc3_dec_plc_save(const lc3_cfg &c, lc3_dec_lds &L, int lane, lc3_dec_state *g) {
    for (int k = lane; k < c.ne; k += LC3_WAVE) g->plc_last_good[k] = L.spec[k];
    if (lane == 0) {
        L.st.plc_num_lost = 0;
        L.st.plc_alpha = 1.0f;
    }
}
__device__ __noinline__ void lc3_dec_plc_load(const lc3_cfg &c, lc3_dec_lds &L, int lane, const lc3_dec_state *g) {
    // The sign-scrambling LCG seed_k = (16831 + seed_{k-1} * 12821) & 0xFFFF is affine mod 2^16, so lane l can
    // jump straight to its elements k = l, l + 64, ...: seed_{k+64} = A64 * seed_k + C64 (integer, exact).
    const int ne = c.ne;
    const uint32_t seed0 = L.st.plc_seed;
    const int num_lost = L.st.plc_num_lost;
    float alpha = L.st.plc_alpha;
    if (num_lost >= 4) alpha *= num_lost < 8 ? 0.9f : 0.85f;
    LC3_SYNC();
    uint32_t a64 = 1, c64 = 0;
    for (int i = 0; i < 64; i++) {
        c64 = (16831u + c64 * 12821u) & 0xFFFFu;
        a64 = (a64 * 12821u) & 0xFFFFu;
    }
    uint32_t seed = seed0;
    for (int i = 0; i <= lane; i++) seed = (16831u + seed * 12821u) & 0xFFFFu;  // seed_{lane}
    for (int k = lane; k < ne; k += LC3_WAVE) {
        const float lg = g->plc_last_good[k];
        L.spec[k] = seed < 0x8000u ? lg * alpha : lg * -alpha;
        if (k == ne - 1) {
            L.st.plc_seed = seed;  // the reference leaves the seed after ne steps
            L.st.plc_alpha = alpha;
            L.st.plc_num_lost = num_lost + 1;
            L.st.plc_events += 1;
        }
        seed = (a64 * seed + c64) & 0xFFFFu;
    }
    LC3_SYNC();
}

// ------------------------------------------------------------------------------------------
// DecoderChannel::decode (decoder/lc3_decoder.rs:73-154): one frame of one stream on one wave.
// in: nbytes in HBM; pcm_out: nf samples in HBM (4-byte aligned); g: the stream's state blob in HBM.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lc3_decode_frame_wave(const lc3_cfg &c, lc3_dec_lds &L, int lane, const uint8_t *in,
                                                      int nbytes, int16_t *pcm_out, int force_plc, lc3_dec_state *g) {
    const int nf = c.nf, nbits = nbytes * 8;
    LC3_STAMP(L, lane, 16);
    const int ok = lc3_dec_read_frame(c, L, lane, in, nbytes, force_plc);
    LC3_STAMP(L, lane, 17);
    int ltpf_active = 0, pitch_index = 0;
    if (ok) {
        ltpf_active = L.ism[SI_LTPF_ACTIVE];
        pitch_index = L.ism[SI_PITCH_INDEX];
        lc3_dec_spectrum(c, L, lane, nbits);
        lc3_dec_plc_save(c, L, lane, g);
    } else {
        lc3_dec_plc_load(c, L, lane, g);
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 18);
    lc3_dec_imdct(c, L, lane);
    LC3_STAMP(L, lane, 19);
    lc3_dec_ltpf(c, L, lane, ltpf_active, pitch_index, nbits);
    LC3_STAMP(L, lane, 20);
    // output_scaling::scale_and_round (decoder/output_scaling.rs:13-25); two samples per 32-bit store
    {
        uint32_t *o32 = (uint32_t *)pcm_out;
        for (int i = lane; i < nf / 2; i += LC3_WAVE) {
            int32_t v[2];
            for (int j = 0; j < 2; j++) {
                const float x = L.spec[2 * i + j];
                int32_t tmp = x > 0.0f ? lc3_f2i32(x + 0.5f) : lc3_f2i32(x - 0.5f);
                tmp = tmp > 32767 ? 32767 : tmp;
                tmp = tmp < -32768 ? -32768 : tmp;
                v[j] = tmp;
            }
            o32[i] = ((uint32_t)v[0] & 0xffffu) | ((uint32_t)v[1] << 16);
        }
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 21);
}
