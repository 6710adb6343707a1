// LC3 batched decoder for MI355X -- device-side stages (one wavefront per stream).
// Mirrors DecoderChannel::decode (reference decoder/lc3_decoder.rs:73-154) stage by stage.
// See lc3_dev_common.h for the execution model and the bit-exactness contract.
#pragma once
#include "lc3_dev_common.h"
#include "lc3_dev_dec_parse.h"

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
    int pad[2];                    // keeps the blob a multiple of 16 bytes (128-bit copies, aligned LDS arrays)
};
static_assert(sizeof(lc3_dec_core) % 16 == 0, "decoder state blob must stay a multiple of 16 bytes");
struct lc3_dec_state {
    lc3_dec_core core;
    float plc_last_good[LC3_MAX_NE];  // decoder/packet_loss_concealment.rs:7-22
};
#define LC3_DEC_CORE_WORDS ((int)(sizeof(lc3_dec_core) / 4))

// LDS working set of one decoder wave (~13 KB -> 12 waves per CU)
struct __attribute__((aligned(16))) lc3_dec_lds {
    lc3_dec_core st;
    float spec[LC3_MAX_NF];        // spec_lines, then freq_samples
    lc3_cpx fa[LC3_MAX_NF / 2];    // FFT in   | t_hat_mdct[0 .. nf)
    lc3_cpx fb[LC3_MAX_NF / 2];    // integer spectrum xi     | FFT work | t_hat_mdct[nf .. 2nf)  (contiguous with fa)
    uint8_t in[LC3_MAX_NE];        // frame bytes (residual bits are read from them)
    float sm[192];
    int ism[64];
    unsigned long long prof_last;  // diagnostic build: time of the previous stage stamp
#ifdef LC3_PROFILE
    unsigned long long prof_acc[32];  // diagnostic build: per-wave stage totals, flushed once per launch
#endif
};
LC3_LDS_DECL(lc3_dec_lds, lc3_dec_wg)

__device__ __forceinline__ void lc3_dec_state_init(lc3_dec_lds &L, int lane, lc3_dec_state *g, int valid) {
    if (valid)
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
    lc3_wave_copy_in16(&L.st, &g->core, (int)(sizeof(lc3_dec_core) / 16), lane);
    LC3_SYNC();
}
__device__ __forceinline__ void lc3_dec_state_store(lc3_dec_lds &L, int lane, lc3_dec_state *g) {
    LC3_SYNC();
    lc3_wave_copy_out16(&g->core, &L.st, (int)(sizeof(lc3_dec_core) / 16), lane);
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
__device__ __noinline__ void lc3_dec_imdct(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_dec_lds) int lane) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_dec_lds, lc3_dec_wg);
    const int nf = c.nf, ne = c.ne, z = c.z, h = nf / 2;
    const uint32_t *w = lc3_window_bits(c);
    float *freq = L.spec;
    float *t = (float *)L.fa;  // t_hat_mdct[2*nf] aliases the FFT buffers (free once the DCT-IV has finished)
    for (int n = ne + lane; n < nf; n += LC3_WAVE) freq[n] = 0.0f;
    LC3_SYNC();
    lc3_dct4_wave(c, lane, freq, L.fa, L.fb);
    LC3_STAMP(L, lane, 25);
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
// compute_filter / compute_filter_mem (:380-415) over the samples [n_begin, n_end) of the current frame:
//   x_hat[blk + n] = input(n) - ramp(n) * ( sum_k cn[k] * input(n - k)  -  sum_k cd[k] * x_hat[blk + n - pitch_int + l_den/2 - k] )
// with input(j) = inp[j] for j >= 0 and hist[l_num + j] for the l_num samples before the frame (the reference's
// x_hat_mem ring is only ever read there), ramp: 0 none, 1 fade-in n / norm for n < s25, 2 fade-out 1 - n / norm.
// Samples are computed in blocks of min(64, pitch_int - l_den/2): inside a block no output depends on another.
__device__ __forceinline__ void lc3_ltpf_run(const lc3_cfg &c, lc3_dec_lds &L, int lane, int n_begin, int n_end,
                                             const float *inp, const float *hist, const float *cn, const float *cd,
                                             int pitch_int, int ramp) {
    const int blk = L.st.block_start_index, l_num = c.l_num, l_den = c.l_den, ring = c.num_mem_blocks * c.nf;
    int bsz = pitch_int - l_den / 2;
    bsz = bsz < 1 ? 1 : (bsz > LC3_WAVE ? LC3_WAVE : bsz);
    float *xh = L.st.x_hat_ltpf_mem;
    for (int n0 = n_begin; n0 < n_end; n0 += bsz) {
        const int n = n0 + lane;
        if (lane < bsz && n < n_end) {
            float acc = 0.0f;
            for (int k = 0; k <= l_num; k++) {
                const int j = n - k;
                acc += cn[k] * (j >= 0 ? inp[j] : hist[l_num + j]);
            }
            const int sden = blk + n - pitch_int + l_den / 2;
            for (int k = 0; k <= l_den; k++) {
                int idx = sden - k;
                idx = idx < 0 ? idx + ring : idx;  // :244-250
                acc -= cd[k] * xh[idx];
            }
            if (ramp == 1) {
                if (n < c.s25) acc *= (float)n / (float)c.norm;
            } else if (ramp == 2) {
                acc *= 1.0f - ((float)n / (float)c.norm);
            }
            xh[blk + n] = inp[n] - acc;
        }
        LC3_SYNC();
    }
}

__device__ __noinline__ void lc3_dec_ltpf(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_dec_lds) int lane, int is_active, int pitch_index,
                                             int nbits) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_dec_lds, lc3_dec_wg);
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
    // The IIR recursion feeds back x_hat delayed by at least pitch_int - l_den/2 samples, so that many consecutive
    // outputs are independent: they are computed one per lane (lc3_ltpf_run), each with the reference's tap order.
    if (trans == 1) {
        // inactive -> inactive: plain copy
        for (int n = lane; n < nf; n += LC3_WAVE) L.st.x_hat_ltpf_mem[blk + n] = freq[n];
    } else if (trans == 2) {
        lc3_ltpf_run(c, L, lane, 0, nf, freq, L.st.x_tail, L.st.c_num, L.st.c_den, pitch_int, 1);
    } else if (trans == 4) {
        lc3_ltpf_run(c, L, lane, 0, nf, freq, L.st.x_tail, L.st.c_num, L.st.c_den, pitch_int, 0);
    } else {
        // deactive_first_2p5ms :417-424
        lc3_ltpf_run(c, L, lane, 0, s25, freq, L.st.x_tail, cnm, cdm, p_int_mem, 2);
        if (trans == 3) {
            for (int n = s25 + lane; n < nf; n += LC3_WAVE) L.st.x_hat_ltpf_mem[blk + n] = freq[n];
        } else {
            // activate_first_2p5ms_from_mem :345-378; cnm/cdm are no longer needed: scratch may overlap them
            const int l_num = c.l_num;
            LC3_SYNC();
            for (int i = lane; i < l_num + c.norm; i += LC3_WAVE) {
                int src;
                if (blk < l_num) src = i < l_num ? c.num_mem_blocks * nf - l_num + i : i - l_num;
                else src = blk - l_num + i;
                scratch[i] = L.st.x_hat_ltpf_mem[src];
            }
            LC3_SYNC();
            lc3_ltpf_run(c, L, lane, 0, s25, scratch + l_num, scratch, L.st.c_num, L.st.c_den, pitch_int, 1);
            lc3_ltpf_run(c, L, lane, s25, nf, freq, L.st.x_tail, L.st.c_num, L.st.c_den, pitch_int, 0);
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
// D0: pick up one parsed frame (lc3_dev_dec_parse.h) from its HBM plane column: side information -> L.ism,
// integer spectrum -> xi, frame bytes -> L.in (the residual bits are still read from them).  Then the order-free
// integer epilogue of arithmetic_codec::decode, lane-parallel: residual-bit count and its bounds check, the
// noise-filling seed  sum |x_k| * k  (:140-145, wrapping) and the zero-frame flag.
// Returns 1 when the frame parsed, 0 -> conceal.
// ------------------------------------------------------------------------------------------
__device__ __noinline__ int lc3_dec_load_frame(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_dec_lds) int lane, const uint8_t *in, int nbytes,
                                               const int32_t *plane, int stride) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_dec_lds, lc3_dec_wg);
    int *si = L.ism;
    int32_t *xi = (int32_t *)L.fb;  // 400 ints
    const int ne = c.ne;
    LC3_HBM_CONST(uint8_t) gin = (LC3_HBM_CONST(uint8_t))in;
    LC3_HBM_CONST(int32_t) gplane = (LC3_HBM_CONST(int32_t))plane;
    (void)stride;  // frame-major planes: the column is contiguous
    {
        // frame bytes (<= 400: seven per lane) and the plane column (side information + integer spectrum, contiguous
        // 16-byte units: two per lane), every load issued before the first LDS store
        uint8_t bv[7];
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const int i = lane + LC3_WAVE * u;
            bv[u] = i < nbytes ? gin[i] : (uint8_t)0;
        }
        LC3_HBM_CONST(lc3_i4) p4 = (LC3_HBM_CONST(lc3_i4))(gplane + LC3_PLANE_SI);
        const int n_si4 = (LC3_PLANE_X - LC3_PLANE_SI) / 4, n4 = n_si4 + ne / 4;
        lc3_i4 pv[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int i = lane + LC3_WAVE * u;
            if (i < n4) pv[u] = p4[i];
        }
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const int i = lane + LC3_WAVE * u;
            if (i < nbytes) L.in[i] = bv[u];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int i = lane + LC3_WAVE * u;
            if (i < n_si4) ((lc3_i4 *)si)[i] = pv[u];
            else if (i < n4) ((lc3_i4 *)xi)[i - n_si4] = pv[u];
        }
    }
    LC3_SYNC();
    if (!si[AD_OK]) return 0;
    uint32_t nnz = 0, seed = 0;
    for (int k = lane; k < ne; k += LC3_WAVE) {
        const int32_t v = xi[k];
        nnz += v != 0;
        seed += (uint32_t)(v < 0 ? -v : v) * (uint32_t)k;
    }
    nnz = lc3_wave_sum_u32(nnz, lane);
    seed = lc3_wave_sum_u32(seed, lane);
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
// Every wave of the workgroup calls this (it contains a serial phase); `ok` = the frame parsed (L.ism[AD_OK]), a
// stream whose frame did not parse skips the work and conceals afterwards.
__device__ __noinline__ void lc3_dec_spectrum(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_dec_lds) int lane, int nbits, int ok) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_dec_lds, lc3_dec_wg);
    const int ne = c.ne;
    int *si = L.ism;
    const int32_t *xi = (const int32_t *)L.fb;
    if (ok) {
    for (int k = lane; k < ne; k += LC3_WAVE) L.spec[k] = (float)xi[k];
    LC3_SYNC();
    // residual_spectrum::decode (decoder/residual_spectrum.rs:13-39): the j-th non-zero line takes residual bit j
    // (tail bit AD_TAIL0 + j, read_tail_bool: bit (pos % 8) of byte len - 1 - pos / 8), for j < AD_NRES.
    // noise_filling::apply_noise_filling (decoder/noise_filling.rs:18-56): the j-th line whose +-width neighbourhood
    // is all zero takes state j + 1 of the LCG s <- (13849 + 31821 s) & 0xFFFF.  Both ranks are prefix counts, and
    // the LCG is affine mod 2^16, so each lane owns 7 consecutive lines and jumps straight to its first state.
    {
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
        int rank_nz = (int)lc3_wave_exscan_u32((uint32_t)__builtin_popcount(nzmask), lane);
        const int rank_fill = (int)lc3_wave_exscan_u32((uint32_t)__builtin_popcount(fillmask), lane);
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
    LC3_STAMP(L, lane, 22);
    // global_gain::apply_global_gain (decoder/global_gain.rs:15-25)
    {
        const int fs = c.fs_ind + 1, q = nbits / (10 * fs);
        const int gg_off = -(q < 115 ? q : 115) - 105 - (5 * fs);
        const float gg = lc3_pow10f(((float)si[SI_GG] + (float)gg_off) / 28.0f);
        for (int k = lane; k < ne; k += LC3_WAVE) L.spec[k] *= gg;
    }
    }  // if (ok)
    // Sixteen lanes.  Lane 0: the TNS synthesis lattice (decoder/temporal_noise_shaping.rs:24-137, recursive in n,
    // state shared across both filters) and the pulse de-enumeration of the SNS shape; then the sixteen together: scale
    // factors (decoder/spectral_noise_shaping.rs:21-151).  Kept on the stream's own wave: the lattice takes anything
    // from nothing to ~100 k cycles depending on the frame's TNS orders (measured: gathering the workgroup's four
    // streams on one wave made every stream wait for the slowest, profiles/r01_v6_notes.txt).
    LC3_LOCAL_BEGIN(lane, LC3_WAVE)  // the whole wave enters (it syncs inside); lanes 16.. only keep step
    int *y = (int *)(L.sm + 96);          // [16] pulses
    float *scf = L.sm, *sfi = L.sm + 16;  // 16 + 64
    if (sub == 0 && ok) {
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
        int *zv = y + 16;
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
        L.sm[128] = gain;
    }
    LC3_SYNC();
    if (ok && sub < 16) {
        // scale factor sub = codebook entry + gain * (y . D[sub][:]), 16-term sum in order
        const int n = sub;
        const float gain = L.sm[128];
        float factor = 0.0f;
        for (int col = 0; col < 16; col++) factor += (float)y[col] * lc3_f(&LC3T_D_BITS[n][0], col);
        const float st1 = n < 8 ? lc3_f(&LC3T_LFCB_BITS[si[SI_IND_LF]][0], n) : lc3_f(&LC3T_HFCB_BITS[si[SI_IND_HF]][0], n - 8);
        scf[n] = st1 + gain * factor;
    }
    LC3_SYNC();
    if (ok && sub < 16) {
        // interpolation :75-98 -- four of the 64 values per lane
        const int n = sub;
        if (n == 0) {
            sfi[0] = scf[0];
            sfi[1] = scf[0];
            sfi[62] = scf[15] + 1.0f / 8.0f * (scf[15] - scf[14]);
            sfi[63] = scf[15] + 3.0f / 8.0f * (scf[15] - scf[14]);
        }
        if (n <= 14) {
            const float fn = scf[n], d = scf[n + 1] - fn;
            sfi[4 * n + 2] = fn + (1.0f / 8.0f * d);
            sfi[4 * n + 3] = fn + (3.0f / 8.0f * d);
            sfi[4 * n + 4] = fn + (5.0f / 8.0f * d);
            sfi[4 * n + 5] = fn + (7.0f / 8.0f * d);
        }
    }
    LC3_SYNC();
    if (ok && sub == 0) {
        const int n2 = 64 - c.nb;
        if (n2 != 0) {  // :100-111 (decoder form)
            for (int b = 0; b < n2; b++) sfi[b] = (sfi[2 * b] + sfi[2 * b + 1]) / 2.0f;
            for (int b = n2; b < c.nb; b++) sfi[b] = sfi[b + n2];
        }
    }
    LC3_LOCAL_END
    if (ok) {
    LC3_STAMP(L, lane, 24);
    // band gains via fast_math::exp2_raw and spectral shaping -- one lane per band
    if (lane < c.nb) {
        const uint16_t *ifs = lc3_band_index(c);
        const float g = lc3_exp2_raw(L.sm[16 + lane]);
        for (int k = ifs[lane]; k < ifs[lane + 1]; k++) L.spec[k] *= g;
    }
    }  // if (ok)
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
__device__ __noinline__ void lc3_dec_plc_load(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_dec_lds) int lane, const lc3_dec_state *g) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_dec_lds, lc3_dec_wg);
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
// in: nbytes in HBM; pcm_out: nf samples in HBM (4-byte aligned); plane/stride: the frame's parsed column
// (lc3_dev_dec_parse.h); g: the stream's state blob in HBM.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lc3_decode_frame_wave(LC3_CFG_PARAM, lc3_dec_lds &L, int lane, const uint8_t *in,
                                                      int nbytes, int16_t *pcm_out, const int32_t *plane, int stride,
                                                      lc3_dec_state *g, int valid) {
    LC3_CFG_BIND;
    const int nf = c.nf, nbits = nbytes * 8;
    LC3_STAMP(L, lane, 16);
    const int ok = lc3_dec_load_frame(LC3_CFG_PASS, LC3_LDS_PASS lane, in, nbytes, plane, stride);
    LC3_STAMP(L, lane, 17);
    int ltpf_active = 0, pitch_index = 0;
    lc3_dec_spectrum(LC3_CFG_PASS, LC3_LDS_PASS lane, nbits, ok);
    if (ok) {
        ltpf_active = L.ism[SI_LTPF_ACTIVE];
        pitch_index = L.ism[SI_PITCH_INDEX];
        if (valid) lc3_dec_plc_save(c, L, lane, g);
    } else {
        lc3_dec_plc_load(LC3_CFG_PASS, LC3_LDS_PASS lane, g);
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 18);
    lc3_dec_imdct(LC3_CFG_PASS, LC3_LDS_PASS lane);
    LC3_STAMP(L, lane, 19);
    lc3_dec_ltpf(LC3_CFG_PASS, LC3_LDS_PASS lane, ltpf_active, pitch_index, nbits);
    LC3_STAMP(L, lane, 20);
    // output_scaling::scale_and_round (decoder/output_scaling.rs:13-25); two samples per 32-bit store
    {
        uint32_t *o32 = (uint32_t *)pcm_out;
        for (int i = lane; valid && i < nf / 2; i += LC3_WAVE) {
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
