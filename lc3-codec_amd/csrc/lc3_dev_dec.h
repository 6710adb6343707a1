// LC3 batched decoder for MI355X -- synthesis stages (one wavefront per stream): concealment, IMDCT, LTPF, output.
// The first half of DecoderChannel::decode (reference decoder/lc3_decoder.rs:73-154: parsing and the spectrum
// reconstruction D1-D8) runs one lane per frame in lc3_dev_dec_parse.h; this file is the stateful second half.
// See lc3_dev_common.h for the execution model and the bit-exactness contract.
#pragma once
#include "lc3_dev_common.h"
#include "lc3_dev_dec_parse.h"

// The decoder's stage functions are real calls, except the IMDCT (everything inlined, the kernel needs 128 VGPRs plus spills
// and runs 1.3x slower).  A call has a price the code is arranged around: every outstanding load and store is waited for
// before a call, at the callee's entry and before its return.  So stores are issued where a long stretch without a call
// follows (the output stage), and the loads of the next frame's plane column -- issued at the top of a frame -- get the inlined
// IMDCT's worth of work before the first call (the LTPF) drains them (IMDCT as a call: synthesis 0.146 ms, inlined 0.134 ms).
#ifndef LC3_DEC_STAGE
#define LC3_DEC_STAGE __noinline__
#define LC3_DEC_STAGE_HOT __forceinline__
#endif

// Persistent per-stream decoder state.  `core` is what a wave keeps resident in LDS while it
// works on the stream; the IMDCT overlap memory (read once and written once per frame, element n by the same lane) and
// plc_last_good (written once per good frame, read only when concealing) stay in HBM.
struct lc3_dec_core {
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
    float mem_ola[304];               // IMDCT overlap memory, nf - z used (decoder/modified_dct.rs:30,149)
    float plc_last_good[LC3_MAX_NE];  // decoder/packet_loss_concealment.rs:7-22
};
#define LC3_DEC_CORE_WORDS ((int)(sizeof(lc3_dec_core) / 4))

// LDS working set of one decoder wave: 8.5 KB; with the workgroup's transform tables four workgroups (16 streams) per CU
struct __attribute__((aligned(16))) lc3_dec_lds {
    lc3_dec_core st;
    float spec[LC3_MAX_NF];        // spec_lines -> FFT work array -> time samples
    lc3_cpx fa[LC3_MAX_NF / 2];    // pre-twiddled FFT input -> DCT-IV output
    float sm[32];                  // LTPF: the previous frame's coefficients
    int ism[64];
    // The LTPF output ring of the state blob is fetched LAZILY (lc3_dec_ring_fill): a launch starts with the ring's LDS copy undefined and
    // only a frame whose filter is on, or was on in the frame before, pulls in what earlier launches left -- minus the blocks this launch has
    // written by then.  A stream whose filter stays off (every stream at the higher bit rates) never reads the 4 KB at all, and the launch
    // stores only the blocks it wrote.
    int ring_loaded;   // the LDS ring holds everything the state blob held at launch start (or the launch started from a fresh state)
    int ring_blk0;     // block_start_index at launch start: the first block this launch writes
    int ring_written;  // blocks written by this launch so far, saturating at num_mem_blocks (a fresh state starts saturated: all is stored)
    int ring_pad_;
    unsigned long long prof_last;  // diagnostic build: time of the previous stage stamp
#ifdef LC3_PROFILE
    unsigned long long prof_acc[32];  // diagnostic build: per-wave stage totals, flushed once per launch
#endif
};
LC3_LDS_DECL(lc3_dec_lds, lc3_dec_wg)
#define LC3_DEC_RING_BYTES ((int)sizeof(((lc3_dec_core *)0)->x_hat_ltpf_mem))
static_assert(offsetof(lc3_dec_core, x_hat_ltpf_mem) == 0 && LC3_DEC_RING_BYTES % 16 == 0, "the ring leads the core blob, in 16-byte units");
static_assert(offsetof(lc3_dec_lds, spec) % 16 == 0 && offsetof(lc3_dec_lds, fa) % 16 == 0 && offsetof(lc3_dec_lds, ism) % 16 == 0,
              "128-bit LDS accesses need aligned buffers");

__device__ __forceinline__ void lc3_dec_state_init(lc3_dec_lds &L, int lane, lc3_dec_state *g, int valid) {
    if (valid) {
        for (int i = lane; i < LC3_MAX_NE; i += LC3_WAVE) g->plc_last_good[i] = 0.0f;
        for (int i = lane; i < 304; i += LC3_WAVE) g->mem_ola[i] = 0.0f;
    }
    int *w = (int *)&L.st;
    for (int i = lane; i < LC3_DEC_CORE_WORDS; i += LC3_WAVE) w[i] = 0;
    LC3_SYNC();
    if (lane == 0) {
        L.st.plc_seed = 24607;  // packet_loss_concealment.rs:31
        L.st.plc_alpha = 1.0f;
        L.ring_loaded = 1;  // zeros: a fresh stream's ring
        L.ring_blk0 = 0;
        L.ring_written = 1 << 20;  // ... and all of it goes to the state blob at the end of the launch
    }
    LC3_SYNC();
}
// everything of the core but the LTPF output ring (see lc3_dec_lds::ring_loaded)
__device__ __forceinline__ void lc3_dec_state_load(lc3_dec_lds &L, int lane, const lc3_dec_state *g) {
    lc3_wave_copy_in16((char *)&L.st + LC3_DEC_RING_BYTES, (const char *)&g->core + LC3_DEC_RING_BYTES,
                       (int)((sizeof(lc3_dec_core) - LC3_DEC_RING_BYTES) / 16), lane);
    LC3_SYNC();
    if (lane == 0) {
        L.ring_loaded = 0;
        L.ring_blk0 = L.st.block_start_index;
        L.ring_written = 0;
    }
    LC3_SYNC();
}
// the same in two steps (see lc3_fft_tables_image_issue): the 12 units are requested, then -- behind whatever else the caller has requested
// in between -- written to LDS
static_assert((sizeof(lc3_dec_core) - LC3_DEC_RING_BYTES) / 16 <= LC3_WAVE, "one unit per lane");
__device__ __forceinline__ lc3_i4 lc3_dec_state_issue(int lane, const lc3_dec_state *g) {
    const int n = (int)((sizeof(lc3_dec_core) - LC3_DEC_RING_BYTES) / 16);
    LC3_HBM_CONST(lc3_i4) src = (LC3_HBM_CONST(lc3_i4))((const char *)&g->core + LC3_DEC_RING_BYTES);
    return src[lane < n ? lane : 0];
}
__device__ __forceinline__ void lc3_dec_state_commit(lc3_dec_lds &L, int lane, const lc3_i4 &v) {
    const int n = (int)((sizeof(lc3_dec_core) - LC3_DEC_RING_BYTES) / 16);
    if (lane < n) ((lc3_i4 *)((char *)&L.st + LC3_DEC_RING_BYTES))[lane] = v;
    LC3_SYNC();
    if (lane == 0) {
        L.ring_loaded = 0;
        L.ring_blk0 = L.st.block_start_index;
        L.ring_written = 0;
    }
    LC3_SYNC();
}
// The ring's blocks this launch has NOT written, from the state blob (whole 16-byte units: a block is nf floats, nf a multiple of 4).
template <class CC>
__device__ __forceinline__ void lc3_dec_ring_fill(const CC &c, lc3_dec_lds &L, int lane, const lc3_dec_state *g) {
    const int nb = c.num_mem_blocks, u_blk = c.nf / 4, b0 = L.ring_blk0 / c.nf, nw = L.ring_written;
    LC3_HBM_CONST(lc3_i4) src = (LC3_HBM_CONST(lc3_i4))g->core.x_hat_ltpf_mem;
    lc3_i4 *dst = (lc3_i4 *)L.st.x_hat_ltpf_mem;
    for (int b = 0; b < nb; b++) {
        int age = b - b0;  // block b is the age-th block this launch writes
        age += age < 0 ? nb : 0;
        if (age < nw) continue;  // newer here than in the state blob
        for (int u = lane; u < u_blk; u += LC3_WAVE) dst[b * u_blk + u] = src[b * u_blk + u];
    }
    LC3_SYNC();
    if (lane == 0) L.ring_loaded = 1;
    LC3_SYNC();
}
// the scalars and filter memories, and the ring blocks this launch wrote
template <class CC>
__device__ __forceinline__ void lc3_dec_state_store(const CC &c, lc3_dec_lds &L, int lane, lc3_dec_state *g) {
    LC3_SYNC();
    lc3_wave_copy_out16((char *)&g->core + LC3_DEC_RING_BYTES, (const char *)&L.st + LC3_DEC_RING_BYTES,
                        (int)((sizeof(lc3_dec_core) - LC3_DEC_RING_BYTES) / 16), lane);
    const int nb = c.num_mem_blocks, u_blk = c.nf / 4, b0 = L.ring_blk0 / c.nf, nw = L.ring_written;
    if (nw >= nb) {
        lc3_wave_copy_out16(g->core.x_hat_ltpf_mem, L.st.x_hat_ltpf_mem, nb * u_blk, lane);
    } else {
        for (int j = 0; j < nw; j++) {
            int b = b0 + j;
            b -= b >= nb ? nb : 0;
            lc3_wave_copy_out16(g->core.x_hat_ltpf_mem + b * c.nf, L.st.x_hat_ltpf_mem + b * c.nf, u_blk, lane);
        }
    }
}

// ------------------------------------------------------------------------------------------
// D10: IMDCT + window + overlap-add (decoder/modified_dct.rs:76-151); spec -> time samples in spec
// ------------------------------------------------------------------------------------------
// The overlap memory (decoder/modified_dct.rs:30,149; nf - z <= 300 samples): element n belongs to lane n % 64 in every frame, so it
// stays in five registers per lane over the frames of a launch and touches the state blob only at the launch's ends.
struct lc3_ola5 { float v[5]; };
template <class CC>
__device__ __forceinline__ lc3_ola5 lc3_dec_ola_load(const CC &c, int lane, const lc3_dec_state *g) {
    LC3_HBM_CONST(float) ola = (LC3_HBM_CONST(float))g->mem_ola;
    lc3_ola5 m;
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int n = lane + LC3_WAVE * r;
        m.v[r] = n < c.nf - c.z ? ola[n] : 0.0f;
    }
    return m;
}
// whole rounds of 64 under a wave-uniform condition, one base address with constant offsets
template <class CC>
__device__ __forceinline__ void lc3_dec_ola_store(const CC &c, int lane, lc3_dec_state *g, int valid, const lc3_ola5 &m) {
    const int nv = LC3_UNIFORM_I32(valid) ? c.nf - c.z : 0;
    int lane_o = lane;  // (an opaque copy: the per-lane address is formed here, after the frame loop, not carried -- spilled -- across it)
    LC3_KEEP_PER_FRAME(lane_o);
    LC3_HBM(float) ob = (LC3_HBM(float))g->mem_ola + lane_o;
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int rem = nv - LC3_WAVE * r;
        if (rem >= LC3_WAVE) ob[LC3_WAVE * r] = m.v[r];
        else if (lane_o < rem) ob[LC3_WAVE * r] = m.v[r];
    }
}
// mo: the previous frame's overlap memory; returns the new one
LC3_CFG_TEMPLATE __device__ LC3_DEC_STAGE_HOT lc3_ola5 lc3_dec_imdct(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_dec_lds) int lane, const lc3_ola5 mo) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_dec_lds, lc3_dec_wg);
    const int nf = c.nf, ne = c.ne, z = c.z, h = nf / 2;
    LC3_HBM_CONST(uint32_t) w = LC3_UNIFORM_PTR(LC3_HBM_CONST(uint32_t), lc3_window_bits(c));  // scalar base: the hoisted per-lane addresses are one register each
    float *freq = L.spec;
    float *u = (float *)L.fa;  // DCT-IV output
    // the window coefficients this lane will need are requested now, as one batch of independent loads, and used after the
    // transform
    float wa[5], wb[5], wc[3];  // nf - z <= 300, z <= 180
    // (the table offsets are formed here, frame by frame, from an opaque copy of the lane number: formed once ahead of the frame loop
    // they were thirteen live registers, some of them spilled, and every reload from scratch drained the wave's memory queue)
    int lane_w = lane;
    LC3_KEEP_PER_FRAME(lane_w);
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int n = lane_w + LC3_WAVE * r, in = n < nf - z;
        wa[r] = in ? lc3_from_bits(w[2 * nf - 1 - (z + n)]) : 0.0f;
        wb[r] = in ? lc3_from_bits(w[2 * nf - 1 - (nf + z + n)]) : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int n = lane_w + LC3_WAVE * r;
        wc[r] = n < z ? lc3_from_bits(w[2 * nf - 1 - (nf + n)]) : 0.0f;
    }
    for (int n = ne + lane; n < nf; n += LC3_WAVE) freq[n] = 0.0f;
    LC3_SYNC();
    lc3_dct4_wave_ab(c, lane, freq, u);
    LC3_STAMP(L, lane, 25);
    // unfold :97-136, gain, reversed window :89-91 and overlap_add :138-151 in one pass: the reference's
    // t_hat_mdct[i] = unfolded(i) * gain * w[2nf - 1 - i] is evaluated where it is consumed (each element is used once)
    const float gain = 1.0f / lc3_sqrtf(2.0f * (float)nf);
#define LC3_UNFOLD(i) ((i) < h ? u[h + (i)] : ((i) < nf ? -u[nf - 1 - ((i) - h)] : ((i) < 3 * h ? -u[h - 1 - ((i) - nf)] : -u[(i) - 3 * h])))
    lc3_ola5 keep;  // the new overlap memory
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int n = lane + LC3_WAVE * r;
        keep.v[r] = 0.0f;
        if (n < nf - z) {
            freq[n] = mo.v[r] + (LC3_UNFOLD(z + n) * gain) * wa[r];
            keep.v[r] = (LC3_UNFOLD(nf + z + n) * gain) * wb[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int n = lane + LC3_WAVE * r;
        if (n < z) freq[nf - z + n] = (LC3_UNFOLD(nf + n) * gain) * wc[r];
    }
#undef LC3_UNFOLD
    LC3_SYNC();
    return keep;
}
// ------------------------------------------------------------------------------------------
// D11: long-term post-filter synthesis (decoder/long_term_post_filter.rs:142-424)
// ------------------------------------------------------------------------------------------
// compute_filter / compute_filter_mem (:380-415) over the samples [n_begin, n_end) of the current frame:
//   x_hat[blk + n] = input(n) - ramp(n) * ( sum_k cn[k] * input(n - k)  -  sum_k cd[k] * x_hat[blk + n - pitch_int + l_den/2 - k] )
// with input(j) = inp[j] for j >= 0 and hist[l_num + j] for the l_num samples before the frame (the reference's
// x_hat_mem ring is only ever read there), ramp: 0 none, 1 fade-in n / norm for n < s25, 2 fade-out 1 - n / norm.
// Samples are computed in blocks of min(64, pitch_int - l_den/2): inside a block no output depends on another.
template <class CC>
__device__ __forceinline__ void lc3_ltpf_run(const CC &c, lc3_dec_lds &L, int lane, int n_begin, int n_end,
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

LC3_CFG_TEMPLATE __device__ LC3_DEC_STAGE void lc3_dec_ltpf(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_dec_lds) int lane, int is_active, int pitch_index,
                                             int nbits, const lc3_dec_state *g) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_dec_lds, lc3_dec_wg);
    const int nf = c.nf, blk = L.st.block_start_index, s25 = c.s25;
    const int ncn = c.l_num + 1, ncd = c.l_den + 1;
    float *freq = L.spec;
    float *cnm = L.sm, *cdm = L.sm + 16;  // c_num_mem / c_den_mem
    float *scratch = (float *)L.fa + 304;  // activate_first_2p5ms scratch, l_num + norm <= 130 floats (fa[0..304) holds the new overlap memory)
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
    const int ring_missing = !L.ring_loaded;
    LC3_SYNC();
    // every transition but "off -> off" reads samples the ring held before this frame
    if (LC3_UNIFORM_I32(trans != 1 && ring_missing)) lc3_dec_ring_fill(c, L, lane, g);
    // compute_filter_coeffs :192-242.  The previous frame's coefficients (c_num_mem / c_den_mem) are only read when that frame's filter
    // was on (transitions 3 and 5); a filter that stays off (transition 1: every frame of a stream at the higher bitrates) finds the
    // zeros it would write already there -- they were written when the filter went off, or at initialisation -- and touches nothing
    if (prev_active) {
        if (lane < ncn) cnm[lane] = L.st.c_num[lane];
        if (lane < ncd) cdm[lane] = L.st.c_den[lane];
    }
    LC3_SYNC();
    if (!is_active) {
        if (prev_active) {
            if (lane < ncn) L.st.c_num[lane] = 0.0f;
            if (lane < ncd) L.st.c_den[lane] = 0.0f;
        }
    } else if (lane == 0) {
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
            // activate_first_2p5ms_from_mem :345-378
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
        if (L.ring_written < c.num_mem_blocks) L.ring_written += 1;
        L.st.ltpf_active_prev = is_active;
        L.st.p_int_mem = pitch_int;
        L.st.p_fr_mem = pitch_frac;
    }
    LC3_SYNC();
}

// ------------------------------------------------------------------------------------------
// D0: pick up one frame as the lane-per-frame stage (lc3_dev_dec_parse.h) left it in its HBM plane column: side
// information words -> L.ism, reconstructed spectrum (f32, D4-D8 applied) -> L.spec.  One 16-byte unit per lane and load,
// both loads in flight together.  Returns 1 when the frame is usable, 0 -> conceal.
// ------------------------------------------------------------------------------------------
// The plane words of a frame in flight (two 16-byte units per lane): the kernel issues the loads of frame t + 1 before
// it works on frame t, so the memory latency hides behind a frame's worth of work.
struct lc3_plane_fetch { lc3_i4 u[2]; };
template <class CC>
__device__ __forceinline__ void lc3_dec_issue_frame(const CC &c, int lane, const int32_t *plane, lc3_plane_fetch &m, int late = 0) {
    LC3_HBM_CONST(lc3_i4) p4 = (LC3_HBM_CONST(lc3_i4))((LC3_HBM_CONST(int32_t))plane + LC3_PLANE_SI);
    const int n4 = (LC3_PLANE_X - LC3_PLANE_SI) / 4 + c.ne / 4;  // <= 112
    // late reconstruction: the residual bit mask (16 words at LC3_PLANE_LEV = unit 112) rides along on four otherwise idle lanes
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = lane + LC3_WAVE * u;
        if (i < n4 || (late && i >= LC3_PLANE_LEV / 4 && i < LC3_PLANE_LEV / 4 + 4)) m.u[u] = p4[i];
    }
}
template <class CC>
__device__ __forceinline__ int lc3_dec_load_frame(const CC &c, lc3_dec_lds &L, int lane, const lc3_plane_fetch &m, int late = 0) {
    const int n_si4 = (LC3_PLANE_X - LC3_PLANE_SI) / 4, n4 = n_si4 + c.ne / 4;
    static_assert(LC3_PLANE_X == 48 && LC3_PLANE_LEV == 448, "ism[48 .. 64) takes the residual bit mask of a late reconstruction");
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = lane + LC3_WAVE * u;
        if (i < n_si4) ((lc3_i4 *)L.ism)[i] = m.u[u];
        else if (i < n4) ((lc3_i4 *)L.spec)[i - n_si4] = m.u[u];
        else if (late && i >= LC3_PLANE_LEV / 4 && i < LC3_PLANE_LEV / 4 + 4) ((lc3_i4 *)L.ism)[n_si4 + i - LC3_PLANE_LEV / 4] = m.u[u];
    }
    LC3_SYNC();
    const int ok = L.ism[AD_OK];
    LC3_SYNC();
    return ok;
}

// TNS synthesis (decoder/temporal_noise_shaping.rs:60-137) of lines [lo, hi) in place with a filter of ORDER stages:
//   t = x - rc[ORDER-1] * st[ORDER-1];  q = ORDER-2 .. 0: t -= rc[q] * st[q]; st[q+1] = rc[q] * t + st[q];  x = st[0] = t
// four lines per LDS round trip (the ranges are multiples of four lines long and start at a multiple of four, except 7.5 ms
// frames whose start 9 is handled line by line)
template <int ORDER>
__device__ __forceinline__ void lc3_dec_tns_lattice(float *x, int lo, int hi, const float (&rq)[8], float (&st)[8]) {
    int k = lo;
    for (; k < hi && (k & 3) != 0; k++) {
        float t = x[k];
#pragma unroll
        for (int q = ORDER - 1; q >= 0; q--) {
            t = t - rq[q] * st[q];
            if (q < ORDER - 1) st[q + 1] = rq[q] * t + st[q];
        }
        x[k] = t;
        st[0] = t;
    }
    for (; k + 4 <= hi; k += 4) {
        const lc3_f4 in = *(const lc3_f4 *)(x + k);
        float v[4] = {in.x, in.y, in.z, in.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float t = v[u];
#pragma unroll
            for (int q = ORDER - 1; q >= 0; q--) {
                t = t - rq[q] * st[q];
                if (q < ORDER - 1) st[q + 1] = rq[q] * t + st[q];
            }
            v[u] = t;
            st[0] = t;
        }
        lc3_f4 o;
        o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
        *(lc3_f4 *)(x + k) = o;
    }
    for (; k < hi; k++) {
        float t = x[k];
#pragma unroll
        for (int q = ORDER - 1; q >= 0; q--) {
            t = t - rq[q] * st[q];
            if (q < ORDER - 1) st[q + 1] = rq[q] * t + st[q];
        }
        x[k] = t;
        st[0] = t;
    }
}

// ------------------------------------------------------------------------------------------
// D4-D8 for launches of a few frames, with the 64 lanes of the stream's wave instead of one lane of the parse kernel
// (lc3_reconstruct_frame, lc3_dev_dec_parse.h, is the same arithmetic line by line: int -> f32, residual bit or noise value,
// global gain, TNS synthesis, band gain).  Lane l owns lines 8l .. 8l+7; what the reference carries from line to line becomes
// a prefix count: the rank of a non-zero line among the non-zero lines picks its residual bit, the rank of a noise-filled line
// among the filled ones picks its LCG state (the LCG is affine mod 2^16: f^n in log n steps).  Only the TNS lattice stays a
// walk over the lines (lane 0, frames with an active filter).
// In: L.ism = side information + residual bit mask (ism[48..)), L.spec = the parsed integers.  Out: L.spec = shaped spectrum.
// ------------------------------------------------------------------------------------------
LC3_CFG_TEMPLATE __device__ LC3_DEC_STAGE void lc3_dec_reconstruct_wave(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_dec_lds) int lane, int nbytes, float *dbg = nullptr) {
    LC3_CFG_BIND;
    LC3_LDS_BIND(lc3_dec_lds, lc3_dec_wg);
    const int ne = c.ne, nbits = nbytes * 8;
    const int lastnz = L.ism[SI_LASTNZ], gg_ind = L.ism[SI_GG], n_res = L.ism[AD_NRES], bw = L.ism[SI_BW];
    const uint32_t *resw = (const uint32_t *)(L.ism + LC3_PLANE_X);
    float *sc = (float *)L.fa;  // scratch until the IMDCT: scale factors [0, 16), band gains [16, 80)
    lc3_recon_ctx r;
    r.scf = sc;
    r.sstride = 1;
    r.mpvq = &LC3T_MPVQ_OFFSETS[0][0];
    r.ifs = lc3_band_index(c);
    // spectral_noise_shaping::decode (:21-73): the pulse vector on every lane (wave-uniform inputs), one scale factor per lane
    {
        int y[16];
#pragma unroll
        for (int n = 0; n < 16; n++) y[n] = 0;
        const int shape_j = (L.ism[SI_SUB_MSB] << 1) + L.ism[SI_SUB_LSB];
        const int ls_a = L.ism[SI_LS_A];
        const uint32_t idx_a = (uint32_t)L.ism[SI_IDX_A];
        if (shape_j == 0) {
            lc3_r_deenum(r, 10, 10, ls_a, idx_a, y, 0);
            lc3_r_deenum(r, 6, 1, L.ism[SI_LS_B], (uint32_t)L.ism[SI_IDX_B], y, 10);
        } else if (shape_j == 1) lc3_r_deenum(r, 10, 10, ls_a, idx_a, y, 0);
        else if (shape_j == 2) lc3_r_deenum(r, 16, 8, ls_a, idx_a, y, 0);
        else lc3_r_deenum(r, 16, 6, ls_a, idx_a, y, 0);
        float y_norm = 0.0f;
#pragma unroll
        for (int n = 0; n < 16; n++) y_norm += (float)y[n] * (float)y[n];
        y_norm = lc3_sqrtf(y_norm);
        float gain;
        const int gi = L.ism[SI_G_IND];
        if (shape_j == 0) gain = lc3_f(LC3T_SNS_VQ_REG_ADJ_GAINS_BITS, gi & 1);
        else if (shape_j == 1) gain = lc3_f(LC3T_SNS_VQ_REG_LF_ADJ_GAINS_BITS, gi & 3);
        else if (shape_j == 2) gain = lc3_f(LC3T_SNS_VQ_NEAR_ADJ_GAINS_BITS, gi & 3);
        else gain = lc3_f(LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS, gi & 7);
        if (y_norm != 0.0f) gain /= y_norm;
        const int ind_lf = L.ism[SI_IND_LF], ind_hf = L.ism[SI_IND_HF];
        const int n = lane & 15;
        float factor = 0.0f;
#pragma unroll
        for (int col = 0; col < 16; col++) factor += (float)y[col] * lc3_f(&LC3T_D_BITS[n][0], col);
        const float st1 = n < 8 ? lc3_f(&LC3T_LFCB_BITS[ind_lf][0], n) : lc3_f(&LC3T_HFCB_BITS[ind_hf][0], n - 8);
        if (lane < 16) sc[n] = st1 + gain * factor;
    }
    LC3_SYNC();
    if (lane < c.nb) sc[16 + lane] = lc3_r_band_gain(r, lane, c.nb);
    // global gain :15-25
    float gg;
    {
        const int fs = c.fs_ind + 1, q = nbits / (10 * fs);
        const int gg_off = -(q < 115 ? q : 115) - 105 - (5 * fs);
        gg = LC3_POW10_GG(gg_ind + gg_off);
    }
    // the lane's eight lines (integers; words at and beyond lastnz are stale)
    const int k0 = 8 * lane;
    int32_t xi[8];
    {
        const lc3_i4 a = ((const lc3_i4 *)L.spec)[2 * lane], b = ((const lc3_i4 *)L.spec)[2 * lane + 1];
        int32_t w[8];
        __builtin_memcpy(w, &a, 16);
        __builtin_memcpy(w + 4, &b, 16);
#pragma unroll
        for (int j = 0; j < 8; j++) xi[j] = k0 + j < lastnz ? w[j] : 0;
    }
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
    const float level = (8.0f - (float)L.ism[SI_NF]) / 16.0f;
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
    // the LCG state before the lane's first filled line: f^R(seed), f(x) = 13849 + 31821 x mod 2^16, R = filled lines below
    {
        uint32_t R = lc3_wave_exscan_u32((uint32_t)__builtin_popcount(fillmask), lane);
        uint32_t pa = 31821u, pc = 13849u, A = 1u, C = 0u;  // f^(2^i) = pa x + pc; accumulated map A x + C
        for (int i = 0; i < 9; i++) {  // R < 512
            if (R & 1u) {
                A = (pa * A) & 0xFFFFu;
                C = (pa * C + pc) & 0xFFFFu;
            }
            pc = (pa * pc + pc) & 0xFFFFu;
            pa = (pa * pa) & 0xFFFFu;
            R >>= 1;
        }
        lcg = (A * lcg + C) & 0xFFFFu;
    }
    int rank_nz = (int)lc3_wave_exscan_u32((uint32_t)__builtin_popcount(nzmask), lane);
    float v8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float v = (float)xi[j];
        {   // residual_spectrum::decode: the j-th non-zero line takes residual bit j while j < n_res
            const int nz = (int)((nzmask >> j) & 1u), take = nz && rank_nz < n_res;
            const int rr = take ? rank_nz : 0;
            const int bit = (int)((resw[rr >> 5] >> (rr & 31)) & 1u);
            const float up = v > 0.0f ? 0.3125f : 0.1875f, dn = v > 0.0f ? 0.1875f : 0.3125f;
            const float v_adj = bit ? v + up : v - dn;
            v = take ? v_adj : v;
            rank_nz += nz;
        }
        {
            const int fill = (int)((fillmask >> j) & 1u);
            const uint32_t lcg_n = (13849u + LC3_MUL24(lcg, 31821u)) & 0xFFFFu;
            lcg = fill ? lcg_n : lcg;
            v = fill ? (lcg_n < 0x8000u ? level : -level) : v;
        }
        v8[j] = v * gg;
    }
    if (dbg && k0 < ne)
        for (int j = 0; j < 8; j++) {
            dbg[LC3_DBG_INT + k0 + j] = (float)xi[j];
            dbg[LC3_DBG_GAIN + k0 + j] = v8[j];
        }
    // TNS :24-137
    const int nbands = bw < 3 ? 1 : 2, num_tns = L.ism[SI_NUM_TNS];
    const int ord0 = (0 < nbands && 0 < num_tns) ? L.ism[AD_ORD0] : 0;
    const int ord1 = (1 < nbands && 1 < num_tns) ? L.ism[AD_ORD0 + 1] : 0;
    LC3_SYNC();  // every lane holds its lines: L.spec may be overwritten
    if (ord0 > 0 || ord1 > 0) {  // wave-uniform
        if (k0 < ne) {
            lc3_f4 o0, o1;
            o0.x = v8[0]; o0.y = v8[1]; o0.z = v8[2]; o0.w = v8[3];
            o1.x = v8[4]; o1.y = v8[5]; o1.z = v8[6]; o1.w = v8[7];
            ((lc3_f4 *)L.spec)[2 * lane] = o0;
            ((lc3_f4 *)L.spec)[2 * lane + 1] = o1;
        }
        LC3_SYNC();
        if (lane == 0) {  // the all-pole lattice, its state shared across the two filters
            float st[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            for (int f = 0; f < nbands; f++) {
                const int order = f == 0 ? ord0 : ord1;
                if (order == 0) continue;
                const int lo = c.n_ms_10 ? LC3C_TNSDEC10[bw][2 * f] : LC3C_TNSDEC75[bw][2 * f];
                const int hi = c.n_ms_10 ? LC3C_TNSDEC10[bw][2 * f + 1] : LC3C_TNSDEC75[bw][2 * f + 1];
                float rq[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int ri = L.ism[AD_RCI + 8 * f + q];
                    rq[q] = ri != 0 ? LC3_TNS_SIN_DEC(ri) : 0.0f;  // sin(step * (ri - 8)); SURVEY A12
                }
                // the order is the same for every line of a filter: one straight-line body per order (no per-stage selects)
                switch (order) {
                case 1: lc3_dec_tns_lattice<1>(L.spec, lo, hi, rq, st); break;
                case 2: lc3_dec_tns_lattice<2>(L.spec, lo, hi, rq, st); break;
                case 3: lc3_dec_tns_lattice<3>(L.spec, lo, hi, rq, st); break;
                case 4: lc3_dec_tns_lattice<4>(L.spec, lo, hi, rq, st); break;
                case 5: lc3_dec_tns_lattice<5>(L.spec, lo, hi, rq, st); break;
                case 6: lc3_dec_tns_lattice<6>(L.spec, lo, hi, rq, st); break;
                case 7: lc3_dec_tns_lattice<7>(L.spec, lo, hi, rq, st); break;
                default: lc3_dec_tns_lattice<8>(L.spec, lo, hi, rq, st); break;
                }
            }
        }
        LC3_SYNC();
        if (k0 < ne) {
            const lc3_f4 i0 = ((const lc3_f4 *)L.spec)[2 * lane], i1 = ((const lc3_f4 *)L.spec)[2 * lane + 1];
            v8[0] = i0.x; v8[1] = i0.y; v8[2] = i0.z; v8[3] = i0.w;
            v8[4] = i1.x; v8[5] = i1.y; v8[6] = i1.z; v8[7] = i1.w;
        }
        LC3_SYNC();
    }
    if (dbg && k0 < ne)
        for (int j = 0; j < 8; j++) dbg[LC3_DBG_TNS + k0 + j] = v8[j];
    // SNS :113-151: the band gain of each line
    if (k0 < ne) {
        const uint8_t *lb = c.line_band + k0;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int b = k0 + j < c.nf ? (int)lb[j] : 255;
            o[j] = v8[j] * sc[16 + (b < 64 ? b : 0)];
        }
        lc3_f4 o0, o1;
        o0.x = o[0]; o0.y = o[1]; o0.z = o[2]; o0.w = o[3];
        o1.x = o[4]; o1.y = o[5]; o1.z = o[6]; o1.w = o[7];
        ((lc3_f4 *)L.spec)[2 * lane] = o0;
        ((lc3_f4 *)L.spec)[2 * lane + 1] = o1;
    }
    LC3_SYNC();
}

// ------------------------------------------------------------------------------------------
// D9: packet loss concealment (decoder/packet_loss_concealment.rs:49-85).  plc_last_good lives in HBM.
// ------------------------------------------------------------------------------------------
// a good frame: the concealment counters restart (save :49-61); the spectrum itself is copied to the state blob only where
// nothing else holds it -- on the last frame of a launch (lc3_decode_frame_wave)
template <class CC>
__device__ __forceinline__ void lc3_dec_plc_save(const CC &c, lc3_dec_lds &L, int lane, lc3_dec_state *g, int to_blob) {
    if (to_blob) {  // ne <= 400: seven rounds, all LDS reads ahead of the stores
        float v[7];
        int lane_s = lane;  // (an opaque copy: formed ahead of the frame loop, the last round's clamped address was a register spilled to scratch)
        LC3_KEEP_PER_FRAME(lane_s);
#pragma unroll
        for (int r = 0; r < 7; r++) v[r] = L.spec[lane_s + LC3_WAVE * r < LC3_MAX_NE ? lane_s + LC3_WAVE * r : 0];
        LC3_HBM(float) dst = (LC3_HBM(float))g->plc_last_good + lane;
#pragma unroll
        for (int r = 0; r < 7; r++)
            if (lane + LC3_WAVE * r < c.ne) dst[LC3_WAVE * r] = v[r];
    }
    if (lane == 0) {
        L.st.plc_num_lost = 0;
        L.st.plc_alpha = 1.0f;
    }
}
// last_good: the last good frame's spectrum (ne f32): the state blob's copy, or the plane column of an earlier frame of this launch
LC3_CFG_TEMPLATE __device__ LC3_DEC_STAGE void lc3_dec_plc_load(LC3_CFG_PARAM, LC3_LDS_PARAM(lc3_dec_lds) int lane, const float *last_good) {
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
        const float lg = ((LC3_HBM_CONST(float))last_good)[k];
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

// The interleaved output path as a call: inlined, its eight per-lane addresses were computed ahead of the frame loop and spilled
// (the planar path, which every full batch takes, paid for them with scratch traffic and a full memory wait at each reload).
__device__ LC3_DEC_STAGE void lc3_dec_store_strided(int16_t *pcm_out, int stride, int lane, int nv, uint32_t w0, uint32_t w1, uint32_t w2,
                                                    uint32_t w3) {
    LC3_HBM(uint16_t) o16 = (LC3_HBM(uint16_t))pcm_out;
    const uint32_t ow[4] = {w0, w1, w2, w3};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = lane + LC3_WAVE * r;
        if (i < nv) {
            o16[(size_t)(2 * i) * (size_t)stride] = (uint16_t)(ow[r] & 0xffffu);
            o16[(size_t)(2 * i + 1) * (size_t)stride] = (uint16_t)(ow[r] >> 16);
        }
    }
}

// ------------------------------------------------------------------------------------------
// DecoderChannel::decode (decoder/lc3_decoder.rs:73-154): one frame of one stream on one wave.
// in: nbytes in HBM; pcm_out: nf samples in HBM (4-byte aligned); plane/stride: the frame's parsed column
// (lc3_dev_dec_parse.h); g: the stream's state blob in HBM.
// ------------------------------------------------------------------------------------------
// plc_src / save_good: see lc3_decode_stream_wave.  Returns 1 for a good frame, 0 for a concealed one.
LC3_CFG_TEMPLATE __device__ __forceinline__ int lc3_decode_frame_wave(LC3_CFG_PARAM, lc3_dec_lds &L, int lane, int nbytes,
                                                      int16_t *pcm_out, const lc3_plane_fetch &fetched, lc3_dec_state *g,
                                                      int valid, int stride, const float *plc_src, int save_good, lc3_ola5 &ola, int late = 0,
                                                      float *dbg = nullptr, int dbg_flags = 0) {
    LC3_CFG_BIND;
    const int nf = c.nf, nbits = nbytes * 8;
    LC3_STAMP(L, lane, 16);
    const int ok = lc3_dec_load_frame(c, L, lane, fetched, late);
    if (ok && late) lc3_dec_reconstruct_wave(LC3_CFG_PASS, LC3_LDS_PASS lane, nbytes, (dbg_flags & LC3_DBG_DUMP) ? dbg : nullptr);  // launches of a few frames: D4-D8 here
    if (dbg_flags & (LC3_DBG_SPEC_IN | LC3_DBG_DUMP)) {  // diagnostic entry points: the spectrum comes from / goes to the caller's buffer
        for (int k = lane; k < c.ne; k += LC3_WAVE) {
            if (dbg_flags & LC3_DBG_SPEC_IN) L.spec[k] = dbg[LC3_DBG_SPEC + k];
            else if (ok) dbg[LC3_DBG_SPEC + k] = L.spec[k];
        }
        LC3_SYNC();
    }
    LC3_STAMP(L, lane, 17);
    int ltpf_active = 0, pitch_index = 0;
    if (ok) {
        ltpf_active = L.ism[SI_LTPF_ACTIVE];
        pitch_index = L.ism[SI_PITCH_INDEX];
        lc3_dec_plc_save(c, L, lane, g, valid && save_good);
    } else {
        lc3_dec_plc_load(LC3_CFG_PASS, LC3_LDS_PASS lane, plc_src);
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 18);
    ola = lc3_dec_imdct(LC3_CFG_PASS, LC3_LDS_PASS lane, ola);
    LC3_STAMP(L, lane, 19);
    if (dbg_flags & (LC3_DBG_TIME_IN | LC3_DBG_DUMP)) {
        for (int n = lane; n < nf; n += LC3_WAVE) {
            if (dbg_flags & LC3_DBG_TIME_IN) L.spec[n] = dbg[LC3_DBG_IMDCT + n];  // (the filter is fed directly)
            else dbg[LC3_DBG_IMDCT + n] = L.spec[n];
        }
        LC3_SYNC();
    }
    lc3_dec_ltpf(LC3_CFG_PASS, LC3_LDS_PASS lane, ltpf_active, pitch_index, nbits, g);
    if (dbg_flags & LC3_DBG_DUMP) {
        for (int n = lane; n < nf; n += LC3_WAVE) dbg[LC3_DBG_LTPF + n] = L.spec[n];
    }
    LC3_STAMP(L, lane, 20);
    // output_scaling::scale_and_round (decoder/output_scaling.rs:13-25); two samples per 32-bit store
    {
        LC3_HBM(uint32_t) o32 = (LC3_HBM(uint32_t))pcm_out;
        uint32_t ow[4];  // nf / 2 <= 240 words: all four computed, then stored together
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = lane + LC3_WAVE * r;
            int32_t v[2] = {0, 0};
            if (i < nf / 2) {
                for (int j = 0; j < 2; j++) {
                    const float x = L.spec[2 * i + j];
                    int32_t tmp = lc3_f2i32(x + (x > 0.0f ? 0.5f : -0.5f));  // x - 0.5 == x + (-0.5)
                    tmp = tmp > 32767 ? 32767 : tmp;
                    tmp = tmp < -32768 ? -32768 : tmp;
                    v[j] = tmp;
                }
            }
            ow[r] = ((uint32_t)v[0] & 0xffffu) | ((uint32_t)v[1] << 16);
        }
        const int nv = LC3_UNIFORM_I32(valid) ? nf / 2 : 0;
        if (stride == 1) {
            LC3_HBM(uint32_t) ob = o32 + lane;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rem = nv - LC3_WAVE * r;
                if (rem >= LC3_WAVE) ob[LC3_WAVE * r] = ow[r];
                else if (lane < rem) ob[LC3_WAVE * r] = ow[r];
            }
        } else {  // interleaved PCM out (examples/decode.rs:93-112 interleaves on the host): sample n at pcm_out[n * stride]
            lc3_dec_store_strided(pcm_out, stride, lane, nv, ow[0], ow[1], ow[2], ow[3]);
        }
    }
    LC3_SYNC();
    LC3_STAMP(L, lane, 21);
    return ok;
}

// All frames of one stream of a launch.  Packet-loss concealment needs the spectrum of the last good frame
// (decoder/packet_loss_concealment.rs:49-85).  Within a launch that spectrum already lies in HBM -- the plane column the
// lane-per-frame stage wrote for that frame -- so a good frame copies nothing; only the last frame of the launch saves its
// spectrum to the state blob (or, when that frame is lost, the plane column of the launch's last good frame is copied there).
// Frame t's samples go to pcm0 + t * frame_step, `stride` elements apart.
struct lc3_no_prologue { __device__ __forceinline__ void operator()() const {} };
// prologue: called once, after the first frame's column and the overlap memory have been REQUESTED and before anything is used -- the
// kernel body writes its tables and the stream state (requested before the call) to LDS there, so that all of a launch's first loads are
// on their way from HBM together
LC3_CFG_TEMPLATE_AND(class PROLOGUE = lc3_no_prologue)
__device__ __forceinline__ void lc3_decode_stream_wave(LC3_CFG_PARAM, lc3_dec_lds &L, int lane, int nbytes, const int32_t *planes,
                                                       size_t fbase, int n_frames, lc3_dec_state *g, int valid, int16_t *pcm0,
                                                       size_t frame_step, int stride, int late = 0, float *dbg = nullptr, int dbg_flags = 0,
                                                       PROLOGUE prologue = PROLOGUE(), int fresh_in_prologue = 0) {
    LC3_CFG_BIND;
    const auto &c0 = c;
    lc3_plane_fetch cur, nxt;
    // late: the plane columns hold integers (lc3_dec_reconstruct_wave): the last good SPECTRUM of the stream is then kept in the
    // state blob by every good frame
    if (n_frames > 0) lc3_dec_issue_frame(c0, lane, LC3_PLANE_COL(planes, fbase, LC3_PLANE_WORDS), cur, late);
    int t_good = -1, last_ok = 1;
    lc3_ola5 ola = lc3_dec_ola_load(c0, lane, g);
    prologue();
    if (LC3_UNIFORM_I32(fresh_in_prologue)) {  // the prologue has just reset the stream (lc3_dec_state_init): what was requested above is the OLD overlap memory
#pragma unroll
        for (int r = 0; r < 5; r++) ola.v[r] = 0.0f;
    }
    for (int t = 0; t < n_frames; t++) {
        const size_t f = fbase + (size_t)t;
        if (t + 1 < n_frames) lc3_dec_issue_frame(c0, lane, LC3_PLANE_COL(planes, f + 1, LC3_PLANE_WORDS), nxt, late);
        int16_t *out = pcm0 + (size_t)t * frame_step;
        const float *plc_src = (t_good >= 0 && !late) ? (const float *)(LC3_PLANE_COL(planes, fbase + (size_t)t_good, LC3_PLANE_WORDS) + LC3_PLANE_X * LC3_PLANE_STRIDE)
                                           : (const float *)g->plc_last_good;
        last_ok = lc3_decode_frame_wave(LC3_CFG_PASS, L, lane, nbytes, out, cur, g, valid, stride, plc_src, late || t == n_frames - 1, ola, late,
                                        dbg, valid ? dbg_flags : 0);
        if (last_ok) t_good = t;
        cur = nxt;
    }
    lc3_dec_ola_store(c0, lane, g, valid, ola);
    if (valid && !last_ok && t_good >= 0 && !late) {  // the launch ended in a lost frame: its last good spectrum moves to the state blob
        LC3_HBM_CONST(float) src = (LC3_HBM_CONST(float))(LC3_PLANE_COL(planes, fbase + (size_t)t_good, LC3_PLANE_WORDS) + LC3_PLANE_X * LC3_PLANE_STRIDE);
        for (int k = lane; k < c0.ne; k += LC3_WAVE) g->plc_last_good[k] = src[k];
    }
}
