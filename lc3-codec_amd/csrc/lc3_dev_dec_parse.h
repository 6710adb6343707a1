// LC3 batched decoder for MI355X -- frame parser, ONE LANE PER FRAME.
//
// Parsing an LC3 frame (side information + range decoder, reference decoder/lc3_decoder.rs:165-177,
// decoder/side_info_reader.rs:29-200, decoder/arithmetic_codec.rs:57-405) is a serial integer state machine with
// no floating point and no dependence on earlier frames.  One wavefront per frame leaves 63 lanes idle and is
// bound by the scalar-issue rate (measured: ~41 k SALU instructions per frame, profiles/r01_pmc_*.csv), so this
// stage is run with the opposite mapping: every lane parses its own frame and a wave advances 64 frames per
// instruction.  The context/cumulative-frequency tables and the frame bytes sit in LDS; the results go to HBM
// "planes" (one column of LC3_PLANE_WORDS words per frame).  The same lane then rebuilds the frame's spectrum from
// the integers (lc3_reconstruct_frame: residual refinement, noise filling, global gain, TNS synthesis, SNS -- all of
// them recursions or prefix-dependent walks over the lines, D4-D8 of the reference) and leaves it as f32 in the plane;
// the wave-per-stream synthesis kernel (lc3_dev_dec.h) picks it up for concealment, IMDCT and LTPF.
//
// Plane words of one frame (int32):  [0, 48) side info (enum below)   [48, 448) spectrum x[k]: integers while
//                                    parsing, f32 bit patterns once reconstructed
//                                    [448, 648) save_lev per tuple (lsb_mode only)
#pragma once
#include "lc3_dev_common.h"

#define LC3_PLANE_SI 0
#define LC3_PLANE_X 48
#define LC3_PLANE_LEV 448
#define LC3_PLANE_WORDS 648

// decoded side information
enum {
    SI_BW = 0, SI_LASTNZ, SI_LSB_MODE, SI_GG, SI_NUM_TNS, SI_ORD0, SI_ORD1, SI_IND_LF, SI_IND_HF, SI_LS_A, SI_LS_B,
    SI_IDX_A, SI_IDX_B, SI_SUB_LSB, SI_SUB_MSB, SI_G_IND, SI_PITCH_PRESENT, SI_LTPF_ACTIVE, SI_PITCH_INDEX, SI_NF,
    AD_ORD0, AD_ORD1, AD_NRES, AD_SEED, AD_ZERO, AD_OK, AD_RCI /* 16 entries */, AD_TAIL0 = AD_RCI + 16, AD_NRES_MAX,
    AD_HEAD, AD_Y /* 3 words: the SNS pulse vector, 16 x 5 bits (lc3_parse_pulses) */, SI_WORDS = AD_Y + 3
};

// Stage dumps of the decoder's diagnostic entry points (lc3gpu_decode_frame_debug, include/lc3gpu.h): float offsets
enum {
    LC3_DBG_INT = 0,        // [400] the integers after the range decoder (values, as floats)
    LC3_DBG_SPEC = 400,     // [400] the spectrum after reconstruction (input of the inverse transform)
    LC3_DBG_IMDCT = 800,    // [480] time samples after IMDCT, window and overlap-add
    LC3_DBG_LTPF = 1280,    // [480] time samples after the long-term post-filter
    LC3_DBG_GAIN = 1760,    // [400] after residual bits, noise filling and global gain (input of the TNS filter)
    LC3_DBG_TNS = 2160,     // [400] after the TNS filter (input of the SNS band gains)
    LC3_DBG_FLOATS = 2560
};
enum { LC3_DBG_DUMP = 1, LC3_DBG_SPEC_IN = 2, LC3_DBG_TIME_IN = 4 };  // what the synthesis kernel does with the buffer

struct lc3_parse_ctx {
    float *dbg;              // stage dumps (diagnostic entry point, one frame) or null
    const uint8_t *bytes;    // this frame's bytes
    int len;
    const uint8_t *lookup;   // AC_SPEC_LOOKUP[4096]
    const uint32_t *cf;      // [64][LC3_DCF_ROW_WORDS] cum | freq << 16 of the spectral model (lc3_dcf_word), 16-byte aligned
    const uint32_t *tns;     // [2][8] TNS order models then [8][17] TNS coefficient models, packed the same way
    int32_t *plane;          // this frame's plane column: word w at plane[w * stride]
    int stride;
    int head, tail;          // BufferReader cursors (decoder/buffer_reader.rs:11-15)
    // the spectral loop keeps the bytes its next reads will need in registers (lc3_p_prime): bytes[head], bytes[head + 1]
    // and the byte under the tail cursor plus the one after it, each fetched one step before its use
    uint32_t hb0, hb1, tcur, tnext;
    uint32_t nnz, seed;      // running count of non-zero lines and sum |x_k| * k (noise-filling seed :140-145, wrapping)
#ifdef LC3_PROFILE
    unsigned long long plast, pt[8];  // diagnostic build: section stamps
#endif
};
#ifndef LC3_PSTAMP
#define LC3_PSTAMP(c, id)
#endif

static_assert(LC3_PLANE_STRIDE == 1, "the parser's 64 / 128-bit plane accesses need frame-major planes");
typedef int32_t lc3_i2 __attribute__((vector_size(8)));  // two plane words (a line pair), 8-byte aligned
__device__ __forceinline__ void lc3_px_set(lc3_parse_ctx &c, int word, int32_t v) { c.plane[word * c.stride] = v; }
__device__ __forceinline__ int32_t lc3_px_get(const lc3_parse_ctx &c, int word) { return c.plane[word * c.stride]; }

// read_tail_usize (decoder/buffer_reader.rs:63-98)
__device__ __forceinline__ int lc3_p_tail(lc3_parse_ctx &c, int num_bits, uint32_t &val) {
    const int byte_index = c.tail >> 3, bit_index = c.tail & 7;
    const int bits_left = 8 - bit_index;
    const int add_bytes = (num_bits > bits_left && num_bits < 8) ? 2 : 1;
    const int num_bytes = (num_bits >> 3) + add_bytes;
    if (c.len - c.head - byte_index - num_bytes < 0) return -1;
    const int from = c.len - byte_index - num_bytes;
    uint32_t value = 0;
    if (num_bytes <= 4)
        for (int i = 0; i < num_bytes; i++) value = (value << 8) | (uint32_t)c.bytes[from + i];
    const int shift_by = 32 - num_bits - bit_index;
    value <<= shift_by;
    value >>= shift_by + bit_index;
    c.tail += num_bits;
    val = value;
    return 0;
}
// read_tail_bool (:100-116)
__device__ __forceinline__ int lc3_p_bool(lc3_parse_ctx &c, int &bit) {
    const int byte_index = c.tail >> 3, bit_index = c.tail & 7;
    if (c.len - c.head - byte_index + 2 < 0) return -1;
    const int from = c.len - byte_index - 1;
    if (from < 0) return -1;  // the reference would panic on this index; treated as a read error (-> PLC)
    bit = ((uint32_t)c.bytes[from] >> bit_index) & 1u;
    c.tail += 1;
    return 0;
}

#define LC3_PT(nb, dst) do { if (lc3_p_tail(c, (nb), (dst))) return -1; } while (0)
#define LC3_PB(dst) do { if (lc3_p_bool(c, (dst))) return -1; } while (0)

// side_info_reader::read (decoder/side_info_reader.rs:29-200).  WR = 0: parse only, nothing goes to the plane (the producer wave of a
// producer / consumer pair needs the flags, its consumer writes the words)
// the SNS vector quantiser's side information (read_sns_vq :131-200), as the scale-factor computation takes it
struct lc3_sns_side {
    int ind_lf, ind_hf, sub_msb, sub_lsb, g_ind, ls_a, ls_b;
    uint32_t idx_a, idx_b;
};
template <int WR = 1>
__device__ __forceinline__ int lc3_parse_side_info(lc3_parse_ctx &c, int fs_ind, int ne, int &lastnz_out, int &lsb_mode_out,
                                                   int &num_tns_out, int ord[2], lc3_sns_side *sns = nullptr) {
#define lc3_px_set(c_, w_, v_) do { if (WR) (lc3_px_set)(c_, w_, v_); } while (0)
    uint32_t v;
    int b, p_bw = 0, lastnz_bits = 0;
    const int nbits_bw = LC3C_NBITS_BW[fs_ind];
    if (nbits_bw > 0) {
        LC3_PT(nbits_bw, v);
        if (fs_ind < (int)v) return -2;  // BandwidthIdxOutOfRange
        p_bw = (int)v;
    }
    while ((1 << lastnz_bits) < ne / 2) lastnz_bits++;
    LC3_PT(lastnz_bits, v);
    const int lastnz = (int)((v + 1) << 1);
    lc3_px_set(c, SI_LASTNZ, lastnz);
    if (lastnz > ne) return -3;
    LC3_PB(b);
    lc3_px_set(c, SI_LSB_MODE, b);
    lsb_mode_out = b;
    LC3_PT(8, v);
    lc3_px_set(c, SI_GG, (int)v);
    const int num_tns = p_bw < 3 ? 1 : 2;
    lc3_px_set(c, SI_NUM_TNS, num_tns);
    ord[0] = 0;
    ord[1] = 0;
    for (int f = 0; f < num_tns; f++) {
        LC3_PB(b);
        ord[f] = b;
    }
    lc3_px_set(c, SI_ORD0, ord[0]);
    lc3_px_set(c, SI_ORD1, ord[1]);
    LC3_PB(b);
    const int pitch_present = b;
    lc3_px_set(c, SI_PITCH_PRESENT, b);
    // read_sns_vq :131-200
    LC3_PT(5, v);
    lc3_px_set(c, SI_IND_LF, (int)v);
    const int lc3_sns_ind_lf_ = (int)v;
    LC3_PT(5, v);
    lc3_px_set(c, SI_IND_HF, (int)v);
    const int lc3_sns_ind_hf_ = (int)v;
    LC3_PB(b);
    const int submode_msb = b;
    if (submode_msb == 0) LC3_PT(1, v);
    else LC3_PT(2, v);
    int g_ind = (int)v;
    LC3_PB(b);
    lc3_px_set(c, SI_LS_A, b);
    const int lc3_sns_ls_a_ = b;
    int submode_lsb = 0, ls_indb = 0;
    uint32_t idx_a, idx_b = 0;
    if (submode_msb == 0) {
        uint32_t tmp;
        LC3_PT(25, tmp);
        if (tmp >= 33460056u) return -4;  // PlcTriggerSns1OutOfRange
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
        LC3_PT(24, tmp);
        if (tmp >= 16708096u) return -5;  // PlcTriggerSns2OutOfRange
        if (tmp >= 15158272u) {
            tmp -= 15158272u;
            submode_lsb = 1;
            g_ind = (g_ind << 1) + (int)(tmp & 1u);
            idx_a = tmp >> 1;
        } else idx_a = tmp;
    }
    lc3_px_set(c, SI_LS_B, ls_indb);
    lc3_px_set(c, SI_IDX_A, (int)idx_a);
    lc3_px_set(c, SI_IDX_B, (int)idx_b);
    lc3_px_set(c, SI_SUB_LSB, submode_lsb);
    lc3_px_set(c, SI_SUB_MSB, submode_msb);
    lc3_px_set(c, SI_G_IND, g_ind);
    if (sns) {
        sns->ind_lf = lc3_sns_ind_lf_;
        sns->ind_hf = lc3_sns_ind_hf_;
        sns->sub_msb = submode_msb;
        sns->sub_lsb = submode_lsb;
        sns->g_ind = g_ind;
        sns->ls_a = lc3_sns_ls_a_;
        sns->ls_b = ls_indb;
        sns->idx_a = idx_a;
        sns->idx_b = idx_b;
    }
    // read_long_term_post_filter_info :106-129
    int ltpf_active = 0, pitch_index = 0;
    if (pitch_present) {
        LC3_PB(b);
        ltpf_active = b;
        LC3_PT(9, v);
        pitch_index = (int)v;
    }
    lc3_px_set(c, SI_LTPF_ACTIVE, ltpf_active);
    lc3_px_set(c, SI_PITCH_INDEX, pitch_index);
    LC3_PT(3, v);
    lc3_px_set(c, SI_NF, (int)v);
    lc3_px_set(c, SI_BW, p_bw);
    lastnz_out = lastnz;
    num_tns_out = num_tns;
    return 0;
#undef lc3_px_set
}

// ---- range decoder and bit reader, written with selects.  Every lane is another frame, so a branch on frame data diverges
// and costs more scalar bookkeeping than the operations it skips.  These never return early: a failed bound check sets
// the sticky `err` (the frame is concealed, exactly as when the reference returns Err at that point), reads fall back to
// a safe index, and the caller tests `err` once after the loop.
// Register copies of the bytes the next reads need.  A lane of this kernel is latency-bound: a byte fetched from LDS at
// its point of use costs a full LDS round trip per bit / per renormalisation.  Here every read consumes a byte that was
// requested one step earlier and requests the one the following step may need.
__device__ __forceinline__ uint32_t lc3_p_head_byte(const lc3_parse_ctx &c, int at) {
    return (uint32_t)c.bytes[at < c.len ? at : c.len - 1];
}
__device__ __forceinline__ uint32_t lc3_p_tail_byte(const lc3_parse_ctx &c, int byte_index) {
    const int from = c.len - byte_index - 1;
    return (uint32_t)c.bytes[from < 0 ? 0 : from];
}
__device__ __forceinline__ void lc3_p_prime(lc3_parse_ctx &c) {
    c.hb0 = lc3_p_head_byte(c, c.head);
    c.hb1 = lc3_p_head_byte(c, c.head + 1);
    c.tcur = lc3_p_tail_byte(c, c.tail >> 3);
    c.tnext = lc3_p_tail_byte(c, (c.tail >> 3) + 1);
}
// read_tail_bool (:100-116) when `want`; returns the bit (0 when not wanted) and advances the cursor by `want`
__device__ __forceinline__ int lc3_p_bool_sel(lc3_parse_ctx &c, int want, int &err) {
    const int byte_index = c.tail >> 3, bit_index = c.tail & 7;
    const int from = c.len - byte_index - 1;
    const int bad = (c.len - c.head - byte_index + 2 < 0) | (from < 0);
    err |= want & bad;
    const int bit = (int)((c.tcur >> bit_index) & 1u);
    c.tail += want;
    const int wrap = want & (bit_index == 7);
    c.tcur = wrap ? c.tnext : c.tcur;
    c.tnext = lc3_p_tail_byte(c, (c.tail >> 3) + 1);  // consumed at the next wrap, not here
    return want ? bit : 0;
}
// two consecutive read_tail_bool (:100-116), each when its `want`: the cursor and its byte window advance once.  The reference's bound
// checks (a bit's byte must lie inside the frame and not more than two bytes below the head cursor, :100-108) are monotone: the cursors
// only move on, so the last bit that was read stands for all of them.  `slack` keeps len + 2 - head - (byte of the last bit read) as of
// the last call that read a bit; the caller tests it, and the byte of the final cursor against the frame, once after its loop.
__device__ __forceinline__ void lc3_p_bool2_sel(lc3_parse_ctx &c, int w0, int w1, int &slack, int &b0, int &b1) {
    const int n = w0 + w1, bit_index = c.tail & 7;
    const int byte_last = (c.tail + n - 1) >> 3;
    slack = n > 0 ? c.len + 2 - c.head - byte_last : slack;
    const uint32_t win = (c.tcur | (c.tnext << 8)) >> bit_index;
    b0 = w0 ? (int)(win & 1u) : 0;
    b1 = w1 ? (int)((win >> w0) & 1u) : 0;
    const int wrap = bit_index + n >= 8;
    c.tail += n;
    c.tcur = wrap ? c.tnext : c.tcur;
    c.tnext = lc3_p_tail_byte(c, (c.tail >> 3) + 1);  // consumed at the next wrap, not here
}
// the two renormalisation steps of ac_decode (:88-95) from the head bytes held in registers; REFILL = 0 leaves the request for the
// next two head bytes to the caller (lc3_p_head_refill)
__device__ __forceinline__ void lc3_p_head_refill(lc3_parse_ctx &c) {
    c.hb0 = lc3_p_head_byte(c, c.head);
    c.hb1 = lc3_p_head_byte(c, c.head + 1);
}
// CHECK = 0: the caller tests the head cursor against the frame once after its loop (read_head_byte :42-50 fails when the cursor is at or
// beyond the end; the cursor moves by one per byte read, so "some byte was read beyond the end" is "the cursor ended beyond the end + 0")
template <int REFILL = 1, int CHECK = 1>
__device__ __forceinline__ void lc3_p_ac_renorm_sel(lc3_parse_ctx &c, lc3_acdec &st, int &err) {
    const int need0 = st.range < 0x10000u;
    if (CHECK) err |= need0 & (c.head >= c.len);  // read_head_byte :42-50
    st.low = need0 ? ((st.low << 8) & 0x00ffffffu) + c.hb0 : st.low;
    st.range = need0 ? st.range << 8 : st.range;
    c.head += need0;
    if (st.range < 0x10000u) {  // a second byte only after a symbol of probability below 2^-8: usually no lane of the wave
        if (CHECK) err |= c.head >= c.len;
        st.low = ((st.low << 8) & 0x00ffffffu) + c.hb1;
        st.range <<= 8;
        c.head += 1;
    }
    if (REFILL) lc3_p_head_refill(c);
}
// ac_decode (decoder/arithmetic_codec.rs:67-97) over a packed (cum | freq << 16) model row with symbols 0..HI: the
// reference scans from the top for the largest j with low >= tmp * cum[j] (:81-84; cum is non-decreasing, cum[0] = 0),
// here a STEPS-step binary search.  After range = tmp * freq the range is at least 64 (tmp >= 64 because range >= 2^16
// on entry, freq >= 1), so the reference's renormalisation loop runs at most twice.
template <int HI, int STEPS>
__device__ __forceinline__ int lc3_p_ac_decode_sel(lc3_parse_ctx &c, lc3_acdec &st, const uint32_t *row, int &err) {
    const uint32_t tmp = st.range >> 10, limit = tmp << 10;
    err |= st.low >= limit;
    int lo = 0, hi = HI;
#pragma unroll
    for (int it = 0; it < STEPS; it++) {
        const int mid = (lo + hi + 1) >> 1;
        const int ge = st.low >= LC3_MUL24(tmp, row[mid] & 0xffffu);
        lo = ge ? mid : lo;
        hi = ge ? hi : mid - 1;
    }
    const uint32_t sv = row[lo];
    st.low -= LC3_MUL24(tmp, sv & 0xffffu);
    st.range = LC3_MUL24(tmp, sv >> 16);
    lc3_p_ac_renorm_sel(c, st, err);
    return lo;
}
// The spectral model rows for the decoder: LC3_DCF_ROW_WORDS = 20 words per context, words 0..16 the symbols' (cum | freq
// << 16), words 17..19 copies of symbols 4, 8 and 12.  One aligned 16-byte read of words 16..19 decides "escape" and which
// group of four symbols holds the answer, a second one fetches that group: two dependent LDS reads per symbol instead of
// the binary search's six.
#define LC3_DCF_ROW_WORDS 20
__device__ __forceinline__ uint32_t lc3_dcf_word(int i) {
    const int p = i / LC3_DCF_ROW_WORDS, w = i % LC3_DCF_ROW_WORDS;
    const int j = w < 17 ? w : 4 * (w - 16);
    return (uint32_t)(int)LC3T_AC_SPEC_CUMFREQ[p][j] | ((uint32_t)(int)LC3T_AC_SPEC_FREQ[p][j] << 16);
}
// pv: words 16..19 of the row (symbols 16, 4, 8, 12), fetched by the caller one symbol ahead
__device__ __forceinline__ int lc3_p_ac_decode_spec_sel(lc3_parse_ctx &c, lc3_acdec &st, const uint32_t *row, const lc3_i4 pv, int &err) {
    const uint32_t tmp = st.range >> 10, limit = tmp << 10;
    err |= st.low >= limit;
    const int ge16 = st.low >= LC3_MUL24(tmp, (uint32_t)pv[0] & 0xffffu);
    const int g = (int)(st.low >= LC3_MUL24(tmp, (uint32_t)pv[1] & 0xffffu)) + (int)(st.low >= LC3_MUL24(tmp, (uint32_t)pv[2] & 0xffffu)) +
                  (int)(st.low >= LC3_MUL24(tmp, (uint32_t)pv[3] & 0xffffu));
    const lc3_i4 q = ((const lc3_i4 *)row)[g];   // symbols 4g .. 4g + 3; symbol 4g is known to satisfy the test
    const int n = (int)(st.low >= LC3_MUL24(tmp, (uint32_t)q[1] & 0xffffu)) + (int)(st.low >= LC3_MUL24(tmp, (uint32_t)q[2] & 0xffffu)) +
                  (int)(st.low >= LC3_MUL24(tmp, (uint32_t)q[3] & 0xffffu));
    // q[n] as a two-level select on the bits of n (a chain of comparisons came out as nested branches)
    const uint32_t q01 = (n & 1) ? (uint32_t)q[1] : (uint32_t)q[0], q23 = (n & 1) ? (uint32_t)q[3] : (uint32_t)q[2];
    uint32_t sv = (n & 2) ? q23 : q01;
    sv = ge16 ? (uint32_t)pv[0] : sv;
    const int lo = ge16 ? 16 : 4 * g + n;
    st.low -= LC3_MUL24(tmp, sv & 0xffffu);
    st.range = LC3_MUL24(tmp, sv >> 16);
    lc3_p_ac_renorm_sel<0, 0>(c, st, err);  // the caller requests the next head bytes (after it has consumed its tail bits) and checks the cursor
    return lo;
}

// read_res_bit (decoder/arithmetic_codec.rs:339-383) on a line whose value `xv` is already in a register; the caller stores it back
template <int COUNT>
__device__ __forceinline__ int lc3_p_res_bit(lc3_parse_ctx &c, int idx, int32_t &xv, int &nbits_res, int &cont) {
    int bit;
    if (nbits_res == 0) { cont = 0; return 0; }
    if (lc3_p_bool(c, bit)) return -1;
    nbits_res -= 1;
    if (bit) {
        if (xv > 0) xv += 1;
        else if (xv < 0) xv -= 1;
        else {
            if (nbits_res == 0) { cont = 0; return 0; }
            if (lc3_p_bool(c, bit)) return -1;
            nbits_res -= 1;
            xv = bit ? -1 : 1;
            if (COUNT) c.nnz += 1;
        }
        if (COUNT) c.seed += (uint32_t)idx;  // |x| grew by one
    }
    cont = 1;
    return 0;
}

// What follows the spectral data in arithmetic_codec::decode: the number of residual bits (calc_num_residual_bits :385-405) and, in
// LSB mode, the refinement of the integers in place (decode_residual_bits :184-206).  range: the range decoder's final range; ntup:
// pairs decoded; lev_end: 1 + the last pair that was coded with escape levels
template <int COUNT>
__device__ __forceinline__ int lc3_parse_finish(lc3_parse_ctx &c, uint32_t range, int nbits, int ntup, int lev_end, int lsb_mode, int ne) {
    // lines lastnz .. ne-1 are zero (:131-133): not stored, lc3_reconstruct_frame substitutes zeros when it reads them
    // calc_num_residual_bits :385-405
    {
        const int nbits_side = c.tail - 8;
        const int nbits_ari = (c.head + 1 - 3) * 8 + 25 - lc3_ilog2(range);
        if (nbits < nbits_side + nbits_ari) return -6;  // NegativeResidualNumBits
        int nres = nbits - nbits_side - nbits_ari, cont;
        lc3_px_set(c, AD_TAIL0, c.tail);
        lc3_px_set(c, AD_NRES_MAX, nres);
        lc3_px_set(c, AD_HEAD, c.head);
        if (lsb_mode) {  // decode_residual_bits :184-206: refines the integers in place
            // save_lev is read by LINE index k = 0, 2, 4 .. but was written by TUPLE index: entries at or beyond the
            // number of tuples were never written (zero in the reference), so the walk ends at ntup.  The levels are
            // fetched eight at a time.
            // The levels and the line pairs they refine are fetched eight pairs at a time (a lane of this kernel is latency-bound: a
            // read-modify-write of a plane word per bit cost a memory round trip each).
            struct q2 { int32_t v[2]; };
            int stop = 0;
            const int walk_end = lev_end < ntup ? lev_end : ntup;  // entries at and beyond lev_end are 0: skipped without reading a bit
            for (int k0 = 0; k0 < walk_end && !stop; k0 += 16) {
                int lv[8];
                q2 xp[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = k0 + 2 * j, in = k < ntup;  // lines k, k + 1 < 2 * ntup <= ne
                    lv[j] = in ? lc3_px_get(c, LC3_PLANE_LEV + k) : 0;
                    xp[j] = __builtin_bit_cast(q2, *(const lc3_i2 *)(c.plane + (LC3_PLANE_X + (in ? k : 0)) * LC3_PLANE_STRIDE));
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (!stop && lv[j] > 0) {
                        const int k = k0 + 2 * j;
                        if (lc3_p_res_bit<COUNT>(c, k, xp[j].v[0], nres, cont)) return -7;
                        if (!cont) stop = 1;
                        else {
                            if (lc3_p_res_bit<COUNT>(c, k + 1, xp[j].v[1], nres, cont)) return -7;
                            if (!cont) stop = 1;
                        }
                        *(lc3_i2 *)(c.plane + (LC3_PLANE_X + k) * LC3_PLANE_STRIDE) = __builtin_bit_cast(lc3_i2, xp[j]);
                    }
                }
            }
        }
    }
    LC3_PSTAMP(c, 3);
    return 0;
}

// read_frame: side info + arithmetic_codec::decode up to (not including) the non-lsb residual bits, the noise seed
// and the zero-frame flag, which the synthesis kernel derives lane-parallel from x.  Returns 0 when the frame parsed.
// COUNT: keep the running count of non-zero lines and the noise-filling seed (the reconstruction kernels of a full batch derive both
// from the integers, wave-parallel: the symbol loop then carries ten operations less per symbol, two of them quarter-rate multiplies).
template <int COUNT = 1>
__device__ __forceinline__ int lc3_parse_frame(lc3_parse_ctx &c, int ne, int fs_ind, int n_ms_10) {
    int lastnz = 0, lsb_mode = 0, num_tns = 0, ord[2];
    c.nnz = 0;
    c.seed = 0;
    int rc = lc3_parse_side_info(c, fs_ind, ne, lastnz, lsb_mode, num_tns, ord);
    if (rc) return rc;
    LC3_PSTAMP(c, 0);
    const int nbits = c.len * 8;
    lc3_acdec st;
    int sym = 0;
    // ac_dec_init :57-65
    if (!(c.head + 2 < c.len)) return -1;
    st.low = ((uint32_t)c.bytes[c.head] << 16) | ((uint32_t)c.bytes[c.head + 1] << 8) | (uint32_t)c.bytes[c.head + 2];
    c.head += 3;
    st.range = 0x00ffffffu;
    lc3_p_prime(c);  // from here to the end of the spectral data the readers run on the register copies
    // decode_tns_data :304-337
    {
        const int wt = nbits < (n_ms_10 ? 480 : 360);
        for (int k = 0; k < 16; k++) lc3_px_set(c, AD_RCI + k, 0);
        for (int f = 0; f < 2; f++) {
            int order = ord[f];
            if (f < num_tns && order > 0) {
                int err = 0;
                order = lc3_p_ac_decode_sel<7, 3>(c, st, c.tns + wt * 8, err) + 1;
                for (int k = 0; k < order; k++)
                    lc3_px_set(c, AD_RCI + f * 8 + k, lc3_p_ac_decode_sel<16, 5>(c, st, c.tns + 16 + k * 17, err));
                if (err) return -2;
            }
            lc3_px_set(c, AD_ORD0 + f, order);
        }
    }
    LC3_PSTAMP(c, 1);
    // decode_spectral_data :211-302
    const int ntup = lastnz / 2;
    int lev_end = 0;  // 1 + the last pair that was coded with escape levels (lsb_mode's refinement walk has nothing to do beyond it)
    {
        const int rate_flag = nbits > (160 + fs_ind * 160) ? 512 : 0;
        // One symbol per iteration and lane: a lane decodes the escape symbols of its pair (each followed by the pair's next
        // bit plane) and then the main symbol (followed by the signs), and moves on to its next pair by itself.  With a pair
        // index common to the wave the iterations add up to sum over pairs of (1 + deepest escape level of any lane); here
        // to max over lanes of (sum over pairs of 1 + level), 1.45x fewer on the benchmark's frames.
        // Reference loop per pair: `while lev < 14 { decode; if sym < 16 break; [two LSBs]; lev += 1 }` -- a pair that reaches
        // level 14 ends without another symbol and keeps the escape symbol (16).
        // The model row of a symbol follows from the symbol before it (context) through two dependent LDS reads (context -> row index,
        // row -> its words 16..19) and a third one picks the group of four: the first two are made one symbol AHEAD, as soon as the
        // current symbol is known, and land while the lane consumes the tail bits and stores the pair -- one exposed LDS round trip
        // per symbol instead of three.
        int cctx = 0, err = 0, tup = 0, lev = 0, slack = 0x7fffffff;
        int32_t xk = 0, xk1 = 0;
        const int hi_from = ne / 2;  // pairs with 2 * tup > ne / 2 use the upper half of the contexts
        const uint32_t *row = c.cf + (int)c.lookup[rate_flag + (0 > hi_from ? 256 : 0)] * LC3_DCF_ROW_WORDS;
        lc3_i4 pv = ((const lc3_i4 *)row)[4];
        while (tup < ntup) {
            if (lev < 14) sym = lc3_p_ac_decode_spec_sel(c, st, row, pv, err);
            const int esc = sym >= 16 && lev < 14;
            const int a = sym & 3, b = sym >> 2;
            const int lv = lev < 3 ? lev : 3;
            const int32_t m0 = xk + (int32_t)((uint32_t)a << lev), m1 = xk1 + (int32_t)((uint32_t)b << lev);  // if this is the main symbol
            // two tail bits: after an escape symbol the pair's next bit plane (when it is transmitted), after the main symbol
            // the signs of the non-zero values
            const int want_e = !lsb_mode || lev > 0;
            int bit0, bit1;
            lc3_p_bool2_sel(c, esc ? want_e : m0 > 0, esc ? want_e : m1 > 0, slack, bit0, bit1);
            // where the lane goes next, and that symbol's row index (requested behind the last use of a prefetched byte: LDS returns
            // in order, and such a use waits for everything in flight)
            const int n_cctx = esc ? cctx : (cctx & 15) * 16 + (lv <= 1 ? 1 + ((a + b) << lv) : 12 + lv);  // (a + b) * (lv + 1) for lv = 0, 1
            const int n_tup = tup + !esc, n_lev = esc ? lev + 1 : 0;
            const int n_lv = n_lev < 3 ? n_lev : 3;
            const int n_row = (int)c.lookup[n_cctx + rate_flag + ((n_tup * 2) > hi_from ? 256 : 0) + n_lv * 1024];
            const int32_t v0 = bit0 ? -m0 : m0, v1 = bit1 ? -m1 : m1;
            // (an escape step stores its partial values too: the pair's main step overwrites them)
            {   // the pair as one 64-bit store (8-byte aligned: the column, LC3_PLANE_X and 2 * tup are even numbers of words)
                lc3_i2 pr;
                pr[0] = v0;
                pr[1] = v1;
                *(lc3_i2 *)(c.plane + (LC3_PLANE_X + 2 * tup) * LC3_PLANE_STRIDE) = pr;
            }
            if (lsb_mode && !esc) lc3_px_set(c, LC3_PLANE_LEV + tup, lev);  // written by TUPLE index, read by LINE index (:184-195)
            lev_end = (!esc && lev > 0) ? tup + 1 : lev_end;
            row = c.cf + n_row * LC3_DCF_ROW_WORDS;
            pv = ((const lc3_i4 *)row)[4];
            lc3_p_head_refill(c);
            asm volatile("" ::: "memory");  // (keeps the three requests HERE: left alone, the row read sinks into the next iteration's `lev < 14` block)
            if (COUNT) {
                c.nnz += esc ? 0u : (uint32_t)(m0 != 0) + (uint32_t)(m1 != 0);
                c.seed += esc ? 0u : (uint32_t)m0 * (uint32_t)(2 * tup) + (uint32_t)m1 * (uint32_t)(2 * tup + 1);
            }
            cctx = n_cctx;
            xk = esc ? xk + (int32_t)((uint32_t)bit0 << lev) : 0;
            xk1 = esc ? xk1 + (int32_t)((uint32_t)bit1 << lev) : 0;
            tup = n_tup;
            lev = n_lev;
        }
        // the loop's deferred bound checks: a head byte read at or beyond the end of the frame, a tail bit more than two bytes below the
        // head cursor as of its read, a tail bit before the start of the frame
        err |= (c.head > c.len) | (slack < 0) | (c.len - ((c.tail - 1) >> 3) - 1 < 0);
        if (err) return -4;
    }
    LC3_PSTAMP(c, 2);
    return lc3_parse_finish<COUNT>(c, st.range, nbits, ntup, lev_end, lsb_mode, ne);
}

// the reconstruction context and the scale factors (used by the consumer wave below and by lc3_reconstruct_frame further down)
struct lc3_recon_ctx {
    float *scf;            // 16 scale factors of this lane, element n at scf[n * sstride] (LDS, dynamically indexed)
    int sstride;
    const uint32_t *mpvq;  // MPVQ_OFFSETS[16][11] (LDS copy)
    const uint16_t *ifs;   // band index table of the configuration, nb + 1 entries (LDS copy: read at every band boundary of
                           // the line loop, and a table word fetched from HBM there waits for every outstanding plane access)
};

// mpvq_deenum (decoder/spectral_noise_shaping.rs:155-235) writing the pulses into scf slots as floats is not possible
// (they are needed as integers first), so the pulses live in a 16-entry register array filled by static unrolling.
__device__ __forceinline__ void lc3_r_deenum(const lc3_recon_ctx &r, int dim_in, int k_val_in, int ls_ind, uint32_t mpvq_ind,
                                             int (&vec)[16], int base) {
    int leading_sign = ls_ind == 0 ? 1 : -1, k_max_local = k_val_in, done = 0;
    uint32_t ind = mpvq_ind;
#pragma unroll
    for (int pos = 0; pos < 16; pos++) {
        if (pos < dim_in && !done) {
            const uint32_t *h_row = r.mpvq + (dim_in - 1 - pos) * 11;
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
                const int k_delta = k_max_local - k_acc;
                if (k_delta != 0) {
                    vec[base + pos] = leading_sign < 0 ? -k_delta : k_delta;
                    leading_sign = (ind & 1u) ? -1 : 1;
                    ind >>= 1;
                    k_max_local -= k_delta;
                }
            } else {
                vec[base + pos] = leading_sign < 0 ? -k_max_local : k_max_local;
                done = 1;
            }
        }
    }
}


// spectral_noise_shaping::decode: the 16 scale factors scf[n] = codebook + gain * (y . D) (:21-73).  TO_REGS = 0: into the lane's LDS
// slots r.scf; 1: into out[16] (registers: fully unrolled)
template <int TO_REGS>
__device__ __forceinline__ void lc3_recon_scf(const lc3_recon_ctx &r, const lc3_sns_side &sv, float *out) {
    int y[16];
#pragma unroll
    for (int n = 0; n < 16; n++) y[n] = 0;
    const int shape_j = (sv.sub_msb << 1) + sv.sub_lsb;
    const int ls_a = sv.ls_a;
    const uint32_t idx_a = sv.idx_a;
    if (shape_j == 0) {
        lc3_r_deenum(r, 10, 10, ls_a, idx_a, y, 0);
        lc3_r_deenum(r, 6, 1, sv.ls_b, sv.idx_b, y, 10);
    } else if (shape_j == 1) lc3_r_deenum(r, 10, 10, ls_a, idx_a, y, 0);
    else if (shape_j == 2) lc3_r_deenum(r, 16, 8, ls_a, idx_a, y, 0);
    else lc3_r_deenum(r, 16, 6, ls_a, idx_a, y, 0);
    float y_norm = 0.0f;
#pragma unroll
    for (int n = 0; n < 16; n++) y_norm += (float)y[n] * (float)y[n];
    y_norm = lc3_sqrtf(y_norm);
    float gain;
    const int gi = sv.g_ind;
    if (shape_j == 0) gain = lc3_f(LC3T_SNS_VQ_REG_ADJ_GAINS_BITS, gi & 1);
    else if (shape_j == 1) gain = lc3_f(LC3T_SNS_VQ_REG_LF_ADJ_GAINS_BITS, gi & 3);
    else if (shape_j == 2) gain = lc3_f(LC3T_SNS_VQ_NEAR_ADJ_GAINS_BITS, gi & 3);
    else gain = lc3_f(LC3T_SNS_VQ_FAR_ADJ_GAINS_BITS, gi & 7);
    if (y_norm != 0.0f) gain /= y_norm;
    const int ind_lf = sv.ind_lf, ind_hf = sv.ind_hf;
    if (TO_REGS) {
#pragma unroll
        for (int n = 0; n < 16; n++) {
            float factor = 0.0f;
#pragma unroll
            for (int col = 0; col < 16; col++) factor += (float)y[col] * lc3_f(&LC3T_D_BITS[n][0], col);
            const float st1 = n < 8 ? lc3_f(&LC3T_LFCB_BITS[ind_lf][0], n) : lc3_f(&LC3T_HFCB_BITS[ind_hf][0], n - 8);
            out[n] = st1 + gain * factor;
        }
    } else {
        for (int n = 0; n < 16; n++) {
            float factor = 0.0f;
#pragma unroll
            for (int col = 0; col < 16; col++) factor += (float)y[col] * lc3_f(&LC3T_D_BITS[n][0], col);
            const float st1 = n < 8 ? lc3_f(&LC3T_LFCB_BITS[ind_lf][0], n) : lc3_f(&LC3T_HFCB_BITS[ind_hf][0], n - 8);
            r.scf[n * r.sstride] = st1 + gain * factor;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The spectral data of a frame decoded by a PRODUCER / CONSUMER pair of waves (full batches: lc3_parse_pc_kernel, lc3gpu.hip).
//
// A 65 536-frame launch gives every SIMD one wave of this kernel, a lone wave issues one instruction per four cycles whatever the
// instruction, and the symbol loop above is a chain of ~165 of them per symbol with three dependent LDS round trips.  Only part of
// that chain is the range decoder's recurrence: (low, range, context) -> symbol -> next context and model row.  The rest -- the
// pair's magnitudes, the tail bits that follow a symbol (bit planes of an escape, signs), the pair's store, the escape levels, the
// running count of non-zero lines and the noise-filling seed -- only CONSUMES symbols.  So the frames of a wave are walked by two
// waves: the producer runs the recurrence and leaves one word per symbol (the symbol and the head bytes its renormalisation took)
// in a ring in LDS; the consumer, on the same SIMD where the hardware places it there, replays the symbols a few iterations behind
// and does everything else, in the issue slots and LDS waits the producer leaves.  Same arithmetic, same order of every read of the
// frame and every store to the plane per frame; the iteration count is the wave's (every lane of both waves steps once per
// iteration, a lane that has finished its frame idles), so ring entry i of lane l is that lane's i-th symbol in both.
// The producer publishes its iteration count every LC3_PC_CHUNK iterations (stores of one wave reach LDS in order: the count
// follows the entries it covers) and waits when the consumer falls a ring behind; the consumer publishes what it has taken.
// ------------------------------------------------------------------------------------------------------------------
// the producer: side information (flags only), range decoder start, TNS data, then the symbols
// returns non-zero when it gave up waiting for the consumer (LC3_PC_SPIN_LIMIT polls: a partner that died; counted by the kernel)
__device__ __forceinline__ int lc3_pc_produce(lc3_parse_ctx &c, const lc3_pc_link &k, int ne, int fs_ind, int n_ms_10, int rc_in) {
    int lastnz = 0, lsb_mode = 0, num_tns = 0, ord[2] = {0, 0};
    int dead = rc_in != 0;
    if (!dead) dead = lc3_parse_side_info<0>(c, fs_ind, ne, lastnz, lsb_mode, num_tns, ord) != 0;
    const int nbits = c.len * 8;
    lc3_acdec st;
    st.low = 0;
    st.range = 0x00ffffffu;
    // ac_dec_init :57-65
    if (!dead && !(c.head + 2 < c.len)) dead = 1;
    if (!dead) {
        st.low = ((uint32_t)c.bytes[c.head] << 16) | ((uint32_t)c.bytes[c.head + 1] << 8) | (uint32_t)c.bytes[c.head + 2];
        c.head += 3;
        lc3_p_prime(c);
        // decode_tns_data :304-337
        const int wt = nbits < (n_ms_10 ? 480 : 360);
        for (int q = 0; q < 16; q++) lc3_px_set(c, AD_RCI + q, 0);
        for (int f = 0; f < 2; f++) {
            int order = ord[f];
            if (f < num_tns && order > 0) {
                int err = 0;
                order = lc3_p_ac_decode_sel<7, 3>(c, st, c.tns + wt * 8, err) + 1;
                for (int q = 0; q < order; q++) lc3_px_set(c, AD_RCI + f * 8 + q, lc3_p_ac_decode_sel<16, 5>(c, st, c.tns + 16 + q * 17, err));
                if (err) dead = 1;
            }
            lc3_px_set(c, AD_ORD0 + f, order);
        }
    }
    // start values for the consumer, then the first publication
    k.fin[0] = (uint32_t)c.head;
    k.fin[3 * k.fstride] = (uint32_t)dead;
    LC3_PC_STORE(k.p_count, 0);
    // decode_spectral_data :211-302 -- the recurrence only (see lc3_parse_frame for the whole of it)
    const int ntup = dead ? 0 : lastnz / 2;
    int err = 0, it = 0, gave_up = 0;
    {
        const int rate_flag = nbits > (160 + fs_ind * 160) ? 512 : 0;
        const int hi_from = ne / 2;
        int cctx = 0, tup = 0, lev = 0, sym = 0;
        const uint32_t *row = c.cf + (int)c.lookup[rate_flag + (0 > hi_from ? 256 : 0)] * LC3_DCF_ROW_WORDS;
        lc3_i4 pv = ((const lc3_i4 *)row)[4];
        int c_seen = 0, spins = 0;
        // A chunk of LC3_PC_CHUNK iterations at a time, straight-line (the loop-carried values of a lane then move between
        // registers once per chunk, not once per symbol); a lane that has finished its frame idles, and the count that is published
        // is always whole chunks: behind the last symbol of the wave the consumer finds every lane idle as well
        while (LC3_WAVE_ANY(tup < ntup)) {
#pragma unroll
            for (int u = 0; u < LC3_PC_CHUNK; u++) {
                if (tup < ntup) {
                    const int head0 = c.head;
                    if (lev < 14) sym = lc3_p_ac_decode_spec_sel(c, st, row, pv, err);
                    const int esc = sym >= 16 && lev < 14;
                    k.ring[((it + u) & k.mask) * k.stride] = (uint32_t)sym | ((uint32_t)(c.head - head0) << 5);
                    const int a = sym & 3, b = sym >> 2;
                    const int lv = lev < 3 ? lev : 3;
                    const int n_cctx = esc ? cctx : (cctx & 15) * 16 + (lv <= 1 ? 1 + ((a + b) << lv) : 12 + lv);
                    const int n_tup = tup + !esc, n_lev = esc ? lev + 1 : 0;
                    const int n_lv = n_lev < 3 ? n_lev : 3;
                    const int n_row = (int)c.lookup[n_cctx + rate_flag + ((n_tup * 2) > hi_from ? 256 : 0) + n_lv * 1024];
                    row = c.cf + n_row * LC3_DCF_ROW_WORDS;
                    pv = ((const lc3_i4 *)row)[4];
                    lc3_p_head_refill(c);
                    cctx = n_cctx;
                    tup = n_tup;
                    lev = n_lev;
                }
            }
            it += LC3_PC_CHUNK;
            LC3_PC_STORE(k.p_count, it);
            // the next chunk's entries must not land on entries the consumer has not taken yet
            while (it + LC3_PC_CHUNK - c_seen > k.mask + 1 && spins < LC3_PC_SPIN_LIMIT) {
                c_seen = LC3_PC_LOAD(k.c_count);
                if (it + LC3_PC_CHUNK - c_seen > k.mask + 1) {
                    LC3_PC_PAUSE();
                    spins++;
                }
            }
        }
        gave_up = spins >= LC3_PC_SPIN_LIMIT;
        err |= gave_up;
    }
    k.fin[k.fstride] = st.range;
    k.fin[2 * k.fstride] = (uint32_t)c.head;
    k.fin[3 * k.fstride] = (uint32_t)(dead | (err != 0));
    LC3_PC_RELEASE();  // the TNS words in the plane and the hand-over words are out before ...
    LC3_PC_STORE(k.p_count, it | LC3_PC_DONE);
    return gave_up;
}

// the consumer: side information (to the plane), then everything of decode_spectral_data that only consumes symbols, then the rest of
// arithmetic_codec::decode (lc3_parse_finish).  Returns 0 when the frame parsed (as lc3_parse_frame<COUNT>)
// r / scf_out (optional): the reconstruction context and 16 registers for the frame's scale factors, which only need the side
// information: computed here while the producer is still at its start (the ring lies in the lane's scale-factor slots in LDS, so
// they stay in registers until the symbols are through)
// timed_out (optional): set when the consumer gave up waiting for the producer (the frame then counts as one that did not parse: it is
// concealed, and the kernel counts the event -- lc3gpu_decoder_pair_timeouts)
template <int COUNT>
__device__ __forceinline__ int lc3_pc_consume(lc3_parse_ctx &c, const lc3_pc_link &k, int ne, int fs_ind, int rc_in,
                                              const lc3_recon_ctx *r = nullptr, float *scf_out = nullptr, int *timed_out = nullptr) {
    int lastnz = 0, lsb_mode = 0, num_tns = 0, ord[2] = {0, 0};
    c.nnz = 0;
    c.seed = 0;
    int rc = rc_in;
    lc3_sns_side sns = {0, 0, 0, 0, 0, 0, 0, 0u, 0u};
    if (rc == 0) rc = lc3_parse_side_info<1>(c, fs_ind, ne, lastnz, lsb_mode, num_tns, ord, &sns);
    if (scf_out) {
        if (rc != 0) sns = lc3_sns_side{0, 0, 0, 0, 0, 0, 0, 0u, 0u};  // (a frame that does not parse: any valid indices)
        lc3_recon_scf<1>(*r, sns, scf_out);
    }
    const int nbits = c.len * 8;
    // the producer's start values
    int spins = 0, pc;
    while ((pc = LC3_PC_LOAD(k.p_count)) < 0 && spins < LC3_PC_SPIN_LIMIT) {
        LC3_PC_PAUSE();
        spins++;
    }
    const int never = pc < 0;  // the producer never answered: nothing in the ring or the hand-over words means anything
    c.head = never ? 0 : (int)k.fin[0];
    const int dead = rc != 0 || never || k.fin[3 * k.fstride] != 0u;  // (the producer's own start can fail: ac_dec_init, the TNS data)
    const int ntup = dead ? 0 : lastnz / 2;
    int lev_end = 0, err = 0, it = 0;
    {
        int tup = 0, lev = 0, slack = 0x7fffffff;
        int32_t xk = 0, xk1 = 0;
        c.tcur = lc3_p_tail_byte(c, c.tail >> 3);
        c.tnext = lc3_p_tail_byte(c, (c.tail >> 3) + 1);
        // whole chunks of LC3_PC_CHUNK iterations, straight-line (see the producer); the producer's count is whole chunks, the wave's last
        // one padded with iterations in which every lane idles
        for (; !never;) {
            while ((((pc = LC3_PC_LOAD(k.p_count)) & (LC3_PC_DONE - 1)) <= it) && !(pc & LC3_PC_DONE) && spins < LC3_PC_SPIN_LIMIT) {
                LC3_PC_PAUSE();
                spins++;
            }
            if ((pc & (LC3_PC_DONE - 1)) <= it) break;  // the producer has ended (or never answered)
#pragma unroll
            for (int u = 0; u < LC3_PC_CHUNK; u++) {
                if (tup < ntup) {
                    const uint32_t w = k.ring[((it + u) & k.mask) * k.stride];
                    const int sym = (int)(w & 31u);
                    c.head += (int)(w >> 5);
                    const int esc = sym >= 16 && lev < 14;
                    const int a = sym & 3, b = sym >> 2;
                    const int32_t m0 = xk + (int32_t)((uint32_t)a << lev), m1 = xk1 + (int32_t)((uint32_t)b << lev);  // if this is the main symbol
                    // two tail bits: after an escape symbol the pair's next bit plane (when it is transmitted), after the main symbol
                    // the signs of the non-zero values
                    const int want_e = !lsb_mode || lev > 0;
                    int bit0, bit1;
                    lc3_p_bool2_sel(c, esc ? want_e : m0 > 0, esc ? want_e : m1 > 0, slack, bit0, bit1);
                    const int32_t v0 = bit0 ? -m0 : m0, v1 = bit1 ? -m1 : m1;
                    {   // (an escape step stores its partial values too: the pair's main step overwrites them)
                        lc3_i2 pr;
                        pr[0] = v0;
                        pr[1] = v1;
                        *(lc3_i2 *)(c.plane + (LC3_PLANE_X + 2 * tup) * LC3_PLANE_STRIDE) = pr;
                    }
                    if (lsb_mode && !esc) lc3_px_set(c, LC3_PLANE_LEV + tup, lev);  // written by TUPLE index, read by LINE index (:184-195)
                    lev_end = (!esc && lev > 0) ? tup + 1 : lev_end;
                    if (COUNT) {
                        c.nnz += esc ? 0u : (uint32_t)(m0 != 0) + (uint32_t)(m1 != 0);
                        // sum |x_k| * k mod 2^16 (:140-145): (m0 + m1) * 2 tup + m1, the factors below 2^17 and 2^9
                        c.seed += esc ? 0u : LC3_MUL24((uint32_t)(m0 + m1), (uint32_t)(2 * tup)) + (uint32_t)m1;
                    }
                    xk = esc ? xk + (int32_t)((uint32_t)bit0 << lev) : 0;
                    xk1 = esc ? xk1 + (int32_t)((uint32_t)bit1 << lev) : 0;
                    tup += !esc;
                    lev = esc ? lev + 1 : 0;
                }
            }
            it += LC3_PC_CHUNK;
            LC3_PC_STORE(k.c_count, it);
        }
        err |= LC3_WAVE_ANY(tup < ntup) && tup < ntup;  // the producer ended short of this lane's frame: cannot happen while both walk the same frames
        // the loop's deferred bound checks (see lc3_parse_frame)
        err |= (slack < 0) | (c.len - ((c.tail - 1) >> 3) - 1 < 0);
    }
    LC3_PC_STORE(k.c_count, LC3_PC_DONE - 1);  // (a producer waiting for ring space after an abandoned loop goes on)
    // the producer's end values
    while (!((pc = LC3_PC_LOAD(k.p_count)) & LC3_PC_DONE) && spins < LC3_PC_SPIN_LIMIT) {
        LC3_PC_PAUSE();
        spins++;
    }
    LC3_PC_ACQUIRE();
    const uint32_t range = k.fin[k.fstride];
    c.head = (int)k.fin[2 * k.fstride];
    err |= (int)k.fin[3 * k.fstride] | (spins >= LC3_PC_SPIN_LIMIT) | (c.head > c.len);
    if (timed_out) *timed_out = spins >= LC3_PC_SPIN_LIMIT;
    if (rc) return rc;
    if (dead) return -2;
    if (err) return -4;
    return lc3_parse_finish<COUNT>(c, range, nbits, ntup, lev_end, lsb_mode, ne);
}

// ------------------------------------------------------------------------------------------------------------------
// D4-D8 of a parsed frame on the same lane: what is left of arithmetic_codec::decode (residual-bit count and bounds,
// noise seed, zero-frame flag; decoder/arithmetic_codec.rs:134-183), residual_spectrum::decode
// (decoder/residual_spectrum.rs:13-39), noise_filling::apply_noise_filling (decoder/noise_filling.rs:18-56),
// global_gain::apply_global_gain (decoder/global_gain.rs:15-25), temporal_noise_shaping::apply_temporal_noise_shaping
// (decoder/temporal_noise_shaping.rs:24-137) and spectral_noise_shaping::decode
// (decoder/spectral_noise_shaping.rs:21-151).  Every one of them walks the lines in order carrying state (bit rank, LCG,
// lattice memory, band pointer), and every line is touched by them in this order, so ONE pass over k = 0..ne-1 applies
// them all: int -> f32, residual or noise value, gain, lattice, band gain, store.  The f32 operations on a line are the
// reference's, in the reference's order.
// Returns 1 (and sets AD_OK) when the frame is usable, 0 -> the synthesis kernel conceals it.
// ------------------------------------------------------------------------------------------------------------------
// interpolated scale factor of band slot b (0..63) before the nb < 64 folding (:75-98)
__device__ __forceinline__ float lc3_r_sfi(const lc3_recon_ctx &r, int b) {
    if (b < 2) return r.scf[0];
    if (b >= 62) {
        const float s15 = r.scf[15 * r.sstride], s14 = r.scf[14 * r.sstride];
        return s15 + (b == 62 ? 1.0f / 8.0f : 3.0f / 8.0f) * (s15 - s14);
    }
    const int n = (b - 2) >> 2, q = (b - 2) & 3;
    const float fn = r.scf[n * r.sstride], d = r.scf[(n + 1) * r.sstride] - fn;
    const float w = q == 0 ? 1.0f / 8.0f : (q == 1 ? 3.0f / 8.0f : (q == 2 ? 5.0f / 8.0f : 7.0f / 8.0f));
    return fn + (w * d);
}
// gain of band bi (0..nb-1): exp2_raw of the scale factor after the decoder's nb < 64 folding (:100-111)
__device__ __forceinline__ float lc3_r_band_gain(const lc3_recon_ctx &r, int bi, int nb) {
    const int n2 = 64 - nb;
    float sf;
    if (n2 == 0) sf = lc3_r_sfi(r, bi);
    else if (bi < n2) sf = (lc3_r_sfi(r, 2 * bi) + lc3_r_sfi(r, 2 * bi + 1)) / 2.0f;
    else sf = lc3_r_sfi(r, bi + n2);
    return lc3_exp2_raw(sf);
}

// Launches of a few frames leave the reconstruction to the synthesis kernel, which does it with the 64 lanes of the stream's wave
// (lc3_dec_reconstruct_wave, lc3_dev_dec.h): a lane walking the 400 lines of one frame alone takes 0.13 ms whatever the launch
// size.  What only this kernel can supply is checked / extracted here: the residual-bit count and its bounds (:168-183) and the
// residual bits themselves, tail bits tail0 .. tail0 + n_res - 1 of the frame, as a bit mask in the (then unused) level words.
// Returns 1 when the frame is usable.
__device__ __forceinline__ int lc3_reconstruct_prepare_late(lc3_parse_ctx &c) {
    const int nbytes = c.len;
    const int lsb_mode = lc3_px_get(c, SI_LSB_MODE), tail0 = lc3_px_get(c, AD_TAIL0), nres_max = lc3_px_get(c, AD_NRES_MAX),
              head = lc3_px_get(c, AD_HEAD);
    int n_res = 0;
    if (!lsb_mode) {
        n_res = (int)c.nnz < nres_max ? (int)c.nnz : nres_max;
        if (n_res > 480) return 0;  // ResidualBoolDataOverflow (Vec<bool, 480>)
        if (n_res > 0) {
            const int last_byte = (tail0 + n_res - 1) / 8;
            if (nbytes - head - last_byte + 2 < 0) return 0;
            if (nbytes - last_byte - 1 < 0) return 0;
        }
        for (int w = 0; 32 * w < n_res; w++) {  // 32 tail bits at a time: five bytes, shifted (bits beyond n_res are never looked at)
            const int pos = tail0 + 32 * w, bi = pos >> 3;
            unsigned long long v = 0;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int idx = nbytes - 1 - bi - i;
                v |= (unsigned long long)(idx >= 0 ? c.bytes[idx] : (uint8_t)0) << (8 * i);
            }
            lc3_px_set(c, LC3_PLANE_LEV + w, (int32_t)(uint32_t)(v >> (pos & 7)));
        }
    }
    lc3_px_set(c, AD_NRES, n_res);
    return 1;
}

// For the wave-per-frame reconstruction kernel (lc3_dev_dec_recon.h) the parser also de-enumerates the SNS pulse vector
// (decoder/spectral_noise_shaping.rs:21-59,155-235): serial integer work, a few hundred operations on a lane that has just walked
// ~300 symbols, against a chain of wave-uniform scalar steps in every one of the 65 536 waves there.  16 values in -10 .. 10, five
// bits each, six to a word.
__device__ __forceinline__ void lc3_parse_pulses(lc3_parse_ctx &c, const lc3_recon_ctx &r) {
    int y[16];
#pragma unroll
    for (int n = 0; n < 16; n++) y[n] = 0;
    const int shape_j = (lc3_px_get(c, SI_SUB_MSB) << 1) + lc3_px_get(c, SI_SUB_LSB);
    const int ls_a = lc3_px_get(c, SI_LS_A), ls_b = lc3_px_get(c, SI_LS_B);
    const uint32_t idx_a = (uint32_t)lc3_px_get(c, SI_IDX_A), idx_b = (uint32_t)lc3_px_get(c, SI_IDX_B);
    if (shape_j == 0) {
        lc3_r_deenum(r, 10, 10, ls_a, idx_a, y, 0);
        lc3_r_deenum(r, 6, 1, ls_b, idx_b, y, 10);
    } else if (shape_j == 1) lc3_r_deenum(r, 10, 10, ls_a, idx_a, y, 0);
    else if (shape_j == 2) lc3_r_deenum(r, 16, 8, ls_a, idx_a, y, 0);
    else lc3_r_deenum(r, 16, 6, ls_a, idx_a, y, 0);
    uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll
    for (int n = 0; n < 16; n++) w[n / 6] |= ((uint32_t)y[n] & 31u) << (5 * (n % 6));
#pragma unroll
    for (int i = 0; i < 3; i++) lc3_px_set(c, AD_Y + i, (int32_t)w[i]);
}
// element n of the packed pulse vector (words w0..w2 = AD_Y .. AD_Y + 2)
__device__ __forceinline__ int lc3_pulse_unpack(uint32_t w0, uint32_t w1, uint32_t w2, int n) {
    const uint32_t w = n < 6 ? w0 : (n < 12 ? w1 : w2);
    const uint32_t f = (w >> (5 * (n % 6))) & 31u;
    return (int)(f ^ 16u) - 16;
}

// ------------------------------------------------------------------------------------------------------------------
// TNS synthesis lattice (decoder/temporal_noise_shaping.rs:60-137) of four consecutive lines k0 .. k0+3 (k0 wave-uniform) of ONE LANE's
// frame, in place in v.  The lane's first filter covers [lo0, lo1) with the coefficients in rq, its second one -- coefficients rq1,
// taken over at line lo1 together with the reset of the lattice memory beyond ord0 -- covers [lo1, hi) (lo1 = hi: one filter; lo0 = hi:
// none).  Per line:   t = x - rc[7] * st[7];  q = 6 .. 0: t -= rc[q] * st[q]; st[q+1] = rc[q] * t + st[q];  x = st[0] = t
// always over all eight stages, without per-stage selects: the coefficients are ZERO beyond a filter's order, a stage with rc = 0
// subtracts rc * st = +-0 from t, which leaves t as it is (t is never -0: the inputs are converted integers, +-level and their
// products with positive gains, and a difference of equal values is +0), and what such stages write into the lattice memory beyond
// the order is discarded where it could be read (the reset at lo1 restores the zeros the reference still has there).
// The bounds are per lane (they follow the frame's bandwidth): lanes whose range the group lies inside take the wavefront form, lanes
// it straddles a boundary of walk it line by line, both under the execution mask; a group no lane's range touches costs a few compares.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lc3_tns_take_second(int ord0, float (&rq)[8], const float (&rq1)[8], float (&st)[8]) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
        rq[q] = rq1[q];
        st[q] = q >= ord0 ? 0.0f : st[q];
    }
}
__device__ __forceinline__ void lc3_tns_lattice4(int k0, int lo0, int lo1, int hi, int ord0, float (&rq)[8], const float (&rq1)[8],
                                                 float (&st)[8], float (&v)[4]) {
    if (k0 == lo1 && lo1 < hi) lc3_tns_take_second(ord0, rq, rq1, st);  // the group starts the second filter
    const int inside = (k0 >= lo0 && k0 + 3 < lo1) || (k0 >= lo1 && k0 + 3 < hi);
    if (inside) {
        // The four lines as a skewed wavefront: line j runs its stage q at step (7 - q) + 2 j.  Stage q of a line reads st[q] as the line
        // before left it (written by that line's stage q - 1, one step earlier) and writes st[q + 1], which the line before has read two
        // steps earlier: the order of every read and write of the lattice memory is that of the line-by-line walk, with up to four
        // independent operations in flight instead of one chain of dependent ones.
#pragma unroll
        for (int step = 0; step < 8 + 2 * 3; step++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int q = 7 - (step - 2 * j);
                if (q >= 0 && q <= 7) {
                    v[j] = v[j] - rq[q] * st[q];
                    LC3_KEEP_SCALAR(v[j]);  // (paired into packed-f32 operations the chain gains moves and hazard nops)
                    if (q < 7) {
                        st[q + 1] = rq[q] * v[j] + st[q];
                        LC3_KEEP_SCALAR(st[q + 1]);
                    }
                    if (q == 0) st[0] = v[j];
                }
            }
        }
    } else if (k0 + 3 >= lo0 && k0 < hi) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = k0 + j;
            if (k == lo1 && j > 0 && lo1 < hi) lc3_tns_take_second(ord0, rq, rq1, st);  // (j == 0: done above)
            if (k >= lo0 && k < hi) {
                float t = v[j];
#pragma unroll
                for (int q = 7; q >= 0; q--) {
                    t = t - rq[q] * st[q];
                    if (q < 7) st[q + 1] = rq[q] * t + st[q];
                }
                st[0] = t;
                v[j] = t;
            }
        }
    }
}

// Hand-over to the reconstruction kernels of a full batch (lc3_dev_dec_recon.h), which count the non-zero lines themselves: the tail
// bits that may turn out to be residual bits -- at most one per line below lastnz, at most AD_NRES_MAX, at most 480 -- as a bit mask in
// the level words; the count and its bound checks (:168-183) happen there.
__device__ __forceinline__ void lc3_reconstruct_prepare_wave(lc3_parse_ctx &c) {
    const int nbytes = c.len;
    const int lsb_mode = lc3_px_get(c, SI_LSB_MODE), tail0 = lc3_px_get(c, AD_TAIL0), nres_max = lc3_px_get(c, AD_NRES_MAX),
              lastnz = lc3_px_get(c, SI_LASTNZ);
    int n = nres_max < lastnz ? nres_max : lastnz;
    n = lsb_mode ? 0 : (n < 480 ? n : 480);
    for (int w = 0; 32 * w < n; w++) {  // 32 tail bits at a time: five bytes, shifted (a byte beyond the frame's start reads as 0)
        const int pos = tail0 + 32 * w, bi = pos >> 3;
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int idx = nbytes - 1 - bi - i;
            v |= (unsigned long long)(idx >= 0 ? c.bytes[idx] : (uint8_t)0) << (8 * i);
        }
        lc3_px_set(c, LC3_PLANE_LEV + w, (int32_t)(uint32_t)(v >> (pos & 7)));
    }
}

template <class CC>
__device__ __forceinline__ int lc3_reconstruct_frame(lc3_parse_ctx &c, const lc3_recon_ctx &r, const CC &cfg, const float *scf_pre = nullptr) {
    const int ne = cfg.ne, nbytes = c.len, nbits = nbytes * 8;
    // the side-information words of the column in one batch of independent loads (a lane of this kernel is
    // latency-bound: a word fetched at its point of use costs a full memory round trip)
    int32_t siw[LC3_PLANE_X];  // the SI_WORDS words and the padding up to the spectrum, fetched as 128-bit units
    {
        struct q4 { int32_t v[4]; };
        static_assert(SI_WORDS <= LC3_PLANE_X && LC3_PLANE_X % 4 == 0, "side-information region");
        const lc3_i4 *s4 = (const lc3_i4 *)c.plane;
#pragma unroll
        for (int i = 0; i < LC3_PLANE_X / 4; i++) {
            const q4 w = __builtin_bit_cast(q4, s4[i]);
#pragma unroll
            for (int j = 0; j < 4; j++) siw[4 * i + j] = w.v[j];
        }
    }
#define LC3_SIW(word) siw[word]
    const int lsb_mode = LC3_SIW(SI_LSB_MODE), lastnz = LC3_SIW(SI_LASTNZ), gg_ind = LC3_SIW(SI_GG);
    const int tail0 = LC3_SIW(AD_TAIL0), nres_max = LC3_SIW(AD_NRES_MAX), head = LC3_SIW(AD_HEAD);
    const int bw = LC3_SIW(SI_BW);
    // decode_residual_bits :168-183: one tail bit per non-zero line, at most nres_max.  Every read_tail_bool bound check
    // is monotone in the bit position, so checking the last position covers all of them.
    int n_res = 0;
    if (!lsb_mode) {
        n_res = (int)c.nnz < nres_max ? (int)c.nnz : nres_max;
        if (n_res > 480) return 0;  // ResidualBoolDataOverflow (Vec<bool, 480>)
        if (n_res > 0) {
            const int last_byte = (tail0 + n_res - 1) / 8;
            if (nbytes - head - last_byte + 2 < 0) return 0;
            if (nbytes - last_byte - 1 < 0) return 0;
        }
    }
    const int x0 = lc3_px_get(c, LC3_PLANE_X), x1 = lc3_px_get(c, LC3_PLANE_X + 1);  // lastnz >= 2: both were stored
    const int do_fill = !(lastnz == 2 && x0 == 0 && x1 == 0 && gg_ind == 0);  // zero frame :147-151
    uint32_t lcg = c.seed & 0xFFFFu;
    // spectral_noise_shaping::decode: scale factors scf[16] = codebook + gain * (y . D) (:21-73) -- or the values the caller has
    // computed already (the consumer wave of a producer / consumer pair does, while it waits for the first symbols)
    if (scf_pre) {
#pragma unroll
        for (int n = 0; n < 16; n++) r.scf[n * r.sstride] = scf_pre[n];
    } else {
        lc3_sns_side sv;
        sv.ind_lf = LC3_SIW(SI_IND_LF); sv.ind_hf = LC3_SIW(SI_IND_HF); sv.sub_msb = LC3_SIW(SI_SUB_MSB); sv.sub_lsb = LC3_SIW(SI_SUB_LSB);
        sv.g_ind = LC3_SIW(SI_G_IND); sv.ls_a = LC3_SIW(SI_LS_A); sv.ls_b = LC3_SIW(SI_LS_B);
        sv.idx_a = (uint32_t)LC3_SIW(SI_IDX_A); sv.idx_b = (uint32_t)LC3_SIW(SI_IDX_B);
        lc3_recon_scf<0>(r, sv, nullptr);
    }
    // global gain :15-25
    float gg;
    {
        const int fs = cfg.fs_ind + 1, q = nbits / (10 * fs);
        const int gg_off = -(q < 115 ? q : 115) - 105 - (5 * fs);
        gg = LC3_POW10_GG(gg_ind + gg_off);  // 10^(((float)gg_ind + (float)gg_off) / 28)
    }
    // TNS :24-137: all-pole lattice, state shared across both filters
    const int nbands = bw < 3 ? 1 : 2, num_tns = LC3_SIW(SI_NUM_TNS);
    const int ord0 = (0 < nbands && 0 < num_tns) ? LC3_SIW(AD_ORD0) : 0;
    const int ord1 = (1 < nbands && 1 < num_tns) ? LC3_SIW(AD_ORD0 + 1) : 0;
    const int lo0 = cfg.n_ms_10 ? LC3C_TNSDEC10[bw][0] : LC3C_TNSDEC75[bw][0];
    const int hi0 = cfg.n_ms_10 ? LC3C_TNSDEC10[bw][1] : LC3C_TNSDEC75[bw][1];
    const int lo1 = cfg.n_ms_10 ? LC3C_TNSDEC10[bw][2] : LC3C_TNSDEC75[bw][2];
    const int hi1 = cfg.n_ms_10 ? LC3C_TNSDEC10[bw][3] : LC3C_TNSDEC75[bw][3];
    // the two filters' coefficients, zero beyond the order (lc3_tns_lattice4); this lane's line ranges, empty without an active filter
    float st[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, rq[8], rq1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int r0 = LC3_SIW(AD_RCI + k), r1 = LC3_SIW(AD_RCI + 8 + k);
        rq[k] = (r0 != 0 && k < ord0) ? LC3_TNS_SIN_DEC(r0) : 0.0f;  // sin(step * (ri - 8)); SURVEY A12
        rq1[k] = (r1 != 0 && k < ord1) ? LC3_TNS_SIN_DEC(r1) : 0.0f;
    }
    const int tns_on = ord0 > 0 || ord1 > 0;
    const int t_lo0 = tns_on ? lo0 : 0, t_hi = tns_on ? (nbands == 2 ? hi1 : hi0) : 0, t_lo1 = tns_on ? (nbands == 2 ? lo1 : hi0) : 0;
    const int tns_wave = LC3_WAVE_ANY(tns_on);  // no lane of the wave has a filter: the lattice is skipped altogether
    // noise filling :18-56
    const int bw_stop = cfg.n_ms_10 ? LC3C_BWSTOP10[bw] : LC3C_BWSTOP75[bw];
    const int nf_start = cfg.n_ms_10 ? 24 : 18, nf_width = cfg.n_ms_10 ? 3 : 2;
    const int lim = bw_stop < ne ? bw_stop : ne;
    const float level = (8.0f - (float)LC3_SIW(SI_NF)) / 16.0f;
    LC3_PSTAMP(c, 4);
    // window of non-zero flags: bit (j + 3) <-> line k + j, j = -3 .. 3, lines at or beyond bw_stop count as zero.
    // The integers are fetched from the plane four lines at a time, two groups ahead of their use (xw holds lines
    // k0 .. k0+11 of the current group of four), so that the loads of a group have a whole group's work to land.
    uint32_t nzwin = 0;
    int32_t xw[12];
    {   // 128-bit loads (the column and LC3_PLANE_X are 16-byte aligned); words at or beyond lastnz are stale and masked
        struct q4 { int32_t v[4]; };
        const lc3_i4 *x4 = (const lc3_i4 *)(c.plane + LC3_PLANE_X * LC3_PLANE_STRIDE);
#pragma unroll
        for (int g4 = 0; g4 < 3; g4++) {
            const q4 w = __builtin_bit_cast(q4, x4[g4]);
#pragma unroll
            for (int j = 0; j < 4; j++) xw[4 * g4 + j] = 4 * g4 + j < lastnz ? w.v[j] : 0;  // lastnz <= ne
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (xw[j] != 0 && j < bw_stop) nzwin |= 1u << (j + 3);
    const uint32_t winmask = nf_width == 3 ? 0x7Fu : 0x3Eu;  // lines k-3..k+3 or k-2..k+2
    const uint16_t *ifs = r.ifs;
    int bi = 0, rank_nz = 0;
    int band_end = (int)ifs[1];  // first line of the next band (kept in a register: one table read per band, not per line)
    float g_band = lc3_r_band_gain(r, 0, cfg.nb);
    for (int k0 = 0; k0 < ne; k0 += 4) {  // ne is a multiple of 4
        // residual bits go to the first n_res non-zero lines of a frame (a few dozen lines in): once no lane of the wave has bits
        // left, the refinement code is skipped for the rest of the pass (a wave-uniform branch)
        const int res_live = LC3_WAVE_ANY(rank_nz < n_res);
        int32_t xnext[4];
        {   // one 128-bit load per group of four lines (up to 11 words past the spectrum: still inside the column)
            struct q4 { int32_t v[4]; };
            const q4 w = __builtin_bit_cast(q4, *(const lc3_i4 *)(c.plane + (LC3_PLANE_X + k0 + 12) * LC3_PLANE_STRIDE));
#pragma unroll
            for (int j = 0; j < 4; j++) xnext[j] = k0 + 12 + j < lastnz ? w.v[j] : 0;
        }
        float vout[4];
        // The per-line work is written with selects, not branches: the conditions differ from lane to lane (each lane is
        // another frame), and a divergent branch costs more scalar bookkeeping than the few operations it would skip.
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = k0 + j;
            const int32_t xi = xw[j];
            float v = (float)xi;
            if (res_live) {  // residual_spectrum::decode: the j-th non-zero line takes tail bit tail0 + j while j < n_res
                const int nz = xi != 0, take = nz && rank_nz < n_res;
                const int pos = tail0 + rank_nz;
                const int bidx = take ? nbytes - 1 - (pos >> 3) : 0;
                const int bit = (c.bytes[bidx] >> (pos & 7)) & 1;
                const float up = v > 0.0f ? 0.3125f : 0.1875f, dn = v > 0.0f ? 0.1875f : 0.3125f;
                const float v_adj = bit ? v + up : v - dn;
                v = take ? v_adj : v;
                rank_nz += nz;
            }
            {   // noise filling: next LCG state and a +-level line where the neighbourhood is empty
                const int fill = do_fill && k >= nf_start && k < lim && (nzwin & winmask) == 0;
                const uint32_t lcg_n = (13849u + LC3_MUL24(lcg, 31821u)) & 0xFFFFu;
                lcg = fill ? lcg_n : lcg;
                v = fill ? (lcg_n < 0x8000u ? level : -level) : v;
            }
            vout[j] = v * gg;
            // slide the window: drop line k - 3, bring in line k + 4
            nzwin = (nzwin >> 1) | ((xw[j + 4] != 0 && k + 4 < bw_stop) ? 64u : 0u);
        }
        if (c.dbg)
            for (int j = 0; j < 4; j++) c.dbg[LC3_DBG_GAIN + k0 + j] = vout[j];
        if (tns_wave) lc3_tns_lattice4(k0, t_lo0, t_lo1, t_hi, ord0, rq, rq1, st, vout);  // TNS synthesis, the group's four lines together
        if (c.dbg)
            for (int j = 0; j < 4; j++) c.dbg[LC3_DBG_TNS + k0 + j] = vout[j];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = k0 + j;
            while (k >= band_end) {  // band of line k (bands are contiguous and non-empty below ne)
                bi++;
                band_end = (int)ifs[bi + 1];
                g_band = lc3_r_band_gain(r, bi, cfg.nb);
            }
            vout[j] *= g_band;
        }
        {   // the group's four reconstructed lines leave as one 128-bit store
            lc3_f4 o;
            o.x = vout[0]; o.y = vout[1]; o.z = vout[2]; o.w = vout[3];
            *(lc3_f4 *)(c.plane + (LC3_PLANE_X + k0) * LC3_PLANE_STRIDE) = o;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) xw[j] = xw[j + 4];
#pragma unroll
        for (int j = 0; j < 4; j++) xw[8 + j] = xnext[j];
    }
    LC3_PSTAMP(c, 5);
#undef LC3_SIW
    return 1;
}
