// LC3 batched encoder for MI355X -- bitstream writer, ONE LANE PER FRAME.
//
// BitstreamEncoding::encode (reference encoder/bitstream_encoding.rs:77-429) with its BufferWriter
// (encoder/buffer_writer.rs:11-67) is a serial integer state machine: side information, a 24-bit range coder
// growing forward, sign/LSB/residual bits growing backward.  It needs nothing but integers the analysis stages
// produced, so -- like the decoder's parser (lc3_dev_dec_parse.h) -- it runs with every lane writing its own frame
// in the reference's exact operation order.  The wave-per-stream analysis kernel (lc3_dev_enc.h) leaves one
// "plane" column per frame in HBM (EP_WORDS contiguous words, lc3_dev_common.h); this stage reads it and produces the
// frame bytes in an LDS staging area that the workgroup then copies out with coalesced stores.
#pragma once
#include "lc3_dev_common.h"

// plane words of one frame (int32)
enum {
    EP_BW = 0, EP_NBITS_BW, EP_LASTNZ_TRUNC, EP_LSB_MODE, EP_GG_IND, EP_NUM_TNS, EP_ORD0, EP_ORD1, EP_LPC_W,
    EP_PITCH_PRESENT, EP_LTPF_ACTIVE, EP_PITCH_INDEX, EP_IND_LF, EP_IND_HF, EP_SHAPE_J, EP_GIND, EP_LS_INDA,
    EP_JOINT, EP_NOISE, EP_RATE_FLAG, EP_N_RES,
    EP_NSYM,                   // number of prepared spectral symbols in EP_SYM (-1: none, the packer derives them from EP_XQ)
    EP_NLSBS,                  // with prepared symbols: the number of LSB-list bits the spectral data implies (:298-312)
    EP_RCI,                    // 16 words: TNS coefficient indices
    EP_RES = EP_RCI + 16,      // 13 words: residual bits, bit j of word j / 32
    EP_XQ = EP_RES + 13,       // 200 words: quantised spectrum, x_q[2k] | x_q[2k+1] << 16
    EP_SYM = EP_XQ + 200,      // LC3_SYM_CAP words: the spectral data as the range coder will see it, one word per symbol (below)
    EP_WORDS = EP_SYM + 448
};
// A prepared symbol (lc3_enc_symbols, wave-parallel in the analysis kernel: what spectral_data :246-326 codes does not depend on
// the coder's state): model interval and the up to two bits that follow it backwards (escape: the pair's next bit plane; main
// symbol: the signs of its non-zero values).  Only launches of up to 16 384 frames prepare symbols: the analysis kernels of a full batch are
// VALU-bound and the work costs them more (+0.10 ms per 65 536 frames) than it saves the packer (-0.03 ms); a launch that does
// not fill the chip has the instruction slots to spare, and its packer -- a lane walking one frame, 0.14 ms whatever the launch
// size -- takes half as long.
#define LC3_SYM_CAP 448
#define LC3_SYM_WORD(cum, freq, nb, bits) ((uint32_t)(cum) | ((uint32_t)(freq) << 10) | ((uint32_t)(nb) << 20) | ((uint32_t)(bits) << 22))


struct lc3_pack_ctx {
    uint8_t *buf;            // this frame's nbytes output bytes (LDS staging)
    uint8_t *sink;           // a byte nobody reads: where the stores of out-of-range writes go (keeps the writers branch-free)
    int nbytes, nbits;
    const uint8_t *lookup;   // AC_SPEC_LOOKUP[4096]
    const uint32_t *cf;      // [64][17] cum | freq << 16
    const uint32_t *tns;     // packed TNS models (lc3_tns_model_word): [2][8] order models, then [8][17] coefficient models
    const int32_t *plane;    // word w at plane[w * stride]
    int stride;
    // BufferWriter (buffer_writer.rs:5-9) + ArithmeticEncoderState (bitstream_encoding.rs:27-34)
    int bp, bp_side, mask_side;
    uint32_t side_acc;       // register mirror of buf[bp_side] (written through): backward bits never read LDS
    uint32_t low, range;
    int cache, carry, carry_count;
#ifdef LC3_PROFILE
    unsigned long long plast, pt[8];  // diagnostic build: section stamps
#endif
};
#ifndef LC3_PSTAMP
#define LC3_PSTAMP(c, id)
#endif

__device__ __forceinline__ int32_t lc3_ep_get(const lc3_pack_ctx &c, int word) { return c.plane[word * c.stride]; }

// The byte the backward writer is filling is mirrored in w.side_acc and written through to LDS on every bit, so a bit
// costs no LDS read (a read-modify-write per bit was half of this kernel's time in memory waits).  The mirror stays exact:
// the staging buffer starts zero-filled, a byte the backward writer moves on to is still zero unless the forward writer
// has already passed it (then it is loaded), and a forward byte landing on the mirrored byte updates the mirror too.
// write_bool_backward (buffer_writer.rs:27-40) when `want`
__device__ __forceinline__ void lc3_pk_bool_backward_sel(lc3_pack_ctx &w, int want, int bit) {
    // bp_side starts at nbytes - 1 and only decreases: in range means not negative
    const uint32_t m = (uint32_t)w.mask_side;
    const uint32_t nw = bit ? (w.side_acc | m) : (w.side_acc & ~m);
    w.side_acc = want ? nw : w.side_acc;
    *(w.bp_side >= 0 ? w.buf + w.bp_side : w.sink) = (uint8_t)w.side_acc;  // without `want` this rewrites the byte with its own value
    const int wrap = want & (w.mask_side == 0x80);
    w.mask_side = want ? (wrap ? 1 : w.mask_side << 1) : w.mask_side;
    w.bp_side -= wrap;
    w.side_acc = wrap ? 0u : w.side_acc;
    if (wrap & (w.bp > w.bp_side)) w.side_acc = w.bp_side >= 0 ? (uint32_t)w.buf[w.bp_side] : 0u;  // rare: the writers crossed
}
__device__ __forceinline__ void lc3_pk_bool_backward(lc3_pack_ctx &w, int bit) { lc3_pk_bool_backward_sel(w, 1, bit); }
// Two consecutive write_bool_backward (:27-40), each when its `want`: the bits form one field (the first written one lowest) that is
// placed with a single multiplication by the mask; the mirrored byte is written through once.  At most one of the two bits can
// complete the byte; when it is the first one (mask 0x80, both wanted) the second bit opens the byte the writer moves on to.
__device__ __forceinline__ void lc3_pk_bool2_backward_sel(lc3_pack_ctx &w, int want0, int bit0, int want1, int bit1) {
    const uint32_t m = (uint32_t)w.mask_side;                           // 1 .. 0x80
    const uint32_t fm = (uint32_t)want0 + (uint32_t)want1 + (uint32_t)(want0 & want1);  // field of 0, 1 or 2 bits: 0, 1, 3
    const uint32_t v = want0 ? ((uint32_t)bit0 | ((uint32_t)bit1 << 1)) : (uint32_t)bit1;
    const uint32_t field = LC3_MUL24(fm, m), val = LC3_MUL24(v & fm, m);  // up to bit 8
    const uint32_t acc = (w.side_acc & ~field) | val;
    const int wrap = (int)((field >> 7) & 1u), spill = (int)(field >> 8);  // a bit landed on mask 0x80 / one bit lies beyond it
    const int at = w.bp_side;
    w.bp_side -= wrap;
    const int moved = (int)(m << (want0 + want1));                      // (a plain shift: as a product this came out as a 64-bit multiply-add in a branch)
    w.mask_side = wrap ? 1 + spill : moved;
    uint32_t next = 0u;                                                 // the byte the writer moves on to: zero ...
    // ... unless the writers crossed (rare).  The read comes BEFORE this call's stores: what follows a conditional LDS read waits for
    // every LDS operation in flight, and a store issued just before it would cost a full round trip on every call.
    if (wrap & (w.bp > w.bp_side)) next = w.bp_side >= 0 ? (uint32_t)w.buf[w.bp_side] : 0u;
    next = spill ? ((next & ~1u) | (acc >> 8)) : next;
    *(at >= 0 ? w.buf + at : w.sink) = (uint8_t)acc;                    // (without a wanted bit this rewrites the byte with its own value)
    *((spill && w.bp_side >= 0) ? w.buf + w.bp_side : w.sink) = (uint8_t)next;
    w.side_acc = wrap ? next : acc;
}
// write_uint_backward (:19-25): nbits (<= 32) bits of val, least significant first -- the reference's bit-by-bit loop done a
// byte at a time: the bits that land in the mirrored byte replace its field (set or cleared, as write_bool_backward does),
// the byte is written through, and a completed byte moves the cursor on.
__device__ __forceinline__ void lc3_pk_uint_backward(lc3_pack_ctx &w, uint32_t val, int nbits) {
    while (nbits > 0) {
        const int o = lc3_ilog2((uint32_t)w.mask_side);
        const int take = nbits < 8 - o ? nbits : 8 - o;
        const uint32_t field = ((1u << take) - 1u) << o;
        w.side_acc = (w.side_acc & ~field) | ((val << o) & field);
        *(w.bp_side >= 0 ? w.buf + w.bp_side : w.sink) = (uint8_t)w.side_acc;
        const int wrap = o + take == 8;
        w.mask_side = wrap ? 1 : w.mask_side << take;
        w.bp_side -= wrap;
        w.side_acc = wrap ? 0u : w.side_acc;
        if (wrap & (w.bp > w.bp_side)) w.side_acc = w.bp_side >= 0 ? (uint32_t)w.buf[w.bp_side] : 0u;  // rare: the writers crossed
        val >>= take;
        nbits -= take;
    }
}
__device__ __forceinline__ void lc3_pk_byte_forward(lc3_pack_ctx &w, int val) {  // :55-58
    *(w.bp < w.nbytes ? w.buf + w.bp : w.sink) = (uint8_t)val;  // bp starts at 0 and only grows
    w.side_acc = w.bp == w.bp_side ? (uint32_t)(val & 0xff) : w.side_acc;
    w.bp += 1;
}
__device__ __forceinline__ void lc3_pk_uint_forward(lc3_pack_ctx &w, unsigned val, int nbits) {  // :42-53
    unsigned mask = 0x80;
    for (int i = 0; i < nbits; i++) {
        if (w.bp >= 0 && w.bp < w.nbytes) {
            if (((val & 0xff) & mask) == 0) w.buf[w.bp] &= (uint8_t)~mask;
            else w.buf[w.bp] |= (uint8_t)mask;
        }
        mask >>= 1;
    }
}
__device__ __forceinline__ void lc3_pk_ac_shift(lc3_pack_ctx &w) {  // bitstream_encoding.rs:397-415
    if (w.low < 0x00ff0000u || w.carry == 1) {
        if (w.cache >= 0) lc3_pk_byte_forward(w, (w.cache + w.carry) & 0xff);
        while (w.carry_count > 0) {
            asm volatile("" ::: "memory");  // a plain loop, not a vectorised one (see lc3_pk_ac_shift_sel)
            lc3_pk_byte_forward(w, (w.carry + 0xff) & 0xff);
            w.carry_count -= 1;
        }
        w.cache = (int)(w.low >> 16);
        w.carry = 0;
    } else w.carry_count += 1;
    w.low <<= 8;
    w.low &= 0x00ffffffu;
}
__device__ __forceinline__ void lc3_pk_ac_encode(lc3_pack_ctx &w, uint32_t cum_freq, uint32_t sym_freq) {  // :417-429
    const uint32_t r = w.range >> 10;
    w.low += LC3_MUL24(r, cum_freq);  // r < 2^14, frequencies <= 2^10
    if ((w.low >> 24) != 0) w.carry = 1;
    w.low &= 0x00ffffffu;
    w.range = LC3_MUL24(r, sym_freq);
    while (w.range < 0x10000u) {
        w.range <<= 8;
        lc3_pk_ac_shift(w);
    }
}

// ---- select-based variants for the spectral loop: every lane is another frame, a branch on frame data diverges and
// costs more scalar bookkeeping than the few operations it skips.
// ac_shift (bitstream_encoding.rs:397-415) when `need`
__device__ __forceinline__ void lc3_pk_ac_shift_sel(lc3_pack_ctx &w, int need) {
    const int flush = need & ((w.low < 0x00ff0000u) | (w.carry == 1));
    {   // the cached byte goes out (to the sink when there is nothing to write)
        const int put = flush & (w.cache >= 0);
        *((put & (w.bp < w.nbytes)) ? w.buf + w.bp : w.sink) = (uint8_t)(w.cache + w.carry);
        w.side_acc = (put & (w.bp == w.bp_side)) ? (uint32_t)((w.cache + w.carry) & 0xff) : w.side_acc;
        w.bp += put;
    }
    while (flush && w.carry_count > 0) {  // rare: a run of 0xff bytes was waiting for the carry
        asm volatile("" ::: "memory");    // keeps this a plain loop (vectorised and unrolled it was a quarter of the loop's code)
        lc3_pk_byte_forward(w, (w.carry + 0xff) & 0xff);
        w.carry_count -= 1;
    }
    w.cache = flush ? (int)(w.low >> 16) : w.cache;
    w.carry = flush ? 0 : w.carry;
    w.carry_count += need & !flush;
    w.low = need ? (w.low << 8) & 0x00ffffffu : w.low;
}
// ac_encode (:417-429).  After range = r * sym_freq the range is at least 64 (r >= 64 because range >= 2^16 on entry,
// sym_freq >= 1), so the reference's renormalisation loop runs at most twice.
__device__ __forceinline__ void lc3_pk_ac_encode_sel(lc3_pack_ctx &w, uint32_t cum_freq, uint32_t sym_freq) {
    const uint32_t r = w.range >> 10;
    w.low += LC3_MUL24(r, cum_freq);
    w.carry = (w.low >> 24) != 0 ? 1 : w.carry;
    w.low &= 0x00ffffffu;
    w.range = LC3_MUL24(r, sym_freq);
    {
        const int need = w.range < 0x10000u;
        w.range = need ? w.range << 8 : w.range;
        lc3_pk_ac_shift_sel(w, need);
    }
    if (w.range < 0x10000u) {  // a second byte only after a symbol of probability below 2^-8: usually no lane of the wave
        w.range <<= 8;
        lc3_pk_ac_shift_sel(w, 1);
    }
}

// one symbol of spectral_data (:262-296) as a function of the quantised pair and the lane's position in it
struct lc3_pk_sym {
    int q0, q1;       // the pair
    unsigned a0, b0;  // its magnitudes
    unsigned a, b;    // ... at the current escape level
    int esc;          // the symbol is an escape (magnitudes >= 4 at this level)
    int lv;           // min(level, 3)
    int idx;          // index of the symbol's context in the lookup table
};
__device__ __forceinline__ lc3_pk_sym lc3_pk_symbol(uint32_t xw, int lev, int cctx, int tup, int rate_flag, int ne) {
    lc3_pk_sym s;
    s.q0 = (int)(int16_t)(xw & 0xffffu);
    s.q1 = (int)(int16_t)(xw >> 16);
    s.a0 = (unsigned)(s.q0 < 0 ? -s.q0 : s.q0);
    s.b0 = (unsigned)(s.q1 < 0 ? -s.q1 : s.q1);
    s.a = s.a0 >> lev;
    s.b = s.b0 >> lev;
    s.esc = (s.a > s.b ? s.a : s.b) >= 4u;
    s.lv = lev < 3 ? lev : 3;
    s.idx = cctx + rate_flag + (2 * tup > ne / 2 ? 256 : 0) + s.lv * 1024;
    return s;
}

// the column's scalar words, TNS indices and residual bit words
struct lc3_pack_head {
    int32_t sw[EP_RES];
    uint32_t rw[13];  // residual bit words (used when the frame is not in LSB mode)
};
#define LC3_EPW(word) h.sw[word]
// the 21 scalar words and the 16 TNS indices of the column: one batch of independent loads (a lane of this kernel
// is latency-bound; every plane word fetched at its point of use would cost a full memory round trip)
__device__ __forceinline__ void lc3_pack_load_head(const lc3_pack_ctx &w, lc3_pack_head &h) {
#pragma unroll
    for (int i = 0; i < EP_RES; i++) h.sw[i] = lc3_ep_get(w, i);
#pragma unroll
    for (int i = 0; i < 13; i++) h.rw[i] = (uint32_t)lc3_ep_get(w, EP_RES + i);
}
// BitstreamEncoding::encode :77-136 up to the spectral data: side information, ac_enc_init, tns_data
__device__ __forceinline__ void lc3_pack_begin(lc3_pack_ctx &w, int ne, const lc3_pack_head &h) {
    w.nbits = w.nbytes * 8;
    w.bp = 0;
    w.bp_side = w.nbytes - 1;
    w.mask_side = 1;
    w.side_acc = 0;
    const int lsb_mode = LC3_EPW(EP_LSB_MODE), lastnz_trunc = LC3_EPW(EP_LASTNZ_TRUNC);
    const int num_tns = LC3_EPW(EP_NUM_TNS), rate_flag = LC3_EPW(EP_RATE_FLAG);
    const int ord0 = LC3_EPW(EP_ORD0), ord1 = LC3_EPW(EP_ORD1);
    // side information :92-112 (layout: SURVEY App. E)
    {
        const int nbits_bw = LC3_EPW(EP_NBITS_BW);
        if (nbits_bw > 0) lc3_pk_uint_backward(w, (uint32_t)LC3_EPW(EP_BW), nbits_bw);
        int nb = 0;
        while ((1 << nb) < ne / 2) nb++;
        lc3_pk_uint_backward(w, (uint32_t)((lastnz_trunc >> 1) - 1), nb);
        lc3_pk_bool_backward(w, lsb_mode);
        lc3_pk_uint_backward(w, (uint32_t)LC3_EPW(EP_GG_IND), 8);
        if (num_tns > 0) lc3_pk_bool_backward(w, ord0 != 0);
        if (num_tns > 1) lc3_pk_bool_backward(w, ord1 != 0);
        const int pitch_present = LC3_EPW(EP_PITCH_PRESENT);
        lc3_pk_bool_backward(w, pitch_present);
        lc3_pk_uint_backward(w, (uint32_t)LC3_EPW(EP_IND_LF), 5);
        lc3_pk_uint_backward(w, (uint32_t)LC3_EPW(EP_IND_HF), 5);
        const int shape_j = LC3_EPW(EP_SHAPE_J);
        const uint32_t joint = (uint32_t)LC3_EPW(EP_JOINT);
        const int submode_msb = (shape_j >> 1) != 0;
        lc3_pk_bool_backward(w, submode_msb);
        lc3_pk_uint_backward(w, (uint32_t)(LC3_EPW(EP_GIND) >> LC3T_SNS_GAIN_LSB_BITS[shape_j]),
                             LC3T_SNS_GAIN_MSB_BITS[shape_j]);
        lc3_pk_bool_backward(w, LC3_EPW(EP_LS_INDA) != 0);
        if (!submode_msb) {
            lc3_pk_uint_backward(w, joint, 13);
            lc3_pk_uint_backward(w, joint >> 13, 12);
        } else {
            lc3_pk_uint_backward(w, joint, 12);
            lc3_pk_uint_backward(w, joint >> 12, 12);
        }
        if (pitch_present) {
            lc3_pk_bool_backward(w, LC3_EPW(EP_LTPF_ACTIVE));
            lc3_pk_uint_backward(w, (uint32_t)LC3_EPW(EP_PITCH_INDEX), 9);
        }
        lc3_pk_uint_backward(w, (uint32_t)LC3_EPW(EP_NOISE), 3);
    }
    LC3_PSTAMP(w, 1);
    // ac_enc_init :216-222
    w.low = 0;
    w.range = 0x00ffffffu;
    w.cache = -1;
    w.carry = 0;
    w.carry_count = 0;
    // tns_data :224-244
    {
        const int wt = LC3_EPW(EP_LPC_W);
        for (int f = 0; f < num_tns; f++) {
            const int order = f == 0 ? ord0 : ord1;
            if (order > 0) {
                const uint32_t so = w.tns[wt * 8 + order - 1];
                lc3_pk_ac_encode(w, so & 0xffffu, so >> 16);
                for (int k = 0; k < order; k++) {
                    int ri = 0;
#pragma unroll
                    for (int q = 0; q < 16; q++)
                        if (q == k + 8 * f) ri = h.sw[EP_RCI + q];
                    ri = ri < 0 ? 0 : (ri > 16 ? 16 : ri);
                    const uint32_t sc = w.tns[16 + k * 17 + ri];
                    lc3_pk_ac_encode(w, sc & 0xffffu, sc >> 16);
                }
            }
        }
    }
}
// ... and behind it: residual_data_and_finalization :328-352, ac_enc_finish :354-395.  nlsbs: the number of LSB-list bits the spectral
// data implies (:298-312)
__device__ __forceinline__ void lc3_pack_end(lc3_pack_ctx &w, int ne, const lc3_pack_head &h, int nlsbs) {
    const int lsb_mode = LC3_EPW(EP_LSB_MODE), lastnz_trunc = LC3_EPW(EP_LASTNZ_TRUNC);
    LC3_PSTAMP(w, 3);
    // residual_data_and_finalization :328-352
    {
        const int nbits_side = w.nbits - (8 * w.bp_side + 8 - lc3_ilog2((uint32_t)w.mask_side));
        int nbits_ari = w.bp * 8 + 25 - lc3_ilog2(w.range);  // nbits_side_forcast :64-75
        if (w.carry >= 0) nbits_ari += 8;
        if (w.carry_count > 0) nbits_ari += w.carry_count * 8;
        int n_enc = w.nbits - (nbits_side + nbits_ari);
        if (n_enc < 0) n_enc = 0;
        if (!lsb_mode) {
            const int n_res = LC3_EPW(EP_N_RES);
            if (n_enc > n_res) n_enc = n_res;
#pragma unroll
            for (int i = 0; i < 13; i++) {  // the words were fetched with the column's scalars
                const int k = 32 * i;
                if (k < n_enc) lc3_pk_uint_backward(w, h.rw[i], n_enc - k < 32 ? n_enc - k : 32);
            }
        } else {
            // lsbs[0 .. nlsbs) in the order spectral_data pushed them (:298-312), regenerated on the fly; the pairs are
            // fetched eight at a time, and a pair's two to four bits (LSB, sign of a value that became 0 >> 1, LSB, sign) go
            // out as one backward write
            if (n_enc > nlsbs) n_enc = nlsbs;
            int written = 0;
            const int ntup = lastnz_trunc / 2;
            for (int tup0 = 0; tup0 < ntup && written < n_enc; tup0 += 8) {
                uint32_t xq8[8];
#pragma unroll
                for (int j = 0; j < 8; j++) xq8[j] = tup0 + j < ne / 2 ? (uint32_t)lc3_ep_get(w, EP_XQ + tup0 + j) : 0u;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int q0 = (int)(int16_t)(xq8[j] & 0xffffu), q1 = (int)(int16_t)(xq8[j] >> 16);
                    const unsigned a = (unsigned)(q0 < 0 ? -q0 : q0), b = (unsigned)(q1 < 0 ? -q1 : q1);
                    if (tup0 + j < ntup && written < n_enc && (a > b ? a : b) >= 4) {
                        uint32_t ev = a & 1u;
                        int ne_ = 1;
                        if ((a >> 1) == 0 && q0 != 0) { ev |= (q0 > 0 ? 0u : 1u) << ne_; ne_++; }
                        ev |= (b & 1u) << ne_;
                        ne_++;
                        if ((b >> 1) == 0 && q1 != 0) { ev |= (q1 > 0 ? 0u : 1u) << ne_; ne_++; }
                        const int m = ne_ < n_enc - written ? ne_ : n_enc - written;
                        lc3_pk_uint_backward(w, ev, m);
                        written += m;
                    }
                }
            }
        }
    }
    LC3_PSTAMP(w, 4);
    // ac_enc_finish :354-395
    {
        int bits = 1;
        while (bits < 24 && (w.range >> (24 - bits)) == 0) bits++;  // (range >= 64 here; the bound only keeps a damaged plane from spinning)
        uint32_t mask = 0x00ffffffu >> bits;
        uint32_t val = w.low + mask;
        const uint32_t over1 = val >> 24;
        const uint32_t high = w.low + w.range;
        const uint32_t over2 = high >> 24;
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
            lc3_pk_ac_shift(w);
            bits -= 8;
        }
        bits += 8;
        if (w.carry_count > 0) {
            lc3_pk_byte_forward(w, w.cache & 0xff);
            while (w.carry_count > 1) {
                lc3_pk_byte_forward(w, 0xff);
                w.carry_count -= 1;
            }
            lc3_pk_uint_forward(w, 0xffu >> (8 - bits), bits);
        } else {
            lc3_pk_uint_forward(w, (unsigned)w.cache, bits);
        }
    }
    LC3_PSTAMP(w, 5);
}

// BitstreamEncoding::encode :77-136; the buffer must be zero-filled (init :138-144)
__device__ __forceinline__ void lc3_pack_frame(lc3_pack_ctx &w, int ne) {
    lc3_pack_head h;
    lc3_pack_load_head(w, h);
    lc3_pack_begin(w, ne, h);
    const int lsb_mode = LC3_EPW(EP_LSB_MODE), lastnz_trunc = LC3_EPW(EP_LASTNZ_TRUNC), rate_flag = LC3_EPW(EP_RATE_FLAG);
    LC3_PSTAMP(w, 2);
    // spectral_data :246-326
    int nlsbs = 0;
    const int nsym = LC3_EPW(EP_NSYM);
    if (LC3_WAVE_ANY(nsym >= 0)) {
        // prepared symbols (small launches): interval, then its bits; the word three symbols ahead is requested every iteration
        const int n = nsym > 0 ? nsym : 0, last = LC3_SYM_CAP - 1;
        uint32_t s0 = (uint32_t)lc3_ep_get(w, EP_SYM), s1 = (uint32_t)lc3_ep_get(w, EP_SYM + 1), s2 = (uint32_t)lc3_ep_get(w, EP_SYM + 2),
                 s3 = (uint32_t)lc3_ep_get(w, EP_SYM + 3);
        for (int i = 0; LC3_WAVE_ANY(i < n); i++) {
            if (i < n) {
                lc3_pk_ac_encode_sel(w, s0 & 0x3ffu, (s0 >> 10) & 0x3ffu);
                const int nb = (int)((s0 >> 20) & 3u);
                lc3_pk_bool2_backward_sel(w, nb > 0, (int)((s0 >> 22) & 1u), nb > 1, (int)((s0 >> 23) & 1u));
            }
            s0 = s1;
            s1 = s2;
            s2 = s3;
            s3 = (uint32_t)lc3_ep_get(w, EP_SYM + (i + 4 < last ? i + 4 : last));
        }
        nlsbs = nsym >= 0 ? LC3_EPW(EP_NLSBS) : 0;
    }
    if (LC3_WAVE_ANY(nsym < 0) && nsym < 0) {

        // One symbol per iteration and lane: every lane walks its own frame's symbol sequence (escape symbols of a pair,
        // then its main symbol) and moves on to its next pair by itself.  With a common pair index the wave spends
        // sum over pairs of (1 + deepest escape level of any lane) iterations, here max over lanes of (sum over pairs of
        // 1 + level): 1.45x fewer on the benchmark's frames.  The pair after next-but-one is requested every iteration.
        const int ntup = lastnz_trunc / 2, last = ne / 2 - 1;
        int tup = 0, lev = 0, cctx = 0;
        uint32_t xw = (uint32_t)lc3_ep_get(w, EP_XQ), x1 = (uint32_t)lc3_ep_get(w, EP_XQ + (1 < last ? 1 : last)),
                 x2 = (uint32_t)lc3_ep_get(w, EP_XQ + (2 < last ? 2 : last)), x3 = (uint32_t)lc3_ep_get(w, EP_XQ + (3 < last ? 3 : last));
        // The symbols depend on the quantised values only, not on the coder's state (:262-296): what the coming iteration
        // encodes (and its model word) is worked out during the iteration before it and carried over.
        lc3_pk_sym cur = lc3_pk_symbol(xw, 0, 0, 0, rate_flag, ne);
        uint32_t sv = w.cf[(int)w.lookup[cur.idx] * 17 + (cur.esc ? 16 : (int)(cur.a + 4u * cur.b))];
        while (tup < ntup) {
            // `cur`: an escape symbol (and two LSBs), or the pair's main symbol (and its signs).  Where the lane will be
            // after it, and that symbol's model row:
            const int adv = !cur.esc;
            const int n_cctx = adv ? (cctx & 15) * 16 + (cur.lv <= 1 ? 1 + (int)((cur.a + cur.b) << cur.lv) : 12 + cur.lv) : cctx;
            const int n_lev = adv ? 0 : lev + 1, n_tup = tup + adv;
            const uint32_t n_xw = adv ? x1 : xw;
            const lc3_pk_sym nxt = lc3_pk_symbol(n_xw, n_lev, n_cctx, n_tup, rate_flag, ne);
            const int rown = (int)w.lookup[nxt.idx];
            // this symbol
            lc3_pk_ac_encode_sel(w, sv & 0xffffu, sv >> 16);
            const int lsb_here = lsb_mode && lev > 0;
            const unsigned a_l = lsb_here ? cur.a0 >> 1 : cur.a0, b_l = lsb_here ? cur.b0 >> 1 : cur.b0;
            const int want_e = !(lsb_mode && lev == 0);
            lc3_pk_bool2_backward_sel(w, cur.esc ? want_e : a_l > 0u, cur.esc ? (cur.a & 1u) == 1u : cur.q0 <= 0,
                                      cur.esc ? want_e : b_l > 0u, cur.esc ? (cur.b & 1u) == 1u : cur.q1 <= 0);
            // the LSB list itself is regenerated below when it is written
            nlsbs += (!cur.esc && lsb_here) ? 2 + (a_l == 0u && cur.q0 != 0) + (b_l == 0u && cur.q1 != 0) : 0;
            sv = w.cf[rown * 17 + (nxt.esc ? 16 : (int)(nxt.a + 4u * nxt.b))];
            cur = nxt;
            cctx = n_cctx;
            lev = n_lev;
            tup = n_tup;
            xw = n_xw;
            x1 = adv ? x2 : x1;
            x2 = adv ? x3 : x2;
            x3 = (uint32_t)lc3_ep_get(w, EP_XQ + (tup + 3 < last ? tup + 3 : last));  // the same word again unless the lane advanced
        }
    }
    lc3_pack_end(w, ne, h, nlsbs);
}
#undef LC3_EPW

// ------------------------------------------------------------------------------------------------------------------
// The packer of a full batch as a PRODUCER / CONSUMER pair of waves (lc3_pack_pc_kernel, lc3gpu.hip; the link: lc3_dev_common.h).
// What spectral_data (:246-326) codes does not depend on the coder's state: the producer walks the frame's quantised pairs and
// leaves one word per symbol -- the model interval and the up to two bits that follow it backwards (the format of the prepared
// symbols, LC3_SYM_WORD, plus a "this lane has a symbol" bit) -- in the ring; the consumer owns the frame: side information, range
// coder and both writers, in the reference's order of writes.  A lone wave of the one-wave form issues ~190 instructions per symbol at
// one per four cycles; here two waves of one SIMD share them.
// ------------------------------------------------------------------------------------------------------------------
#define LC3_PK_SYM_VALID 0x80000000u
// fin words of the link: [0] the number of LSB-list bits (:298-312)
// Both halves return non-zero when they gave up waiting for their partner (LC3_PC_SPIN_LIMIT polls: a partner that died); the kernel
// counts those (lc3gpu_encoder_pair_timeouts) and the consumer leaves such a frame ZERO-FILLED rather than half written.
__device__ __forceinline__ int lc3_pack_produce(const lc3_pack_ctx &w, const lc3_pc_link &k, int ne, int valid) {
    const int lsb_mode = lc3_ep_get(w, EP_LSB_MODE), lastnz_trunc = lc3_ep_get(w, EP_LASTNZ_TRUNC), rate_flag = lc3_ep_get(w, EP_RATE_FLAG);
    const int ntup = valid ? lastnz_trunc / 2 : 0, last = ne / 2 - 1;
    int tup = 0, lev = 0, cctx = 0, nlsbs = 0, it = 0, c_seen = 0, spins = 0;
    uint32_t xw = (uint32_t)lc3_ep_get(w, EP_XQ), x1 = (uint32_t)lc3_ep_get(w, EP_XQ + (1 < last ? 1 : last)),
             x2 = (uint32_t)lc3_ep_get(w, EP_XQ + (2 < last ? 2 : last)), x3 = (uint32_t)lc3_ep_get(w, EP_XQ + (3 < last ? 3 : last));
    lc3_pk_sym cur = lc3_pk_symbol(xw, 0, 0, 0, rate_flag, ne);
    uint32_t sv = w.cf[(int)w.lookup[cur.idx] * 17 + (cur.esc ? 16 : (int)(cur.a + 4u * cur.b))];
    LC3_PC_STORE(k.p_count, 0);
    // whole chunks of LC3_PC_CHUNK iterations, straight-line (see lc3_pc_produce, lc3_dev_dec_parse.h)
    while (LC3_WAVE_ANY(tup < ntup)) {
#pragma unroll
        for (int u = 0; u < LC3_PC_CHUNK; u++) {
            uint32_t word = 0u;
            if (tup < ntup) {
                // where the lane will be after this symbol, and that symbol's model row (see lc3_pack_frame)
                const int adv = !cur.esc;
                const int n_cctx = adv ? (cctx & 15) * 16 + (cur.lv <= 1 ? 1 + (int)((cur.a + cur.b) << cur.lv) : 12 + cur.lv) : cctx;
                const int n_lev = adv ? 0 : lev + 1, n_tup = tup + adv;
                const uint32_t n_xw = adv ? x1 : xw;
                const lc3_pk_sym nxt = lc3_pk_symbol(n_xw, n_lev, n_cctx, n_tup, rate_flag, ne);
                const int rown = (int)w.lookup[nxt.idx];
                // this symbol: interval, then the bits that follow it (escape: the pair's next bit plane; main symbol: the signs)
                const int lsb_here = lsb_mode && lev > 0;
                const unsigned a_l = lsb_here ? cur.a0 >> 1 : cur.a0, b_l = lsb_here ? cur.b0 >> 1 : cur.b0;
                const int want_e = !(lsb_mode && lev == 0);
                const int w0 = cur.esc ? want_e : a_l > 0u, w1 = cur.esc ? want_e : b_l > 0u;
                const uint32_t b0 = cur.esc ? (cur.a & 1u) : (uint32_t)(cur.q0 <= 0), b1 = cur.esc ? (cur.b & 1u) : (uint32_t)(cur.q1 <= 0);
                const uint32_t bits = w0 ? (b0 | (b1 << 1)) : b1;  // the first wanted bit lowest
                word = LC3_PK_SYM_VALID | LC3_SYM_WORD(sv & 0xffffu, sv >> 16, w0 + w1, bits & 3u);
                // (the LSB list itself is regenerated by lc3_pack_end when it is written)
                nlsbs += (!cur.esc && lsb_here) ? 2 + (a_l == 0u && cur.q0 != 0) + (b_l == 0u && cur.q1 != 0) : 0;
                sv = w.cf[rown * 17 + (nxt.esc ? 16 : (int)(nxt.a + 4u * nxt.b))];
                cur = nxt;
                cctx = n_cctx;
                lev = n_lev;
                tup = n_tup;
                xw = n_xw;
                x1 = adv ? x2 : x1;
                x2 = adv ? x3 : x2;
                x3 = (uint32_t)lc3_ep_get(w, EP_XQ + (tup + 3 < last ? tup + 3 : last));  // the same word again unless the lane advanced
            }
            k.ring[((it + u) & k.mask) * k.stride] = word;  // every lane: a lane that has finished its frame says so
        }
        it += LC3_PC_CHUNK;
        LC3_PC_STORE(k.p_count, it);
        while (it + LC3_PC_CHUNK - c_seen > k.mask + 1 && spins < LC3_PC_SPIN_LIMIT) {
            c_seen = LC3_PC_LOAD(k.c_count);
            if (it + LC3_PC_CHUNK - c_seen > k.mask + 1) {
                LC3_PC_PAUSE();
                spins++;
            }
        }
    }
    k.fin[0] = (uint32_t)nlsbs;
    LC3_PC_STORE(k.p_count, it | LC3_PC_DONE);
    return spins >= LC3_PC_SPIN_LIMIT;
}

__device__ __forceinline__ int lc3_pack_consume(lc3_pack_ctx &w, const lc3_pc_link &k, int ne, int valid) {
    lc3_pack_head h;
    lc3_pack_load_head(w, h);
    if (valid) lc3_pack_begin(w, ne, h);
    int it = 0, spins = 0, pc = 0;
    for (;;) {  // whole chunks; the producer's count is whole chunks
        while ((((pc = LC3_PC_LOAD(k.p_count)) & (LC3_PC_DONE - 1)) <= it || pc < 0) && !(pc >= 0 && (pc & LC3_PC_DONE)) && spins < LC3_PC_SPIN_LIMIT) {
            LC3_PC_PAUSE();
            spins++;
        }
        if (pc < 0 || (pc & (LC3_PC_DONE - 1)) <= it) break;  // the producer has finished (or never answered)
#pragma unroll
        for (int u = 0; u < LC3_PC_CHUNK; u++) {
            const uint32_t s0 = k.ring[((it + u) & k.mask) * k.stride];
            if (s0 & LC3_PK_SYM_VALID) {
                lc3_pk_ac_encode_sel(w, s0 & 0x3ffu, (s0 >> 10) & 0x3ffu);
                const int nb = (int)((s0 >> 20) & 3u);
                lc3_pk_bool2_backward_sel(w, nb > 0, (int)((s0 >> 22) & 1u), nb > 1, (int)((s0 >> 23) & 1u));
            }
        }
        it += LC3_PC_CHUNK;
        LC3_PC_STORE(k.c_count, it);
    }
    LC3_PC_STORE(k.c_count, LC3_PC_DONE - 1);
    const int timed_out = spins >= LC3_PC_SPIN_LIMIT;  // (the wave's, not the lane's: the counts are per wave)
    const int nlsbs = (int)k.fin[0];
    if (valid && !timed_out) lc3_pack_end(w, ne, h, nlsbs);
    if (valid && timed_out)  // the symbols never came: no frame rather than a truncated one
        for (int i = 0; i < w.nbytes; i++) w.buf[i] = 0;
    return timed_out;
}
