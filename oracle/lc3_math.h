/* ORACLE (test infrastructure, not product code).
 *
 * Floating-point helpers for the CPU restatement of ninjasource/lc3-codec v0.2.0.
 *
 * The reference is a `#![no_std]` crate: every float method resolves through
 * `num_traits::real::Real` to the `libm` crate (Cargo.toml:17, e.g.
 * src/encoder/spectral_quantization.rs:8-9), a Rust port of musl's
 * FreeBSD-derived msun routines.  `libm` is NOT under /root/reference (crates.io
 * dependency, version unpinned: Cargo.lock is git-ignored), so the published
 * algorithms are restated here in C:
 *   powf   <- e_powf.c    (call sites: encoder/spectral_noise_shaping.rs:218,
 *                          encoder/spectral_quantization.rs:239, decoder/global_gain.rs:20)
 *   log2f  <- e_log2f.c   (encoder/spectral_noise_shaping.rs:232)
 *   exp2f  <- s_exp2f.c   (encoder/spectral_noise_shaping.rs:256)
 *   log10f <- e_log10f.c  (encoder/spectral_quantization.rs:218,393)
 *   asinf  <- e_asinf.c   (encoder/temporal_noise_shaping.rs:272)
 *   sinf   <- s_sinf.c + k_sinf.c/k_cosf.c (encoder/temporal_noise_shaping.rs:273,
 *                          decoder/temporal_noise_shaping.rs:44)
 * and `fast_math::exp2_raw` (fast-math 0.1.1, decoder/spectral_noise_shaping.rs:122).
 *
 * Pinning: powf is pinned by the reference's own goldens
 * (encoder/spectral_quantization.rs:474 gg = 24.7091141 = 0x41C5AC44, which is
 * NOT the correctly rounded value, and decoder/global_gain.rs:33-39);
 * exp2_raw by decoder/spectral_noise_shaping.rs:244-350.  The other five are
 * checked against the goldens produced by the reference's *test* build (which
 * used the host libm for typed receivers); for the shipped libm-crate flavour of
 * those five functions parity is unpinned.
 */
#ifndef LC3_ORACLE_MATH_H_
#define LC3_ORACLE_MATH_H_
#include <stdint.h>

float lc3m_powf(float x, float y);
float lc3m_log2f(float x);
float lc3m_exp2f(float x);
float lc3m_log10f(float x);
float lc3m_asinf(float x);
float lc3m_sinf(float x);
float lc3m_exp2_raw(float x);
/* num_traits pow(): square-and-multiply, `powi` for exp >= 0 (float.rs / pow.rs) */
float lc3m_powi(float base, int exp);

/* Rust `as` casts: saturating, NaN -> 0 */
int32_t lc3m_f32_to_i32(float x);
int16_t lc3m_f32_to_i16(float x);
int8_t lc3m_f32_to_i8(float x);
uint16_t lc3m_f32_to_u16(float x);
uint64_t lc3m_f64_to_usize(double x);

/* f32::max / f32::min semantics (NaN-ignoring) */
static inline float lc3m_maxf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
static inline float lc3m_minf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }

static inline float lc3m_from_bits(uint32_t u) {
    union { uint32_t u; float f; } c; c.u = u; return c.f;
}
static inline uint32_t lc3m_to_bits(float f) {
    union { uint32_t u; float f; } c; c.f = f; return c.u;
}
#endif
