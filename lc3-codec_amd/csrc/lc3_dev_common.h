// LC3 batched codec for MI355X (gfx950) -- device-side common layer.
//
// Execution model of the stream kernels: ONE WAVEFRONT (64 lanes) PER STREAM, four streams per workgroup.  A
// stream is one codec channel (the reference's EncoderChannel / DecoderChannel, encoder/lc3_encoder.rs:42-60,
// decoder/lc3_decoder.rs:62-69); its frames are processed in time order by the same wave with the working data in
// LDS.  Every stage routine takes the caller's lane id and is executed by all 64 lanes of the wave; lane-parallel
// loops stride by 64, inherently serial recurrences run on lane 0 (or a few lanes) while the others wait at the next
// LC3_SYNC().  Stages that are serial per frame AND stateless across frames do not live here at all: they run one
// lane per frame in lc3_dev_enc_pack.h / lc3_dev_dec_parse.h.
//
// Bit-exactness contract (SURVEY.md section 7 "design rule"): parallelise across
// OUTPUTS, never inside one of the reference's f32 summations; every f32
// expression keeps the reference's evaluation order; build with
// -ffp-contract=off so no mul+add is fused.
//
// The including translation unit provides LC3_SYNC() (orders one wave's LDS traffic), the wave primitives
// (lc3_wave_max_i32, lc3_wave_sum_u32, lc3_wave_exscan_u32, lc3_wave_ballot) and the __device__ / __forceinline__ keywords: lc3gpu.hip
// for the GPU, tests/emu/lc3_emu.cpp for the CPU wave emulator of the tests.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "lc3_dev_experiments.h"

#define LC3_TABLE_QUAL static __device__ const
#include "../../tables/lc3_tables.h"

// Optional in-kernel stage stamps (diagnostic build only, -DLC3_PROFILE; see lc3gpu.hip).  The production
// build compiles them away.
#ifndef LC3_STAMP
#define LC3_STAMP(L, lane, id)
#endif

// How the stage functions receive the codec configuration.  Default: by reference.  The HIP translation unit passes
// a slot number into a __constant__ table instead (lc3gpu.hip), so that the fields are scalar loads, and makes the stage
// functions templates over a "configuration view" (LC3_CFG_TEMPLATE): the run-time view reads the table, the view of the
// headline configuration (48 kHz / 10 ms) has its integers as compile-time constants.  Helpers that take the
// configuration as an argument are templates over its type for the same reason.
#ifndef LC3_CFG_PARAM
#define LC3_CFG_TEMPLATE
#define LC3_CFG_TEMPLATE_AND(...) template <__VA_ARGS__>  // a stage function with template parameters of its own
#define LC3_CFG_PARAM const lc3_cfg &c
#define LC3_CFG_BIND
#define LC3_CFG_PASS c
#endif

// How the stage functions reach their stream's LDS working set.  Default: a reference parameter.  The HIP translation
// unit keeps the working sets of a workgroup in a file-scope __shared__ array (declared by LC3_LDS_DECL right after the
// struct) and binds `L` to the calling wave's element inside every function, so that the compiler addresses LDS
// directly (ds_read / ds_write) instead of through generic pointers.
#ifndef LC3_LDS_PARAM
#define LC3_LDS_DECL(T, arr)
#define LC3_LDS_PARAM(T) T &L,
#define LC3_LDS_PASS L,
#define LC3_LDS_BIND(T, arr)
#endif

// The spectral-model tables of the quantiser's bit estimate (AC_SPEC_LOOKUP[4096], AC_SPEC_BITS[64][17]): read
// straight from the constant tables by default; the HIP kernels keep a per-workgroup copy in LDS (data-dependent
// gathers, two dependent lookups per tuple).
#ifndef LC3_SPEC_LOOKUP
#define LC3_SPEC_LOOKUP(i) ((int)LC3T_AC_SPEC_LOOKUP[(i)])
#define LC3_SPEC_BITS(p, j) ((uint32_t)LC3T_AC_SPEC_BITS[(p)][(j)])
#endif

// Code that is serial per stream sits between LC3_SERIAL_BEGIN(T, L, lane, phase, K) and LC3_SERIAL_END: K lanes
// (`sub` = 0..K-1) run it per stream with `L` bound to the stream's working set of type T.  The translation unit decides
// who runs it: by default lanes 0..K-1 of the stream's own wave; the HIP kernels gather the streams of a workgroup on
// one wave (lc3gpu.hip).  Such a block may only use L, launch-uniform values and what it declares itself.
#ifndef LC3_SERIAL_BEGIN
#define LC3_SERIAL_BEGIN(T, L, lane, phase, K) \
    {                                          \
        LC3_SYNC();                            \
        if ((lane) < (K)) {                    \
            const int sub = (lane);            \
            (void)sub;
#define LC3_SERIAL_END \
        }              \
        LC3_SYNC();    \
    }
#endif
#ifndef LC3_SERIAL_WIDE_BEGIN  // the same for K up to 64 lanes per stream (the HIP kernels spread 4 K lanes over more than one wave)
#define LC3_SERIAL_WIDE_BEGIN(T, L, lane, phase, K) LC3_SERIAL_BEGIN(T, L, lane, phase, K)
#endif

// The same bracket without the gathering: lanes 0..K-1 of the stream's own wave run the block (for serial work whose
// duration differs a lot from stream to stream, where waiting for the slowest stream of the workgroup costs more than
// the issue slots the gathering saves).
#define LC3_LOCAL_BEGIN(lane, K)    \
    {                               \
        LC3_SYNC();                 \
        if ((lane) < (K)) {         \
            const int sub = (lane); \
            (void)sub;
#define LC3_LOCAL_END \
        }             \
        LC3_SYNC();   \
    }

// Pointer to read-only HBM data handed to a stage function: the HIP translation unit marks the address space so that
// the loads are global_load rather than flat_load.
// a value the caller knows to be the same on every lane of the wave (the device build moves it to a scalar register)
#ifndef LC3_KEEP_SCALAR  // lc3gpu.hip: keeps the compiler from merging a value's operation with a neighbour's into a packed one
#define LC3_KEEP_SCALAR(x) ((void)0)
#endif
#ifndef LC3_UNIFORM_I32
#define LC3_UNIFORM_I32(x) (x)
#endif
#ifndef LC3_KEEP_PER_FRAME  // lc3gpu.hip: an opaque barrier that keeps what is computed from x inside the loop it is written in
#define LC3_KEEP_PER_FRAME(x) ((void)0)
#endif
#ifndef LC3_UNIFORM_PTR
#define LC3_UNIFORM_PTR(T, p) ((T)(p))
#endif
#ifndef LC3_LDS_BASE
#define LC3_LDS_BASE(p) (p)  // device build: the LDS address as an opaque base register (lc3gpu.hip); for arrays whose alignment is
                            // unknown anyway (the compiler forgets what it knew about the pointer)
#endif
#ifndef LC3_LDS_BASE_ALIGNED
#define LC3_LDS_BASE_ALIGNED(p, bytes) (p)
#endif
#ifndef LC3_HBM_CONST
#define LC3_HBM_CONST(T) const T *
#define LC3_HBM(T) T *
#endif

#define LC3_WAVE 64
// does the predicate hold on any lane of the (lane-per-frame) wave?  Translation units that run frames one at a time: the lane's own.
#ifndef LC3_WAVE_ANY
#define LC3_WAVE_ANY(pred) ((pred) != 0)
#endif
// the largest value any lane of the (lane-per-frame) wave holds, wave-uniform.  Translation units that run frames one at a time: the lane's own.
#ifndef LC3_LANEWAVE_MAX
#define LC3_LANEWAVE_MAX(v) (v)
#endif

// Opt-in corrections of the reference's deviations from the LC3 specification, one bit each; 0 = the
// reference's behaviour, which is what every parity claim is about.  Same values as LC3GPU_SPEC_* (include/lc3gpu.h).
#define LC3_SPEC_8KHZ_ENCODE 1     // A6: an 8 kHz encoder exists (bandwidth detector returns at once, bandwidth_detector.rs:66-71)
#define LC3_SPEC_TNS_SSWB_STOP 2   // A5: 10 ms, bandwidth index 2: TNS filter stops at line 240 instead of 200
#define LC3_SPEC_BW_CUTOFF_DB 4    // A7: cut-off test on 10 log10(eps + E[n-L] / E[n]) (the note at bandwidth_detector.rs:106-107)
#define LC3_SPEC_SNS_LAST_GAIN 8   // A3: the SNS gain search also tries the last gain of every shape
#define LC3_SPEC_NBITS_SPEC_OLD 16 // A1: nbits_spec_old follows nbits_spec
// internal (never part of the ABI's LC3GPU_SPEC_* set; LC3GPU_SEQ_SUMS=1 in the environment at create time): decisions that
// are normally taken from guarded tree sums always take their sequential-sum path, so that tests can run that path
#define LC3_SPEC_TEST_SEQ_SUMS 256
// per launch, set by the host for launches that do not fill the chip: the analysis kernel prepares the packer's symbols
// (lc3_enc_symbols).  LC3GPU_PREP_SYMBOLS=0 / 1 in the environment forces it off / on for every launch (tests).
#define LC3_LAUNCH_PREP_SYMBOLS 512
#ifndef LC3_BISECT_GUARD
#define LC3_BISECT_GUARD 1.0e-4f
#endif
// the CPU emulator of the device code (tests/emu) defines these to run the sequential sum next to every guarded
// decision and to stop when the two disagree
#ifndef LC3_GUARD_SELFCHECK
#define LC3_GUARD_SELFCHECK 0
#define LC3_GUARD_ASSERT(cond) ((void)0)
#endif
// x / d for many x and one (wave-uniform) d.  The GPU build (lc3gpu.hip) supplies the hardware's own correctly rounded
// division sequence with the part that only depends on d done once; here: the plain division.
#ifndef LC3_UNIFORM_DIV
struct lc3_divisor { float d; };
static inline lc3_divisor lc3_divisor_make(float d) { lc3_divisor v = {d}; return v; }
static inline float lc3_div_by(float x, const lc3_divisor &v) { return x / v.d; }
#endif

// HBM "planes" hand frames between the wave-per-stream and the lane-per-frame kernels.  Frame-major: a frame's
// words are contiguous, so the wave side moves them with coalesced 256-byte accesses and the lane side walks its
// own frame sequentially (sectors are merged / re-served by L2).  The lane-major alternative
// ([block of 64 frames][word][lane]) coalesces the lane side instead but costs a 64-byte sector per 4-byte word on
// the wave side (measured: 15 KB written + 33 KB fetched per frame, profiles/r01_v3_hbm_traffic.csv).
#ifndef LC3_PLANE_FRAME_MAJOR
#define LC3_PLANE_FRAME_MAJOR 1
#endif
#if LC3_PLANE_FRAME_MAJOR
#define LC3_PLANE_COL(base, f, words) ((base) + (size_t)(f) * (size_t)(words))
#define LC3_PLANE_STRIDE 1
#else
#define LC3_PLANE_COL(base, f, words) ((base) + ((size_t)(f) >> 6) * (size_t)((words) * 64) + ((size_t)(f) & 63))
#define LC3_PLANE_STRIDE 64
#endif
#define LC3_MAX_NF 480
#define LC3_MAX_NE 400

// four consecutive floats at a 16-byte aligned address (one ds_read_b128 / global_load_dwordx4)
struct __attribute__((aligned(16))) lc3_f4 { float x, y, z, w; };
typedef int lc3_i4 __attribute__((vector_size(16)));  // builtin vector: assignable across address spaces

struct lc3_cpx {
    float r, i;
};

// Per-configuration constants (reference common/config.rs:18-100) plus the derived
// transform plan.  Passed to kernels by value (wave-uniform -> SGPRs).
struct lc3_cfg {
    int fs, fs_ind, nf, ne, nb, z, n_ms_10;
    // mixed-radix plan of the nf/2-point complex FFT (common/kissfft.rs:47-76)
    int nfft, n_stages;
    int radix[6], m[6], fstride[6];
    int inv_m[6];  // ceil(2^16 / m[s]): u / m[s] == (u * inv_m[s]) >> 16 for every butterfly index u < nfft (host plan checks)
    // device tables owned by the codec handle
    const lc3_cpx *fft_tw;  // exp(-2*pi*i*k/nfft), f64 -> f32 (kissfft.rs:19-27)
    const lc3_cpx *dct_tw;  // exp(-i*pi*(8n+1)/(8*nf)), f64 -> f32 (dct_iv.rs:30-35)
    const uint16_t *perm;   // leaf gather order of kf_work (kissfft.rs:101-108)
    // encoder LTPF (encoder/long_term_post_filter.rs:93-127)
    int len12, len6, delay12, p_up, hist;
    float resamp_scale;  // p as f32 * resampling_factor
    // polyphase layout of the resampling low-pass: row ph (0..p_up-1) holds taps k = -lim..lim of phase ph, zero padded
    // to resamp_nt (multiple of 4) taps, rows resamp_stride floats apart (lc3_resamp_poly_value)
    int resamp_lim, resamp_nt, resamp_stride;
    int inv_p;     // ceil(2^16 / p_up): x / p_up == (x * inv_p) >> 16 for x = 15 n, n < len12
    const float *resamp_poly;
    // width (in lines, as f32) of the band each spectral line belongs to, ne entries (lc3_line_width_value)
    const float *line_width;
    // band (0..nb-1) each spectral line belongs to, 255 for lines >= ne; nf entries, read four at a time (lc3_line_band_value)
    const uint8_t *line_band;
    // The tables a wave-per-stream kernel stages in LDS once per workgroup, as ONE image in the LDS layout (lc3_fft_tables, then
    // lc3_front_tables; written on the device when the configuration is registered): staging is a 16-byte copy instead of six strided loops
    const void *stage_image;
    // decoder LTPF (decoder/long_term_post_filter.rs:104-134)
    int l_den, l_num, num_mem_blocks, norm, s25;
};

__device__ __forceinline__ float lc3_f(const uint32_t *tab, int i) { return __builtin_bit_cast(float, tab[i]); }
__device__ __forceinline__ float lc3_from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t lc3_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ double lc3_d_from_bits(uint64_t u) { return __builtin_bit_cast(double, u); }

// f32::max / f32::min: NaN-ignoring (Rust core::f32)
__device__ __forceinline__ float lc3_maxf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
__device__ __forceinline__ float lc3_minf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
__device__ __forceinline__ float lc3_absf(float a) { return lc3_from_bits(lc3_bits(a) & 0x7fffffffu); }

// Rust `as` casts are saturating with NaN -> 0.  Branch-free: NaN replaced by 0, the value clamped in the
// float domain (both bounds are exact floats), then truncated -- three or four instructions instead of a chain of divergent
// early returns.
__device__ __forceinline__ float lc3_nan_to_zero(float x) { return x != x ? 0.0f : x; }
__device__ __forceinline__ float lc3_clampf(float x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }
__device__ __forceinline__ int32_t lc3_f2i32(float x) {
    const float xc = lc3_nan_to_zero(x);
    const int32_t r = (int32_t)lc3_clampf(xc, -2147483648.0f, 2147483520.0f);  // 2147483520 = the largest float below 2^31
    return xc >= 2147483648.0f ? 2147483647 : r;
}
__device__ __forceinline__ int32_t lc3_f2i16(float x) { return (int32_t)lc3_clampf(lc3_nan_to_zero(x), -32768.0f, 32767.0f); }
// the same cast for an operand the caller knows is not a NaN (a finite value divided by a positive finite one): no NaN test
__device__ __forceinline__ int32_t lc3_f2i16_no_nan(float x) { return (int32_t)lc3_clampf(x, -32768.0f, 32767.0f); }
// |mag| with the sign bit of x (x = -0.0 counts as negative: callers add the result to a value for which that makes no difference)
__device__ __forceinline__ float lc3_with_sign_of(float mag, float x) { return lc3_from_bits((lc3_bits(mag) & 0x7fffffffu) | (lc3_bits(x) & 0x80000000u)); }
__device__ __forceinline__ int32_t lc3_f2i8(float x) { return (int32_t)lc3_clampf(lc3_nan_to_zero(x), -128.0f, 127.0f); }
__device__ __forceinline__ int32_t lc3_f2u16(float x) { return (int32_t)lc3_clampf(lc3_nan_to_zero(x), 0.0f, 65535.0f); }
// The transform tables of the configuration (FFT twiddles, DCT-IV twiddles, leaf gather order).  The device build can
// redirect them to a copy staged in LDS (lc3gpu.hip: lc3_fft_tab); the default reads the configuration's HBM tables.
#ifdef LC3_FFT_TABLES_IN_LDS
struct lc3_fft_tables {
    lc3_cpx fft_tw[LC3_MAX_NF / 2], dct_tw[LC3_MAX_NF / 2];
    uint16_t perm[LC3_MAX_NF / 2];
    uint16_t leaf_bfly[64];  // innermost stage: the butterfly whose leaves are l, l + nb, l + 2 nb ... (nb = nfft / radix <= 48 of them)
};
__shared__ lc3_fft_tables lc3_fft_tab;  // one copy per workgroup (4.2 KB), filled by lc3_fft_tables_stage
#ifdef LC3_TABLES_IN_GLOBAL  // experiment build: the tables read from the configuration's stage image in global memory (no LDS copy)
#define LC3_FFT_IMG(c) ((const lc3_fft_tables *)(c).stage_image)
#define LC3_FFT_TW(c) (LC3_FFT_IMG(c)->fft_tw)
#define LC3_DCT_TW(c) (LC3_FFT_IMG(c)->dct_tw)
#define LC3_FFT_PERM(c) (LC3_FFT_IMG(c)->perm)
#define LC3_FFT_LEAF_BFLY(c, l) ((int)LC3_FFT_IMG(c)->leaf_bfly[(l)])
#else
#define LC3_FFT_TW(c) (lc3_fft_tab.fft_tw)
#define LC3_DCT_TW(c) (lc3_fft_tab.dct_tw)
#define LC3_FFT_PERM(c) (lc3_fft_tab.perm)
#define LC3_FFT_LEAF_BFLY(c, l) ((int)lc3_fft_tab.leaf_bfly[(l)])
#endif
// all threads of the workgroup; ends with a workgroup barrier
template <class CC>
__device__ __forceinline__ void lc3_fft_tables_stage(const CC &c) {
    const int n = c.nfft;  // complex elements are copied as two 32-bit words, the gather order as 16-bit words
    const uint32_t *ft = (const uint32_t *)c.fft_tw, *dt = (const uint32_t *)c.dct_tw;
    uint32_t *lf = (uint32_t *)lc3_fft_tab.fft_tw, *ld = (uint32_t *)lc3_fft_tab.dct_tw;
    for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) {
        lf[i] = ft[i];
        ld[i] = dt[i];
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) lc3_fft_tab.perm[i] = c.perm[i];
    {   // the leaves of the innermost stage's butterfly u are perm[p u + k] = perm[p u] + k nb: invert u -> perm[p u]
        const int p = c.radix[c.n_stages - 1], nb = n / p;
        for (int u = threadIdx.x; u < nb; u += blockDim.x) lc3_fft_tab.leaf_bfly[c.perm[u * p]] = (uint16_t)u;
    }
    __syncthreads();
}
// the same from the configuration's image: 16-byte units, all threads of the workgroup; ends with a workgroup barrier
static_assert(sizeof(lc3_fft_tables) % 16 == 0, "image: 16-byte units");
__device__ __forceinline__ void lc3_fft_tables_stage_image(const void *image) {
    const lc3_i4 *src = (const lc3_i4 *)image;
    lc3_i4 *dst = (lc3_i4 *)&lc3_fft_tab;
    for (int i = threadIdx.x; i < (int)(sizeof(lc3_fft_tables) / 16); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}
// ... in two steps, so that a kernel can have the image, its stream state and its first frame on their way from HBM at the same time (a
// launch is only four frames long: three or four trips in a row at its start were ~15 % of the synthesis kernel's time).  Two units per
// thread: enough for the 270 units with 64 * LC3_WG_WAVES >= 135 threads (three waves; the production workgroup has four).
#ifndef LC3_WG_WAVES
#define LC3_WG_WAVES 4
#endif
#define LC3_WG_THREADS (64 * LC3_WG_WAVES)
struct lc3_fft_image_regs { lc3_i4 a, b; };
__device__ __forceinline__ lc3_fft_image_regs lc3_fft_tables_image_issue(const void *image) {
    LC3_HBM_CONST(lc3_i4) src = (LC3_HBM_CONST(lc3_i4))image;
    const int n = (int)(sizeof(lc3_fft_tables) / 16), i = (int)threadIdx.x;
    static_assert(sizeof(lc3_fft_tables) / 16 <= 2 * LC3_WG_THREADS, "two units per thread of the workgroup");
    lc3_fft_image_regs r;
    r.a = src[i < n ? i : 0];
    r.b = src[i + LC3_WG_THREADS < n ? i + LC3_WG_THREADS : 0];
    return r;
}
__device__ __forceinline__ void lc3_fft_tables_image_commit(const lc3_fft_image_regs &r) {  // ends with a workgroup barrier
    lc3_i4 *dst = (lc3_i4 *)&lc3_fft_tab;
    const int n = (int)(sizeof(lc3_fft_tables) / 16), i = (int)threadIdx.x;
    if (i < n) dst[i] = r.a;
    if (i + LC3_WG_THREADS < n) dst[i + LC3_WG_THREADS] = r.b;
    __syncthreads();
}
// Tables only the analysis front half reads, per workgroup: the band width of every spectral line (divisor of the band
// energies) and the fractional-lag interpolation filter of the LTPF analysis -- every frame re-read them from L2 otherwise
struct lc3_front_tables {
    float line_width[LC3_MAX_NE];
    float interp_r[32];
    float resamp_poly[336];  // polyphase rows of the LTPF resampler (lc3_resamp_poly_value), <= 12 rows x 28 floats for an encoder
};
__shared__ __attribute__((aligned(16))) lc3_front_tables lc3_front_tab;  // 3.2 KB, filled by lc3_front_tables_stage
#ifdef LC3_TABLES_IN_GLOBAL
#define LC3_FRONT_IMG(c) ((const lc3_front_tables *)((const char *)(c).stage_image + sizeof(lc3_fft_tables)))
#define LC3_LINE_WIDTH(c, k) (LC3_FRONT_IMG(c)->line_width[(k)])
#define LC3_LTPF_INTERP_R(i) (lc3_f(LC3T_TAB_LTPF_INTERP_R_BITS, (i)))
#define LC3_RESAMP_POLY(c) (LC3_FRONT_IMG(c)->resamp_poly)
#else
#define LC3_LINE_WIDTH(c, k) (lc3_front_tab.line_width[(k)])
#define LC3_LTPF_INTERP_R(i) (lc3_front_tab.interp_r[(i)])
#define LC3_RESAMP_POLY(c) (lc3_front_tab.resamp_poly)
#endif
#define LC3_RESAMP_POLY_IN_LDS 1
template <class CC>
__device__ __forceinline__ void lc3_front_tables_stage(const CC &c) {  // all threads of the workgroup, before a workgroup barrier
    for (int i = threadIdx.x; i < c.ne; i += blockDim.x) lc3_front_tab.line_width[i] = c.line_width[i];
    for (int i = threadIdx.x; i < 31; i += blockDim.x) lc3_front_tab.interp_r[i] = lc3_f(LC3T_TAB_LTPF_INTERP_R_BITS, i);
    const int p_rows = c.p_up * c.resamp_stride;
    for (int i = threadIdx.x; i < p_rows && i < 336; i += blockDim.x) lc3_front_tab.resamp_poly[i] = c.resamp_poly[i];
}
static_assert(sizeof(lc3_front_tables) % 16 == 0, "image: 16-byte units");
__device__ __forceinline__ void lc3_front_tables_stage_image(const void *image) {  // (before a workgroup barrier, like lc3_front_tables_stage)
    const lc3_i4 *src = (const lc3_i4 *)((const char *)image + sizeof(lc3_fft_tables));
    lc3_i4 *dst = (lc3_i4 *)&lc3_front_tab;
    for (int i = threadIdx.x; i < (int)(sizeof(lc3_front_tables) / 16); i += blockDim.x) dst[i] = src[i];
}
#define LC3_STAGE_IMAGE_BYTES (sizeof(lc3_fft_tables) + sizeof(lc3_front_tables))
#else
#define LC3_LINE_WIDTH(c, k) ((c).line_width[(k)])
#define LC3_LTPF_INTERP_R(i) (lc3_f(LC3T_TAB_LTPF_INTERP_R_BITS, (i)))
#define LC3_FFT_TW(c) ((c).fft_tw)
#define LC3_DCT_TW(c) ((c).dct_tw)
#define LC3_FFT_PERM(c) ((c).perm)
#endif
#ifndef LC3_FFT_LEAF_BFLY  // (no LDS copy of the tables: found by search -- the CPU emulator's build)
template <class CC>
__device__ __forceinline__ int lc3_fft_leaf_bfly_search(const CC &c, int l) {
    const int p = c.radix[c.n_stages - 1], nb = c.nfft / p;
    for (int u = 0; u < nb; u++)
        if ((int)c.perm[u * p] == l) return u;
    return 0;
}
#define LC3_FFT_LEAF_BFLY(c, l) lc3_fft_leaf_bfly_search(c, l)
#endif
// a * b for a, b < 2^24 (the device build maps it to the 24-bit multiplier)
#ifndef LC3_MUL24
#define LC3_MUL24(a, b) ((uint32_t)(a) * (uint32_t)(b))
#endif
__device__ __forceinline__ int lc3_ilog2(uint32_t v) { return 31 - __builtin_clz(v | 1u); }

// exact IEEE operations the reference relies on.  With hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt both are correctly rounded.
__device__ __forceinline__ float lc3_sqrtf(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float lc3_floorf(float x) { return __builtin_floorf(x); }
__device__ __forceinline__ float lc3_ceilf(float x) { return __builtin_ceilf(x); }

// ---- the link between the two waves of a producer / consumer pair (parser: lc3_dev_dec_parse.h, packer: lc3_dev_enc_pack.h): a
// ring of one word per lane and iteration in LDS and two wave-level counters.  The iteration count is the wave's -- every lane of both
// waves steps once per iteration, a lane without work idles -- so ring entry i of lane l means the same to both.  The producer
// publishes its count every LC3_PC_CHUNK iterations (a wave's LDS stores execute in order: the count follows the entries it covers)
// and waits when the consumer falls a ring behind; the consumer publishes what it has taken.
#ifndef LC3_PC_STORE   // (the GPU build defines these over LDS with the compiler kept from reordering around them; these are the emulator's)
#define LC3_PC_STORE(p, v) (*(volatile int *)(p) = (v))
#define LC3_PC_LOAD(p) (*(volatile const int *)(p))
#define LC3_PC_PAUSE() ((void)0)
#define LC3_PC_RELEASE() ((void)0)
#define LC3_PC_ACQUIRE() ((void)0)
#endif
#ifndef LC3_PC_CHUNK
#define LC3_PC_CHUNK 8
#endif
#define LC3_PC_DONE 0x40000000
#define LC3_PC_SPIN_LIMIT (1 << 24)  // polls before a wave gives up on its partner (never reached unless the partner died)
struct lc3_pc_link {
    uint32_t *ring;   // this lane's entries: entry i at ring[(i & mask) * stride]
    int mask, stride;
    int *p_count;     // wave-level words: -1 until the producer has left its start values, then the iterations it has published, | LC3_PC_DONE at its end
    int *c_count;     // iterations the consumer has taken
    uint32_t *fin;    // this lane's hand-over words fin[j * fstride], j = 0 .. 3: head cursor after the TNS data, final range, final head cursor, error flag
    int fstride;
};

// range decoder state (decoder/arithmetic_codec.rs:22-26)
struct lc3_acdec { uint32_t low, range; };

// element i (0 .. LC3_TNS_MODEL_WORDS-1) of the packed TNS models (range coder and decoder keep them in LDS): [2][8] order models, then [8][17] coefficient models
#define LC3_TNS_MODEL_WORDS (2 * 8 + 8 * 17)
__device__ __forceinline__ uint32_t lc3_tns_model_word(int i) {
    if (i < 16) return (uint32_t)(int)LC3T_AC_TNS_ORDER_CUMFREQ[i / 8][i % 8] | ((uint32_t)(int)LC3T_AC_TNS_ORDER_FREQ[i / 8][i % 8] << 16);
    const int k = (i - 16) / 17, j = (i - 16) % 17;
    return (uint32_t)(int)LC3T_AC_TNS_COEF_CUMFREQ[k][j] | ((uint32_t)(int)LC3T_AC_TNS_COEF_FREQ[k][j] << 16);
}

// ------------------------------------------------------------------------------------------
// Float library.  The reference is no_std: every f32 method resolves through num_traits to the
// `libm` crate (reference Cargo.toml:17), a port of the FreeBSD msun routines.  These are the same
// published algorithms written for the GPU: plain f32 (f64 where msun uses double), no FMA.
// Call sites are cited where they are used.
// ------------------------------------------------------------------------------------------

// e_powf.c, specialised for the only base the codec uses: x = 10 (interval k = 1, n = 3)
// and finite y with |y| < 2^27 (encoder/spectral_noise_shaping.rs:218,
// encoder/spectral_quantization.rs:239, decoder/global_gain.rs:20).
__device__ __forceinline__ float lc3_pow10f(float y) {
    const float L1 = lc3_from_bits(0x3f19999au), L2 = lc3_from_bits(0x3edb6db7u), L3 = lc3_from_bits(0x3eaaaaabu);
    const float L4 = lc3_from_bits(0x3e8ba305u), L5 = lc3_from_bits(0x3e6c3255u), L6 = lc3_from_bits(0x3e53f142u);
    const float P1 = lc3_from_bits(0x3e2aaaabu), P2 = lc3_from_bits(0xbb360b61u), P3 = lc3_from_bits(0x388ab355u);
    const float P4 = lc3_from_bits(0xb5ddea0eu), P5 = lc3_from_bits(0x3331bb4cu);
    const float lg2 = lc3_from_bits(0x3f317218u), lg2_h = lc3_from_bits(0x3f317200u), lg2_l = lc3_from_bits(0x35bfbe8cu);
    const float ovt = 4.2995665694e-08f;
    const float cp = lc3_from_bits(0x3f76384fu), cp_h = lc3_from_bits(0x3f764000u), cp_l = lc3_from_bits(0xb8f623c6u);
    const float dp_h = lc3_from_bits(0x3f15c000u), dp_l = lc3_from_bits(0x35d1cfdcu);
    const float huge = 1.0e30f, tiny = 1.0e-30f;
    const float bp = 1.5f;
    uint32_t hy = lc3_bits(y), iy = hy & 0x7fffffffu;
    if (iy == 0) return 1.0f;
    if (iy > 0x7f800000u) return 10.0f + y;
    if (iy == 0x7f800000u) return (hy >> 31) ? 0.0f : y;
    if (iy == 0x3f800000u) return (hy >> 31) ? 1.0f / 10.0f : 10.0f;
    if (hy == 0x40000000u) return 10.0f * 10.0f;
    if (hy == 0x3f000000u) return lc3_sqrtf(10.0f);
    float t1, t2;
    {
        // ix(10.0f) = 0x41200000: n = 3, j = 0x200000 -> interval k = 1, normalised ax = 1.25
        const int32_t ix = 0x3fa00000;
        const float ax = 1.25f;
        float u = ax - bp;
        float v = 1.0f / (ax + bp);
        float s = u * v;
        float s_h = lc3_from_bits(lc3_bits(s) & 0xfffff000u);
        uint32_t is = (((uint32_t)ix >> 1) & 0xfffff000u) | 0x20000000u;
        float t_h = lc3_from_bits(is + 0x00400000u + (1u << 21));
        float t_l = ax - (t_h - bp);
        float s_l = v * ((u - s_h * t_h) - s_h * t_l);
        float s2 = s * s;
        float r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
        r += s_l * (s_h + s);
        s2 = s_h * s_h;
        t_h = 3.0f + s2 + r;
        t_h = lc3_from_bits(lc3_bits(t_h) & 0xfffff000u);
        t_l = r - ((t_h - 3.0f) - s2);
        u = s_h * t_h;
        v = s_l * t_h + t_l * s;
        float p_h = u + v;
        p_h = lc3_from_bits(lc3_bits(p_h) & 0xfffff000u);
        float p_l = v - (p_h - u);
        float z_h = cp_h * p_h;
        float z_l = cp_l * p_h + p_l * cp + dp_l;
        float t = 3.0f;
        t1 = (((z_h + z_l) + dp_h) + t);
        t1 = lc3_from_bits(lc3_bits(t1) & 0xfffff000u);
        t2 = z_l - (((t1 - t) - dp_h) - z_h);
    }
    float y1 = lc3_from_bits(hy & 0xfffff000u);
    float p_l = (y - y1) * t1 + y * t2;
    float p_h = y1 * t1;
    float z = p_l + p_h;
    int32_t j = (int32_t)lc3_bits(z);
    if (j > 0x43000000) return huge * huge;
    else if (j == 0x43000000) {
        if (p_l + ovt > z - p_h) return huge * huge;
    } else if ((j & 0x7fffffff) > 0x43160000) return tiny * tiny;
    else if ((uint32_t)j == 0xc3160000u) {
        if (p_l <= z - p_h) return tiny * tiny;
    }
    int32_t i = j & 0x7fffffff;
    int32_t k = (i >> 23) - 0x7f;
    int32_t n = 0;
    if (i > 0x3f000000) {
        n = j + (0x00800000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 23) - 0x7f;
        float t = lc3_from_bits((uint32_t)n & ~(0x007fffffu >> k));
        n = ((n & 0x007fffff) | 0x00800000) >> (23 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    float t = p_l + p_h;
    t = lc3_from_bits(lc3_bits(t) & 0xffff8000u);
    float u = t * lg2_h;
    float v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    float w = v - (z - u);
    t = z * z;
    t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    float r = (z * t1) / (t1 - 2.0f) - (w + z * w);
    z = 1.0f - (r - z);
    j = (int32_t)lc3_bits(z);
    j += (int32_t)((uint32_t)n << 23);
    if ((j >> 23) <= 0) {
        // subnormal result: scalbnf(z, n) -- two exact power-of-two scalings (n in [-150, -126))
        float s1 = lc3_from_bits((uint32_t)(n + 60 + 127) << 23);
        z = (z * s1) * lc3_from_bits((uint32_t)(127 - 60) << 23);
    } else z = lc3_from_bits((uint32_t)j);
    return z;
}

// The codec only ever raises 10 to (gg_ind + gg_off) / 28 (an integer in [-256, 255] over 28: spectral_quantization.rs:239,
// global_gain.rs:19-20) and to b * g_tilt / 630 (band b < 64, five tilts: spectral_noise_shaping.rs:216-218).  The HIP build
// tabulates both once per device WITH lc3_pow10f itself (lc3gpu.hip: lc3_pow10_tables_kernel), so a lookup returns the very bits
// the routine computes; translation units without the tables evaluate the routine.
#ifndef LC3_POW10_GG
#define LC3_POW10_GG(k) lc3_pow10f((float)(k) / 28.0f)
#define LC3_POW10_TILT(fs_ind, b) lc3_pow10f((float)(b) * ((float)LC3C_G_TILT[(fs_ind)] / 630.0f))
#endif

// shared reduction of e_log2f.c / e_log10f.c.  Branch-free: the main path is evaluated for every input (garbage in, garbage
// out, no traps) and the special results (zero, negative, infinity / NaN, exactly 1) are selected at the end -- the lanes of a
// wave hold different values, so early returns would only add exec-mask bookkeeping around the common path.
struct lc3_logparts {
    float hi, lo;
    int k;
    int special;   // 1: result already final in `hi`
};
__device__ __forceinline__ lc3_logparts lc3_log_reduce(float x) {
    const float Lg1 = lc3_from_bits(0x3f2aaaaau), Lg2 = lc3_from_bits(0x3eccce13u);
    const float Lg3 = lc3_from_bits(0x3e91e9eeu), Lg4 = lc3_from_bits(0x3e789e26u);
    lc3_logparts p;
    const uint32_t ix0 = lc3_bits(x);
    const int neg = (int)(ix0 >> 31), zero = (ix0 << 1) == 0, sub = ix0 < 0x00800000u || neg;  // x < 2^-126 (or negative)
    const int big = !sub && ix0 >= 0x7f800000u, one = !sub && !big && ix0 == 0x3f800000u;
    // special results, in msun's order of tests
    const float sp = zero ? -1.0f / (x * x) : (neg ? (x - x) / 0.0f : (big ? x : 0.0f));
    p.special = zero | neg | big | one;
    const float xs = sub ? x * 33554432.0f : x;  // subnormal: scale up by 2^25
    uint32_t ix = lc3_bits(xs);
    int k = sub ? -25 : 0;
    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    const float xr = lc3_from_bits(ix);
    const float f = xr - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (Lg2 + w * Lg4);
    const float t2 = z * (Lg1 + w * Lg3);
    const float R = t2 + t1;
    const float hfsq = 0.5f * f * f;
    float hi = f - hfsq;
    hi = lc3_from_bits(lc3_bits(hi) & 0xfffff000u);
    p.hi = p.special ? sp : hi;
    p.lo = (f - hi) - hfsq + s * (hfsq + R);
    p.k = k;
    return p;
}
// e_log2f.c (encoder/spectral_noise_shaping.rs:232)
__device__ __forceinline__ float lc3_log2f(float x) {
    const float ivln2hi = lc3_from_bits(0x3fb8b000u), ivln2lo = lc3_from_bits(0xb9389ad4u);
    const lc3_logparts p = lc3_log_reduce(x);
    const float r = (p.lo + p.hi) * ivln2lo + p.lo * ivln2hi + p.hi * ivln2hi + (float)p.k;
    return p.special ? p.hi : r;
}
// e_log10f.c (encoder/spectral_quantization.rs:218,393)
__device__ __forceinline__ float lc3_log10f(float x) {
    const float ivln10hi = lc3_from_bits(0x3ede6000u), ivln10lo = lc3_from_bits(0xb804ead9u);
    const float log10_2hi = lc3_from_bits(0x3e9a2080u), log10_2lo = lc3_from_bits(0x355427dbu);
    const lc3_logparts p = lc3_log_reduce(x);
    const float dk = (float)p.k;
    const float r = dk * log10_2lo + (p.lo + p.hi) * ivln10lo + p.lo * ivln10hi + p.hi * ivln10hi + dk * log10_2hi;
    return p.special ? p.hi : r;
}

// s_exp2f.c, TBLSIZE = 16, f64 polynomial (encoder/spectral_noise_shaping.rs:256).  exp2ft[i] = 2^((i-8)/16) as f64 bit
// patterns; branch-free (the callers run it one frame per lane: divergent early returns and a 16-way switch would be
// executed by every lane) with the special results selected at the end.
static __device__ const uint64_t LC3C_EXP2FT[16] = {
    0x3fe6a09e667f3bcdull, 0x3fe7a11473eb0187ull, 0x3fe8ace5422aa0dbull, 0x3fe9c49182a3f090ull,
    0x3feae89f995ad3adull, 0x3fec199bdd85529cull, 0x3fed5818dcfba487ull, 0x3feea4afa2a490daull,
    0x3ff0000000000000ull, 0x3ff0b5586cf9890full, 0x3ff172b83c7d517bull, 0x3ff2387a6e756238ull,
    0x3ff306fe0a31b715ull, 0x3ff3dea64c123422ull, 0x3ff4bfdad5362a27ull, 0x3ff5ab07dd485429ull};
__device__ __forceinline__ float lc3_exp2f(float x) {
    const float redux = lc3_from_bits(0x4b400000u) / 16.0f;
    const float P1 = lc3_from_bits(0x3f317218u), P2 = lc3_from_bits(0x3e75fdf0u);
    const float P3 = lc3_from_bits(0x3d6359a4u), P4 = lc3_from_bits(0x3c1d964eu);
    const uint32_t ui = lc3_bits(x), ix = ui & 0x7fffffffu;
    // main path (meaningful for 2^-25 < |x| <= 126; evaluated for every input)
    float uf = x + redux;
    uint32_t i0 = lc3_bits(uf);
    i0 += 8;
    const uint32_t k = i0 / 16;
    const uint64_t uk = (uint64_t)(0x3ffu + k) << 52;
    i0 &= 15;
    uf -= redux;
    const double z = (double)(x - uf);
    double r = lc3_d_from_bits(LC3C_EXP2FT[i0]);
    const double t = r * z;
    r = r + t * ((double)P1 + z * (double)P2) + t * (z * z) * ((double)P3 + z * (double)P4);
    const float res = (float)(r * lc3_d_from_bits(uk));
    // the special ranges, in msun's order
    const int large = ix > 0x42fc0000u;  // |x| > 126
    const int nan = ix > 0x7f800000u, over = ui >= 0x43000000u && ui < 0x80000000u, under = ui >= 0xc3160000u;
    const float sp_large = nan ? x : (over ? x * lc3_from_bits(0x7f000000u) : (under ? 0.0f : res));
    return large ? sp_large : (ix <= 0x33000000u ? 1.0f + x : res);
}

// e_asinf.c (encoder/temporal_noise_shaping.rs:272)
__device__ __forceinline__ float lc3_asinf_R(float z) {
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f;
    const float qS1 = -7.0662963390e-01f;
    float p = z * (pS0 + z * (pS1 + z * pS2));
    float q = 1.0f + z * qS1;
    return p / q;
}
__device__ __forceinline__ float lc3_asinf(float x) {
    const double pio2 = 1.570796326794896558e+00;
    uint32_t hx = lc3_bits(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3f800000u) {
        if (ix == 0x3f800000u) return (float)((double)x * pio2 + 7.5231638452626401e-37);
        return 0.0f / (x - x);
    }
    if (ix < 0x3f000000u) {
        if (ix < 0x39800000u && ix >= 0x00800000u) return x;
        return x + x * lc3_asinf_R(x * x);
    }
    float z = (1.0f - lc3_absf(x)) * 0.5f;
    double s = __builtin_sqrt((double)z);
    x = (float)(pio2 - 2.0 * (s + s * (double)lc3_asinf_R(z)));
    if (hx >> 31) return -x;
    return x;
}

// s_sinf.c restricted to |x| <= 3*pi/4 (the codec evaluates sin(k*pi/17), |k| <= 8:
// encoder/temporal_noise_shaping.rs:273, decoder/temporal_noise_shaping.rs:44)
__device__ __forceinline__ float lc3_sinf_small(float x) {
    const double S1 = -0x15555554cbac77.0p-55, S2 = 0x111110896efbb2.0p-59;
    const double S3 = -0x1a00f9e2cae774.0p-65, S4 = 0x16cd878c3b46a7.0p-71;
    const double C0 = -0x1ffffffd0c5e81.0p-54, C1 = 0x155553e1053a42.0p-57;
    const double C2 = -0x16c087e80f1e27.0p-62, C3 = 0x199342e0ee5069.0p-68;
    const double s1pio2 = 1.5707963267948966;
    uint32_t ix = lc3_bits(x);
    int sign = (int)(ix >> 31);
    ix &= 0x7fffffffu;
    if (ix <= 0x3f490fdau) {
        if (ix < 0x39800000u) return x;
        double xd = (double)x, z = xd * xd, w = z * z, r = S3 + z * S4, s = z * xd;
        return (float)((xd + s * (S1 + z * S2)) + s * w * r);
    }
    double xd = sign ? (double)x + s1pio2 : (double)x - s1pio2;
    double z = xd * xd, w = z * z, r = C2 + z * C3;
    float c = (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
    return sign ? -c : c;
}

// Quantised TNS reflection coefficients: sin(step * (rc_i - 8)), rc_i = 0..16, step = PI / 17 -- formed as (PI as f32) / 17.0 by
// the encoder (encoder/temporal_noise_shaping.rs:268-273) and as (PI / 17.0) as f32 by the decoder
// (decoder/temporal_noise_shaping.rs:41-44).  The HIP build tabulates both with lc3_sinf_small itself (lc3gpu.hip).
__device__ __forceinline__ float lc3_tns_sin_enc_value(int ri) {
    const float step = (float)3.14159265358979323846 / 17.0f;
    return lc3_sinf_small(step * ((float)ri - 8.0f));
}
__device__ __forceinline__ float lc3_tns_sin_dec_value(int ri) {
    const float step = (float)(3.14159265358979323846 / 17.0);
    return lc3_sinf_small(step * (float)(ri - 8));
}
#ifndef LC3_TNS_SIN_ENC
#define LC3_TNS_SIN_ENC(ri) lc3_tns_sin_enc_value(ri)
#define LC3_TNS_SIN_DEC(ri) lc3_tns_sin_dec_value(ri)
#endif

// fast_math::exp2_raw (fast-math 0.1.1; decoder/spectral_noise_shaping.rs:122)
__device__ __forceinline__ float lc3_exp2_raw(float x) {
    const float A = 8388608.0f, E = 1.1920929e-7f;
    const float C0 = (0.3371894346f * E) * E, C1 = 0.657636276f * E, C2 = 1.00172476f;
    float a = A * x;
    int32_t mul = lc3_f2i32(a);
    uint32_t fl = (uint32_t)mul & 0xff800000u;
    float frac = (float)(int32_t)((uint32_t)mul - fl);
    float approx = (C0 * frac + C1) * frac + C2;
    return lc3_from_bits(lc3_bits(approx) + fl);
}

// num_traits pow(): square-and-multiply (used as powi; encoder/spectral_noise_shaping.rs:223-224,
// encoder/temporal_noise_shaping.rs:240)
__device__ __forceinline__ float lc3_powi(float base, int exp) {
    unsigned e;
    if (exp < 0) { base = 1.0f / base; e = (unsigned)(-exp); } else e = (unsigned)exp;
    if (e == 0) return 1.0f;
    while ((e & 1u) == 0) { base = base * base; e >>= 1; }
    if (e == 1) return base;
    float acc = base;
    while (e > 1) {
        e >>= 1;
        base = base * base;
        if (e & 1u) acc = acc * base;
    }
    return acc;
}

// ------------------------------------------------------------------------------------------
// small constant tables of the reference's stage modules (kept in device constant data, not on
// the per-lane stack)
// ------------------------------------------------------------------------------------------
static __device__ const int LC3C_NBITS_BW[5] = {0, 1, 2, 2, 3};  // bandwidth_detector.rs:10, side_info_reader.rs:11
// encoder/bandwidth_detector.rs:5-18
static __device__ const int LC3C_BW_START10[4][4] = {{53, 0, 0, 0}, {47, 59, 0, 0}, {44, 54, 60, 0}, {41, 51, 57, 61}};
static __device__ const int LC3C_BW_STOP10[4][4] = {{63, 0, 0, 0}, {56, 63, 0, 0}, {52, 59, 63, 0}, {49, 55, 60, 63}};
static __device__ const int LC3C_BW_START75[4][4] = {{51, 0, 0, 0}, {45, 58, 0, 0}, {42, 53, 60, 0}, {40, 51, 57, 61}};
static __device__ const int LC3C_BW_STOP75[4][4] = {{63, 0, 0, 0}, {55, 63, 0, 0}, {51, 58, 63, 0}, {48, 55, 60, 63}};
static __device__ const int LC3C_BW_TQ[4] = {20, 10, 10, 10};
static __device__ const int LC3C_BW_TC[4] = {15, 23, 20, 20};
static __device__ const int LC3C_BW_L10[4] = {4, 4, 3, 1};
static __device__ const int LC3C_BW_L75[4] = {4, 4, 3, 2};
static __device__ const int LC3C_G_TILT[5] = {14, 18, 22, 26, 30};  // spectral_noise_shaping.rs:51-57
static __device__ const int LC3C_BWSTOP10[5] = {80, 160, 240, 320, 400};  // noise_level_estimation.rs:22-23, noise_filling.rs:28-29
static __device__ const int LC3C_BWSTOP75[5] = {60, 120, 180, 240, 300};
// encoder/spectral_quantization.rs:351-353
static __device__ const int LC3C_GGA_T1[5] = {80, 230, 380, 530, 680};
static __device__ const int LC3C_GGA_T2[5] = {500, 1025, 1550, 2075, 2600};
static __device__ const int LC3C_GGA_T3[5] = {850, 1700, 2550, 3400, 4250};
// encoder/temporal_noise_shaping.rs:81-84 (f32 literals)
static __device__ const float LC3C_TNS_LAGW[9] = {1.0f, 0.9980280260203829f, 0.9921354055113971f, 0.9823915844707989f,
                                                  0.9689107911912967f, 0.9518498073692735f, 0.9314049334023056f,
                                                  0.9078082299969592f, 0.8813231366694713f};
// TNS parameter sets: {num_filters, start0, start1, stop0, stop1, sub_start[2][3], sub_stop[2][3]}
// encoder/temporal_noise_shaping.rs:117-202 (10 ms p_bw = 2 keeps stop_freq = 200: SURVEY A5)
struct lc3_tns_params { int num, start[2], stop[2], sub_start[2][3], sub_stop[2][3]; };
static __device__ const lc3_tns_params LC3C_TNS10[5] = {
    {1, {12, 160}, {80, 0}, {{12, 34, 57}, {0, 0, 0}}, {{34, 57, 80}, {0, 0, 0}}},
    {1, {12, 160}, {160, 0}, {{12, 61, 110}, {0, 0, 0}}, {{61, 110, 160}, {0, 0, 0}}},
    {1, {12, 160}, {200, 0}, {{12, 88, 164}, {0, 0, 0}}, {{88, 164, 240}, {0, 0, 0}}},
    {2, {12, 160}, {160, 320}, {{12, 61, 110}, {160, 213, 266}}, {{61, 110, 160}, {213, 266, 320}}},
    {2, {12, 200}, {200, 400}, {{12, 74, 137}, {200, 266, 333}}, {{74, 137, 200}, {266, 333, 400}}},
};
static __device__ const lc3_tns_params LC3C_TNS75[5] = {
    {1, {9, 120}, {60, 0}, {{9, 26, 43}, {0, 0, 0}}, {{26, 43, 60}, {0, 0, 0}}},
    {1, {9, 120}, {120, 0}, {{9, 46, 83}, {0, 0, 0}}, {{46, 83, 120}, {0, 0, 0}}},
    {1, {9, 120}, {180, 0}, {{9, 66, 123}, {0, 0, 0}}, {{66, 123, 180}, {0, 0, 0}}},
    {2, {9, 120}, {120, 240}, {{9, 46, 82}, {120, 159, 200}}, {{46, 82, 120}, {159, 200, 240}}},
    {2, {9, 150}, {150, 300}, {{9, 56, 103}, {150, 200, 250}}, {{56, 103, 150}, {200, 250, 300}}},
};
// decoder/temporal_noise_shaping.rs:84-137: filter bands {lo0, hi0, lo1, hi1} per bandwidth
static __device__ const int LC3C_TNSDEC10[5][4] = {{12, 80, 0, 0}, {12, 160, 0, 0}, {12, 240, 0, 0}, {12, 160, 160, 320}, {12, 200, 200, 400}};
static __device__ const int LC3C_TNSDEC75[5][4] = {{9, 60, 0, 0}, {9, 120, 0, 0}, {9, 180, 0, 0}, {9, 120, 120, 240}, {9, 150, 150, 300}};

// ------------------------------------------------------------------------------------------
// static tables selected by configuration
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Coalesced block copies between HBM and a wave's LDS working set.  n4 = number of 16-byte units; both sides 16-byte
// aligned.  All of a batch's loads are issued before the first use, so a copy costs one memory latency per 8 KB.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lc3_wave_copy_in16(void *lds_dst, const void *hbm_src, int n4, int lane) {
    lc3_i4 *d = (lc3_i4 *)lds_dst;
    LC3_HBM_CONST(lc3_i4) s = (LC3_HBM_CONST(lc3_i4))hbm_src;
    for (int i0 = 0; i0 < n4; i0 += 8 * LC3_WAVE) {
        lc3_i4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + LC3_WAVE * u + lane;
            if (i < n4) v[u] = s[i];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + LC3_WAVE * u + lane;
            if (i < n4) d[i] = v[u];
        }
    }
}
__device__ __forceinline__ void lc3_wave_copy_out16(void *hbm_dst, const void *lds_src, int n4, int lane) {
    lc3_i4 *d = (lc3_i4 *)hbm_dst;
    const lc3_i4 *s = (const lc3_i4 *)lds_src;
    for (int i0 = 0; i0 < n4; i0 += 8 * LC3_WAVE) {
        lc3_i4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + LC3_WAVE * u + lane;
            if (i < n4) v[u] = s[i];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + LC3_WAVE * u + lane;
            if (i < n4) d[i] = v[u];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Sequential f32 sums over LDS arrays.  The additions run in index order (the reference's order); the operands are
// fetched eight at a time so that a sum costs one LDS latency per eight terms instead of one per term.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float lc3_sum_seq(const float *a, int n, float acc) {
    // software-pipelined: the next eight operands are requested before the current eight are added (the additions are one
    // dependent chain; their operands' LDS latency is then hidden behind it).  Two register blocks take turns, so no block is
    // ever copied.
    float x[8], y[8];
    const int nb = n & ~7;
    int i = 0;
    if (nb > 0) {
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = a[u];
    }
    for (; i + 16 <= nb; i += 16) {
#pragma unroll
        for (int u = 0; u < 8; u++) y[u] = a[i + 8 + u];
#pragma unroll
        for (int u = 0; u < 8; u++) acc += x[u];
        if (i + 16 < nb) {
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = a[i + 16 + u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += y[u];
    }
    if (i < nb) {  // an odd number of blocks: the last one is in x
#pragma unroll
        for (int u = 0; u < 8; u++) acc += x[u];
    }
    for (i = nb; i < n; i++) acc += a[i];
    return acc;
}
__device__ __forceinline__ float lc3_dot_seq(const float *a, const float *b, int n, float acc) {
    float x[8], y[8], xn[8], yn[8];
    const int nb = n & ~7;
    int i = 0;
    if (nb > 0) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x[u] = a[u];
            y[u] = b[u];
        }
    }
    for (; i + 16 <= nb; i += 16) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            xn[u] = a[i + 8 + u];
            yn[u] = b[i + 8 + u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += x[u] * y[u];
        if (i + 16 < nb) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x[u] = a[i + 16 + u];
                y[u] = b[i + 16 + u];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += xn[u] * yn[u];
    }
    if (i < nb) {
#pragma unroll
        for (int u = 0; u < 8; u++) acc += x[u] * y[u];
    }
    for (i = nb; i < n; i++) acc += a[i] * b[i];
    return acc;
}

// first index of the maximum / minimum of a[0..n) under the reference's scan  `if (a[i] > best) { best = a[i]; idx = i; }`
// (strict comparison, starting from `best`/`idx`), operands fetched eight at a time
__device__ __forceinline__ void lc3_argmax_seq(const float *a, int n, float &best, int &idx) {
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = a[i + u];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (x[u] > best) { best = x[u]; idx = i + u; }
    }
    for (; i < n; i++)
        if (a[i] > best) { best = a[i]; idx = i; }
}
__device__ __forceinline__ void lc3_argmin_seq(const float *a, int n, float &best, int &idx) {
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = a[i + u];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (x[u] < best) { best = x[u]; idx = i + u; }
    }
    for (; i < n; i++)
        if (a[i] < best) { best = a[i]; idx = i; }
}
// a[from..to) *= g, four lines per round trip
__device__ __forceinline__ void lc3_scale_run(float *a, int from, int to, float g) {
    for (int k = from; k < to; k += 4) {
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = k + u < to ? a[k + u] : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (k + u < to) a[k + u] = x[u] * g;
    }
}

// Element i of the polyphase resampler table (encoder/long_term_post_filter.rs:152-166): row ph = i / stride,
// tap j = i % stride stands for k = j - lim of the reference loop, i.e. tab_resamp_filter[119 + p*k - ph] while
// that index lies strictly inside (-120, 120) and k <= lim; zero elsewhere.  A zero tap adds x * 0 = +-0 to the
// running f32 sum, which leaves it unchanged, so the padded sum equals the reference's guarded one bit for bit.
__device__ __forceinline__ float lc3_resamp_poly_value(int p, int lim, int stride, int i) {
    const int ph = i / stride, j = i - ph * stride;
    const int index_h = p * (j - lim) - ph;
    if (j > 2 * lim || index_h <= -120 || index_h >= 120) return 0.0f;
    return lc3_f(LC3T_TAB_RESAMP_FILTER_BITS, 119 + index_h);
}

template <class CC>
__device__ __forceinline__ const uint32_t *lc3_window_bits(const CC &c) {
    if (c.n_ms_10) {
        switch (c.fs_ind) {
        case 0: return LC3T_W_N80_10MS_BITS;
        case 1: return LC3T_W_N160_10MS_BITS;
        case 2: return LC3T_W_N240_10MS_BITS;
        case 3: return LC3T_W_N320_10MS_BITS;
        default: return LC3T_W_N480_10MS_BITS;
        }
    }
    switch (c.fs_ind) {
    case 0: return LC3T_W_N60_7P5MS_BITS;
    case 1: return LC3T_W_N120_7P5MS_BITS;
    case 2: return LC3T_W_N180_7P5MS_BITS;
    case 3: return LC3T_W_N240_7P5MS_BITS;
    default: return LC3T_W_N360_7P5MS_BITS;
    }
}
template <class CC>
__device__ __forceinline__ const uint16_t *lc3_band_index(const CC &c) {
    if (c.n_ms_10) {
        switch (c.fs_ind) {
        case 0: return LC3T_I_8000_10MS;
        case 1: return LC3T_I_16000_10MS;
        case 2: return LC3T_I_24000_10MS;
        case 3: return LC3T_I_32000_10MS;
        default: return LC3T_I_48000_10MS;
        }
    }
    switch (c.fs_ind) {
    case 0: return LC3T_I_8000_7P5MS;
    case 1: return LC3T_I_16000_7P5MS;
    case 2: return LC3T_I_24000_7P5MS;
    case 3: return LC3T_I_32000_7P5MS;
    default: return LC3T_I_48000_7P5MS;
    }
}
// band width of spectral line k (the divisor of apply_energy_estimation, encoder/modified_dct.rs:140-152); fills c.line_width
template <class CC>
__device__ __forceinline__ float lc3_line_width_value(const CC &c, int k) {
    const uint16_t *ifs = lc3_band_index(c);
    int b = 0;
    while (b + 1 < c.nb && (int)ifs[b + 1] <= k) b++;
    return (float)((int)ifs[b + 1] - (int)ifs[b]);
}

// band of spectral line k (0..nb-1), 255 beyond the coded bandwidth; fills c.line_band
template <class CC>
__device__ __forceinline__ int lc3_line_band_value(const CC &c, int k) {
    if (k >= c.ne) return 255;
    const uint16_t *ifs = lc3_band_index(c);
    int b = 0;
    while (b + 1 < c.nb && (int)ifs[b + 1] <= k) b++;
    return b;
}

// ------------------------------------------------------------------------------------------
// Complex FFT (common/kissfft.rs) and DCT-IV (common/dct_iv.rs), wave-parallel over butterflies.
// The reference's recursion kf_work (:86-131) is a decimation-in-time plan: a strided gather at
// the leaves (perm[]) followed by the butterfly stages innermost-first.  Butterflies of one
// stage touch disjoint elements, so one lane per butterfly is exact; the butterfly expression
// trees (:133-256) are kept verbatim.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ lc3_cpx lc3_cmul(lc3_cpx a, lc3_cpx b) {  // common/complex.rs:16-24
    lc3_cpx r;
    r.r = a.r * b.r - a.i * b.i;
    r.i = a.r * b.i + a.i * b.r;
    return r;
}
__device__ __forceinline__ lc3_cpx lc3_cadd(lc3_cpx a, lc3_cpx b) { lc3_cpx r; r.r = a.r + b.r; r.i = a.i + b.i; return r; }
__device__ __forceinline__ lc3_cpx lc3_csub(lc3_cpx a, lc3_cpx b) { lc3_cpx r; r.r = a.r - b.r; r.i = a.i - b.i; return r; }

// one butterfly of radix p: x[k] = element i + k m of the sub-transform (k < p) -> y[k], the reference's expression trees term by term
__device__ __forceinline__ void lc3_bfly_vals(const lc3_cpx *tw, int p, int fstride, int m, int i, const lc3_cpx (&x)[5], lc3_cpx (&y)[5]) {
    if (p == 4) {  // kissfft.rs:143-175
        lc3_cpx s0 = lc3_cmul(x[1], tw[i * fstride]);
        lc3_cpx s1 = lc3_cmul(x[2], tw[i * fstride * 2]);
        lc3_cpx s2 = lc3_cmul(x[3], tw[i * fstride * 3]);
        lc3_cpx f0 = x[0];
        lc3_cpx s5 = lc3_csub(f0, s1);
        f0 = lc3_cadd(f0, s1);
        lc3_cpx s3 = lc3_cadd(s0, s2);
        lc3_cpx s4 = lc3_csub(s0, s2);
        y[2] = lc3_csub(f0, s3);
        y[0] = lc3_cadd(f0, s3);
        y[1].r = s5.r + s4.i;
        y[1].i = s5.i - s4.r;
        y[3].r = s5.r - s4.i;
        y[3].i = s5.i + s4.r;
    } else if (p == 2) {  // :133-141
        lc3_cpx t = lc3_cmul(x[1], tw[i * fstride]);
        y[1] = lc3_csub(x[0], t);
        y[0] = lc3_cadd(x[0], t);
    } else if (p == 3) {  // :177-205
        lc3_cpx epi3 = tw[fstride * m];
        lc3_cpx s1 = lc3_cmul(x[1], tw[i * fstride]);
        lc3_cpx s2 = lc3_cmul(x[2], tw[i * fstride * 2]);
        lc3_cpx s3 = lc3_cadd(s1, s2);
        lc3_cpx s0 = lc3_csub(s1, s2);
        lc3_cpx fi = x[0];
        lc3_cpx fm;
        fm.r = fi.r - (s3.r * 0.5f);
        fm.i = fi.i - (s3.i * 0.5f);
        s0.r *= epi3.i;
        s0.i *= epi3.i;
        y[0] = lc3_cadd(fi, s3);
        y[2].r = fm.r + s0.i;
        y[2].i = fm.i - s0.r;
        y[1].r = fm.r - s0.i;
        y[1].i = fm.i + s0.r;
    } else {  // p == 5, :207-256
        lc3_cpx ya = tw[fstride * m], yb = tw[fstride * 2 * m];
        lc3_cpx s0 = x[0];
        lc3_cpx s1 = lc3_cmul(x[1], tw[i * fstride]);
        lc3_cpx s2 = lc3_cmul(x[2], tw[i * 2 * fstride]);
        lc3_cpx s3 = lc3_cmul(x[3], tw[i * 3 * fstride]);
        lc3_cpx s4 = lc3_cmul(x[4], tw[i * 4 * fstride]);
        lc3_cpx s7 = lc3_cadd(s1, s4), s10 = lc3_csub(s1, s4), s8 = lc3_cadd(s2, s3), s9 = lc3_csub(s2, s3);
        lc3_cpx s5, s6, s11, s12;
        y[0].r = s0.r + (s7.r + s8.r);
        y[0].i = s0.i + (s7.i + s8.i);
        s5.r = s0.r + (s7.r * ya.r) + (s8.r * yb.r);
        s5.i = s0.i + (s7.i * ya.r) + (s8.i * yb.r);
        s6.r = (s10.i * ya.i) + (s9.i * yb.i);
        s6.i = -(s10.r * ya.i) - (s9.r * yb.i);
        y[1] = lc3_csub(s5, s6);
        y[4] = lc3_cadd(s5, s6);
        s11.r = s0.r + (s7.r * yb.r) + (s8.r * ya.r);
        s11.i = s0.i + (s7.i * yb.r) + (s8.i * ya.r);
        s12.r = -(s10.i * yb.i) + (s9.i * ya.i);
        s12.i = (s10.r * yb.i) - (s9.r * ya.i);
        y[2] = lc3_cadd(s11, s12);
        y[3] = lc3_csub(s11, s12);
    }
}
// ... in place at element i of the sub-transform based at f[0]
__device__ __forceinline__ void lc3_bfly(lc3_cpx *f, const lc3_cpx *tw, int p, int fstride, int m, int i) {
    lc3_cpx x[5], y[5];
#pragma unroll
    for (int k = 0; k < 5; k++)
        if (k < p) x[k] = f[i + k * m];
    lc3_bfly_vals(tw, p, fstride, m, i, x, y);
#pragma unroll
    for (int k = 0; k < 5; k++)
        if (k < p) f[i + k * m] = y[k];
}

// DiscreteCosTransformIv::run (common/dct_iv.rs:49-67): in[0..nf) -> out[0..nf) through the complex work array wk[0..nf/2) (all LDS).
// `in` is consumed; wk may be `in` itself (IN_PLACE: the work array takes the input's place) or a third buffer; out may be `in` when wk
// is a third buffer.  Same operations in the same order as the reference, with two of its passes over the array folded away:
//   * the pre-twiddle (:53-56) and the leaf gather of kf_work (kissfft.rs:101-108) happen in the loads of the innermost butterfly
//     stage (m = 1: butterfly u takes the leaves p u .. p u + p - 1; at most 48 butterflies in every configuration, so one pass: every
//     lane has read its leaves before any lane writes, which is what lets the work array overlay the input);
//   * the post-twiddle (:62-66) happens in the stores of the outermost stage (blk = 0: butterfly i produces elements i + k m).
template <int IN_PLACE, class CC>
__device__ __forceinline__ void lc3_dct4_core(const CC &c, int lane, float *in, lc3_cpx *wk, float *out) {
    const int nf = c.nf, cnt = c.nfft;
    {   // innermost stage with the pre-twiddle and the gather in its loads
        const int s = c.n_stages - 1;
        const int p = c.radix[s], fstride = c.fstride[s], nb = cnt / p;  // m = 1
        // Lane l takes the butterfly whose leaves are l, l + nb, ... (the gather order of kf_work is a digit reversal: the leaves of ONE
        // butterfly lie nb apart, and every residue l < nb belongs to exactly one): the loads are then contiguous over the lanes instead
        // of a gather through perm[] that put three lanes on every bank, and the stores (butterfly u -> elements p u ...) go 3 p or so
        // elements apart.  lc3_fft_tables::leaf_bfly names the butterfly.
        lc3_cpx x[5], y[5];
        if (lane < nb) {
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (k < p) {
                    const int n = lane + k * nb;
                    lc3_cpx v;
                    v.r = in[2 * n];
                    v.i = in[nf - 2 * n - 1];
                    x[k] = lc3_cmul(LC3_DCT_TW(c)[n], v);
                }
            lc3_bfly_vals(LC3_FFT_TW(c), p, fstride, 1, 0, x, y);
        }
        const int u = lane < nb ? LC3_FFT_LEAF_BFLY(c, lane) : 0;  // (a table in LDS, not the array being overlaid)
        if (IN_PLACE) LC3_SYNC();
        if (lane < nb) {
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (k < p) wk[u * p + k] = y[k];
        }
        LC3_SYNC();
    }
    for (int s = c.n_stages - 2; s >= 1; s--) {  // the stages between
        const int p = c.radix[s], m = c.m[s], fstride = c.fstride[s];
        const int nb = cnt / p, nblk = nb / m;  // nblk sub-transforms of p m elements, m butterflies each
        // Which butterfly a lane takes is free (butterflies of a stage touch disjoint elements).  Neighbouring lanes take the SAME butterfly
        // index i of neighbouring sub-transforms: their elements lie p m apart (15, 45, 10 ... complex numbers: odd or twice odd, so a
        // pass of sixteen 8-byte accesses spreads over the LDS banks) and they share their twiddles.  With neighbouring lanes on
        // neighbouring butterflies of one sub-transform, i + k m for several sub-transforms at once, the radix-3 stage of the 240-point
        // transform ran three deep in bank conflicts (modelled: 256 -> 88 LDS cycles for its two middle stages).
        for (int u = lane; u < nb; u += LC3_WAVE) {
            const int i = u / nblk, blk = u - i * nblk;  // (compile-time plans: a multiplication and a shift)
            lc3_bfly(wk + blk * p * m, LC3_FFT_TW(c), p, fstride, m, i);
        }
        LC3_SYNC();
    }
    {   // outermost stage with the post-twiddle in its stores
        const int p = c.radix[0], m = c.m[0], fstride = c.fstride[0];  // blk = 0
        for (int i = lane; i < m; i += LC3_WAVE) {
            lc3_cpx x[5], y[5];
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (k < p) x[k] = wk[i + k * m];
            lc3_bfly_vals(LC3_FFT_TW(c), p, fstride, m, i, x, y);
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (k < p) {
                    const int n = i + k * m;
                    const lc3_cpx t = lc3_cmul(LC3_DCT_TW(c)[n], y[k]);
                    out[2 * n] = t.r * 2.0f;
                    out[nf - 2 * n - 1] = -t.i * 2.0f;
                }
        }
        LC3_SYNC();
    }
}
#ifndef LC3_DCT4_CORE  // (lc3gpu.hip: the run-time configuration view picks a compile-time plan by frame length)
#define LC3_DCT4_CORE(IN_PLACE, c, lane, in, wk, out) lc3_dct4_core<IN_PLACE>(c, lane, in, wk, out)
#endif
// on buf[0..nf) in place, scratch fb (nf/2 complex; fa is no longer needed)
template <class CC>
__device__ __forceinline__ void lc3_dct4_wave(const CC &c, int lane, float *buf, lc3_cpx *fa, lc3_cpx *fb) {
    (void)fa;
    LC3_DCT4_CORE(0, c, lane, buf, fb, buf);
}
// a[0..nf) -> b[0..nf) with two buffers: a is destroyed (it serves as the complex work array once the innermost stage has read it)
template <class CC>
__device__ __forceinline__ void lc3_dct4_wave_ab(const CC &c, int lane, float *a, float *b) {
    LC3_DCT4_CORE(1, c, lane, a, (lc3_cpx *)a, b);
}
