// lc3gpu -- kernels and C ABI of the MI355X-native batched LC3 codec.  gfx950 only.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see __graft_entry__.build()).
//
// Six kernels.  Stream kernels (lc3_enc_front_kernel, lc3_enc_back_kernel, lc3_decode_kernel): one workgroup = four
// wavefronts = four streams; a launch covers `n_streams` streams x `n_frames` frames, each wave loads its stream's
// state scalars from HBM into LDS once, runs the frames in time order and writes the state back.  Frame kernels
// (lc3_sns_vq_kernel, lc3_pack_kernel, lc3_parse_kernel): one LANE per frame for the serial, frame-local stages.
// Stages meet in HBM "planes" (one contiguous column of words per frame).  blockIdx -> stream / frame is the identity:
// streams share nothing but read-only tables, so XCD placement only affects table L2 hits.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/lc3gpu.h"

// ---- translation units ------------------------------------------------------------------------------------------------------------
// The library is this one source compiled either whole (no LC3_TU_KIND: the four configuration views of LC3_FOR_EACH_VIEW_BASE; the
// diagnostic and the experiment builds) or as parts, side by side (lc3-codec_amd/api.py; device code generation for ~250 kernel bodies is
// what a build spends its time on, and it is serial inside one unit):
//   -DLC3_TU_KIND=0                     the MAIN unit: the host side and, as device code, exactly what the whole-source build holds --
//                                       the kernels of the four base views and of the run-time view, the mixed kernels with those bodies
//   -DLC3_TU_KIND=1 -DLC3_TU_INDEX=v    the three encoder kernels of one of the other views (lc3_cfg_views.h, v = 5 ..)
//   -DLC3_TU_KIND=3 -DLC3_TU_INDEX=v    its eight decoder kernels
//   -DLC3_TU_KIND=2 -DLC3_TU_INDEX=k    the k-th of the eight mixed-configuration kernels with a body for EVERY view (these are the ones a
//                                       multi-unit library launches: <name>_all; the main unit's stay unused)
// Why the main unit repeats the whole-source module instead of holding only what nobody else does: the front half of the headline
// configuration sits at the edge of its 128 registers, and on which side it lands depends on what else its module holds (117 registers
// and no spill there; 128 and 47 spilled in a unit of its own, 0.448 -> 0.481 ms; profiles/r04_translation_units.txt).
// Every unit but the main one has its own copy of the constant table and of the device-filled tables (`static` there):
// lc3_tu_register_<kind>_<index> fills them, the main unit calls every unit's.
#ifdef LC3_TU_KIND
#define LC3_MULTI_TU 1
#ifndef LC3_TU_INDEX
#define LC3_TU_INDEX 0
#endif
#else
#define LC3_MULTI_TU 0
#define LC3_TU_KIND 0
#define LC3_TU_INDEX 0
#endif
#if LC3_TU_KIND == 0
#define LC3_TU_STATIC
#else
#define LC3_TU_STATIC static
#endif
#define LC3_IN_HOST_TU (LC3_TU_KIND == 0)
#define LC3_IN_MIXED_TU(k) (LC3_TU_KIND == 0 || (LC3_TU_KIND == 2 && LC3_TU_INDEX == (k)))
// a mixed kernel's name: in its own unit <name>_all (a body per view), in the main unit the plain name (base views); what the host launches
#if LC3_TU_KIND == 2
#define LC3_MIXED_KERNEL(name) name##_all
#else
#define LC3_MIXED_KERNEL(name) name
#endif
#if LC3_MULTI_TU
#define LC3_MIXED_LAUNCH(name) name##_all
#else
#define LC3_MIXED_LAUNCH(name) name
#endif
#define LC3_CAT_(a, b) a##b
#define LC3_CAT(a, b) LC3_CAT_(a, b)

// A workgroup is LC3_WG_WAVES wavefronts, one stream each.  LC3_SYNC orders the LDS traffic of ONE wave (its lanes
// exchange data through the stream's LDS working set): the hardware executes a wave's LDS instructions in order, so
// only the compiler has to be kept from moving accesses across the point.  LC3_SERIAL_BEGIN/END bracket code that is
// serial per stream: the workgroup meets at a barrier, one wave (rotating with `phase`, so that every SIMD gets its
// share) runs the block for all streams of the workgroup at once -- lane q*K+sub works on stream q with `L` rebound
// to that stream's working set -- and a second barrier releases the others.  Everything such a block reads or writes
// therefore lives in LDS (or is a launch-uniform value); per-stream register values must be staged through L first.
#ifndef LC3_WG_WAVES
#define LC3_WG_WAVES 4
#endif
#define LC3_SYNC()                                            \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#define LC3_WAVE_ID() ((int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)))
// lanes of a wave exchanging data through HBM (state rings): wait for the wave's outstanding stores/loads; the CU's
// vector L1 serves all lanes of the wave, so no cache maintenance is needed
#define LC3_HBM_FENCE()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)
#ifndef LC3_BACK_WAVES
#define LC3_BACK_WAVES 4
#endif
#ifndef LC3_FRONT_WAVES
#define LC3_FRONT_WAVES 4  // waves per SIMD the register allocation of the front half aims at (tuning experiments)
#endif
#ifndef LC3_FRONT_WAVES_75
#define LC3_FRONT_WAVES_75 3  // ... of the 48 kHz / 7.5 ms view (168 registers: no spills; at 4 it spills 47)
#endif
#ifndef LC3_SYNTH_WAVES
#define LC3_SYNTH_WAVES 4
#endif
// Wave priority of the lane-per-frame kernels (s_setprio 0..3).  Their waves walk one dependent chain per lane and leave most issue
// slots empty; beside another handle's wave-per-stream kernels (the two-stream arrangement) a raised priority lets them keep their own
// pace -- and leave the compute unit sooner -- while the other kernel's waves take the slots in between.
// Round 4 (one or two caller streams): no difference between 0, 1 and 3.  Round 5, four caller streams (`quad`, bench.py's default): 1 is worth
// +1.5 % sustained (55.2 -> 56.0 M frames/s; profiles/r05_arrangements.txt), 3 the same.
#ifndef LC3_LANE_PRIO
#define LC3_LANE_PRIO 1
#endif
#define LC3_LANE_KERNEL_BEGIN() do { if (LC3_LANE_PRIO) __builtin_amdgcn_s_setprio(LC3_LANE_PRIO); } while (0)
#ifndef LC3_SPEC_IN_LDS
#define LC3_SPEC_IN_LDS 1
#endif
// per-workgroup LDS copy of the spectral-model tables used by the analysis kernel's bit estimate (6.1 KB)
struct lc3_spec_tables {
    uint8_t lookup[4096];
    uint16_t bits[64 * 17];
};
__shared__ lc3_spec_tables lc3_spec_tab;
#if LC3_SPEC_IN_LDS
#define LC3_SPEC_LOOKUP(i) ((int)lc3_spec_tab.lookup[(i)])
#define LC3_SPEC_BITS(p, j) ((uint32_t)lc3_spec_tab.bits[(p) * 17 + (j)])
#endif
// FFT / DCT-IV twiddles and the gather order: one LDS copy per workgroup instead of HBM table reads inside every stage
#define LC3_FFT_TABLES_IN_LDS 1
#define LC3_LDS_DECL(T, arr) __shared__ T arr[LC3_WG_WAVES];
#define LC3_LDS_PARAM(T)
#define LC3_LDS_PASS
#define LC3_LDS_BIND(T, arr) T &L = arr[LC3_WAVE_ID()]
// (LC3_NO_GATHER, experiment: every wave runs the serial blocks of its own stream on its own lanes -- the definitions of lc3_dev_common.h,
// no workgroup barriers -- instead of one wave running them for the workgroup's four streams)
#ifndef LC3_NO_GATHER
// (LC3_SERIAL_PRIO, experiment: the wave that runs a gathered block does so at a raised wave priority -- three waves of its workgroup wait for
// it while the SIMD's other workgroups compete with it for issue slots; 0 = off)
#ifndef LC3_SERIAL_PRIO
#define LC3_SERIAL_PRIO 0
#endif
#define LC3_SERIAL_PRIO_UP() do { if (LC3_SERIAL_PRIO) __builtin_amdgcn_s_setprio(LC3_SERIAL_PRIO); } while (0)
#define LC3_SERIAL_PRIO_DOWN() do { if (LC3_SERIAL_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)
#define LC3_SERIAL_BEGIN(T, L, lane, phase, K)                                               \
    {                                                                                        \
        T *lc3_wg_base_ = &(L) - LC3_WAVE_ID();                                              \
        __syncthreads();                                                                     \
        if (LC3_WAVE_ID() == ((phase) % LC3_WG_WAVES)) LC3_SERIAL_PRIO_UP();                 \
        if (LC3_WAVE_ID() == ((phase) % LC3_WG_WAVES) && (lane) < LC3_WG_WAVES * (K)) {      \
            T &L = lc3_wg_base_[(lane) / (K)];                                               \
            const int sub = (lane) % (K);                                                    \
            (void)sub;
#define LC3_SERIAL_END \
        }              \
        LC3_SERIAL_PRIO_DOWN(); \
        __syncthreads(); \
    }
// The same for blocks of up to 64 lanes per stream (K x streams beyond one wave): virtual lane v = lane + 64 j runs on the j-th wave after
// wave `phase`, stream v / K, sub = v % K.  (K = 17: wave `phase` carries 64 of the workgroup's 68 lanes, the next one 4.)
#define LC3_SERIAL_WIDE_BEGIN(T, L, lane, phase, K)                                                           \
    {                                                                                                         \
        T *lc3_wg_base_ = &(L) - LC3_WAVE_ID();                                                               \
        __syncthreads();                                                                                      \
        const int lc3_v_ = (lane) + 64 * ((LC3_WAVE_ID() + LC3_WG_WAVES - ((phase) % LC3_WG_WAVES)) % LC3_WG_WAVES); \
        if (lc3_v_ - (lane) < LC3_WG_WAVES * (K)) LC3_SERIAL_PRIO_UP();                                       \
        if (lc3_v_ < LC3_WG_WAVES * (K)) {                                                                    \
            T &L = lc3_wg_base_[lc3_v_ / (K)];                                                                \
            const int sub = lc3_v_ % (K);                                                                     \
            (void)sub;
#endif  // LC3_NO_GATHER
#define LC3_HBM_CONST(T) const __attribute__((address_space(1))) T *
#define LC3_HBM(T) __attribute__((address_space(1))) T *
#define LC3_UNIFORM_I32(x) __builtin_amdgcn_readfirstlane((int)(x))
// a pointer the caller knows to be wave-uniform, in scalar registers: loads through it take the (scalar base + 32-bit lane offset) form
#define LC3_UNIFORM_PTR(T, p)                                                                                             \
    ((T)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)(uintptr_t)(p) >> 32)) << 32) | \
         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(p))))
#define LC3_KEEP_SCALAR(x) asm("" : "+v"(x))
// An LDS pointer as a base register the compiler knows nothing about: the accesses p[0], p[1], ... then carry their distance as the
// instruction's immediate offset.  Without it the compiler folds the array's position inside the working set (a constant beyond the 8-bit
// offsets of ds_read2_b32) into every access and spends one v_add_u32 per LDS instruction on it.
static __device__ __forceinline__ const float *lc3_lds_base(const float *p) {
    unsigned a = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p;
    asm("" : "+v"(a));
    return (const float *)(const __attribute__((address_space(3))) float *)(uintptr_t)a;
}
#define LC3_LDS_BASE(p) lc3_lds_base(p)
// ... of a known alignment (the mask is what tells the compiler: it tracks known-zero address bits, not assumptions, through the cast)
template <unsigned BYTES>
static __device__ __forceinline__ const float *lc3_lds_base_aligned(const float *p) {
    unsigned a = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p;
    asm("" : "+v"(a));
    a &= ~(BYTES - 1u);
    return (const float *)(const __attribute__((address_space(3))) float *)(uintptr_t)a;
}
#define LC3_LDS_BASE_ALIGNED(p, bytes) lc3_lds_base_aligned<bytes>(p)
#define LC3_KEEP_PER_FRAME(x) asm volatile("" : "+v"(x))
// repeat builds (lc3_dev_experiments.h): a 2 in a scalar register that the compiler cannot fold
static __device__ __forceinline__ int lc3_exp_two() {
    int n = 2;
    asm volatile("" : "+s"(n) : : "memory");
    return n;
}
#define LC3_EXP_TWO() lc3_exp_two()
static __device__ __forceinline__ void lc3_exp_clobber() { asm volatile("" : : : "memory"); }
#define LC3_EXP_CLOBBER() lc3_exp_clobber()
// product of two values below 2^24 (range-coder steps): the full-rate 24-bit multiplier instead of the quarter-rate v_mul_lo_u32
#define LC3_MUL24(a, b) __umul24((a), (b))
#define LC3_WAVE_ANY(pred) (__ballot((pred) != 0) != 0ull)
// the producer / consumer link of the parser (lc3_dev_dec_parse.h): words in LDS read and written by two waves of a workgroup.  A wave's
// LDS operations execute in issue order; what these add is that the compiler keeps its own order around them
// (LDS address space spelled out: through a generic pointer these become flat accesses with system-scope cache bits; the loaded value is
// the same in every lane -- one word per wave pair -- and is handed on in a scalar register, so the loops it controls stay wave-uniform)
#define LC3_PC_WORD(p) ((volatile __attribute__((address_space(3))) int *)(p))
static __device__ __forceinline__ int lc3_pc_load_(const int *p) {
    asm volatile("" ::: "memory");
    const int v = *LC3_PC_WORD(p);
    asm volatile("" ::: "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
#define LC3_PC_STORE(p, v)                 \
    do {                                   \
        asm volatile("" ::: "memory");     \
        *LC3_PC_WORD(p) = (v);             \
        asm volatile("" ::: "memory");     \
    } while (0)
#define LC3_PC_LOAD(p) lc3_pc_load_((const int *)(p))
#define LC3_PC_PAUSE() __builtin_amdgcn_s_sleep(2)
#define LC3_PC_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
#define LC3_PC_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup")
#define LC3_LANEWAVE_MAX(v) lc3_wave_max_i32((v), 0)
// 10^x tables of the two argument families the codec uses (lc3_dev_common.h: LC3_POW10_GG / LC3_POW10_TILT), filled on the device
// by lc3_pow10f itself when a device's first configuration is registered
LC3_TU_STATIC __device__ float lc3_pow10_gg_tab[512];       // [k + 256] = 10^(k / 28), k = gg_ind + gg_off
LC3_TU_STATIC __device__ float lc3_pow10_tilt_tab[5 * 64];  // [fs_ind * 64 + b] = 10^(b * (g_tilt[fs_ind] / 630))
#define LC3_POW10_GG(k) ((unsigned)((k) + 256) < 512u ? lc3_pow10_gg_tab[(k) + 256] : lc3_pow10f((float)(k) / 28.0f))
#define LC3_POW10_TILT(fs_ind, b) (lc3_pow10_tilt_tab[(fs_ind) * 64 + (b)])
// the 17 quantised TNS reflection coefficients in the encoder's form of the step (lc3_dev_common.h: LC3_TNS_SIN_*)
LC3_TU_STATIC __device__ float lc3_tns_sin_tab[17];
#define LC3_TNS_SIN_ENC(ri) ((unsigned)(ri) < 17u ? lc3_tns_sin_tab[(ri)] : lc3_tns_sin_enc_value(ri))
// (the decoder's lane-per-frame parser evaluates the routine: a per-lane table fetch from memory costs it more than the arithmetic)
#define LC3_TNS_SIN_DEC(ri) lc3_tns_sin_dec_value(ri)
// x / d for many x and one d: the compiler's f32 division is v_div_scale x2, v_rcp, two Newton steps on the reciprocal,
// q = n * r, two residual corrections, v_div_fmas, v_div_fixup; scale and fix-up only act on zero / infinite / NaN operands,
// a denominator outside 2^+-126, an exponent difference of 96 or more, and numerators below 2^-103 or quotients below 2^-126.
// With d a finite normal number (the global gain, 1.8e-9 .. 1.5e5) and |x| < 2^60 only the last two can occur, and a numerator
// below 2^-103 gives a quotient below 2^-74, which the callers (quantiser: trunc(q + 0.375)) cannot tell from 0 either way; what
// is left is this sequence with the reciprocal refined once per d.  Measured on the device over 2^24 random pairs of that range
// (tests/test_gpu_parity.py::test_device_math_on_the_device): bit-identical to the IEEE quotient for every |x| >= 2^-103, one
// unit in the last place off for 0.13 % of the numerators below 2^-110.
#define LC3_UNIFORM_DIV 1
struct lc3_divisor { float d, r; };
__device__ __forceinline__ lc3_divisor lc3_divisor_make(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    lc3_divisor v = {d, r};
    return v;
}
__device__ __forceinline__ float lc3_div_by(float x, const lc3_divisor &v) {
    float q = x * v.r;
    float e = __builtin_fmaf(-v.d, q, x);
    q = __builtin_fmaf(e, v.r, q);
    e = __builtin_fmaf(-v.d, q, x);
    return __builtin_fmaf(e, v.r, q);
}
#define LC3_DCT4_CORE(IN_PLACE, c, lane, in, wk, out) lc3_dct4_select<IN_PLACE>(c, lane, in, wk, out)
struct lc3_cpx;
template <int IN_PLACE, class CC>
__device__ __forceinline__ void lc3_dct4_select(const CC &c, int lane, float *in, lc3_cpx *wk, float *out);
#include "lc3_dev_common.h"
// ---- configuration slots ----------------------------------------------------------------------------------------
// Every (sampling rate, frame duration) pair owns one slot of a __constant__ table; handles register their
// configuration there (cfg_acquire below) and the kernels pass the slot number down to the stage functions, which
// bind `c` to the slot through a wave-uniform index: every field is then a scalar (s_load) read that no LDS or HBM
// store can alias, instead of a load from a by-value copy of the struct in scratch.
#define LC3_CFG_SLOTS 12
LC3_TU_STATIC __constant__ lc3_cfg lc3_cfg_table[LC3_CFG_SLOTS];
// Configuration views.  A kernel (and every stage function under it) is instantiated once per view: lc3_cfg_any binds
// `c` to the slot's entry of the constant table; lc3_cfg_48k10 -- the configuration the headline benchmark runs -- carries
// the integers of that configuration as compile-time constants (loop bounds, index arithmetic and the FFT plan fold at
// compile time: back half -7 %, synthesis -27 %) and takes only the device pointers from the table.  cfg_acquire checks the
// constants against the plan it computes and falls back to the run-time view if they ever disagree.
template <class CV> struct lc3_cfg_slot { int id; };
struct lc3_cfg_any {
    typedef const lc3_cfg &bind_t;
    static __device__ __forceinline__ bind_t bind(const lc3_cfg &r) { return r; }
};
// One struct per view (lc3_cfg_views.h, generated by tools/gen_views.py): the configuration's integers as constants, the
// device pointers taken from the constant table's slot, and matches() -- host side -- to check the constants against the plan
// cfg_acquire computes.
#define LC3_ARR(...) {__VA_ARGS__}
#define LC3_DEFINE_CFG_VIEW(NAME, FS, FS_IND, NF, NE, NB, Z, N10, NFFT, NST, RADIX, M, FSTRIDE, INV_M, LEN12, LEN6, DELAY12, P_UP, HIST,  \
                            RSCALE, RLIM, RNT, RSTRIDE, INV_P, L_DEN, L_NUM, NMB, NORM, S25)                                             \
    struct NAME {                                                                                                                      \
        static constexpr int fs = FS, fs_ind = FS_IND, nf = NF, ne = NE, nb = NB, z = Z, n_ms_10 = N10, nfft = NFFT, n_stages = NST;    \
        static constexpr int radix[6] = RADIX, m[6] = M, fstride[6] = FSTRIDE, inv_m[6] = INV_M;                                        \
        static constexpr int len12 = LEN12, len6 = LEN6, delay12 = DELAY12, p_up = P_UP, hist = HIST;                                   \
        static constexpr float resamp_scale = RSCALE;                                                                                  \
        static constexpr int resamp_lim = RLIM, resamp_nt = RNT, resamp_stride = RSTRIDE, inv_p = INV_P;                                \
        static constexpr int l_den = L_DEN, l_num = L_NUM, num_mem_blocks = NMB, norm = NORM, s25 = S25;                                \
        const lc3_cpx *fft_tw, *dct_tw;                                                                                                \
        const uint16_t *perm;                                                                                                          \
        const float *resamp_poly, *line_width;                                                                                         \
        const uint8_t *line_band;                                                                                                      \
        const void *stage_image;                                                                                                       \
        __device__ __forceinline__ explicit NAME(const lc3_cfg &r)                                                                     \
            : fft_tw(r.fft_tw), dct_tw(r.dct_tw), perm(r.perm), resamp_poly(r.resamp_poly), line_width(r.line_width),                  \
              line_band(r.line_band), stage_image(r.stage_image) {}                                                                    \
        typedef const NAME bind_t;                                                                                                     \
        static __device__ __forceinline__ NAME bind(const lc3_cfg &r) { return NAME(r); }                                              \
        static bool matches(const lc3_cfg &r) {                                                                                        \
            bool ok = r.fs == fs && r.fs_ind == fs_ind && r.nf == nf && r.ne == ne && r.nb == nb && r.z == z && r.n_ms_10 == n_ms_10 && \
                      r.nfft == nfft && r.n_stages == n_stages && r.len12 == len12 && r.len6 == len6 && r.delay12 == delay12 &&         \
                      r.p_up == p_up && r.hist == hist && r.resamp_scale == resamp_scale && r.resamp_lim == resamp_lim &&               \
                      r.resamp_nt == resamp_nt && r.resamp_stride == resamp_stride && r.inv_p == inv_p && r.l_den == l_den &&           \
                      r.l_num == l_num && r.num_mem_blocks == num_mem_blocks && r.norm == norm && r.s25 == s25;                         \
            for (int i = 0; i < 6; i++)                                                                                                \
                ok = ok && r.radix[i] == radix[i] && r.m[i] == m[i] && r.fstride[i] == fstride[i] && r.inv_m[i] == inv_m[i];            \
            return ok;                                                                                                                 \
        }                                                                                                                              \
    };
// an FFT / DCT-IV plan by frame length (lc3_cfg_views.h): the integers lc3_dct4_core reads of a configuration, as constants
#define LC3_DEFINE_FFT_PLAN(NAME, NF, NFFT, NST, RADIX, M, FSTRIDE, INV_M)                         \
    struct NAME {                                                                                 \
        static constexpr int nf = NF, nfft = NFFT, n_stages = NST;                                \
        static constexpr int radix[6] = RADIX, m[6] = M, fstride[6] = FSTRIDE, inv_m[6] = INV_M;  \
    };
#include "lc3_cfg_views.h"
// LC3_FOR_EACH_VIEW / LC3_LAUNCH_CASES: the views the HOST knows (every view in a multi-unit library); LC3_VIEW_CASES: the bodies a mixed
// kernel of THIS unit carries
#if LC3_MULTI_TU
#define LC3_FOR_EACH_VIEW(X) LC3_FOR_EACH_VIEW_ALL(X)
#define LC3_LAUNCH_CASES LC3_LAUNCH_CASES_ALL
#define LC3_N_VIEWS LC3_N_VIEWS_ALL
#else
#define LC3_FOR_EACH_VIEW(X) LC3_FOR_EACH_VIEW_BASE(X)
#define LC3_LAUNCH_CASES LC3_LAUNCH_CASES_BASE
#define LC3_N_VIEWS LC3_N_VIEWS_BASE
#endif
#if LC3_TU_KIND == 2
#define LC3_VIEW_CASES LC3_VIEW_CASES_ALL
#else
#define LC3_VIEW_CASES LC3_VIEW_CASES_BASE
#endif
// The DCT-IV of a wave (lc3_dct4_wave / lc3_dct4_wave_ab, lc3_dev_common.h).  A compile-time configuration view IS its plan.  The run-time
// view picks the plan of its frame length -- with the radices, strides and loop bounds as run-time values the transform took 0.130 ms
// against 0.034 ms in the synthesis kernel and 0.193 against 0.106 ms in the front half (48 kHz sizes, profiles/r04_knockout_generic.txt)
template <int IN_PLACE, class CC>
__device__ __forceinline__ void lc3_dct4_select(const CC &c, int lane, float *in, lc3_cpx *wk, float *out) {
    lc3_dct4_core<IN_PLACE>(c, lane, in, wk, out);
}
template <int IN_PLACE>
__device__ __forceinline__ void lc3_dct4_select(const lc3_cfg &c, int lane, float *in, lc3_cpx *wk, float *out) {
#ifdef LC3_TABLES_IN_GLOBAL  // (experiment build: the plan structs carry no table pointer; the run-time view runs its run-time plan)
    lc3_dct4_core<IN_PLACE>(c, lane, in, wk, out);
#else
    switch (c.nf) {
    case 480: lc3_dct4_core<IN_PLACE>(lc3_fft_plan_480(), lane, in, wk, out); break;
    case 360: lc3_dct4_core<IN_PLACE>(lc3_fft_plan_360(), lane, in, wk, out); break;
    case 320: lc3_dct4_core<IN_PLACE>(lc3_fft_plan_320(), lane, in, wk, out); break;
    case 240: lc3_dct4_core<IN_PLACE>(lc3_fft_plan_240(), lane, in, wk, out); break;
    case 180: lc3_dct4_core<IN_PLACE>(lc3_fft_plan_180(), lane, in, wk, out); break;
    case 160: lc3_dct4_core<IN_PLACE>(lc3_fft_plan_160(), lane, in, wk, out); break;
    case 120: lc3_dct4_core<IN_PLACE>(lc3_fft_plan_120(), lane, in, wk, out); break;
    case 80: lc3_dct4_core<IN_PLACE>(lc3_fft_plan_80(), lane, in, wk, out); break;
    case 60: lc3_dct4_core<IN_PLACE>(lc3_fft_plan_60(), lane, in, wk, out); break;
    default: lc3_dct4_core<IN_PLACE>(c, lane, in, wk, out); break;
    }
#endif
}
#undef LC3_CFG_TEMPLATE
#undef LC3_CFG_TEMPLATE_AND
#undef LC3_CFG_PARAM
#undef LC3_CFG_BIND
#undef LC3_CFG_PASS
#define LC3_CFG_TEMPLATE template <class CV>
#define LC3_CFG_TEMPLATE_AND(...) template <class CV, __VA_ARGS__>
#define LC3_CFG_PARAM lc3_cfg_slot<CV> cslot
#define LC3_CFG_BIND typename CV::bind_t c = CV::bind(lc3_cfg_table[__builtin_amdgcn_readfirstlane(cslot.id)])
#define LC3_CFG_PASS cslot

// ---- wave-level primitives used by the stage code (64 lanes, results wave-uniform unless noted) -----------------
// Integer max / sum over the wave and exclusive prefix sum (any evaluation order is exact for integers).  Built on
// DPP lane permutes (no LDS crossbar round trips): four steps reduce/scan inside each row of 16 lanes, the four rows
// are then combined through v_readlane (reductions) or row broadcasts (scan).
#define LC3_DPP(old, src, ctrl, row_mask, bound) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (row_mask), 0xf, (bound))
#define LC3_DPP_QUAD_1032 0xB1
#define LC3_DPP_QUAD_2301 0x4E
#define LC3_DPP_ROW_HALF_MIRROR 0x141
#define LC3_DPP_ROW_MIRROR 0x140
#define LC3_DPP_ROW_SHR(n) (0x110 + (n))
#define LC3_DPP_ROW_BCAST15 0x142
#define LC3_DPP_ROW_BCAST31 0x143
__device__ __forceinline__ int lc3_wave_max_i32(int v, int lane) {
    (void)lane;
    int w;
    w = LC3_DPP(v, v, LC3_DPP_QUAD_1032, 0xf, false); v = w > v ? w : v;
    w = LC3_DPP(v, v, LC3_DPP_QUAD_2301, 0xf, false); v = w > v ? w : v;
    w = LC3_DPP(v, v, LC3_DPP_ROW_HALF_MIRROR, 0xf, false); v = w > v ? w : v;
    w = LC3_DPP(v, v, LC3_DPP_ROW_MIRROR, 0xf, false); v = w > v ? w : v;
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}
__device__ __forceinline__ uint32_t lc3_wave_sum_u32(uint32_t u, int lane) {
    (void)lane;
    int v = (int)u;
    v += LC3_DPP(v, v, LC3_DPP_QUAD_1032, 0xf, false);
    v += LC3_DPP(v, v, LC3_DPP_QUAD_2301, 0xf, false);
    v += LC3_DPP(v, v, LC3_DPP_ROW_HALF_MIRROR, 0xf, false);
    v += LC3_DPP(v, v, LC3_DPP_ROW_MIRROR, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane(v, 0) + (uint32_t)__builtin_amdgcn_readlane(v, 16) +
           (uint32_t)__builtin_amdgcn_readlane(v, 32) + (uint32_t)__builtin_amdgcn_readlane(v, 48);
}
// f32 sum over the wave in an UNSPECIFIED order (a tree): only for decisions that are guarded against the rounding
// difference to the reference's sequential sum (see lc3_enc_quant)
__device__ __forceinline__ float lc3_wave_sum_f32_any(float v, int lane) {
    (void)lane;
    int w;
    w = LC3_DPP(0, __builtin_bit_cast(int, v), LC3_DPP_QUAD_1032, 0xf, false); v += __builtin_bit_cast(float, w);
    w = LC3_DPP(0, __builtin_bit_cast(int, v), LC3_DPP_QUAD_2301, 0xf, false); v += __builtin_bit_cast(float, w);
    w = LC3_DPP(0, __builtin_bit_cast(int, v), LC3_DPP_ROW_HALF_MIRROR, 0xf, false); v += __builtin_bit_cast(float, w);
    w = LC3_DPP(0, __builtin_bit_cast(int, v), LC3_DPP_ROW_MIRROR, 0xf, false); v += __builtin_bit_cast(float, w);
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
// bit l of the result = lane l's predicate (wave-uniform)
// the value lane - 1 holds (lane 0: unspecified); one DPP move across the whole wave
__device__ __forceinline__ float lc3_wave_shr1_f32(float v, int lane) {
    (void)lane;
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    return __builtin_bit_cast(float, r);
}
// integer form; lane 0 receives 0
__device__ __forceinline__ int lc3_wave_shr1_i32(int v, int lane) {
    (void)lane;
    return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// the value lane + 1 holds; lane 63 receives 0
__device__ __forceinline__ int lc3_wave_shl1_i32(int v, int lane) {
    (void)lane;
    return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
// the value lane `src` holds (src the same on every lane), on every lane
__device__ __forceinline__ float lc3_wave_read_f32(float v, int src, int lane) {
    (void)lane;
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), __builtin_amdgcn_readfirstlane(src)));
}
__device__ __forceinline__ int lc3_wave_read_i32(int v, int src, int lane) {
    (void)lane;
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}
// lane 0's value on every lane (all lanes active): a scalar register, no LDS round trip
__device__ __forceinline__ int lc3_wave_bcast0_i32(int v, int lane) {
    (void)lane;
    return __builtin_amdgcn_readlane(v, 0);
}
__device__ __forceinline__ float lc3_wave_bcast0_f32(float v, int lane) {
    (void)lane;
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
}
__device__ __forceinline__ unsigned long long lc3_wave_ballot(int pred, int lane) {
    (void)lane;
    return __ballot(pred);
}
// returns the sum over lanes < lane (per lane)
__device__ __forceinline__ uint32_t lc3_wave_exscan_u32(uint32_t u, int lane) {
    (void)lane;
    int inc = (int)u;
    inc += LC3_DPP(0, inc, LC3_DPP_ROW_SHR(1), 0xf, true);  // lanes shifted in from outside the row read 0
    inc += LC3_DPP(0, inc, LC3_DPP_ROW_SHR(2), 0xf, true);
    inc += LC3_DPP(0, inc, LC3_DPP_ROW_SHR(4), 0xf, true);
    inc += LC3_DPP(0, inc, LC3_DPP_ROW_SHR(8), 0xf, true);
    inc += LC3_DPP(0, inc, LC3_DPP_ROW_BCAST15, 0xa, false);  // rows 1,3 += total of rows 0,2
    inc += LC3_DPP(0, inc, LC3_DPP_ROW_BCAST31, 0xc, false);  // rows 2,3 += total of rows 0..1
    return (uint32_t)inc - u;
}

#ifdef LC3_PROFILE
// Diagnostic build (liblc3gpu_prof.so): lane 0 of every wave adds the shader-clock cycles between consecutive
// stage stamps into a per-wave table in LDS (stamp i accumulates the time since the previous stamp into slot i);
// the table is flushed to a global one with one atomic per slot at the end of the launch, so the stamps do not put
// memory traffic inside the stages.  Never timed as a whole; read its SHARES (cdna_hip_programming.md section 7).
LC3_TU_STATIC __device__ unsigned long long lc3_prof_acc[64];  // 0..31 stage sums; 32/33/34 enc wave time sum/max/count, 35/36/37 dec; 40..47 parse kernel; 48..55 pack kernel
#undef LC3_STAMP
#define LC3_STAMP(L, lane, id)                                                     \
    do {                                                                           \
        if ((lane) == 0) {                                                         \
            unsigned long long t_ = clock64();                                     \
            if ((id) != 0 && (id) != 16) (L).prof_acc[(id)] += t_ - (L).prof_last; \
            (L).prof_last = t_;                                                    \
        }                                                                          \
    } while (0)
// lane-per-frame parse kernel: wave-level sections (lane 0 of each wave reports), slots 40..47
#define LC3_PSTAMP(c, id)                                              \
    do {                                                               \
        const unsigned long long t_ = clock64();                       \
        if ((id) >= 0) (c).pt[(id)] += t_ - (c).plast;                 \
        (c).plast = t_;                                                \
    } while (0)
#define LC3_PROF_BEGIN(L, lane)                                   \
    do {                                                          \
        if ((lane) < 32) (L).prof_acc[(lane)] = 0;                \
        if ((lane) == 0) (L).prof_acc[0] = clock64();             \
        __syncthreads();                                          \
    } while (0)
#define LC3_PROF_MARK(L, lane, slot)                                                    \
    do {                                                                                \
        if ((lane) == 0) {                                                              \
            const unsigned long long t_ = clock64();                                    \
            atomicAdd(&lc3_prof_acc[(slot)], t_ - ((L).prof_acc[16] ? (L).prof_acc[16] : (L).prof_acc[0])); \
            (L).prof_acc[16] = t_;                                                      \
        }                                                                               \
    } while (0)
#define LC3_PROF_END(L, lane, base)                                                                     \
    do {                                                                                                \
        __syncthreads();                                                                                \
        if ((lane) == 0) {                                                                              \
            const unsigned long long d_ = clock64() - (L).prof_acc[0];                                  \
            atomicAdd(&lc3_prof_acc[(base)], d_);                                                       \
            atomicMax(&lc3_prof_acc[(base) + 1], d_);                                                   \
            atomicAdd(&lc3_prof_acc[(base) + 2], 1ull);                                                 \
        } else if ((lane) < 32 && (L).prof_acc[(lane)])                                                 \
            atomicAdd(&lc3_prof_acc[(lane)], (L).prof_acc[(lane)]);                                     \
    } while (0)
#else
#define LC3_PROF_BEGIN(L, lane)
#define LC3_PROF_END(L, lane, base)
#define LC3_PROF_MARK(L, lane, slot)
#endif
#include "lc3_dev_dec.h"
#include "lc3_dev_dec_recon.h"
#include "lc3_dev_enc.h"
#include "lc3_host_plan.h"

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// How a launch finds a stream's PCM, frame bytes and bad-frame flags.  Three buffer layouts (stream s, frame t, T frames):
//   planar       int16[stream][frame][nf], uint8[stream][frame][nbytes], flags[stream][frame]                (default)
//   interleaved  int16[frame][nf][ilv streams], uint8[frame][ilv][nbytes], flags[frame][ilv]: the WAV / .lc3 file order the
//                reference's callers convert from and to on the host (examples/encode.rs:95-115, examples/decode.rs:86-112)
//   ragged       a mixed-configuration handle: stream i's PCM at element T * tab[i].pcm_off1, its bytes at T * tab[i].byte_off1,
//                its flags at tab[i].flag_idx * T (the streams in the caller's order, each planar)
struct lc3_stream_io {
    long long pcm_off1, byte_off1;  // per frame of the batch: sum of nf / of nbytes over the caller's earlier streams
    int flag_idx, pad;              // the stream's index in the caller's order
};
struct lc3_io {
    int ilv;
    const lc3_stream_io *tab;  // indexed by the handle's internal stream index (first_channel + s)
};
__device__ __forceinline__ const int16_t *lc3_io_pcm(const lc3_io &io, const int16_t *pcm, int nf, int first, int s, int t, int T,
                                                     int *stride) {
    *stride = 1;
    if (io.tab) return pcm + (size_t)T * (size_t)io.tab[first + s].pcm_off1 + (size_t)t * (size_t)nf;
    if (io.ilv) {
        *stride = io.ilv;
        return pcm + (size_t)t * (size_t)nf * (size_t)io.ilv + (size_t)s;
    }
    return pcm + ((size_t)s * (size_t)T + (size_t)t) * (size_t)nf;
}
__device__ __forceinline__ size_t lc3_io_byte_off(const lc3_io &io, int nbytes, int first, size_t s, size_t t, int T) {
    if (io.tab) return (size_t)T * (size_t)io.tab[(size_t)first + s].byte_off1 + t * (size_t)nbytes;
    if (io.ilv) return (t * (size_t)io.ilv + s) * (size_t)nbytes;
    return (s * (size_t)T + t) * (size_t)nbytes;
}
__device__ __forceinline__ size_t lc3_io_flag_idx(const lc3_io &io, int first, size_t s, size_t t, int T) {
    if (io.tab) return (size_t)io.tab[(size_t)first + s].flag_idx * (size_t)T + t;
    if (io.ilv) return t * (size_t)io.ilv + s;
    return s * (size_t)T + t;
}

// runs BODY<view>(slot{g.slot}, args...) with the group's configuration view (a mixed batch is BASELINE config 4's whole point: its
// 48 kHz / 7.5 ms, 32 kHz and 16 kHz groups get the compile-time views the uniform handles of those configurations get)
#define LC3_GROUP_VIEW(BODY, g, ...)                                                         \
    do {                                                                                     \
        switch ((g).fixed) {                                                                 \
            LC3_VIEW_CASES(BODY, (g).slot, __VA_ARGS__)                                      \
        default: BODY<lc3_cfg_any>(lc3_cfg_slot<lc3_cfg_any>{(g).slot}, __VA_ARGS__); break; \
        }                                                                                    \
    } while (0)
// A mixed-configuration handle keeps its streams sorted by configuration; a "group" is one run of streams of equal
// (rate, duration, frame bytes).  Every kernel of a mixed batch is ONE launch: a workgroup finds its group from its index.
#define LC3_MAX_GROUPS 24
struct lc3_group {
    int slot, fixed;              // configuration slot; the compile-time view that applies (lc3_cfg_views.h: 1..4), 0 = the run-time view
    int first_stream, n_streams;  // [first_stream, first_stream + n_streams) in the handle's internal order
    int wg_stream, wg_frame;      // the group's first workgroup in a stream-kernel / frame-kernel launch
    int nbytes, ne, nb, pad;
    long long frame_base;         // first plane column of the group
};
struct lc3_groups {
    int n, pad;
    lc3_group g[LC3_MAX_GROUPS];
};
__device__ __forceinline__ int lc3_find_group(const lc3_groups &G, unsigned wg, int frame_kernel) {
    int gi = 0;
    while (gi + 1 < G.n && wg >= (unsigned)(frame_kernel ? G.g[gi + 1].wg_frame : G.g[gi + 1].wg_stream)) gi++;
    return __builtin_amdgcn_readfirstlane(gi);
}

// Analysis, front half: one wave per stream (four streams per workgroup): MDCT, band energies, bandwidth, attack,
// SNS targets, LTPF analysis.  Leaves the mid-plane column (spectrum, targets, flags) and the first packer-plane words.
// (OUTLINE_LTPF: the LTPF stage as a function of its own, i.e. a register allocation of its own.  With the stage inlined the kernel sits at
// the edge of its 128 registers, and on which side it lands depends on what ELSE the module holds: 117 registers and no spill in the whole-
// source module (= the main unit), 128 and 37 - 47 spilled for every view in a translation unit of its own, from nearly the same IR.  The
// main unit keeps the inlined form it was tuned in; in the other units the 10 ms views outline it (103 - 120 registers, no spill); the
// 7.5 ms views keep it inline with a larger budget, lc3_front_waves -- their LTPF stage saves and restores ~50 registers per call when
// outlined, 0.473 -> 0.625 ms.  The mixed kernels, a body per view, outline all.  profiles/r04_translation_units.txt)
#if LC3_TU_KIND == 0
template <class CV> struct lc3_front_outline { static constexpr int value = 0; };  // (the whole-source module as tuned: inline, 117 registers)
#else
template <class CV> struct lc3_front_outline { static constexpr int value = CV::n_ms_10 ? 1 : 0; };
#endif
template <class CV, int OUTLINE_LTPF = lc3_front_outline<CV>::value>
__device__ __forceinline__ void lc3_enc_front_body(lc3_cfg_slot<CV> cfg, unsigned wg, lc3_enc_state *states, int first_channel,
                                                   int n_streams, const int16_t *pcm, float *mid, int32_t *planes, int nbytes,
                                                   int n_frames, int fresh, float *dbg, lc3_io io, int spec_flags) {
    const int lane = threadIdx.x & 63, wave = LC3_WAVE_ID();
    lc3_enc_lds &L = lc3_enc_wg[wave];
    // stream index inside this launch; the waves past the end of the launch shadow the last stream and store nothing
    const int s_raw = (int)wg * LC3_WG_WAVES + wave;
    const int valid = s_raw < n_streams;
    const int s = valid ? s_raw : n_streams - 1;
    typename CV::bind_t c0 = CV::bind(lc3_cfg_table[cfg.id]);
    const int nf = c0.nf, z = c0.z;
    lc3_enc_state *gst = states + (size_t)(first_channel + s);
#ifndef LC3_TABLES_IN_GLOBAL
    lc3_front_tables_stage_image(c0.stage_image);
    lc3_fft_tables_stage_image(c0.stage_image);  // ends with the workgroup barrier
#endif
    LC3_PROF_BEGIN(L, lane);
    if (lane == 0) L.spec_flags = spec_flags;
    if (fresh) lc3_enc_state_init(L, lane, gst, valid);
    else lc3_enc_state_load(L, lane, gst);
    for (int t = 0; t < n_frames; t++) {
        const size_t f = (size_t)s * (size_t)n_frames + (size_t)t;
        int32_t *plane = valid ? LC3_PLANE_COL(planes, f, EP_WORDS) : nullptr;
        float *mcol = valid ? mid + f * (size_t)MP_WORDS : nullptr;
        int stride;
        const int16_t *frame = lc3_io_pcm(io, pcm, nf, first_channel, s, t, n_frames, &stride);
        // MDCT history: the tail of the previous frame of this launch, else the state blob's copy (silence when fresh)
        const int16_t *hist = t > 0 ? frame - (size_t)(nf - z) * (size_t)stride : (fresh ? nullptr : gst->hist);
        // (phase: the frame's number modulo the workgroup's waves -- they take turns at the gathered blocks -- and, bit 8, "not one of the
        // launch's last two frames": lc3_enc_ltpf skips work whose only use is the state blob's memory of the last two frames.  The number is
        // reduced HERE: a launch may have any number of frames, and frame 256's own bit 8 must not read as the flag)
        lc3_encode_front_wave(cfg, L, lane, frame, hist, gst, mcol, plane, LC3_PLANE_STRIDE, nbytes, valid ? dbg : nullptr, stride,
                              t > 0 ? stride : 1, (t % LC3_WG_WAVES) + (t + 2 < n_frames ? 0x100 : 0), OUTLINE_LTPF);
    }
    if (valid) {
        int stride = 1;
        const int16_t *last = n_frames > 0 ? lc3_io_pcm(io, pcm, nf, first_channel, s, n_frames - 1, n_frames, &stride) : nullptr;
        lc3_enc_state_store(c0, L, lane, gst, last, stride);
    }
    LC3_PROF_END(L, lane, 32);
}
// waves per SIMD the front half's register allocation aims at, by configuration view: within 128 registers (four waves) the 48 kHz / 7.5 ms
// view spilled 47 of them (176 bytes of scratch per lane, front half 0.599 ms per 65 536 frames); with the budget of three waves it
// spills none and takes 0.462 ms -- it still fits four (LDS allows no more).  The other views fit 128 registers as they are (the
// run-time view gets slower with the larger budget: 0.641 -> 0.733 ms)
template <class CV> struct lc3_front_waves { static constexpr int value = CV::n_ms_10 ? LC3_FRONT_WAVES : LC3_FRONT_WAVES_75; };
template <> struct lc3_front_waves<lc3_cfg_any> { static constexpr int value = LC3_FRONT_WAVES; };
template <class CV>
__global__ __launch_bounds__(64 * LC3_WG_WAVES, lc3_front_waves<CV>::value) void lc3_enc_front_kernel(lc3_cfg_slot<CV> cfg, lc3_enc_state *states,
                                                                             int first_channel, int n_streams,
                                                                             const int16_t *pcm, float *mid, int32_t *planes,
                                                                             int nbytes, int n_frames, int fresh, float *dbg,
                                                                             lc3_io io, int spec_flags) {
    lc3_enc_front_body<CV>(cfg, blockIdx.x, states, first_channel, n_streams, pcm, mid, planes, nbytes, n_frames, fresh, dbg, io,
                           spec_flags);
}
template <class CV>
__device__ __forceinline__ void lc3_enc_front_body_mixed(lc3_cfg_slot<CV> cfg, unsigned wg, lc3_enc_state *states, int first_channel,
                                                         int n_streams, const int16_t *pcm, float *mid, int32_t *planes, int nbytes,
                                                         int n_frames, int fresh, float *dbg, lc3_io io, int spec_flags) {
    lc3_enc_front_body<CV, 1>(cfg, wg, states, first_channel, n_streams, pcm, mid, planes, nbytes, n_frames, fresh, dbg, io, spec_flags);
}
#if LC3_IN_MIXED_TU(0)
__global__ __launch_bounds__(64 * LC3_WG_WAVES, LC3_FRONT_WAVES) void LC3_MIXED_KERNEL(lc3_enc_front_mixed_kernel)(lc3_groups G, lc3_enc_state *states,
                                                                                   const int16_t *pcm, float *mid, int32_t *planes,
                                                                                   int n_frames, int fresh, lc3_io io, int spec_flags) {
    const lc3_group &g = G.g[lc3_find_group(G, blockIdx.x, 0)];
    float *m = mid + (size_t)g.frame_base * (size_t)MP_WORDS;
    int32_t *p = planes + (size_t)g.frame_base * (size_t)EP_WORDS;
    LC3_GROUP_VIEW(lc3_enc_front_body_mixed, g, blockIdx.x - g.wg_stream, states, g.first_stream, g.n_streams, pcm, m, p, g.nbytes, n_frames, fresh,
                   (float *)nullptr, io, spec_flags);
}
#endif
#if LC3_MULTI_TU && LC3_TU_KIND == 0
__global__ __launch_bounds__(64 * LC3_WG_WAVES, LC3_FRONT_WAVES) void lc3_enc_front_mixed_kernel_all(lc3_groups G, lc3_enc_state *states,
                                                                                   const int16_t *pcm, float *mid, int32_t *planes,
                                                                                   int n_frames, int fresh, lc3_io io, int spec_flags);
#endif

// SNS vector quantiser, one LANE per frame (lc3_dev_enc_vq.h): 16 targets -> indices (packer plane) + 64 band gains.
__device__ __forceinline__ void lc3_sns_vq_body(unsigned wg, int nb, float *mid, int32_t *planes, int n_frames, int spec_flags) {
    LC3_LANE_KERNEL_BEGIN();
    __shared__ uint32_t s_mpvq[16 * 11];
    for (int i = threadIdx.x; i < 16 * 11; i += blockDim.x) s_mpvq[i] = LC3T_MPVQ_OFFSETS[i / 11][i % 11];
    __syncthreads();
    const size_t f = (size_t)wg * blockDim.x + threadIdx.x;
    if (f < (size_t)n_frames) {
        lc3_vq_ctx v;
        v.mid = mid + f * (size_t)MP_WORDS;
        v.gains = mid + f * (size_t)MP_WORDS + MP_G;
        v.plane = LC3_PLANE_COL(planes, f, EP_WORDS);
        v.stride = LC3_PLANE_STRIDE;
        v.mpvq = s_mpvq;
        v.nb = nb;
        v.spec_flags = spec_flags;
        lc3_sns_vq_frame(v);
    }
}
#if LC3_IN_HOST_TU
__global__ __launch_bounds__(256) void lc3_sns_vq_kernel(int nb, float *mid, int32_t *planes, int n_frames, int spec_flags) {
    lc3_sns_vq_body(blockIdx.x, nb, mid, planes, n_frames, spec_flags);
}
#endif
#if LC3_IN_HOST_TU
__global__ __launch_bounds__(256) void lc3_sns_vq_mixed_kernel(lc3_groups G, float *mid, int32_t *planes, int n_frames, int spec_flags) {
    const lc3_group &g = G.g[lc3_find_group(G, blockIdx.x, 1)];
    lc3_sns_vq_body(blockIdx.x - g.wg_frame, g.nb, mid + (size_t)g.frame_base * (size_t)MP_WORDS,
                    planes + (size_t)g.frame_base * (size_t)EP_WORDS, g.n_streams * n_frames, spec_flags);
}
#endif

// Analysis, back half: one wave per stream: spectral shaping with the quantised gains, TNS, quantiser (stateful),
// residual bits, noise level.  Completes the packer plane column.
template <class CV>
__device__ __forceinline__ void lc3_enc_back_body(lc3_cfg_slot<CV> cfg, unsigned wg, lc3_enc_state *states, int first_channel,
                                                  int n_streams, const float *mid, int32_t *planes, int nbytes, int n_frames,
                                                  float *dbg, int spec_flags) {
    const int lane = threadIdx.x & 63, wave = LC3_WAVE_ID();
    lc3_enc_lds &L = lc3_enc_wg[wave];
    const int s_raw = (int)wg * LC3_WG_WAVES + wave;
    const int valid = s_raw < n_streams;
    const int s = valid ? s_raw : n_streams - 1;
    lc3_enc_state *gst = states + (size_t)(first_channel + s);
#if LC3_SPEC_IN_LDS
    {   // spectral-model tables -> LDS, once per workgroup
        const uint32_t *lk = (const uint32_t *)LC3T_AC_SPEC_LOOKUP, *bt = (const uint32_t *)&LC3T_AC_SPEC_BITS[0][0];
        uint32_t *dl = (uint32_t *)lc3_spec_tab.lookup, *db = (uint32_t *)lc3_spec_tab.bits;
        for (int i = threadIdx.x; i < 1024; i += 64 * LC3_WG_WAVES) dl[i] = lk[i];
        for (int i = threadIdx.x; i < 64 * 17 / 2; i += 64 * LC3_WG_WAVES) db[i] = bt[i];
        __syncthreads();
    }
#endif
    LC3_PROF_BEGIN(L, lane);
    if (lane == 0) L.spec_flags = spec_flags;
    lc3_enc_state_load(L, lane, gst);  // the front half has stored (or initialised) the scalars
    typename CV::bind_t c0 = CV::bind(lc3_cfg_table[cfg.id]);
    const size_t fbase = (size_t)s * (size_t)n_frames;
    (void)c0;
    lc3_encode_back_stream(cfg, L, lane, mid, planes, fbase, n_frames, nbytes, valid, valid ? dbg : nullptr);
    if (valid) lc3_enc_state_store(c0, L, lane, gst, nullptr);
    LC3_PROF_END(L, lane, 32);
}
template <class CV>
__global__ __launch_bounds__(64 * LC3_WG_WAVES, LC3_BACK_WAVES) void lc3_enc_back_kernel(lc3_cfg_slot<CV> cfg, lc3_enc_state *states,
                                                                            int first_channel, int n_streams,
                                                                            const float *mid, int32_t *planes, int nbytes,
                                                                            int n_frames, float *dbg, int spec_flags) {
    lc3_enc_back_body<CV>(cfg, blockIdx.x, states, first_channel, n_streams, mid, planes, nbytes, n_frames, dbg, spec_flags);
}
#if LC3_IN_MIXED_TU(1)
__global__ __launch_bounds__(64 * LC3_WG_WAVES, LC3_BACK_WAVES) void LC3_MIXED_KERNEL(lc3_enc_back_mixed_kernel)(lc3_groups G, lc3_enc_state *states,
                                                                                                const float *mid, int32_t *planes,
                                                                                                int n_frames, int spec_flags) {
    const lc3_group &g = G.g[lc3_find_group(G, blockIdx.x, 0)];
    const float *m = mid + (size_t)g.frame_base * (size_t)MP_WORDS;
    int32_t *p = planes + (size_t)g.frame_base * (size_t)EP_WORDS;
    LC3_GROUP_VIEW(lc3_enc_back_body, g, blockIdx.x - g.wg_stream, states, g.first_stream, g.n_streams, m, p, g.nbytes, n_frames, (float *)nullptr,
                   spec_flags);
}
#endif
#if LC3_MULTI_TU && LC3_TU_KIND == 0
__global__ __launch_bounds__(64 * LC3_WG_WAVES, LC3_BACK_WAVES) void lc3_enc_back_mixed_kernel_all(lc3_groups G, lc3_enc_state *states,
                                                                                                const float *mid, int32_t *planes,
                                                                                                int n_frames, int spec_flags);
#endif

// Bitstream packer, one LANE per frame (lc3_dev_enc_pack.h).  blockDim.x frames per workgroup; context lookup and the
// packed spectral model in LDS, every lane builds its frame in an LDS staging slot, then the workgroup copies the
// frames out with coalesced stores (planar layout: one contiguous run; other layouts: frame by frame).
// Dynamic LDS: 4096 + 64*17*4 + 152*4 (TNS models) + blockDim.x * nbytes (rounded up to 4) + 4 (sink).
#define LC3_PACK_LDS_FIXED (4096 + 64 * 17 * 4 + LC3_TNS_MODEL_WORDS * 4)
__device__ __forceinline__ void lc3_pack_body(unsigned wg, int ne, const int32_t *planes, uint8_t *out, int nbytes, int n_frames,
                                              int T, int first_channel, lc3_io io) {
    LC3_LANE_KERNEL_BEGIN();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *s_lookup = smem;
    uint32_t *s_cf = (uint32_t *)(smem + 4096);
    uint32_t *s_tns = (uint32_t *)(smem + 4096 + 64 * 17 * 4);
    uint8_t *s_bytes = smem + LC3_PACK_LDS_FIXED;
    const int tid = threadIdx.x, fpb = blockDim.x;
    const size_t f0 = (size_t)wg * (size_t)fpb;
    const size_t remaining = (size_t)n_frames - f0;
    const int nfr = remaining < (size_t)fpb ? (int)remaining : fpb;
    const int total = nfr * nbytes;
#ifdef LC3_PROFILE
    const unsigned long long pk_t0 = clock64();
    unsigned long long pk_t1 = 0, pk_t2 = 0, pk_pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    {
        const uint32_t *lk32 = (const uint32_t *)LC3T_AC_SPEC_LOOKUP;
        uint32_t *d32 = (uint32_t *)s_lookup;
        for (int i = tid; i < 1024; i += fpb) d32[i] = lk32[i];
        for (int i = tid; i < 64 * 17; i += fpb) {
            const int p = i / 17, j = i - 17 * p;
            s_cf[i] = (uint32_t)(int)LC3T_AC_SPEC_CUMFREQ[p][j] | ((uint32_t)(int)LC3T_AC_SPEC_FREQ[p][j] << 16);
        }
        for (int i = tid; i < LC3_TNS_MODEL_WORDS; i += fpb) s_tns[i] = lc3_tns_model_word(i);
        uint32_t *b32 = (uint32_t *)s_bytes;  // init :138-144: frames start zero-filled
        for (int i = tid; i < (total + 3) / 4; i += fpb) b32[i] = 0;
    }
    __syncthreads();
    const size_t f = f0 + (size_t)tid;
    if (f < (size_t)n_frames) {
        lc3_pack_ctx c;
        c.buf = s_bytes + tid * nbytes;
        c.sink = s_bytes + (((size_t)fpb * (size_t)nbytes + 3) & ~(size_t)3);
        c.nbytes = nbytes;
        c.lookup = s_lookup;
        c.cf = s_cf;
        c.tns = s_tns;
        c.plane = LC3_PLANE_COL(planes, f, EP_WORDS);
        c.stride = LC3_PLANE_STRIDE;
#ifdef LC3_PROFILE
        for (int i = 0; i < 8; i++) c.pt[i] = 0;
        c.plast = clock64();
        pk_t1 = c.plast;
#endif
        lc3_pack_frame(c, ne);
#ifdef LC3_PROFILE
        for (int i = 0; i < 8; i++) pk_pt[i] = c.pt[i];
#endif
    }
#ifdef LC3_PROFILE
    pk_t2 = clock64();
#endif
    __syncthreads();
    if (io.ilv || io.tab) {  // frame f = s * T + t has its own place: one frame after the other, its bytes spread over the threads
        for (int j = 0; j < nfr; j++) {
            const size_t fj = f0 + (size_t)j, s = fj / (size_t)T, t = fj - s * (size_t)T;
            uint8_t *d = out + lc3_io_byte_off(io, nbytes, first_channel, s, t, T);
            for (int b = tid; b < nbytes; b += fpb) d[b] = s_bytes[j * nbytes + b];
        }
    } else {
        uint8_t *dst = out + f0 * (size_t)nbytes;
        if ((((uintptr_t)dst) & 3u) == 0) {
            const uint32_t *b32 = (const uint32_t *)s_bytes;
            uint32_t *d32 = (uint32_t *)dst;
            for (int i = tid; i < total / 4; i += fpb) d32[i] = b32[i];
            for (int i = (total & ~3) + tid; i < total; i += fpb) dst[i] = s_bytes[i];
        } else {
            for (int i = tid; i < total; i += fpb) dst[i] = s_bytes[i];
        }
    }
#ifdef LC3_PROFILE
    if ((tid & 63) == 0 && pk_t1) {  // 48: staging; 49..53: sections of lc3_pack_frame; 54: barrier wait + copy-out; 55: waves
        const unsigned long long t3 = clock64();
        atomicAdd(&lc3_prof_acc[48], pk_t1 - pk_t0);
        for (int i = 1; i <= 5; i++) atomicAdd(&lc3_prof_acc[48 + i], pk_pt[i]);
        atomicAdd(&lc3_prof_acc[54], t3 - pk_t2);
        atomicAdd(&lc3_prof_acc[55], 1ull);
    }
#endif
}
#if LC3_IN_HOST_TU
__global__ __launch_bounds__(256) void lc3_pack_kernel(int ne, const int32_t *planes, uint8_t *out, int nbytes, int n_frames, int T,
                                                       lc3_io io) {
    lc3_pack_body(blockIdx.x, ne, planes, out, nbytes, n_frames, T, 0, io);
}
#endif
#if LC3_IN_HOST_TU
__global__ __launch_bounds__(256) void lc3_pack_mixed_kernel(lc3_groups G, const int32_t *planes, uint8_t *out, int T, lc3_io io) {
    const lc3_group &g = G.g[lc3_find_group(G, blockIdx.x, 1)];
    lc3_pack_body(blockIdx.x - g.wg_frame, g.ne, planes + (size_t)g.frame_base * (size_t)EP_WORDS, out, g.nbytes, g.n_streams * T, T,
                  g.first_stream, io);
}
#endif

// The packer of a full batch as PRODUCER / CONSUMER wave pairs (lc3_pack_produce / lc3_pack_consume, lc3_dev_enc_pack.h): a workgroup of
// 2 x fpb threads packs fpb frames; wave w of its first half derives the symbol words of 64 frames, wave w of the second half (the same
// SIMD where the hardware deals a workgroup's waves round the SIMDs) owns their writers and range coders.  Dynamic LDS: as lc3_pack_kernel +
// per pair of waves a ring of LC3_PKPC_RING x 64 words, 64 hand-over words and two counters (lc3_pack_pc_lds).  All 2 x fpb threads copy
// the frames out.
#define LC3_PKPC_RING 16
static __host__ __device__ inline size_t lc3_pack_pc_lds(unsigned fpb, int nbytes) {
    const size_t base = (LC3_PACK_LDS_FIXED + (size_t)fpb * (size_t)nbytes + 4 + 15) & ~(size_t)15;  // ... + the packer's sink byte
    return base + (size_t)(fpb / 64) * (LC3_PKPC_RING * 64 * 4 + 64 * 4) + 64;
}
// A pair half gave up on its partner: count it in the handle's sticky device word (pc[0]; lc3gpu_*_pair_timeouts) and raise the handle's
// flag in pinned HOST memory (its address sits in pc[2..3]), which the next batch call of the handle reads without any synchronisation
// and reports as LC3GPU_EPAIR -- a caller that never polls the counter must not ship zero-filled frames as LC3 payload unnoticed.
__device__ __forceinline__ void lc3_pc_gave_up(unsigned *pc) {
    atomicAdd(pc, 1u);
    volatile unsigned *flag = (volatile unsigned *)(((unsigned long long)pc[3] << 32) | (unsigned long long)pc[2]);
    if (flag) {
        *flag = 1u;
        __threadfence_system();
    }
}
// pc_timeouts: the handle's sticky count of pair halves that gave up on their partner (lc3gpu_encoder_pair_timeouts)
__device__ __forceinline__ void lc3_pack_pc_body(unsigned wg, int ne, const int32_t *planes, uint8_t *out, int nbytes, int n_frames, int T,
                                                 int first_channel, lc3_io io, unsigned *pc_timeouts) {
    LC3_LANE_KERNEL_BEGIN();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *s_lookup = smem;
    uint32_t *s_cf = (uint32_t *)(smem + 4096);
    uint32_t *s_tns = (uint32_t *)(smem + 4096 + 64 * 17 * 4);
    uint8_t *s_bytes = smem + LC3_PACK_LDS_FIXED;
    const int tid = threadIdx.x, nt = blockDim.x, fpb = nt >> 1;
    const int role = tid >= fpb;      // 0: producer, 1: consumer
    const int ft = tid - role * fpb;  // the frame's slot in the workgroup
    const int pair = ft >> 6, lane = tid & 63, npairs = fpb >> 6;
    uint32_t *s_ring = (uint32_t *)(smem + ((LC3_PACK_LDS_FIXED + (size_t)fpb * (size_t)nbytes + 4 + 15) & ~(size_t)15));
    uint32_t *s_fin = s_ring + npairs * LC3_PKPC_RING * 64;
    int *s_cnt = (int *)(s_fin + npairs * 64);
    const size_t f0 = (size_t)wg * (size_t)fpb;
    const size_t remaining = (size_t)n_frames - f0;
    const int nfr = remaining < (size_t)fpb ? (int)remaining : fpb;
    const int total = nfr * nbytes;
    {
        if (tid < npairs) {
            s_cnt[2 * tid] = -1;
            s_cnt[2 * tid + 1] = 0;
        }
        const uint32_t *lk32 = (const uint32_t *)LC3T_AC_SPEC_LOOKUP;
        uint32_t *d32 = (uint32_t *)s_lookup;
        for (int i = tid; i < 1024; i += nt) d32[i] = lk32[i];
        for (int i = tid; i < 64 * 17; i += nt) {
            const int p = i / 17, j = i - 17 * p;
            s_cf[i] = (uint32_t)(int)LC3T_AC_SPEC_CUMFREQ[p][j] | ((uint32_t)(int)LC3T_AC_SPEC_FREQ[p][j] << 16);
        }
        for (int i = tid; i < LC3_TNS_MODEL_WORDS; i += nt) s_tns[i] = lc3_tns_model_word(i);
        uint32_t *b32 = (uint32_t *)s_bytes;  // init :138-144: frames start zero-filled
        for (int i = tid; i < (total + 3) / 4; i += nt) b32[i] = 0;
    }
    __syncthreads();
    if (f0 + (size_t)pair * 64 < (size_t)n_frames) {  // (wave-uniform) a pair of waves past the end of the launch has nothing to do
        const size_t f = f0 + (size_t)ft;
        const int valid = f < (size_t)n_frames;
        lc3_pack_ctx c;
        c.buf = s_bytes + (valid ? ft : 0) * nbytes;
        c.sink = s_bytes + (((size_t)fpb * (size_t)nbytes + 3) & ~(size_t)3);
        c.nbytes = nbytes;
        c.lookup = s_lookup;
        c.cf = s_cf;
        c.tns = s_tns;
        c.plane = LC3_PLANE_COL(planes, valid ? f : f0, EP_WORDS);  // (a lane past the end reads a column that exists and writes nothing)
        c.stride = LC3_PLANE_STRIDE;
        lc3_pc_link k;
        k.ring = s_ring + pair * (LC3_PKPC_RING * 64) + lane;
        k.mask = LC3_PKPC_RING - 1;
        k.stride = 64;
        k.fstride = 64;
        k.p_count = s_cnt + 2 * pair;
        k.c_count = s_cnt + 2 * pair + 1;
        k.fin = s_fin + pair * 64 + lane;
        const int gave_up = role == 0 ? lc3_pack_produce(c, k, ne, valid) : lc3_pack_consume(c, k, ne, valid);
        if (gave_up && lane == 0) lc3_pc_gave_up(pc_timeouts);  // (per wave: the link's counts are the wave's)
    }
    __syncthreads();
    if (io.ilv || io.tab) {  // frame f = s * T + t has its own place: one frame after the other, its bytes spread over the threads
        for (int j = 0; j < nfr; j++) {
            const size_t fj = f0 + (size_t)j, s = fj / (size_t)T, t = fj - s * (size_t)T;
            uint8_t *d = out + lc3_io_byte_off(io, nbytes, first_channel, s, t, T);
            for (int b = tid; b < nbytes; b += nt) d[b] = s_bytes[j * nbytes + b];
        }
    } else {
        uint8_t *dst = out + f0 * (size_t)nbytes;
        if ((((uintptr_t)dst) & 3u) == 0) {
            const uint32_t *b32 = (const uint32_t *)s_bytes;
            uint32_t *d32 = (uint32_t *)dst;
            for (int i = tid; i < total / 4; i += nt) d32[i] = b32[i];
            for (int i = (total & ~3) + tid; i < total; i += nt) dst[i] = s_bytes[i];
        } else {
            for (int i = tid; i < total; i += nt) dst[i] = s_bytes[i];
        }
    }
}
#if LC3_IN_HOST_TU
__global__ __launch_bounds__(512) void lc3_pack_pc_kernel(int ne, const int32_t *planes, uint8_t *out, int nbytes, int n_frames, int T,
                                                          lc3_io io, unsigned *pc_timeouts) {
    lc3_pack_pc_body(blockIdx.x, ne, planes, out, nbytes, n_frames, T, 0, io, pc_timeouts);
}
#endif
#if LC3_IN_HOST_TU
__global__ __launch_bounds__(512) void lc3_pack_pc_mixed_kernel(lc3_groups G, const int32_t *planes, uint8_t *out, int T, lc3_io io,
                                                                unsigned *pc_timeouts) {
    const lc3_group &g = G.g[lc3_find_group(G, blockIdx.x, 1)];
    lc3_pack_pc_body(blockIdx.x - g.wg_frame, g.ne, planes + (size_t)g.frame_base * (size_t)EP_WORDS, out, g.nbytes, g.n_streams * T, T,
                     g.first_stream, io, pc_timeouts);
}
#endif

// The packer's symbols as a stage of its own (lc3_enc_symbols_frame, lc3_dev_enc.h): one WAVE per frame.  A workgroup stages the
// context lookup table once (lc3_spec_tab) and walks frames wg * 4 + wave, + 4 * gridDim.x, ...; eight waves per SIMD.  Selectable
// (LC3GPU_PREP_SYMBOLS=2); measured against the two other forms in DESIGN.md section 6.
template <class CV>
__global__ __launch_bounds__(64 * LC3_WG_WAVES, 8) void lc3_symbols_kernel(lc3_cfg_slot<CV> cfg, int32_t *planes, int n_frames) {
    typename CV::bind_t c0 = CV::bind(lc3_cfg_table[cfg.id]);
    {
        const uint32_t *sl = (const uint32_t *)LC3T_AC_SPEC_LOOKUP;
        uint32_t *dl = (uint32_t *)lc3_spec_tab.lookup;
        for (int i = (int)threadIdx.x; i < 1024; i += 64 * LC3_WG_WAVES) dl[i] = sl[i];
    }
    __syncthreads();
    const int lane = (int)(threadIdx.x & 63);
    for (size_t f = (size_t)blockIdx.x * LC3_WG_WAVES + (size_t)LC3_WAVE_ID(); f < (size_t)n_frames; f += (size_t)gridDim.x * LC3_WG_WAVES)
        lc3_enc_symbols_frame(c0.ne, lane, LC3_UNIFORM_PTR(int32_t *, LC3_PLANE_COL(planes, f, EP_WORDS)));
}

// Frame parser, one LANE per frame (lc3_dev_dec_parse.h).  blockDim.x frames per workgroup (256, fewer for long frames so that
// the staging fits 64 KB of dynamic LDS); the context lookup, the packed (cum | freq) spectral model and the frames' bytes are
// staged in LDS with coalesced loads.
// Dynamic LDS: 4096 (context lookup) + 64*20*4 (spectral model, lc3_dcf_word) + 16*11*4 (MPVQ offsets) + 152*4 (TNS models) + 144 (band index table) + 16*4*blockDim.x (scale
// factors, [n][lane]) + blockDim.x * nbytes (frame bytes).
#define LC3_PARSE_LDS_FIXED (4096 + 64 * LC3_DCF_ROW_WORDS * 4 + 16 * 11 * 4 + 4 * 152 + 144)
template <class CV>
__device__ __forceinline__ void lc3_parse_body(lc3_cfg_slot<CV> cfg, unsigned wg, const uint8_t *in, const uint8_t *bad,
                                               int32_t *planes, int nbytes, int n_frames, int T, int first_channel, lc3_io io, int late,
                                               float *dbg = nullptr) {
    LC3_LANE_KERNEL_BEGIN();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typename CV::bind_t c0 = CV::bind(lc3_cfg_table[cfg.id]);
    const int ne = c0.ne, fs_ind = c0.fs_ind, n_ms_10 = c0.n_ms_10;
    const int tid = threadIdx.x, fpb = blockDim.x;
    uint8_t *s_lookup = smem;
    uint32_t *s_cf = (uint32_t *)(smem + 4096);
    uint32_t *s_mpvq = (uint32_t *)(smem + 4096 + 64 * LC3_DCF_ROW_WORDS * 4);
    uint32_t *s_tns = (uint32_t *)(smem + 4096 + 64 * LC3_DCF_ROW_WORDS * 4 + 16 * 11 * 4);
    uint16_t *s_ifs = (uint16_t *)(smem + 4096 + 64 * LC3_DCF_ROW_WORDS * 4 + 16 * 11 * 4 + 4 * 152);  // 65 entries, 144 bytes reserved
    float *s_scf = (float *)(smem + LC3_PARSE_LDS_FIXED);
    uint8_t *s_bytes = smem + LC3_PARSE_LDS_FIXED + 16 * 4 * fpb;
    const size_t f0 = (size_t)wg * (size_t)fpb;
    {
        for (int i = tid; i < 16 * 11; i += fpb) s_mpvq[i] = LC3T_MPVQ_OFFSETS[i / 11][i % 11];
        for (int i = tid; i < LC3_TNS_MODEL_WORDS; i += fpb) s_tns[i] = lc3_tns_model_word(i);
        for (int i = tid; i <= c0.nb; i += fpb) s_ifs[i] = lc3_band_index(c0)[i];
        const uint32_t *lk32 = (const uint32_t *)LC3T_AC_SPEC_LOOKUP;
        uint32_t *d32 = (uint32_t *)s_lookup;
        for (int i = tid; i < 1024; i += fpb) d32[i] = lk32[i];
        for (int i = tid; i < 64 * LC3_DCF_ROW_WORDS; i += fpb) s_cf[i] = lc3_dcf_word(i);
        const size_t remaining = (size_t)n_frames - f0;
        const int nfr = remaining < (size_t)fpb ? (int)remaining : fpb;
        const int total = nfr * nbytes;
        const uint8_t *src = in + f0 * (size_t)nbytes;
        if (io.ilv || io.tab) {  // frame f = s * T + t is fetched from its own place (examples/decode.rs:86-92 for the file order)
            for (int j = 0; j < nfr; j++) {
                const size_t fj = f0 + (size_t)j, s = fj / (size_t)T, t = fj - s * (size_t)T;
                const uint8_t *q = in + lc3_io_byte_off(io, nbytes, first_channel, s, t, T);
                for (int b = tid; b < nbytes; b += fpb) s_bytes[j * nbytes + b] = q[b];
            }
        } else if ((((uintptr_t)src) & 3u) == 0) {
            const uint32_t *s32 = (const uint32_t *)src;
            uint32_t *b32 = (uint32_t *)s_bytes;
            for (int i = tid; i < total / 4; i += fpb) b32[i] = s32[i];
            for (int i = (total & ~3) + tid; i < total; i += fpb) s_bytes[i] = src[i];
        } else {
            for (int i = tid; i < total; i += fpb) s_bytes[i] = src[i];
        }
    }
    __syncthreads();
    const size_t f = f0 + (size_t)tid;
    if (f < (size_t)n_frames) {
        lc3_parse_ctx c;
        c.dbg = f == 0 ? dbg : nullptr;
        c.bytes = s_bytes + tid * nbytes;
        c.len = nbytes;
        c.lookup = s_lookup;
        c.cf = s_cf;
        c.tns = s_tns;
        c.plane = LC3_PLANE_COL(planes, f, LC3_PLANE_WORDS);
        c.stride = LC3_PLANE_STRIDE;
        c.head = 0;
        c.tail = 0;
#ifdef LC3_PROFILE
        for (int i = 0; i < 8; i++) c.pt[i] = 0;
        c.plast = clock64();
#endif
        const size_t fb = lc3_io_flag_idx(io, first_channel, f / (size_t)T, f % (size_t)T, T);  // the flag array follows the frame layout
        int rc;
        if (late == 2) rc = (bad && bad[fb]) ? -100 : lc3_parse_frame<0>(c, ne, fs_ind, n_ms_10);  // (late is launch-uniform)
        else rc = (bad && bad[fb]) ? -100 : lc3_parse_frame<1>(c, ne, fs_ind, n_ms_10);
        int ok = rc == 0;
        if (ok && late == 2) {  // D4-D8 in the reconstruction kernels of a full batch
            lc3_recon_ctx r;
            r.scf = nullptr;
            r.sstride = 0;
            r.mpvq = s_mpvq;
            r.ifs = nullptr;
            lc3_reconstruct_prepare_wave(c);
            lc3_parse_pulses(c, r);
        } else if (ok && late) {  // D4-D8 in the synthesis kernel (launches of a few frames)
            ok = lc3_reconstruct_prepare_late(c);
        } else if (ok) {
            lc3_recon_ctx r;
            r.scf = s_scf + tid;
            r.sstride = fpb;
            r.mpvq = s_mpvq;
            r.ifs = s_ifs;
            ok = lc3_reconstruct_frame(c, r, c0);
        }
        lc3_px_set(c, AD_OK, ok);
#ifdef LC3_PROFILE
        LC3_PSTAMP(c, 6);
        if ((tid & 63) == 0)
            for (int i = 0; i < 8; i++) atomicAdd(&lc3_prof_acc[40 + i], c.pt[i]);
#endif
    }
}
template <class CV>
__global__ __launch_bounds__(256) void lc3_parse_kernel(lc3_cfg_slot<CV> cfg, const uint8_t *in, const uint8_t *bad,
                                                        int32_t *planes, int nbytes, int n_frames, int T, lc3_io io, int late) {
    lc3_parse_body<CV>(cfg, blockIdx.x, in, bad, planes, nbytes, n_frames, T, 0, io, late);
}
template <class CV>
__global__ __launch_bounds__(256) void lc3_parse_debug_kernel(lc3_cfg_slot<CV> cfg, const uint8_t *in, int32_t *planes, int nbytes, int late,
                                                              float *dbg) {
    lc3_io io = {0, nullptr};
    lc3_parse_body<CV>(cfg, 0, in, nullptr, planes, nbytes, 1, 1, 0, io, late, dbg);
}
#if LC3_IN_MIXED_TU(2)
__global__ __launch_bounds__(256) void LC3_MIXED_KERNEL(lc3_parse_mixed_kernel)(lc3_groups G, const uint8_t *in, const uint8_t *bad, int32_t *planes,
                                                              int T, lc3_io io, int late) {
    const lc3_group &g = G.g[lc3_find_group(G, blockIdx.x, 1)];
    int32_t *p = planes + (size_t)g.frame_base * (size_t)LC3_PLANE_WORDS;
    LC3_GROUP_VIEW(lc3_parse_body, g, blockIdx.x - g.wg_frame, in, bad, p, g.nbytes, g.n_streams * T, T, g.first_stream, io, late);
}
#endif
#if LC3_MULTI_TU && LC3_TU_KIND == 0
__global__ __launch_bounds__(256) void lc3_parse_mixed_kernel_all(lc3_groups G, const uint8_t *in, const uint8_t *bad, int32_t *planes,
                                                              int T, lc3_io io, int late);
#endif

// The parser of a full batch as PRODUCER / CONSUMER wave pairs (lc3_pc_produce / lc3_pc_consume, lc3_dev_dec_parse.h): a workgroup of
// 2 x fpb threads parses fpb frames; wave w of its first half runs the range decoder's recurrence for 64 frames, wave w of the second half
// (the same SIMD where the hardware deals a workgroup's waves round the SIMDs) everything else of those frames, the spectrum
// reconstruction included.  Dynamic LDS: as lc3_parse_kernel + per pair of waves 4 x 64 hand-over words and two counters
// (lc3_parse_pc_lds).  The ring of a pair -- LC3_PC_RING = 16 entries per lane -- lies in the pair's own scale-factor columns
// ([16][frames of the workgroup] floats), which only the consumer's reconstruction uses, after the symbols.
#define LC3_PC_RING 16
static __host__ __device__ inline size_t lc3_parse_pc_lds(unsigned fpb, int nbytes) {
    const size_t base = (LC3_PARSE_LDS_FIXED + (size_t)fpb * (size_t)(64 + nbytes) + 15) & ~(size_t)15;
    return base + (size_t)(fpb / 64) * (4 * 64 * 4) + 64;
}
template <class CV>
__device__ __forceinline__ void lc3_parse_pc_body(lc3_cfg_slot<CV> cfg, unsigned wg, const uint8_t *in, const uint8_t *bad, int32_t *planes,
                                                  int nbytes, int n_frames, int T, int first_channel, lc3_io io, unsigned *pc_timeouts) {
    LC3_LANE_KERNEL_BEGIN();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typename CV::bind_t c0 = CV::bind(lc3_cfg_table[cfg.id]);
    const int ne = c0.ne, fs_ind = c0.fs_ind, n_ms_10 = c0.n_ms_10;
    const int tid = threadIdx.x, nt = blockDim.x, fpb = nt >> 1;
    const int role = tid >= fpb;      // 0: producer, 1: consumer
    const int ft = tid - role * fpb;  // the frame's slot in the workgroup
    const int pair = ft >> 6, lane = tid & 63, npairs = fpb >> 6;
    uint8_t *s_lookup = smem;
    uint32_t *s_cf = (uint32_t *)(smem + 4096);
    uint32_t *s_mpvq = (uint32_t *)(smem + 4096 + 64 * LC3_DCF_ROW_WORDS * 4);
    uint32_t *s_tns = (uint32_t *)(smem + 4096 + 64 * LC3_DCF_ROW_WORDS * 4 + 16 * 11 * 4);
    uint16_t *s_ifs = (uint16_t *)(smem + 4096 + 64 * LC3_DCF_ROW_WORDS * 4 + 16 * 11 * 4 + 4 * 152);
    float *s_scf = (float *)(smem + LC3_PARSE_LDS_FIXED);
    uint8_t *s_bytes = smem + LC3_PARSE_LDS_FIXED + 16 * 4 * fpb;
    uint32_t *s_fin = (uint32_t *)(smem + ((LC3_PARSE_LDS_FIXED + (size_t)fpb * (size_t)(64 + nbytes) + 15) & ~(size_t)15));
    int *s_cnt = (int *)(s_fin + npairs * 4 * 64);
    const size_t f0 = (size_t)wg * (size_t)fpb;
    {
        if (tid < npairs) {
            s_cnt[2 * tid] = -1;
            s_cnt[2 * tid + 1] = 0;
        }
        for (int i = tid; i < 16 * 11; i += nt) s_mpvq[i] = LC3T_MPVQ_OFFSETS[i / 11][i % 11];
        for (int i = tid; i < LC3_TNS_MODEL_WORDS; i += nt) s_tns[i] = lc3_tns_model_word(i);
        for (int i = tid; i <= c0.nb; i += nt) s_ifs[i] = lc3_band_index(c0)[i];
        const uint32_t *lk32 = (const uint32_t *)LC3T_AC_SPEC_LOOKUP;
        uint32_t *d32 = (uint32_t *)s_lookup;
        for (int i = tid; i < 1024; i += nt) d32[i] = lk32[i];
        for (int i = tid; i < 64 * LC3_DCF_ROW_WORDS; i += nt) s_cf[i] = lc3_dcf_word(i);
        const size_t remaining = (size_t)n_frames - f0;
        const int nfr = remaining < (size_t)fpb ? (int)remaining : fpb;
        const int total = nfr * nbytes;
        const uint8_t *src = in + f0 * (size_t)nbytes;
        if (io.ilv || io.tab) {  // frame f = s * T + t is fetched from its own place (examples/decode.rs:86-92 for the file order)
            for (int j = 0; j < nfr; j++) {
                const size_t fj = f0 + (size_t)j, s = fj / (size_t)T, t = fj - s * (size_t)T;
                const uint8_t *q = in + lc3_io_byte_off(io, nbytes, first_channel, s, t, T);
                for (int b = tid; b < nbytes; b += nt) s_bytes[j * nbytes + b] = q[b];
            }
        } else if ((((uintptr_t)src) & 3u) == 0) {
            const uint32_t *s32 = (const uint32_t *)src;
            uint32_t *b32 = (uint32_t *)s_bytes;
            for (int i = tid; i < total / 4; i += nt) b32[i] = s32[i];
            for (int i = (total & ~3) + tid; i < total; i += nt) s_bytes[i] = src[i];
        } else {
            for (int i = tid; i < total; i += nt) s_bytes[i] = src[i];
        }
    }
    __syncthreads();
    if (f0 + (size_t)pair * 64 >= (size_t)n_frames) return;  // (wave-uniform) a pair of waves past the end of the launch
    const size_t f = f0 + (size_t)ft;
    const int valid = f < (size_t)n_frames;
    lc3_parse_ctx c;
    c.dbg = nullptr;
    c.bytes = s_bytes + (valid ? ft : 0) * nbytes;
    c.len = nbytes;
    c.lookup = s_lookup;
    c.cf = s_cf;
    c.tns = s_tns;
    c.plane = LC3_PLANE_COL(planes, valid ? f : f0, LC3_PLANE_WORDS);  // (a lane past the end parses nothing and stores nothing)
    c.stride = LC3_PLANE_STRIDE;
    c.head = 0;
    c.tail = 0;
    lc3_pc_link k;
    k.ring = (uint32_t *)s_scf + ft;  // entry i of this lane at [i][ft]: the lane's 16 scale-factor slots
    k.mask = LC3_PC_RING - 1;
    k.stride = fpb;
    k.fstride = 64;
    k.p_count = s_cnt + 2 * pair;
    k.c_count = s_cnt + 2 * pair + 1;
    k.fin = s_fin + pair * (4 * 64) + lane;
    int rc_in = -100;
    if (valid) {
        const size_t fb = lc3_io_flag_idx(io, first_channel, f / (size_t)T, f % (size_t)T, T);  // the flag array follows the frame layout
        rc_in = (bad && bad[fb]) ? -100 : 0;
    }
    if (role == 0) {  // pc_timeouts: the handle's sticky count of pair halves that gave up on their partner (lc3gpu_decoder_pair_timeouts)
        if (lc3_pc_produce(c, k, ne, fs_ind, n_ms_10, rc_in) && lane == 0) lc3_pc_gave_up(pc_timeouts);
        return;
    }
    lc3_recon_ctx r;
    r.scf = s_scf + ft;
    r.sstride = fpb;
    r.mpvq = s_mpvq;
    r.ifs = s_ifs;
    float scf[16];
    int gave_up = 0;
    int ok = lc3_pc_consume<1>(c, k, ne, fs_ind, rc_in, &r, scf, &gave_up) == 0;
    if (gave_up && lane == 0) lc3_pc_gave_up(pc_timeouts);  // (per wave: the link's counts are the wave's; its frames are concealed)
    if (ok) ok = lc3_reconstruct_frame(c, r, c0, scf);
    if (valid) lc3_px_set(c, AD_OK, ok);
}
template <class CV>
__global__ __launch_bounds__(512) void lc3_parse_pc_kernel(lc3_cfg_slot<CV> cfg, const uint8_t *in, const uint8_t *bad, int32_t *planes,
                                                           int nbytes, int n_frames, int T, lc3_io io, unsigned *pc_timeouts) {
    lc3_parse_pc_body<CV>(cfg, blockIdx.x, in, bad, planes, nbytes, n_frames, T, 0, io, pc_timeouts);
}
#if LC3_IN_MIXED_TU(3)
__global__ __launch_bounds__(512) void LC3_MIXED_KERNEL(lc3_parse_pc_mixed_kernel)(lc3_groups G, const uint8_t *in, const uint8_t *bad, int32_t *planes, int T,
                                                                 lc3_io io, unsigned *pc_timeouts) {
    const lc3_group &g = G.g[lc3_find_group(G, blockIdx.x, 1)];
    int32_t *p = planes + (size_t)g.frame_base * (size_t)LC3_PLANE_WORDS;
    LC3_GROUP_VIEW(lc3_parse_pc_body, g, blockIdx.x - g.wg_frame, in, bad, p, g.nbytes, g.n_streams * T, T, g.first_stream, io, pc_timeouts);
}
#endif
#if LC3_MULTI_TU && LC3_TU_KIND == 0
__global__ __launch_bounds__(512) void lc3_parse_pc_mixed_kernel_all(lc3_groups G, const uint8_t *in, const uint8_t *bad, int32_t *planes, int T,
                                                                 lc3_io io, unsigned *pc_timeouts);
#endif

// Spectrum reconstruction D4-D8 of a full batch (lc3_dev_dec_recon.h), between the parser and the synthesis kernel:
//   lc3_recon_kernel  one WAVE per frame: residual bits, noise filling, gain, scale factors, band gains.  Frames are independent; a
//                     workgroup (LC3_WG_WAVES waves) stages the tables once and walks frames wg * 4 + wave, + 4 * gridDim.x, ...;
//   lc3_tns_kernel    one LANE per frame: the TNS lattice and the band gains of the filter range, for the frames that have a filter.
#ifndef LC3_RECON_WAVES
#define LC3_RECON_WAVES 6  // waves per SIMD the register allocation aims at (80 registers: 5 spilled; at 8 waves 16 were, and the kernel was slower)
#endif
__shared__ lc3_recon_tables lc3_recon_tab;
__shared__ lc3_recon_wave lc3_recon_wv[LC3_WG_WAVES];
template <class CV>
__device__ __forceinline__ void lc3_recon_body(lc3_cfg_slot<CV> cfg, unsigned wg, unsigned n_wg, int32_t *planes, int nbytes, size_t n_frames) {
    const int lane = threadIdx.x & 63, wave = LC3_WAVE_ID();
    typename CV::bind_t c0 = CV::bind(lc3_cfg_table[cfg.id]);
    lc3_recon_tables_stage(c0, lc3_recon_tab, (int)threadIdx.x, 64 * LC3_WG_WAVES);
    __syncthreads();
    for (size_t f = (size_t)wg * LC3_WG_WAVES + (size_t)wave; f < n_frames; f += (size_t)n_wg * LC3_WG_WAVES)
        lc3_recon_frame_direct(c0, lc3_recon_tab, lc3_recon_wv[wave], lane, LC3_PLANE_COL(planes, f, LC3_PLANE_WORDS), nbytes);
}
template <class CV>
__global__ __launch_bounds__(64 * LC3_WG_WAVES, LC3_RECON_WAVES) void lc3_recon_kernel(lc3_cfg_slot<CV> cfg, int32_t *planes, int nbytes, int n_frames) {
    lc3_recon_body<CV>(cfg, blockIdx.x, gridDim.x, planes, nbytes, (size_t)n_frames);
}
// mixed batch: one frame per wave, G's frame-kernel workgroup numbers computed for LC3_WG_WAVES frames per workgroup
#if LC3_IN_MIXED_TU(4)
__global__ __launch_bounds__(64 * LC3_WG_WAVES, LC3_RECON_WAVES) void LC3_MIXED_KERNEL(lc3_recon_mixed_kernel)(lc3_groups G, int32_t *planes, int T) {
    const lc3_group &g = G.g[lc3_find_group(G, blockIdx.x, 1)];
    int32_t *p = planes + (size_t)g.frame_base * (size_t)LC3_PLANE_WORDS;
    const unsigned wg = blockIdx.x - g.wg_frame;
    const size_t nfr = (size_t)g.n_streams * (size_t)T, left = nfr - (size_t)wg * LC3_WG_WAVES;  // this workgroup's four frames only
    LC3_GROUP_VIEW(lc3_recon_body, g, 0u, 1u, p + (size_t)wg * LC3_WG_WAVES * (size_t)LC3_PLANE_WORDS, g.nbytes,
                   left < LC3_WG_WAVES ? left : (size_t)LC3_WG_WAVES);
}
#endif
#if LC3_MULTI_TU && LC3_TU_KIND == 0
__global__ __launch_bounds__(64 * LC3_WG_WAVES, LC3_RECON_WAVES) void lc3_recon_mixed_kernel_all(lc3_groups G, int32_t *planes, int T);
#endif
// LC3_TNS_FPB frames per workgroup, one wave per 64 of them: 16 KB of (dynamic) LDS per wave for its frames' band gains, band-major.
// Four waves per workgroup so that the 1 024 waves of a full batch land one per SIMD (single-wave workgroups are packed several to a CU).
#define LC3_TNS_FPB 256
#define LC3_TNS_LDS (LC3_TNS_FPB * 64 * 4 + 20 * 4 + LC3_MAX_NF)
template <class CV>
__device__ __forceinline__ void lc3_tns_body(lc3_cfg_slot<CV> cfg, unsigned wg, int32_t *planes, size_t n_frames) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float *s_gains = (float *)smem;
    float *s_sin = (float *)(smem + LC3_TNS_FPB * 64 * 4);
    uint32_t *s_lb = (uint32_t *)(smem + LC3_TNS_FPB * 64 * 4 + 20 * 4);
    const int tid = threadIdx.x;
    typename CV::bind_t c0 = CV::bind(lc3_cfg_table[cfg.id]);
    if (tid < 17) s_sin[tid] = lc3_tns_sin_dec_value(tid);
    for (int i = tid; i < LC3_MAX_NF / 4; i += LC3_TNS_FPB) s_lb[i] = lc3_line_band_word(c0, i);
    __syncthreads();
    const size_t f = (size_t)wg * LC3_TNS_FPB + (size_t)tid;
    const int valid = f < n_frames;
    lc3_tns_lane_ctx x;
    x.col = LC3_PLANE_COL(planes, valid ? f : 0, LC3_PLANE_WORDS);
    x.gains = s_gains + (tid >> 6) * (64 * 64) + (tid & 63);
    x.gstride = 64;
    x.sin_tab = s_sin;
    x.line_band = s_lb;
    lc3_tns_lane_frame(c0, x, valid);
}
template <class CV>
__global__ __launch_bounds__(LC3_TNS_FPB) void lc3_tns_kernel(lc3_cfg_slot<CV> cfg, int32_t *planes, int n_frames) {
    lc3_tns_body<CV>(cfg, blockIdx.x, planes, (size_t)n_frames);
}
// G: frame-kernel workgroup numbers computed for LC3_TNS_FPB frames per workgroup
#if LC3_IN_MIXED_TU(5)
__global__ __launch_bounds__(LC3_TNS_FPB) void LC3_MIXED_KERNEL(lc3_tns_mixed_kernel)(lc3_groups G, int32_t *planes, int T) {
    const lc3_group &g = G.g[lc3_find_group(G, blockIdx.x, 1)];
    int32_t *p = planes + (size_t)g.frame_base * (size_t)LC3_PLANE_WORDS;
    LC3_GROUP_VIEW(lc3_tns_body, g, blockIdx.x - g.wg_frame, p, (size_t)g.n_streams * (size_t)T);
}
#endif
#if LC3_MULTI_TU && LC3_TU_KIND == 0
__global__ __launch_bounds__(LC3_TNS_FPB) void lc3_tns_mixed_kernel_all(lc3_groups G, int32_t *planes, int T);
#endif

// LATE: the launch reconstructs the spectrum here (lc3_dec_reconstruct_wave) -- a compile-time switch, so that the kernels of full
// batches carry none of it
template <class CV, int LATE>
__device__ __forceinline__ void lc3_decode_body(lc3_cfg_slot<CV> cfg, unsigned wg, lc3_dec_state *states, int first_channel,
                                                int n_streams, const int32_t *planes, int16_t *pcm, int nbytes, int n_frames,
                                                int fresh, lc3_io io, float *dbg = nullptr, int dbg_flags = 0) {
    const int lane = threadIdx.x & 63, wave = LC3_WAVE_ID();
    lc3_dec_lds &L = lc3_dec_wg[wave];
    const int s_raw = (int)wg * LC3_WG_WAVES + wave;
    const int valid = s_raw < n_streams;
    const int s = valid ? s_raw : n_streams - 1;  // see lc3_enc_front_body
    typename CV::bind_t c0 = CV::bind(lc3_cfg_table[cfg.id]);
    const int nf = c0.nf;
    lc3_dec_state *gst = states + (size_t)(first_channel + s);
    // the launch's first loads -- table image, stream state, first frame's column, overlap memory -- are all requested before any is used
    // (lc3_decode_stream_wave's prologue hook)
#ifndef LC3_TABLES_IN_GLOBAL
    const lc3_fft_image_regs tab_regs = lc3_fft_tables_image_issue(c0.stage_image);
#endif
    LC3_PROF_BEGIN(L, lane);
    lc3_i4 st_regs = {0, 0, 0, 0};
    if (!fresh) st_regs = lc3_dec_state_issue(lane, gst);
    const size_t fbase = (size_t)s * (size_t)n_frames;
    int stride;
    int16_t *pcm0 = (int16_t *)lc3_io_pcm(io, pcm, nf, first_channel, s, 0, n_frames, &stride);
    lc3_decode_stream_wave(cfg, L, lane, nbytes, planes, fbase, n_frames, gst, valid, pcm0, (size_t)nf * (size_t)stride, stride, LATE, dbg,
                           dbg_flags, [&]() {
#ifndef LC3_TABLES_IN_GLOBAL
                               lc3_fft_tables_image_commit(tab_regs);
#endif
                               if (fresh) lc3_dec_state_init(L, lane, gst, valid);
                               else lc3_dec_state_commit(L, lane, st_regs);
                               LC3_PROF_MARK(L, lane, 38);  // state load
                           }, fresh);
    LC3_PROF_MARK(L, lane, 39);  // frames (incl. everything between the stage stamps)
    if (valid) lc3_dec_state_store(c0, L, lane, gst);
    LC3_PROF_END(L, lane, 35);
}
template <class CV>
__global__ __launch_bounds__(64 * LC3_WG_WAVES, LC3_SYNTH_WAVES) void lc3_decode_kernel(lc3_cfg_slot<CV> cfg, lc3_dec_state *states,
                                                                          int first_channel, int n_streams, const int32_t *planes,
                                                                          int16_t *pcm, int nbytes, int n_frames, int fresh,
                                                                          lc3_io io) {
    lc3_decode_body<CV, 0>(cfg, blockIdx.x, states, first_channel, n_streams, planes, pcm, nbytes, n_frames, fresh, io);
}
template <class CV>
__global__ __launch_bounds__(64 * LC3_WG_WAVES, 4) void lc3_decode_late_kernel(lc3_cfg_slot<CV> cfg, lc3_dec_state *states,
                                                                               int first_channel, int n_streams, const int32_t *planes,
                                                                               int16_t *pcm, int nbytes, int n_frames, int fresh,
                                                                               lc3_io io) {
    lc3_decode_body<CV, 1>(cfg, blockIdx.x, states, first_channel, n_streams, planes, pcm, nbytes, n_frames, fresh, io);
}
// the diagnostic entry points' instantiation (lc3gpu_decode_frame_debug, lc3gpu_decoder_synth_debug): one stream, stage dumps
template <class CV>
__global__ __launch_bounds__(64 * LC3_WG_WAVES, 4) void lc3_decode_debug_kernel(lc3_cfg_slot<CV> cfg, lc3_dec_state *states, int channel,
                                                                                const int32_t *planes, int16_t *pcm, int nbytes, int late,
                                                                                float *dbg, int dbg_flags) {
    lc3_io io = {0, nullptr};
    if (late) lc3_decode_body<CV, 1>(cfg, 0, states, channel, 1, planes, pcm, nbytes, 1, 0, io, dbg, dbg_flags);
    else lc3_decode_body<CV, 0>(cfg, 0, states, channel, 1, planes, pcm, nbytes, 1, 0, io, dbg, dbg_flags);
}
template <class CV>
__device__ __forceinline__ void lc3_decode_body_now(lc3_cfg_slot<CV> cfg, unsigned wg, lc3_dec_state *states, int first_channel, int n_streams,
                                                    const int32_t *planes, int16_t *pcm, int nbytes, int n_frames, int fresh, lc3_io io) {
    lc3_decode_body<CV, 0>(cfg, wg, states, first_channel, n_streams, planes, pcm, nbytes, n_frames, fresh, io);
}
template <class CV>
__device__ __forceinline__ void lc3_decode_body_late(lc3_cfg_slot<CV> cfg, unsigned wg, lc3_dec_state *states, int first_channel, int n_streams,
                                                     const int32_t *planes, int16_t *pcm, int nbytes, int n_frames, int fresh, lc3_io io) {
    lc3_decode_body<CV, 1>(cfg, wg, states, first_channel, n_streams, planes, pcm, nbytes, n_frames, fresh, io);
}
template <int LATE>
__device__ __forceinline__ void lc3_decode_mixed_body(const lc3_groups &G, lc3_dec_state *states, const int32_t *planes, int16_t *pcm,
                                                      int n_frames, int fresh, lc3_io io) {
    const lc3_group &g = G.g[lc3_find_group(G, blockIdx.x, 0)];
    const int32_t *p = planes + (size_t)g.frame_base * (size_t)LC3_PLANE_WORDS;
    if (LATE) LC3_GROUP_VIEW(lc3_decode_body_late, g, blockIdx.x - g.wg_stream, states, g.first_stream, g.n_streams, p, pcm, g.nbytes, n_frames, fresh, io);
    else LC3_GROUP_VIEW(lc3_decode_body_now, g, blockIdx.x - g.wg_stream, states, g.first_stream, g.n_streams, p, pcm, g.nbytes, n_frames, fresh, io);
}
#if LC3_IN_MIXED_TU(6)
__global__ __launch_bounds__(64 * LC3_WG_WAVES, 4) void LC3_MIXED_KERNEL(lc3_decode_mixed_kernel)(lc3_groups G, lc3_dec_state *states,
                                                                                const int32_t *planes, int16_t *pcm, int n_frames,
                                                                                int fresh, lc3_io io) {
    lc3_decode_mixed_body<0>(G, states, planes, pcm, n_frames, fresh, io);
}
#endif
#if LC3_MULTI_TU && LC3_TU_KIND == 0
__global__ __launch_bounds__(64 * LC3_WG_WAVES, 4) void lc3_decode_mixed_kernel_all(lc3_groups G, lc3_dec_state *states,
                                                                                const int32_t *planes, int16_t *pcm, int n_frames,
                                                                                int fresh, lc3_io io);
#endif
#if LC3_IN_MIXED_TU(7)
__global__ __launch_bounds__(64 * LC3_WG_WAVES, 4) void LC3_MIXED_KERNEL(lc3_decode_mixed_late_kernel)(lc3_groups G, lc3_dec_state *states,
                                                                                     const int32_t *planes, int16_t *pcm, int n_frames,
                                                                                     int fresh, lc3_io io) {
    lc3_decode_mixed_body<1>(G, states, planes, pcm, n_frames, fresh, io);
}
#endif
#if LC3_MULTI_TU && LC3_TU_KIND == 0
__global__ __launch_bounds__(64 * LC3_WG_WAVES, 4) void lc3_decode_mixed_late_kernel_all(lc3_groups G, lc3_dec_state *states,
                                                                                     const int32_t *planes, int16_t *pcm, int n_frames,
                                                                                     int fresh, lc3_io io);
#endif

// ---------------------------------------------------------------------------------------------
// per translation unit: the unit's kernels, its copy of the tables
// ---------------------------------------------------------------------------------------------
template <int I> struct lc3_view_by_index { typedef lc3_cfg_any type; };
#define LC3_X(i, V) template <> struct lc3_view_by_index<i> { typedef V type; };
LC3_FOR_EACH_VIEW_ALL(LC3_X)
#undef LC3_X
#if LC3_MULTI_TU
// the kernels of a configuration view -- three of the encoder, eight of the decoder --: T = `extern` declares them (the host unit), T
// empty instantiates them (the view's units; encoder and decoder apart: besides halving the longest unit, what else a module holds
// changes how the compiler lays out LDS for the stage functions, and with all eleven in one module the front half came out with 47
// spilled registers)
#define LC3_VIEW_ENC_KERNELS(T, CV)                                                                                                          \
    T template __global__ void lc3_enc_front_kernel<CV>(lc3_cfg_slot<CV>, lc3_enc_state *, int, int, const int16_t *, float *, int32_t *, int, int, \
                                                        int, float *, lc3_io, int);                                                           \
    T template __global__ void lc3_enc_back_kernel<CV>(lc3_cfg_slot<CV>, lc3_enc_state *, int, int, const float *, int32_t *, int, int, float *, int); \
    T template __global__ void lc3_symbols_kernel<CV>(lc3_cfg_slot<CV>, int32_t *, int);
#define LC3_VIEW_DEC_KERNELS(T, CV)                                                                                                          \
    T template __global__ void lc3_parse_kernel<CV>(lc3_cfg_slot<CV>, const uint8_t *, const uint8_t *, int32_t *, int, int, int, lc3_io, int);  \
    T template __global__ void lc3_parse_debug_kernel<CV>(lc3_cfg_slot<CV>, const uint8_t *, int32_t *, int, int, float *);                    \
    T template __global__ void lc3_parse_pc_kernel<CV>(lc3_cfg_slot<CV>, const uint8_t *, const uint8_t *, int32_t *, int, int, int, lc3_io, unsigned *);   \
    T template __global__ void lc3_recon_kernel<CV>(lc3_cfg_slot<CV>, int32_t *, int, int);                                                    \
    T template __global__ void lc3_tns_kernel<CV>(lc3_cfg_slot<CV>, int32_t *, int);                                                           \
    T template __global__ void lc3_decode_kernel<CV>(lc3_cfg_slot<CV>, lc3_dec_state *, int, int, const int32_t *, int16_t *, int, int, int, lc3_io); \
    T template __global__ void lc3_decode_late_kernel<CV>(lc3_cfg_slot<CV>, lc3_dec_state *, int, int, const int32_t *, int16_t *, int, int, int, \
                                                          lc3_io);                                                                            \
    T template __global__ void lc3_decode_debug_kernel<CV>(lc3_cfg_slot<CV>, lc3_dec_state *, int, const int32_t *, int16_t *, int, int, float *, int);
#if LC3_TU_KIND == 0
#define LC3_X(i, V) LC3_VIEW_ENC_KERNELS(extern, V) LC3_VIEW_DEC_KERNELS(extern, V)
LC3_FOR_EACH_VIEW_EXTRA(LC3_X)
#undef LC3_X
#elif LC3_TU_KIND == 1
typedef lc3_view_by_index<LC3_TU_INDEX>::type lc3_tu_view;
LC3_VIEW_ENC_KERNELS(, lc3_tu_view)
#elif LC3_TU_KIND == 3
typedef lc3_view_by_index<LC3_TU_INDEX>::type lc3_tu_view;
LC3_VIEW_DEC_KERNELS(, lc3_tu_view)
#endif
#endif
namespace {
__global__ void lc3_pow10_tables_kernel() {
    for (int i = threadIdx.x; i < 512; i += blockDim.x) lc3_pow10_gg_tab[i] = lc3_pow10f((float)(i - 256) / 28.0f);
    for (int i = threadIdx.x; i < 5 * 64; i += blockDim.x) {
        const int fsi = i / 64, b = i % 64;
        lc3_pow10_tilt_tab[i] = lc3_pow10f((float)b * ((float)LC3C_G_TILT[fsi] / 630.0f));
    }
    for (int i = threadIdx.x; i < 17; i += blockDim.x) {
        lc3_tns_sin_tab[i] = lc3_tns_sin_enc_value(i);
    }
}
}  // namespace
// this unit's copy of the tables: slot >= 0 publishes a configuration in its constant table, fill_tables runs its table-filling kernel
// (once per device).  Returns a hipError_t as int (0 = success).
#define LC3_TU_REGISTER_NAME(kind, index) LC3_CAT(LC3_CAT(lc3_tu_register_, kind), LC3_CAT(_, index))
extern "C" __attribute__((visibility("hidden"))) int LC3_TU_REGISTER_NAME(LC3_TU_KIND, LC3_TU_INDEX)(int slot, const lc3_cfg *c, int fill_tables) {
    if (fill_tables) {
        hipLaunchKernelGGL(lc3_pow10_tables_kernel, dim3(1), dim3(256), 0, nullptr);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) return (int)e;
    }
    if (slot >= 0 && c) {
        const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(lc3_cfg_table), c, sizeof(*c), sizeof(*c) * (size_t)slot, hipMemcpyHostToDevice);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}
typedef int (*lc3_tu_register_fn)(int, const lc3_cfg *, int);
#if LC3_MULTI_TU && LC3_TU_KIND == 0
#define LC3_X(i, V)                                                                                              \
    extern "C" __attribute__((visibility("hidden"))) int LC3_TU_REGISTER_NAME(1, i)(int, const lc3_cfg *, int); \
    extern "C" __attribute__((visibility("hidden"))) int LC3_TU_REGISTER_NAME(3, i)(int, const lc3_cfg *, int);
LC3_FOR_EACH_VIEW_EXTRA(LC3_X)
#undef LC3_X
#define LC3_FOR_EACH_MIXED_TU(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define LC3_X(k) extern "C" __attribute__((visibility("hidden"))) int LC3_TU_REGISTER_NAME(2, k)(int, const lc3_cfg *, int);
LC3_FOR_EACH_MIXED_TU(LC3_X)
#undef LC3_X
static const lc3_tu_register_fn lc3_tu_registers[] = {lc3_tu_register_0_0,
#define LC3_X(i, V) LC3_TU_REGISTER_NAME(1, i), LC3_TU_REGISTER_NAME(3, i),
                                                       LC3_FOR_EACH_VIEW_EXTRA(LC3_X)
#undef LC3_X
#define LC3_X(k) LC3_TU_REGISTER_NAME(2, k),
                                                           LC3_FOR_EACH_MIXED_TU(LC3_X)
#undef LC3_X
};
#elif LC3_TU_KIND == 0
static const lc3_tu_register_fn lc3_tu_registers[] = {lc3_tu_register_0_0};
#endif

#if LC3_IN_HOST_TU  // ============================================================================================== the host side
// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
namespace {

thread_local int g_last_hip = 0;

#define HIP_TRY(expr)                       \
    do {                                    \
        hipError_t e_ = (expr);             \
        if (e_ != hipSuccess) {             \
            g_last_hip = (int)e_;           \
            return LC3GPU_EHIP;             \
        }                                   \
    } while (0)

struct HostCfg {
    lc3_cfg c;       // host copy (device pointers inside are valid on the device it was registered on)
    int slot = 0;    // its slot in lc3_cfg_table
    int view = 0;    // 0: run-time view (lc3_cfg_any); 1..4: the compile-time view of lc3_cfg_views.h that matches this configuration
};
// launches kern<view>(slot, args...) with the view the configuration allows
#define LC3_LAUNCH_VIEW(kern, V, h, grid, block, lds, stream, ...) \
    hipLaunchKernelGGL(kern<V>, grid, block, lds, stream, lc3_cfg_slot<V>{(h).slot}, __VA_ARGS__)
#define LC3_LAUNCH_CFG(kern, h, grid, block, lds, stream, ...)                                       \
    do {                                                                                             \
        switch ((h).view) {                                                                          \
            LC3_LAUNCH_CASES(kern, h, grid, block, lds, stream, __VA_ARGS__)                         \
        default: LC3_LAUNCH_VIEW(kern, lc3_cfg_any, h, grid, block, lds, stream, __VA_ARGS__); break; \
        }                                                                                            \
    } while (0)

int make_config(lc3_cfg &c, int frame_us, int fs_hz) {
    return lc3_make_config(c, frame_us, fs_hz) ? LC3GPU_EINVAL : LC3GPU_OK;
}

// fills the line -> band width table of a configuration on the device (lc3_line_width_value)
__global__ void lc3_line_width_kernel(float *out, lc3_cfg c) {
    for (int k = threadIdx.x; k < c.ne; k += blockDim.x) out[k] = lc3_line_width_value(c, k);
}
// fills the line -> band table of a configuration on the device (lc3_line_band_value)
__global__ void lc3_line_band_kernel(uint8_t *out, lc3_cfg c) {
    for (int k = threadIdx.x; k < c.nf; k += blockDim.x) out[k] = (uint8_t)lc3_line_band_value(c, k);
}
// Writes a configuration's stage image (lc3_cfg::stage_image): one workgroup stages the tables in LDS exactly as the kernels did before the
// image existed (lc3_front_tables_stage, lc3_fft_tables_stage over a zeroed array) and copies the two LDS structs out.
__global__ __launch_bounds__(256) void lc3_stage_image_kernel(uint32_t *image, lc3_cfg c) {
    uint32_t *lf = (uint32_t *)&lc3_fft_tab, *lt = (uint32_t *)&lc3_front_tab;
    const int nf_w = (int)(sizeof(lc3_fft_tables) / 4), nt_w = (int)(sizeof(lc3_front_tables) / 4);
    for (int i = threadIdx.x; i < nf_w; i += blockDim.x) lf[i] = 0u;
    for (int i = threadIdx.x; i < nt_w; i += blockDim.x) lt[i] = 0u;
    __syncthreads();
    lc3_front_tables_stage(c);
    lc3_fft_tables_stage(c);  // ends with the workgroup barrier
    for (int i = threadIdx.x; i < nf_w; i += blockDim.x) image[i] = lf[i];
    for (int i = threadIdx.x; i < nt_w; i += blockDim.x) image[nf_w + i] = lt[i];
}
// fills the polyphase resampler table of a configuration on the device (lc3_resamp_poly_value)
__global__ void lc3_resamp_poly_kernel(float *out, int p, int lim, int stride) {
    const int n = p * stride;
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = lc3_resamp_poly_value(p, lim, stride, i);
}

// Process-wide registry of configuration slots, per device: the plan (lc3_host_plan.h) is built on the host once per
// (device, rate, duration), its tables are uploaded in one allocation that lives as long as the process, and the
// filled lc3_cfg is published in the device's lc3_cfg_table.  Encoders and decoders of equal configuration share it.
#define LC3_MAX_DEVICES 64
struct CfgRegistry {
    std::mutex mu;
    bool tables[LC3_MAX_DEVICES] = {};  // the device's 10^x tables are filled
    bool ready[LC3_MAX_DEVICES][LC3_CFG_SLOTS] = {};
    lc3_cfg cfg[LC3_MAX_DEVICES][LC3_CFG_SLOTS];
};
CfgRegistry g_cfgs;

// uploads the tables of one configuration into `base` (owned by the caller until success)
int cfg_upload(lc3_cfg &c, const lc3_host_plan &pl, char *base, size_t bytes_tw, size_t bytes_perm, size_t bytes_poly, size_t bytes_lw,
               size_t bytes_lb, int slot) {
    HIP_TRY(hipMemcpy(base, pl.fft_tw.data(), bytes_tw, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + bytes_tw, pl.dct_tw.data(), bytes_tw, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + 2 * bytes_tw, pl.perm.data(), sizeof(uint16_t) * (size_t)c.nfft, hipMemcpyHostToDevice));
    float *poly = (float *)(base + 2 * bytes_tw + bytes_perm);
    float *lw = (float *)(base + 2 * bytes_tw + bytes_perm + bytes_poly);
    uint8_t *lb = (uint8_t *)(base + 2 * bytes_tw + bytes_perm + bytes_poly + bytes_lw);
    hipLaunchKernelGGL(lc3_resamp_poly_kernel, dim3(1), dim3(256), 0, nullptr, poly, c.p_up, c.resamp_lim, c.resamp_stride);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(lc3_line_width_kernel, dim3(1), dim3(256), 0, nullptr, lw, c);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(lc3_line_band_kernel, dim3(1), dim3(256), 0, nullptr, lb, c);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(nullptr));
    c.fft_tw = (const lc3_cpx *)base;
    c.dct_tw = (const lc3_cpx *)(base + bytes_tw);
    c.perm = (const uint16_t *)(base + 2 * bytes_tw);
    c.resamp_poly = poly;
    c.line_width = lw;
    c.line_band = lb;
    uint32_t *image = (uint32_t *)(base + 2 * bytes_tw + bytes_perm + bytes_poly + bytes_lw + bytes_lb);  // 16-byte aligned: every part is
    c.stage_image = nullptr;
    hipLaunchKernelGGL(lc3_stage_image_kernel, dim3(1), dim3(256), 0, nullptr, image, c);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(nullptr));
    c.stage_image = image;
    for (lc3_tu_register_fn reg : lc3_tu_registers) HIP_TRY((hipError_t)reg(slot, &c, 0));  // every translation unit's constant table
    return LC3GPU_OK;
}

// The run-time view's DCT-IV takes a compile-time FFT plan by frame length (lc3_dct4_select): the plan the host computes for a
// configuration -- whose twiddles and gather order are what gets uploaded -- has to BE that plan, or the transform would read tables
// laid out for other radices.  Checked when a configuration is first registered; a mismatch refuses the configuration instead of
// producing wrong output (advisor, round 4).
template <class P>
static bool fft_plan_equals(const lc3_cfg &c) {
    bool ok = c.nfft == P::nfft && c.n_stages == P::n_stages;
    for (int i = 0; i < 6; i++) ok = ok && c.radix[i] == P::radix[i] && c.m[i] == P::m[i] && c.fstride[i] == P::fstride[i] && c.inv_m[i] == P::inv_m[i];
    return ok;
}
static bool fft_plan_consistent(const lc3_cfg &c) {
    switch (c.nf) {
    case 480: return fft_plan_equals<lc3_fft_plan_480>(c);
    case 360: return fft_plan_equals<lc3_fft_plan_360>(c);
    case 320: return fft_plan_equals<lc3_fft_plan_320>(c);
    case 240: return fft_plan_equals<lc3_fft_plan_240>(c);
    case 180: return fft_plan_equals<lc3_fft_plan_180>(c);
    case 160: return fft_plan_equals<lc3_fft_plan_160>(c);
    case 120: return fft_plan_equals<lc3_fft_plan_120>(c);
    case 80: return fft_plan_equals<lc3_fft_plan_80>(c);
    case 60: return fft_plan_equals<lc3_fft_plan_60>(c);
    default: return true;  // (no plan by length: the run-time view reads the configuration's own)
    }
}

int cfg_acquire(HostCfg &h, int frame_us, int fs_hz) {
    static const int fs_tab[6] = {8000, 16000, 24000, 32000, 44100, 48000};
    int k = -1;
    for (int i = 0; i < 6; i++)
        if (fs_tab[i] == fs_hz) k = i;
    if (k < 0 || (frame_us != 7500 && frame_us != 10000)) return LC3GPU_EINVAL;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= LC3_MAX_DEVICES) return LC3GPU_EINVAL;
    const int slot = 2 * k + (frame_us == 10000);
    std::lock_guard<std::mutex> lock(g_cfgs.mu);
    if (!g_cfgs.tables[dev]) {
        for (lc3_tu_register_fn reg : lc3_tu_registers) HIP_TRY((hipError_t)reg(-1, nullptr, 1));  // every translation unit's tables
        g_cfgs.tables[dev] = true;
    }
    if (!g_cfgs.ready[dev][slot]) {
        lc3_cfg c;
        lc3_host_plan pl;
        if (lc3_make_config(c, frame_us, fs_hz) || lc3_make_plan(c, pl)) return LC3GPU_EINVAL;
        if (!fft_plan_consistent(c)) return LC3GPU_EUNSUPPORTED;
        const size_t bytes_tw = sizeof(lc3_cpx) * (size_t)c.nfft;
        const size_t bytes_perm = (sizeof(uint16_t) * (size_t)c.nfft + 15) & ~(size_t)15;
        const size_t bytes_poly = (sizeof(float) * (size_t)c.p_up * (size_t)c.resamp_stride + 15) & ~(size_t)15;
        const size_t bytes_lw = (sizeof(float) * (size_t)c.ne + 15) & ~(size_t)15;
        const size_t bytes_lb = ((size_t)c.nf + 15) & ~(size_t)15;
        char *base = nullptr;
        HIP_TRY(hipMalloc((void **)&base, 2 * bytes_tw + bytes_perm + bytes_poly + bytes_lw + bytes_lb + LC3_STAGE_IMAGE_BYTES));
        const int rc = cfg_upload(c, pl, base, bytes_tw, bytes_perm, bytes_poly, bytes_lw, bytes_lb, slot);
        if (rc) {
            (void)hipFree(base);  // nothing published: the slot stays unregistered
            return rc;
        }
        g_cfgs.cfg[dev][slot] = c;
        g_cfgs.ready[dev][slot] = true;
    }
    h.c = g_cfgs.cfg[dev][slot];
    h.slot = slot;
    // LC3GPU_GENERIC=1 keeps every configuration on the run-time view (test aid: both instantiations must agree)
    static const bool generic_only = std::getenv("LC3GPU_GENERIC") != nullptr && std::atoi(std::getenv("LC3GPU_GENERIC")) != 0;
    h.view = 0;
    if (!generic_only) {
#define LC3_X(i, V) if (h.view == 0 && V::matches(h.c)) h.view = i;
        LC3_FOR_EACH_VIEW(LC3_X)
#undef LC3_X
    }
    return LC3GPU_OK;
}

// optional per-kernel timing with HIP events recorded on the launch stream (bench.py roofline).  Events come from a pool
// that lives as long as the handle: nothing is created inside a timed region once the pool has warmed up.
// after a kernel launch inside a batch call: on failure the marks of this call are forgotten (the timer's intervals stay aligned) and
// what was already enqueued is still recorded as the handle's work in flight, so that a later call or quiesce() orders behind it
#define LC3_LAUNCH_CHECK(h, stream, t0)               \
    do {                                              \
        hipError_t e_ = hipGetLastError();            \
        if (e_ != hipSuccess) {                       \
            g_last_hip = (int)e_;                     \
            (h)->timer.rollback(t0);                  \
            (void)(h)->order_end(stream);             \
            return LC3GPU_EHIP;                       \
        }                                             \
    } while (0)
// a stage event of the caller behind the kernel just queued (mixed handles; the uniform ones: encode_kernels / decode_kernels)
#define LC3_STAGE_RECORD(h, stage, stream, t0)        \
    do {                                              \
        const int rc_ = (h)->stage_record(stage, stream); \
        if (rc_) {                                    \
            (h)->timer.rollback(t0);                  \
            (void)(h)->order_end(stream);             \
            return rc_;                               \
        }                                             \
    } while (0)
struct KernelTimer {
    bool enabled = false;
    std::vector<hipEvent_t> pool;  // every event ever created for this handle
    std::vector<int8_t> slot;      // per recorded mark: the ms[] slot of the interval that ENDS at it, -1 for the first mark of a chain
    std::vector<int32_t> prev;     // per recorded mark: the mark its interval STARTS at (the one before it on the same HIP stream), -1: none
    size_t used = 0;               // events of the pool holding a recorded mark
    size_t call_start = 0;         // `used` when the current call began
    int last[3] = {-1, -1, -1};    // per chain of the current call (0: the caller's stream, 1 / 2: the handle's internal streams): its latest mark
    double ms[4] = {0.0, 0.0, 0.0, 0.0};
    long launches = 0;             // batch calls that were timed
    int period = 1;                // every period-th batch call is timed (an event after every kernel costs the stream ~4 us each)
    long calls = 0;
    bool active = false;           // the current call is one of them
    // a batch call: arm(); then per HIP stream it launches on (chain), mark(stream, -1, chain) before its first kernel there and
    // mark(stream, slot of that kernel, chain) after every kernel.  begin(stream) = arm + the first mark of chain 0.
    bool arm() {
        active = enabled && (calls++ % (long)period) == 0;
        call_start = used;
        last[0] = last[1] = last[2] = -1;
        if (active) launches += 1;
        return active;
    }
    void begin(hipStream_t s) {
        arm();
        mark(s, -1);
    }
    void set(int enable) {
        enabled = enable != 0;
        period = enable > 1 ? enable : 1;
        calls = 0;
        active = false;
    }
    void mark(hipStream_t s, int sl, int chain = 0) {
        if (!active) return;
        if (used == pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) {  // no event: this call is not timed at all (a dropped mark would charge the
                (void)hipGetLastError();             // next interval, spanning two kernels, to the wrong slot)
                rollback(call_start);
                return;
            }
            pool.push_back(e);
            slot.push_back(0);
            prev.push_back(-1);
        }
        slot[used] = (int8_t)sl;
        prev[used] = sl < 0 ? -1 : last[chain];
        last[chain] = (int)used;
        (void)hipEventRecord(pool[used++], s);
    }
    // a call failed after its first mark: forget the marks it recorded (the events stay in the pool)
    void rollback(size_t to) {
        if (to > used) return;
        if (active) launches -= 1;
        active = false;
        used = to;
    }
    // synchronises; folds the recorded intervals into ms[] and returns the events to the pool
    void collect() {
        for (size_t i = used; i-- > 0;) (void)hipEventSynchronize(pool[i]);
        for (size_t i = 0; i < used; i++) {
            if (slot[i] < 0 || prev[i] < 0) continue;
            float d = 0.f;
            if (hipEventElapsedTime(&d, pool[(size_t)prev[i]], pool[i]) == hipSuccess) ms[slot[i]] += d;
        }
        used = 0;
        call_start = 0;
        active = false;
    }
    void release() {
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
        pool.clear();
        slot.clear();
        prev.clear();
        used = 0;
    }
};

// makes the handle's device current for the duration of a call (handles are bound to the device they were created on)
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { ok = false; return; }
        if (cur != dev) {
            if (hipSetDevice(dev) != hipSuccess) { ok = false; return; }
            prev = cur;
        }
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define LC3_ON_DEVICE(h)                      \
    DeviceGuard guard_((h)->device);          \
    if (!guard_.ok) return LC3GPU_EHIP

// one stream of a mixed-configuration handle, in the caller's order
struct MixedStream {
    int group = 0, internal = 0;  // its group and its index in the handle's internal (sorted) order
};
struct GroupHost {
    HostCfg h;
    int nbytes = 0, first_stream = 0, n_streams = 0;
};

// what encoder and decoder handles share: the device binding, the ordering of launches across HIP streams (every launch of a
// handle uses the handle's plane buffers: a launch on another stream than the previous one first waits for it), the timer,
// and the description of a mixed-configuration handle
struct HandleCommon {
    int device = 0;
    KernelTimer timer;
    HostCfg h;  // the configuration of a uniform handle
    int num_channels = 0;
    bool mixed = false;
    std::vector<GroupHost> groups;
    std::vector<MixedStream> streams;      // caller order
    std::vector<int> caller_of_internal;   // internal index -> caller index
    lc3_stream_io *d_tab = nullptr;        // per internal stream
    unsigned *d_pc_timeouts = nullptr;     // [0] producer / consumer pair halves that gave up on their partner (sticky; *_pair_timeouts), [2..3] address of h_pc_flag
    volatile unsigned *h_pc_flag = nullptr;  // one word of pinned host memory a kernel sets when a pair half gives up (lc3_pc_gave_up)
    int pc_health_alloc() {
        HIP_TRY(hipHostMalloc((void **)&h_pc_flag, sizeof(unsigned), hipHostMallocDefault));
        *h_pc_flag = 0;
        HIP_TRY(hipMalloc((void **)&d_pc_timeouts, 4 * sizeof(unsigned)));
        const unsigned long long a = (unsigned long long)(uintptr_t)h_pc_flag;
        const unsigned init[4] = {0u, 0u, (unsigned)(a & 0xFFFFFFFFull), (unsigned)(a >> 32)};
        HIP_TRY(hipMemcpy(d_pc_timeouts, init, sizeof init, hipMemcpyHostToDevice));
        return LC3GPU_OK;
    }
    bool pc_optin_done = false;            // the pair kernels' dynamic-LDS opt-in has been made for this handle's device (no lock per call)
    hipStream_t last_stream = nullptr;
    hipEvent_t done = nullptr;
    bool has_work = false;
    // lc3gpu_*_bind_stream: the caller promises that every batch call of the handle comes on ONE stream that outlives the handle (the pipeline
    // object's).  Then the per-call event below is not needed -- nothing ever has to be ordered behind the handle's work on another stream,
    // and "everything the handle has launched" is simply that stream: one event record less per call (~0.1 % of a 1.1 ms step each)
    hipStream_t bound_stream = nullptr;
    bool is_bound = false;
    // The split path (lc3_split_parts): a batch call of a full batch runs as two halves of its streams on two internal HIP streams,
    // forked from and joined to the caller's stream by events, so that the lane-per-frame kernels of one half (one wave per SIMD, a
    // third of its issue slots idle) run beside the wave-per-stream kernels of the other.
    hipStream_t sub[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_stage = nullptr, ev_join[2] = {nullptr, nullptr};
    bool last_split = false;  // the handle's latest work ended on the internal streams (ev_join), not on last_stream (done)
    // The host-resident batch path (lc3gpu_encode_host / lc3gpu_decode_host): device staging buffers for two channel ranges in flight, one per
    // internal stream; while such a call runs the in-call split (which owns the same two streams) is off
    void *d_stage_in[2] = {nullptr, nullptr}, *d_stage_out[2] = {nullptr, nullptr}, *d_stage_flag[2] = {nullptr, nullptr};
    size_t stage_in_bytes = 0, stage_out_bytes = 0, stage_flag_bytes = 0;
    bool in_host_call = false;
    // ... and three internal streams of their own: copies in, kernels, copies out -- each direction's copy engine and the compute queue run
    // side by side, tied by events per staging buffer (host_lane_*)
    hipStream_t host_s[3] = {nullptr, nullptr, nullptr};
    hipEvent_t host_in_done[2] = {nullptr, nullptr}, host_k_done[2] = {nullptr, nullptr}, host_out_done[2] = {nullptr, nullptr};
    int ensure_host_lane() {
        if (host_s[0]) return LC3GPU_OK;
        for (int i = 0; i < 3; i++) HIP_TRY(hipStreamCreateWithFlags(&host_s[i], hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipEventCreateWithFlags(&host_in_done[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&host_k_done[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&host_out_done[i], hipEventDisableTiming));
        }
        return LC3GPU_OK;
    }
    int stage_reserve(size_t in_bytes, size_t out_bytes, size_t flag_bytes) {
        auto grow = [](void **buf, size_t &have, size_t want) -> int {
            if (want <= have) return LC3GPU_OK;
            for (int i = 0; i < 2; i++) {
                if (buf[i]) (void)hipFree(buf[i]);
                buf[i] = nullptr;
            }
            have = 0;
            for (int i = 0; i < 2; i++) HIP_TRY(hipMalloc(&buf[i], want));
            have = want;
            return LC3GPU_OK;
        };
        int rc = grow(d_stage_in, stage_in_bytes, in_bytes);
        if (rc == LC3GPU_OK) rc = grow(d_stage_out, stage_out_bytes, out_bytes);
        if (rc == LC3GPU_OK) rc = grow(d_stage_flag, stage_flag_bytes, flag_bytes);
        return rc;
    }
    // Stage events (lc3gpu_*_stage_event): events of the CALLER, recorded behind a stage's kernel(s) of every batch call, so that a caller
    // with several handles on several HIP streams can start another handle's work beside a chosen part of this one's
    hipEvent_t stage_ev[LC3GPU_MAX_STAGES] = {};
    int stage_record(int stage, hipStream_t s) {  // (chain 0 only: the split path records every stage at its join)
        if (stage_ev[stage]) HIP_TRY(hipEventRecord(stage_ev[stage], s));
        return LC3GPU_OK;
    }
    int stage_record_all(hipStream_t s) {
        for (int i = 0; i < LC3GPU_MAX_STAGES; i++) {
            const int rc = stage_record(i, s);
            if (rc) return rc;
        }
        return LC3GPU_OK;
    }
    int stage_set(int stage, void *ev) {
        if (stage < 0 || stage >= LC3GPU_MAX_STAGES) return LC3GPU_EINVAL;
        stage_ev[stage] = (hipEvent_t)ev;
        return LC3GPU_OK;
    }

    // A handle's launches share its scratch planes and its state blobs: a call on another stream than the previous one waits for
    // everything the handle has queued so far.  What stands for "so far" is an event the HANDLE owns, recorded when the work was
    // queued (`done` on the caller's stream, or the two join events of the split path): the caller's stream of an earlier call is never
    // touched again, it may have been destroyed or its address reused since.
    int order_begin(hipStream_t s) {
        // a pair half of an EARLIER call gave up (its frames left zero-filled / concealed): said once, by the first call that sees it;
        // that call launches nothing and may be repeated; the count stays in lc3gpu_*_pair_timeouts
        if (h_pc_flag && *h_pc_flag) {
            *h_pc_flag = 0;
            return LC3GPU_EPAIR;
        }
        if (is_bound) return s == bound_stream ? LC3GPU_OK : LC3GPU_EINVAL;  // (a bound handle takes batch calls on its stream only)
        if (has_work && s != last_stream) {
            if (last_split) {
                HIP_TRY(hipStreamWaitEvent(s, ev_join[0], 0));
                HIP_TRY(hipStreamWaitEvent(s, ev_join[1], 0));
            } else HIP_TRY(hipStreamWaitEvent(s, done, 0));
        }
        return LC3GPU_OK;
    }
    int order_end(hipStream_t s, bool split = false) {
        if (!split && !is_bound) HIP_TRY(hipEventRecord(done, s));  // (the split path has recorded its join events)
        last_stream = s;
        last_split = split;
        has_work = true;
        return LC3GPU_OK;
    }
    // host waits for everything the handle has launched
    int quiesce() {
        if (!has_work) return LC3GPU_OK;
        if (is_bound) {
            HIP_TRY(hipStreamSynchronize(bound_stream));
            return LC3GPU_OK;
        }
        if (last_split) {
            HIP_TRY(hipEventSynchronize(ev_join[0]));
            HIP_TRY(hipEventSynchronize(ev_join[1]));
        } else HIP_TRY(hipEventSynchronize(done));
        return LC3GPU_OK;
    }
    // the internal streams and events of the split path, created at its first use
    int ensure_split() {
        if (sub[0]) return LC3GPU_OK;
        hipStream_t a = nullptr, b = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        if (hipStreamCreateWithFlags(&b, hipStreamNonBlocking) != hipSuccess) {
            (void)hipStreamDestroy(a);
            g_last_hip = (int)hipGetLastError();
            return LC3GPU_EHIP;
        }
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < 4; i++)
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
                g_last_hip = (int)hipGetLastError();
                for (int j = 0; j < i; j++) (void)hipEventDestroy(ev[j]);
                (void)hipStreamDestroy(a);
                (void)hipStreamDestroy(b);
                return LC3GPU_EHIP;
            }
        sub[0] = a;
        sub[1] = b;
        ev_fork = ev[0];
        ev_stage = ev[1];
        ev_join[0] = ev[2];
        ev_join[1] = ev[3];
        return LC3GPU_OK;
    }
    const HostCfg &cfg_of_channel(int ch) const { return mixed ? groups[(size_t)streams[(size_t)ch].group].h : h; }
    int internal_of_channel(int ch) const { return mixed ? streams[(size_t)ch].internal : ch; }
    void release_common() {
        timer.release();
        if (done) (void)hipEventDestroy(done);
        if (d_tab) (void)hipFree(d_tab);
        if (d_pc_timeouts) (void)hipFree(d_pc_timeouts);
        d_pc_timeouts = nullptr;
        if (h_pc_flag) (void)hipHostFree((void *)h_pc_flag);
        h_pc_flag = nullptr;
        for (int i = 0; i < 2; i++) {
            for (void **b : {&d_stage_in[i], &d_stage_out[i], &d_stage_flag[i]}) {
                if (*b) (void)hipFree(*b);
                *b = nullptr;
            }
        }
        stage_in_bytes = stage_out_bytes = stage_flag_bytes = 0;
        for (int i = 0; i < 3; i++) {
            if (host_s[i]) (void)hipStreamDestroy(host_s[i]);
            host_s[i] = nullptr;
        }
        for (int i = 0; i < 2; i++)
            for (hipEvent_t *ev : {&host_in_done[i], &host_k_done[i], &host_out_done[i]}) {
                if (*ev) (void)hipEventDestroy(*ev);
                *ev = nullptr;
            }
        for (hipEvent_t ev : {ev_fork, ev_stage, ev_join[0], ev_join[1]})
            if (ev) (void)hipEventDestroy(ev);
        for (hipStream_t st : sub)
            if (st) (void)hipStreamDestroy(st);
        done = ev_fork = ev_stage = ev_join[0] = ev_join[1] = nullptr;
        sub[0] = sub[1] = nullptr;
        d_tab = nullptr;
    }
};

// Grows a device buffer in stream order: the new block is allocated and the old one freed on `stream`, which (order_begin)
// already waits for every earlier launch of the handle -- no device-wide synchronisation in the middle of a pipeline.
template <class T>
int grow_async(T *&p, size_t bytes, hipStream_t stream) {
    static int async_ok = -1;  // does this runtime have the stream-ordered allocator?  decided at the first allocation
    if (async_ok != 0) {
        T *n = nullptr;
        if (hipMallocAsync((void **)&n, bytes, stream) == hipSuccess) {
            async_ok = 1;
            if (p) HIP_TRY(hipFreeAsync(p, stream));
            p = n;
            return LC3GPU_OK;
        }
        (void)hipGetLastError();
        if (async_ok == 1) { g_last_hip = (int)hipErrorOutOfMemory; return LC3GPU_EHIP; }
        async_ok = 0;
    }
    // no stream-ordered allocator: the conservative path
    HIP_TRY(hipDeviceSynchronize());
    if (p) (void)hipFree(p);
    p = nullptr;
    HIP_TRY(hipMalloc((void **)&p, bytes));
    return LC3GPU_OK;
}

// sorts the streams of a mixed handle by (configuration slot, frame bytes) and builds groups, tables and mappings
// refuse_8k: an encoder without LC3GPU_SPEC_8KHZ_ENCODE; min_bytes: 20 for encoders, 1 for decoders
int build_mixed(HandleCommon &hc, int n, const lc3gpu_stream_desc *descs, bool refuse_8k, int min_bytes) {
    if (!descs || n <= 0) return LC3GPU_EINVAL;
    struct Key { int slot, nbytes, idx; };
    std::vector<Key> keys((size_t)n);
    std::vector<lc3_cfg> cfgs((size_t)n);
    static const int fs_tab[6] = {8000, 16000, 24000, 32000, 44100, 48000};
    for (int i = 0; i < n; i++) {
        int rc = make_config(cfgs[(size_t)i], descs[i].frame_us, descs[i].fs_hz);
        if (rc) return rc;
        if (refuse_8k && descs[i].fs_hz == 8000) return LC3GPU_EUNSUPPORTED;  // bandwidth_detector.rs:36-37
        if (descs[i].nbytes < min_bytes || descs[i].nbytes > LC3_MAX_NE) return LC3GPU_ELENGTH;
        int k = 0;
        for (int q = 0; q < 6; q++)
            if (fs_tab[q] == descs[i].fs_hz) k = q;
        keys[(size_t)i] = {2 * k + (descs[i].frame_us == 10000), descs[i].nbytes, i};
    }
    std::vector<Key> sorted = keys;
    std::stable_sort(sorted.begin(), sorted.end(), [](const Key &a, const Key &b) { return a.slot != b.slot ? a.slot < b.slot : a.nbytes < b.nbytes; });
    hc.streams.assign((size_t)n, MixedStream());
    hc.caller_of_internal.assign((size_t)n, 0);
    hc.groups.clear();
    for (int j = 0; j < n; j++) {
        const Key &k = sorted[(size_t)j];
        if (hc.groups.empty() || hc.groups.back().h.slot != k.slot || hc.groups.back().nbytes != k.nbytes) {
            if (hc.groups.size() == LC3_MAX_GROUPS) return LC3GPU_EINVAL;
            GroupHost g;
            int rc = cfg_acquire(g.h, descs[k.idx].frame_us, descs[k.idx].fs_hz);
            if (rc) return rc;
            g.nbytes = k.nbytes;
            g.first_stream = j;
            hc.groups.push_back(g);
        }
        hc.groups.back().n_streams += 1;
        hc.streams[(size_t)k.idx].group = (int)hc.groups.size() - 1;
        hc.streams[(size_t)k.idx].internal = j;
        hc.caller_of_internal[(size_t)j] = k.idx;
    }
    // per-frame prefix offsets in the caller's order
    std::vector<lc3_stream_io> tab((size_t)n);
    long long po = 0, bo = 0;
    for (int i = 0; i < n; i++) {
        lc3_stream_io &e = tab[(size_t)hc.streams[(size_t)i].internal];
        e.pcm_off1 = po;
        e.byte_off1 = bo;
        e.flag_idx = i;
        e.pad = 0;
        po += cfgs[(size_t)i].nf;
        bo += descs[i].nbytes;
    }
    HIP_TRY(hipMalloc((void **)&hc.d_tab, sizeof(lc3_stream_io) * (size_t)n));
    HIP_TRY(hipMemcpy(hc.d_tab, tab.data(), sizeof(lc3_stream_io) * (size_t)n, hipMemcpyHostToDevice));
    hc.mixed = true;
    hc.num_channels = n;
    return LC3GPU_OK;
}

// the group table of one mixed launch (T frames per stream, fpb frames per workgroup of the frame kernels)
void fill_groups(const HandleCommon &hc, int T, unsigned fpb, lc3_groups &G, unsigned &wg_stream, unsigned &wg_frame, size_t &frames,
                 int &max_nbytes) {
    G.n = (int)hc.groups.size();
    G.pad = 0;
    wg_stream = wg_frame = 0;
    frames = 0;
    max_nbytes = 0;
    for (size_t i = 0; i < hc.groups.size(); i++) {
        const GroupHost &gh = hc.groups[i];
        lc3_group &g = G.g[i];
        g.slot = gh.h.slot;
        g.fixed = gh.h.view;  // the mixed kernels carry one body per configuration view
        g.first_stream = gh.first_stream;
        g.n_streams = gh.n_streams;
        g.wg_stream = (int)wg_stream;
        g.wg_frame = (int)wg_frame;
        g.nbytes = gh.nbytes;
        g.ne = gh.h.c.ne;
        g.nb = gh.h.c.nb;
        g.pad = 0;
        g.frame_base = (long long)frames;
        wg_stream += (unsigned)((gh.n_streams + LC3_WG_WAVES - 1) / LC3_WG_WAVES);
        wg_frame += (unsigned)(((size_t)gh.n_streams * (size_t)T + fpb - 1) / fpb);
        frames += (size_t)gh.n_streams * (size_t)T;
        if (gh.nbytes > max_nbytes) max_nbytes = gh.nbytes;
    }
}

}  // namespace

struct lc3gpu_encoder : HandleCommon {
    int spec_flags = 0;  // LC3GPU_SPEC_* (0 = the reference's behaviour)
    lc3_enc_state *d_states = nullptr;
    // staging for the single-frame host API
    int16_t *d_pcm1 = nullptr;
    uint8_t *d_out1 = nullptr;
    float *d_dbg = nullptr;
    std::vector<uint8_t> fresh_mask;  // per channel (internal order): 1 = state not yet materialised in HBM
    int32_t *d_planes = nullptr;      // packer planes, EP_WORDS words per frame
    float *d_mid = nullptr;           // mid planes (front half -> vector quantiser -> back half), MP_WORDS words per frame
    size_t planes_frames = 0;
};

struct lc3gpu_decoder : HandleCommon {
    lc3_dec_state *d_states = nullptr;
    uint8_t *d_in1 = nullptr;
    int16_t *d_pcm1 = nullptr;
    int32_t *d_planes = nullptr;   // parsed-frame planes, LC3_PLANE_WORDS words per frame
    size_t planes_frames = 0;      // capacity in frames (multiple of 64)
    float *d_dbg = nullptr;        // stage dumps of the diagnostic entry points (LC3_DBG_FLOATS), allocated at first use
    // lc3gpu_decoder_reset only NOTES that every channel is back to the constructed state: the next batch launch over all channels initialises
    // the states inside the synthesis kernel (fresh = 1, as the encoder's fresh_mask does); anything else that looks at the state blobs first
    // (a range or frame call, state_save, plc_events, the diagnostic calls) materialises them with the zero-frame launch (decoder_init_states)
    bool fresh_pending = false;
};

// frames per workgroup of the lane-per-frame kernels (= threads per workgroup).  LC3GPU_FPB overrides (tuning aid).
static unsigned lc3_frame_block(unsigned dflt) {
    static int env = -1;
    if (env < 0) {
        const char *v = std::getenv("LC3GPU_FPB");
        env = v ? std::atoi(v) : 0;
    }
    return (env == 64 || env == 128 || env == 256) ? (unsigned)env : dflt;
}
// ... clamped so that the dynamic LDS (fixed part + per-frame part) fits the default 64 KB
// LC3_LAUNCH_PREP_SYMBOLS for launches of at most 16 384 frames (lc3_dev_enc_pack.h): the preparation costs the back half
// ~24 us per 16 384 frames and saves the packer ~40 us whatever the launch size (measured: 2048 x 1 frames 295 -> 264 us,
// 16 384 x 1 486 -> 469 us, 16 384 x 4 1090 -> 1139 us for the four encoder kernels); LC3GPU_PREP_SYMBOLS=0 / 1 forces it
// off / on (tests)
// LC3GPU_PREP_SYMBOLS=2: the preparation as a kernel of its own between the back half and the packer (lc3_symbols_kernel; uniform
// handles), built to measure the third form
static int lc3_prep_symbols_mode(size_t n_frames_total) {  // 0: the packer derives its symbols, 1: back half, 2: lc3_symbols_kernel
    static const int forced = [] {
        const char *e = std::getenv("LC3GPU_PREP_SYMBOLS");
        return e ? (std::atoi(e) == 2 ? 2 : (std::atoi(e) != 0 ? 1 : 0)) : -1;
    }();
    return forced >= 0 ? forced : (n_frames_total <= 16384 ? 1 : 0);
}
static int lc3_prep_symbols_flag(size_t n_frames_total, bool mixed = false) {
    const int m = lc3_prep_symbols_mode(n_frames_total);
    return (m == 1 || (m == 2 && mixed)) ? LC3_LAUNCH_PREP_SYMBOLS : 0;
}
// Where the spectrum of a parsed frame is reconstructed (D4-D8).  Three forms, the same arithmetic line by line:
//   LC3_RECON_LANE  in the parse kernel, by the lane that parsed the frame (lc3_reconstruct_frame): full batches.  One pass over the
//                   lines with the TNS lattice as a four-line wavefront (lc3_tns_lattice4);
//   LC3_RECON_LATE  in the synthesis kernel, by the stream's wave (lc3_dec_reconstruct_wave): launches of a few frames, where a lane
//                   walking its frame alone is what the caller waits for (measured, parse + synthesis, us: 1 x 1 frames 301 -> 157,
//                   1024 x 1 361 -> 260, 4096 x 4 402 -> 354, 1024 x 16 428 -> 469);
//   LC3_RECON_WAVE  two kernels of its own between parser and synthesis (lc3_dev_dec_recon.h): one WAVE per frame for what is parallel
//                   over a frame's lines, then one LANE per frame for the TNS lattice.  Built in round 3 to take the line-parallel
//                   work out of the one-wave-per-SIMD parser; measured on the 65 536-frame batch (ms): parser 0.220 + wave-per-frame
//                   0.093 + lattice 0.070 = 0.383 against 0.339 for LC3_RECON_LANE -- the two extra trips of every spectrum through
//                   HBM cost more than the occupancy gains.  Kept selectable and tested, not the default.
// LC3GPU_RECON=lane|late|wave forces a form for every launch (tests); LC3GPU_LATE_RECON=0 / 1 is the older spelling of lane / late.
enum { LC3_RECON_LANE = 0, LC3_RECON_LATE = 1, LC3_RECON_WAVE = 2 };
static int lc3_recon_mode(size_t n_frames_total, int frames_per_stream) {
    static const int forced = [] {
        const char *m = std::getenv("LC3GPU_RECON");
        if (m) {
            if (!std::strcmp(m, "lane")) return (int)LC3_RECON_LANE;
            if (!std::strcmp(m, "late")) return (int)LC3_RECON_LATE;
            if (!std::strcmp(m, "wave")) return (int)LC3_RECON_WAVE;
        }
        const char *e = std::getenv("LC3GPU_LATE_RECON");
        return e ? (std::atoi(e) != 0 ? (int)LC3_RECON_LATE : (int)LC3_RECON_LANE) : -1;
    }();
    if (forced >= 0) return forced;
    return (n_frames_total <= 16384 && frames_per_stream <= 4) ? LC3_RECON_LATE : LC3_RECON_LANE;
}
// tuning aid: extra dynamic LDS (bytes) a wave-per-stream kernel is launched with, which lowers how many of its workgroups a CU holds
// (LC3GPU_LDS_PAD_FRONT / _BACK / _SYNTH); 0 = none
static size_t lc3_lds_pad(int which) {
    static const size_t pad[3] = {
        (size_t)(std::getenv("LC3GPU_LDS_PAD_FRONT") ? std::atoi(std::getenv("LC3GPU_LDS_PAD_FRONT")) : 0),
        (size_t)(std::getenv("LC3GPU_LDS_PAD_BACK") ? std::atoi(std::getenv("LC3GPU_LDS_PAD_BACK")) : 0),
        (size_t)(std::getenv("LC3GPU_LDS_PAD_SYNTH") ? std::atoi(std::getenv("LC3GPU_LDS_PAD_SYNTH")) : 0)};
    return pad[which];
}
// The producer / consumer packer (lc3_pack_pc_kernel): the form of full batches (where the packer derives its symbols itself) unless
// LC3GPU_PACK_PC=0; its ring buffers take the workgroup just beyond the default 64 KB of dynamic LDS
static bool lc3_pack_pc_enabled() {
    static const bool on = [] {
        const char *e = std::getenv("LC3GPU_PACK_PC");
        return !(e && std::atoi(e) == 0);
    }();
    return on;
}
static int lc3_pack_pc_optin() {
    static bool done[LC3_MAX_DEVICES] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= LC3_MAX_DEVICES) return LC3GPU_EINVAL;
    if (done[dev]) return LC3GPU_OK;
    HIP_TRY(hipFuncSetAttribute((const void *)lc3_pack_pc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void *)lc3_pack_pc_mixed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done[dev] = true;
    return LC3GPU_OK;
}
// Frames per workgroup of the pair kernels: 128 (four waves, one per SIMD; ~40 KB of LDS at 150-byte frames).  Such a workgroup takes
// the place of exactly ONE workgroup of a wave-per-stream kernel (40 KB, a wave per SIMD) when another handle's call runs beside it on
// another HIP stream; with 256 frames (70 KB, two waves per SIMD) it displaced two for as long as it ran.  Measured: two-stream
// arrangement 48.9 -> 50.3 M frames/s, one stream 46.0 -> 45.9 M (parse 0.281 -> 0.287, pack 0.163 -> 0.159 ms alone;
// profiles/r04_pair_workgroup_size.txt)
static unsigned lc3_pack_pc_fpb(int nbytes) {
    unsigned fpb = lc3_frame_block(128u);
    while (fpb > 64u && lc3_pack_pc_lds(fpb, nbytes) > (size_t)(160 * 1024)) fpb >>= 1;
    return fpb;
}
// The producer / consumer parser (lc3_parse_pc_kernel): the form of full batches unless LC3GPU_PARSE_PC=0.  Its ring buffers take the
// workgroup beyond the default 64 KB of dynamic LDS: opt in once per device (every instantiation); frames per workgroup as many as fit
// the 160 KB of a CU
#define LC3_PC_LDS_MAX (160 * 1024)
static bool lc3_parse_pc_enabled() {
    static const bool on = [] {
        const char *e = std::getenv("LC3GPU_PARSE_PC");
        return !(e && std::atoi(e) == 0);
    }();
    return on;
}
static int lc3_parse_pc_optin() {
    static bool done[LC3_MAX_DEVICES] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= LC3_MAX_DEVICES) return LC3GPU_EINVAL;
    if (done[dev]) return LC3GPU_OK;
    HIP_TRY(hipFuncSetAttribute((const void *)LC3_MIXED_LAUNCH(lc3_parse_pc_mixed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LC3_PC_LDS_MAX));
    HIP_TRY(hipFuncSetAttribute((const void *)lc3_parse_pc_kernel<lc3_cfg_any>, hipFuncAttributeMaxDynamicSharedMemorySize, LC3_PC_LDS_MAX));
#define LC3_X(i, V) HIP_TRY(hipFuncSetAttribute((const void *)lc3_parse_pc_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, LC3_PC_LDS_MAX));
    LC3_FOR_EACH_VIEW(LC3_X)
#undef LC3_X
    done[dev] = true;
    return LC3GPU_OK;
}
static unsigned lc3_parse_pc_fpb(int nbytes) {
    unsigned fpb = lc3_frame_block(128u);  // (see lc3_pack_pc_fpb)
    while (fpb > 64u && lc3_parse_pc_lds(fpb, nbytes) > (size_t)LC3_PC_LDS_MAX) fpb >>= 1;
    return fpb;
}
// the TNS kernels use more than the default 64 KB of dynamic LDS: opt in once per device (every instantiation)
static int lc3_tns_lds_optin() {
    static bool done[LC3_MAX_DEVICES] = {};
    static std::mutex mu;  // handles of different host threads may reach this at the same time
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= LC3_MAX_DEVICES) return LC3GPU_EINVAL;
    if (done[dev]) return LC3GPU_OK;
    HIP_TRY(hipFuncSetAttribute((const void *)lc3_tns_kernel<lc3_cfg_any>, hipFuncAttributeMaxDynamicSharedMemorySize, LC3_TNS_LDS));
#define LC3_X(i, V) HIP_TRY(hipFuncSetAttribute((const void *)lc3_tns_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, LC3_TNS_LDS));
    LC3_FOR_EACH_VIEW(LC3_X)
#undef LC3_X
    HIP_TRY(hipFuncSetAttribute((const void *)LC3_MIXED_LAUNCH(lc3_tns_mixed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LC3_TNS_LDS));
    done[dev] = true;
    return LC3GPU_OK;
}
// workgroups of the reconstruction kernel: eight per CU (LC3GPU_RECON_GRID overrides: tuning aid)
static size_t lc3_recon_grid() {
    static const size_t n = [] {
        const char *e = std::getenv("LC3GPU_RECON_GRID");
        if (e && std::atoi(e) > 0) return (size_t)std::atoi(e);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return (size_t)(cus > 0 ? cus : 256) * 8;
    }();
    return n;
}
static unsigned lc3_frame_block_fit(size_t lds_fixed, size_t lds_per_frame) {
    unsigned fpb = lc3_frame_block(256u);
    while (fpb > 64u && lds_fixed + (size_t)fpb * lds_per_frame + 8 > 65536u) fpb >>= 1;
    return fpb;
}

static int encoder_reserve_planes(lc3gpu_encoder *e, size_t frames, hipStream_t stream) {
    const size_t need = (frames + 63) / 64 * 64;
    if (need <= e->planes_frames) return LC3GPU_OK;
    e->planes_frames = 0;
    int rc = grow_async(e->d_planes, need * (size_t)EP_WORDS * sizeof(int32_t), stream);
    if (rc == LC3GPU_OK) rc = grow_async(e->d_mid, need * (size_t)MP_WORDS * sizeof(float), stream);
    if (rc) return rc;
    e->planes_frames = need;
    return LC3GPU_OK;
}
static int decoder_reserve_planes(lc3gpu_decoder *d, size_t frames, hipStream_t stream) {
    const size_t need = (frames + 63) / 64 * 64;
    if (need <= d->planes_frames) return LC3GPU_OK;
    d->planes_frames = 0;
    int rc = grow_async(d->d_planes, need * (size_t)LC3_PLANE_WORDS * sizeof(int32_t), stream);
    if (rc) return rc;
    d->planes_frames = need;
    return LC3GPU_OK;
}

// ---- state blobs (checkpoint / CPU cross-checks).  Per channel, in the caller's channel order: a 16-byte header that says what
// the payload is (which side, layout version, payload size, the channel's configuration and switches) followed by the state struct.
// A blob from a handle of another configuration, another descriptor order, the other side or another layout version is refused
// (LC3GPU_EINVAL) instead of being taken as state.
struct lc3_state_header {
    uint32_t magic;       // 'LC3E' / 'LC3D'
    uint16_t version;     // layout version of the state structs (LC3_STATE_VERSION)
    uint16_t payload16;   // payload bytes / 16
    uint32_t fs_hz;
    uint16_t frame_us10;  // frame duration in units of 100 us (75 / 100)
    uint16_t spec_flags;  // LC3GPU_SPEC_* of an encoder channel, 0 for decoders
};
static_assert(sizeof(lc3_state_header) == 16, "state blob header");
#define LC3_STATE_VERSION 3
#define LC3_STATE_MAGIC_ENC 0x4533434cu  // "LC3E"
#define LC3_STATE_MAGIC_DEC 0x4433434cu  // "LC3D"
static lc3_state_header lc3_state_header_of(const HandleCommon &hc, int channel, uint32_t magic, size_t payload, int spec_flags) {
    const HostCfg &h = hc.cfg_of_channel(channel);
    lc3_state_header hd;
    hd.magic = magic;
    hd.version = LC3_STATE_VERSION;
    hd.payload16 = (uint16_t)(payload / 16);
    hd.fs_hz = (uint32_t)h.c.fs;
    hd.frame_us10 = (uint16_t)(h.c.n_ms_10 ? 100 : 75);
    hd.spec_flags = (uint16_t)spec_flags;
    return hd;
}
// device states (internal order) -> blobs (caller order), and back after the headers have been checked
template <class ST>
static int state_blobs_save(HandleCommon &hc, const ST *d_states, uint32_t magic, int spec_flags, void *host_dst, size_t nbytes) {
    const size_t per = sizeof(lc3_state_header) + sizeof(ST);
    if (nbytes != per * (size_t)hc.num_channels) return LC3GPU_ELENGTH;
    std::vector<ST> tmp((size_t)hc.num_channels);
    HIP_TRY(hipMemcpy(tmp.data(), d_states, sizeof(ST) * tmp.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < hc.num_channels; i++) {
        char *dst = (char *)host_dst + per * (size_t)i;
        const lc3_state_header hd = lc3_state_header_of(hc, i, magic, sizeof(ST), spec_flags);
        std::memcpy(dst, &hd, sizeof(hd));
        std::memcpy(dst + sizeof(hd), &tmp[(size_t)hc.internal_of_channel(i)], sizeof(ST));
    }
    return LC3GPU_OK;
}
template <class ST>
static int state_blobs_load(HandleCommon &hc, ST *d_states, uint32_t magic, int spec_flags, const void *host_src, size_t nbytes) {
    const size_t per = sizeof(lc3_state_header) + sizeof(ST);
    if (nbytes != per * (size_t)hc.num_channels) return LC3GPU_ELENGTH;
    std::vector<ST> tmp((size_t)hc.num_channels);
    for (int i = 0; i < hc.num_channels; i++) {
        const char *src = (const char *)host_src + per * (size_t)i;
        const lc3_state_header want = lc3_state_header_of(hc, i, magic, sizeof(ST), spec_flags);
        if (std::memcmp(src, &want, sizeof(want)) != 0) return LC3GPU_EINVAL;  // another configuration / order / side / version
        std::memcpy(&tmp[(size_t)hc.internal_of_channel(i)], src + sizeof(want), sizeof(ST));
    }
    HIP_TRY(hipMemcpy(d_states, tmp.data(), sizeof(ST) * tmp.size(), hipMemcpyHostToDevice));
    return LC3GPU_OK;
}

extern "C" {

int lc3gpu_version(void) { return 300; }

const char *lc3gpu_strerror(int code) {
    switch (code) {
    case LC3GPU_OK: return "ok";
    case LC3GPU_EINVAL: return "invalid argument";
    case LC3GPU_ECHANNEL: return "channel index out of range";
    case LC3GPU_ELENGTH: return "buffer length does not match the configuration";
    case LC3GPU_EBITS: return "only 16 bits per audio sample supported";
    case LC3GPU_EHIP: return "HIP runtime error";
    case LC3GPU_ENODEVICE: return "no HIP device";
    case LC3GPU_EUNSUPPORTED: return "configuration not supported by the reference";
    case LC3GPU_EPAIR: return "a producer / consumer pair of an earlier call gave up: frames of that call are zero-filled (encoder) or concealed (decoder)";
    default: return "unknown error";
    }
}

int lc3gpu_last_hip_error(void) { return g_last_hip; }

int lc3gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lc3gpu_config(int frame_us, int fs_hz, int out[7]) {
    lc3_cfg c;
    if (!out) return LC3GPU_EINVAL;
    int rc = make_config(c, frame_us, fs_hz);
    if (rc) return rc;
    out[0] = c.fs_ind; out[1] = c.fs; out[2] = c.ne; out[3] = c.n_ms_10; out[4] = c.nb; out[5] = c.nf; out[6] = c.z;
    return LC3GPU_OK;
}

int lc3gpu_encoder_working_buffer_lengths(int num_channels, int frame_us, int fs_hz, int64_t out[3]) {
    lc3_cfg c;
    if (!out || num_channels < 0) return LC3GPU_EINVAL;
    int rc = make_config(c, frame_us, fs_hz);
    if (rc) return rc;
    // lc3_encoder.rs:194-209, modified_dct.rs:67-71, long_term_post_filter.rs:93-137, dct_iv.rs:69-71
    const int64_t integer_len = 2 * c.nf + (c.hist + c.nf) + c.ne;
    const int64_t scaler_len = (c.len12 + c.delay12 + 232) + (64 + 114) + c.nf + c.nb;
    const int64_t complex_len = c.nf / 2 * 4;
    out[0] = integer_len * num_channels;
    out[1] = scaler_len * num_channels;
    out[2] = complex_len * num_channels;
    return LC3GPU_OK;
}

int lc3gpu_decoder_working_buffer_lengths(int num_channels, int frame_us, int fs_hz, int64_t out[2]) {
    lc3_cfg c;
    if (!out || num_channels < 0) return LC3GPU_EINVAL;
    int rc = make_config(c, frame_us, fs_hz);
    if (rc) return rc;
    // lc3_decoder.rs:155-162, modified_dct.rs:153-166, long_term_post_filter.rs:104-140
    const int64_t dct = c.nf / 2 + (c.nf - c.ne) + (c.nf - c.z) + 2 * c.nf + c.nf;
    const int64_t c_num = c.l_num + 1, c_den = c.l_den + 1, scratch = c.l_num + c.norm;
    const int64_t ltpf = c_den * 3 + c_num * 2 + 2 * (int64_t)c.nf * c.num_mem_blocks + scratch;
    out[0] = (c.ne + c.ne + dct + ltpf) * num_channels;
    out[1] = (int64_t)(c.nf / 2 * 4) * num_channels;
    return LC3GPU_OK;
}

// ---------------------------------------------------------------------------------------------
static int encoder_alloc(lc3gpu_encoder *e) {
    HIP_TRY(hipGetDevice(&e->device));
    HIP_TRY(hipEventCreateWithFlags(&e->done, hipEventDisableTiming));
    { const int rc = e->pc_health_alloc(); if (rc) return rc; }
    HIP_TRY(hipMalloc((void **)&e->d_states, sizeof(lc3_enc_state) * (size_t)e->num_channels));
    // staging of the *_frame calls: pinned host memory the kernels read / write in place (no copy engine round trips)
    HIP_TRY(hipHostMalloc((void **)&e->d_pcm1, sizeof(int16_t) * LC3_MAX_NF, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&e->d_out1, LC3_MAX_NE, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&e->d_dbg, sizeof(float) * LC3_ENC_DBG_FLOATS));
    e->fresh_mask.assign((size_t)e->num_channels, 1);
    return encoder_reserve_planes(e, (size_t)e->num_channels, nullptr);
}

int lc3gpu_encoder_create(lc3gpu_encoder **out, int num_channels, int frame_us, int fs_hz) {
    return lc3gpu_encoder_create_spec(out, num_channels, frame_us, fs_hz, 0);
}

// test hooks read from the environment when a handle is created (LC3GPU_GENERIC, LC3GPU_SEQ_SUMS)
static bool lc3_env_flag(const char *name) {
    const char *v = std::getenv(name);
    return v != nullptr && std::atoi(v) != 0;
}

int lc3gpu_encoder_create_spec(lc3gpu_encoder **out, int num_channels, int frame_us, int fs_hz, int spec_flags) {
    if (!out || num_channels <= 0 || (spec_flags & ~LC3GPU_SPEC_ALL)) return LC3GPU_EINVAL;
    *out = nullptr;
    lc3_cfg c;
    int rc = make_config(c, frame_us, fs_hz);
    if (rc) return rc;
    // the reference cannot construct an 8 kHz encoder (encoder/bandwidth_detector.rs:36-37 indexes [fs_ind - 1])
    if (c.fs_ind == 0 && !(spec_flags & LC3GPU_SPEC_8KHZ_ENCODE)) return LC3GPU_EUNSUPPORTED;
    if (lc3gpu_device_count() <= 0) return LC3GPU_ENODEVICE;
    lc3gpu_encoder *e = new (std::nothrow) lc3gpu_encoder();
    if (!e) return LC3GPU_EINVAL;
    e->num_channels = num_channels;
    e->spec_flags = spec_flags | (lc3_env_flag("LC3GPU_SEQ_SUMS") ? LC3_SPEC_TEST_SEQ_SUMS : 0);
    rc = cfg_acquire(e->h, frame_us, fs_hz);
    if (rc == LC3GPU_OK) rc = encoder_alloc(e);
    if (rc) { lc3gpu_encoder_destroy(e); return rc; }
    *out = e;
    return LC3GPU_OK;
}

int lc3gpu_encoder_create_mixed(lc3gpu_encoder **out, int n_streams, const lc3gpu_stream_desc *descs) {
    return lc3gpu_encoder_create_mixed_spec(out, n_streams, descs, 0);
}

int lc3gpu_encoder_create_mixed_spec(lc3gpu_encoder **out, int n_streams, const lc3gpu_stream_desc *descs, int spec_flags) {
    if (!out || n_streams <= 0 || !descs || (spec_flags & ~LC3GPU_SPEC_ALL)) return LC3GPU_EINVAL;
    *out = nullptr;
    if (lc3gpu_device_count() <= 0) return LC3GPU_ENODEVICE;
    lc3gpu_encoder *e = new (std::nothrow) lc3gpu_encoder();
    if (!e) return LC3GPU_EINVAL;
    e->spec_flags = spec_flags | (lc3_env_flag("LC3GPU_SEQ_SUMS") ? LC3_SPEC_TEST_SEQ_SUMS : 0);
    int rc = build_mixed(*e, n_streams, descs, !(spec_flags & LC3GPU_SPEC_8KHZ_ENCODE), 20);
    if (rc == LC3GPU_OK) rc = encoder_alloc(e);
    if (rc) { lc3gpu_encoder_destroy(e); return rc; }
    *out = e;
    return LC3GPU_OK;
}

int lc3gpu_encoder_destroy(lc3gpu_encoder *e) {
    if (!e) return LC3GPU_OK;
    {
        DeviceGuard g(e->device);
        (void)e->quiesce();
        if (e->d_states) (void)hipFree(e->d_states);
        if (e->d_pcm1) (void)hipHostFree(e->d_pcm1);
        if (e->d_out1) (void)hipHostFree(e->d_out1);
        if (e->d_dbg) (void)hipFree(e->d_dbg);
        if (e->d_planes) (void)hipFree(e->d_planes);
        if (e->d_mid) (void)hipFree(e->d_mid);
        e->release_common();
    }
    delete e;
    return LC3GPU_OK;
}

int lc3gpu_encoder_reset(lc3gpu_encoder *e) {
    if (!e) return LC3GPU_EINVAL;
    // Nothing to wait for: a launch in flight stores its streams' state at its end, and the next launch of a fresh channel never reads
    // that -- it starts from the constructed state inside its kernel and stores what IT leaves, behind the earlier store in stream order
    // (launches on another stream are ordered by order_begin's event; a partial launch materialises the other channels with launches of
    // its own).  A reset between two batch calls therefore costs no synchronisation ("a new encoder per frame", SURVEY 8d Mode A).
    e->fresh_mask.assign((size_t)e->num_channels, 1);
    return LC3GPU_OK;
}

// materialises the state of still-fresh channels [first, first + n) (internal order) with zero-frame launches of the init path
static int encoder_materialise(lc3gpu_encoder *e, int first, int n, hipStream_t stream) {
    lc3_io io = {0, nullptr};
    for (int i = first; i < first + n; i++) {
        if (!e->fresh_mask[(size_t)i]) continue;
        const HostCfg &h = e->mixed ? e->groups[(size_t)e->streams[(size_t)e->caller_of_internal[(size_t)i]].group].h : e->h;
        LC3_LAUNCH_CFG(lc3_enc_front_kernel, h, dim3(1), dim3(64 * LC3_WG_WAVES), 0, stream, e->d_states, i, 1,
                       (const int16_t *)e->d_pcm1, e->d_mid, e->d_planes, 20, 0, 1, (float *)nullptr, io, e->spec_flags);
        HIP_TRY(hipGetLastError());
        e->fresh_mask[(size_t)i] = 0;
    }
    return LC3GPU_OK;
}

// A batch call of a uniform handle as two halves of its streams on the handle's two internal HIP streams, forked from and joined to
// the caller's stream by events (OPT-IN: LC3GPU_SPLIT=1, every launch of at least 16 streams).  The idea (round-3 review): a launch of
// 65 536 frames gives every SIMD exactly ONE wave of the lane-per-frame kernels (vector quantiser, packer, parser), a lone wave leaves
// part of the SIMD's issue slots idle, and as two halves the lane-per-frame kernels of one half could share the chip with the
// wave-per-stream kernels of the other.  Built, byte-identical to the unsplit form in every test -- and measured slower on the
// 65 536-frame batch, 32.4 M frames/s against 43.3 M (gpurun_out/r04_exp_arr.txt -> profiles/r04_split_experiment.txt): a lane-per-frame
// kernel takes as long for half the frames (it is the latency of one wave walking its 64 frames), so the two halves' quantiser, packer
// and parser add up where they do not run at the same moment, and every call pays a fork and a join across HIP streams (~35 us per
// call on this platform, tools/exp_hops.py).  Inside ONE call there is nothing to put beside the last packer / the parsers (DESIGN
// section 6); what does pay is the caller running the encoder handle and the decoder handle on two streams (bench.py --arrangement
// pipelined).  -> number of parts (1 or 2).
static int lc3_split_parts(size_t frames, int n_streams) {
    static const int forced = [] {
        const char *e = std::getenv("LC3GPU_SPLIT");
        return e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
    }();
    (void)frames;
    if (n_streams < 4 * LC3_WG_WAVES) return 1;
    return forced == 1 ? 2 : 1;
}
// first stream of the second half: whole workgroups of the stream kernels, whole waves of the frame kernels where the launch allows
static int lc3_split_point(int n_streams, int n_frames) {
    int a = n_streams / 2;
    const int unit = (a >= 64 && ((size_t)a * (size_t)n_frames) >= 4096) ? 64 : LC3_WG_WAVES;
    a = (a + unit - 1) / unit * unit;
    return a < n_streams ? a : n_streams / 2 / LC3_WG_WAVES * LC3_WG_WAVES;
}
// where the second half's first kernel is held back to (LC3GPU_SPLIT_STAGGER, tuning aid): 0 nothing, 1 (default) until the first
// half's first kernel has finished -- both halves' wave-per-stream kernels side by side would only share the chip between them
static int lc3_split_stagger() {
    static const int v = [] {
        const char *e = std::getenv("LC3GPU_SPLIT_STAGGER");
        return e ? std::atoi(e) : 1;
    }();
    return v;
}

// the four encoder kernels of channels [first, first + n) (internal order) on `stream`; the buffers and planes are those of this range.
// chain: the timer's chain of this stream.  after_front: recorded behind the front half when not null
static int encode_kernels(lc3gpu_encoder *e, const HostCfg &h, int first, int n, const int16_t *d_pcm, uint8_t *d_out, float *mid,
                          int32_t *planes, int nbytes, int n_frames, int fresh, lc3_io io, hipStream_t stream, int chain,
                          hipEvent_t after_front, float *dbg, size_t frames_of_call) {
    const size_t frames = (size_t)n * (size_t)n_frames;
    // analysis front half (wave per stream) -> SNS vector quantiser (lane per frame) -> back half (wave per stream) ->
    // bitstream packing (lane per frame)
    const dim3 wg_grid((unsigned)((n + LC3_WG_WAVES - 1) / LC3_WG_WAVES)), wg_block(64 * LC3_WG_WAVES);
    int rc_stage = 0;
    e->timer.mark(stream, -1, chain);
    LC3_LAUNCH_CFG(lc3_enc_front_kernel, h, wg_grid, wg_block, lc3_lds_pad(0), stream, e->d_states, first, n, d_pcm, mid, planes, nbytes, n_frames, fresh,
                   dbg, io, e->spec_flags);
    HIP_TRY(hipGetLastError());
    e->timer.mark(stream, 0, chain);
    if (after_front) HIP_TRY(hipEventRecord(after_front, stream));
    if (chain == 0 && (rc_stage = e->stage_record(LC3GPU_ENC_STAGE_FRONT, stream)) != 0) return rc_stage;
    hipLaunchKernelGGL(lc3_sns_vq_kernel, dim3((unsigned)((frames + 255) / 256)), dim3(256), 0, stream, h.c.nb, mid, planes, (int)frames,
                       e->spec_flags);
    HIP_TRY(hipGetLastError());
    e->timer.mark(stream, 1, chain);
    if (chain == 0 && (rc_stage = e->stage_record(LC3GPU_ENC_STAGE_VQ, stream)) != 0) return rc_stage;
    LC3_LAUNCH_CFG(lc3_enc_back_kernel, h, wg_grid, wg_block, lc3_lds_pad(1), stream, e->d_states, first, n, (const float *)mid, planes, nbytes, n_frames,
                   dbg, e->spec_flags | lc3_prep_symbols_flag(frames_of_call));
    HIP_TRY(hipGetLastError());
    e->timer.mark(stream, 2, chain);
    if (chain == 0 && (rc_stage = e->stage_record(LC3GPU_ENC_STAGE_BACK, stream)) != 0) return rc_stage;
    if (lc3_prep_symbols_mode(frames_of_call) == 2) {  // (timed together with the packer)
        const size_t wgs = (frames + LC3_WG_WAVES - 1) / LC3_WG_WAVES;
        LC3_LAUNCH_CFG(lc3_symbols_kernel, h, dim3((unsigned)(wgs < lc3_recon_grid() ? wgs : lc3_recon_grid())), dim3(64 * LC3_WG_WAVES), 0, stream,
                       planes, (int)frames);
        HIP_TRY(hipGetLastError());
    }
    if (lc3_prep_symbols_mode(frames_of_call) == 0 && lc3_pack_pc_enabled()) {
        if (!e->pc_optin_done) {  // (once per handle: the opt-in itself takes a process-wide lock)
            int rc = lc3_pack_pc_optin();
            if (rc) return rc;
            e->pc_optin_done = true;
        }
        const unsigned pfpb = lc3_pack_pc_fpb(nbytes);
        hipLaunchKernelGGL(lc3_pack_pc_kernel, dim3((unsigned)((frames + pfpb - 1) / pfpb)), dim3(2 * pfpb), lc3_pack_pc_lds(pfpb, nbytes), stream,
                           h.c.ne, (const int32_t *)planes, d_out, nbytes, (int)frames, n_frames, io, e->d_pc_timeouts);
    } else {
        const unsigned fpb = lc3_frame_block_fit(LC3_PACK_LDS_FIXED, (size_t)nbytes);
        const size_t lds = LC3_PACK_LDS_FIXED + (((size_t)fpb * (size_t)nbytes + 3) & ~(size_t)3) + 4;  // + the packer's sink byte
        hipLaunchKernelGGL(lc3_pack_kernel, dim3((unsigned)((frames + fpb - 1) / fpb)), dim3(fpb), lds, stream, h.c.ne, (const int32_t *)planes,
                           d_out, nbytes, (int)frames, n_frames, io);
    }
    HIP_TRY(hipGetLastError());
    e->timer.mark(stream, 3, chain);
    return LC3GPU_OK;
}

// one configuration, channels [first, first + n) in internal order; the buffers hold only those channels
static int encode_launch(lc3gpu_encoder *e, const HostCfg &h, int first, int n, const int16_t *d_pcm, uint8_t *d_out, int nbytes,
                         int n_frames, int layout, hipStream_t stream, float *dbg) {
    if (!e || !d_pcm || !d_out) return LC3GPU_EINVAL;
    if (first < 0 || n <= 0 || first + n > e->num_channels) return LC3GPU_ECHANNEL;
    if (nbytes < 20 || nbytes > LC3_MAX_NE || n_frames <= 0) return LC3GPU_ELENGTH;
    if (layout != LC3GPU_LAYOUT_PLANAR && layout != LC3GPU_LAYOUT_INTERLEAVED) return LC3GPU_EINVAL;
    if (layout == LC3GPU_LAYOUT_PLANAR ? ((uintptr_t)d_pcm & 3u) != 0 : ((uintptr_t)d_pcm & 1u) != 0) return LC3GPU_EINVAL;
    const size_t frames = (size_t)n * (size_t)n_frames;
    const int nf = h.c.nf;
    int parts = (dbg || e->in_host_call || e->is_bound) ? 1 : lc3_split_parts(frames, n);
    const int na = parts == 2 ? lc3_split_point(n, n_frames) : n;
    // planar PCM is read as 32-bit words: the second half starts na * n_frames * nf samples in (nf is even), its bytes are copied out as
    // words when aligned and as bytes otherwise
    if (parts == 2 && (na <= 0 || na >= n)) parts = 1;
    int rc = e->order_begin(stream);
    if (rc == LC3GPU_OK) rc = encoder_reserve_planes(e, frames, stream);
    if (rc == LC3GPU_OK && parts == 2) rc = e->ensure_split();
    if (rc) return rc;
    // a range is launched "fresh" only if every channel in it is still fresh
    int fresh = 1;
    for (int i = first; i < first + n; i++) fresh &= e->fresh_mask[(size_t)i];
    if (!fresh) {
        rc = encoder_materialise(e, first, n, stream);
        if (rc) return rc;
    }
    lc3_io io = {layout == LC3GPU_LAYOUT_INTERLEAVED ? n : 0, nullptr};
    const size_t t0 = e->timer.used;
    e->timer.arm();
    if (parts == 1) {
        rc = encode_kernels(e, h, first, n, d_pcm, d_out, e->d_mid, e->d_planes, nbytes, n_frames, fresh, io, stream, 0, nullptr, dbg, frames);
    } else {
        // two halves [first, first + na) and [first + na, first + n) on the handle's streams: fork behind everything the caller's stream
        // holds so far, join back into it
        rc = LC3GPU_OK;
        if (hipEventRecord(e->ev_fork, stream) != hipSuccess || hipStreamWaitEvent(e->sub[0], e->ev_fork, 0) != hipSuccess ||
            hipStreamWaitEvent(e->sub[1], e->ev_fork, 0) != hipSuccess) {
            g_last_hip = (int)hipGetLastError();
            rc = LC3GPU_EHIP;
        }
        const size_t fa = (size_t)na * (size_t)n_frames;
        const bool ilv = layout == LC3GPU_LAYOUT_INTERLEAVED;
        const int stagger = lc3_split_stagger();
        if (rc == LC3GPU_OK)
            rc = encode_kernels(e, h, first, na, d_pcm, d_out, e->d_mid, e->d_planes, nbytes, n_frames, fresh, io, e->sub[0], 1,
                                stagger == 1 ? e->ev_stage : nullptr, nullptr, frames);
        if (rc == LC3GPU_OK && stagger == 1 && hipStreamWaitEvent(e->sub[1], e->ev_stage, 0) != hipSuccess) {
            g_last_hip = (int)hipGetLastError();
            rc = LC3GPU_EHIP;
        }
        if (rc == LC3GPU_OK)
            rc = encode_kernels(e, h, first + na, n - na, ilv ? d_pcm + na : d_pcm + fa * (size_t)nf, ilv ? d_out + (size_t)na * (size_t)nbytes : d_out + fa * (size_t)nbytes,
                                e->d_mid + fa * (size_t)MP_WORDS, e->d_planes + fa * (size_t)EP_WORDS, nbytes, n_frames, fresh, io, e->sub[1], 2,
                                nullptr, nullptr, frames);
        // whatever was queued, the caller's stream (and the handle's next call) orders behind it
        for (int i = 0; i < 2; i++)
            if (hipEventRecord(e->ev_join[i], e->sub[i]) != hipSuccess || hipStreamWaitEvent(stream, e->ev_join[i], 0) != hipSuccess) {
                g_last_hip = (int)hipGetLastError();
                if (rc == LC3GPU_OK) rc = LC3GPU_EHIP;
            }
        if (rc == LC3GPU_OK) rc = e->stage_record_all(stream);
    }
    if (rc) {
        e->timer.rollback(t0);
        (void)e->order_end(stream, parts == 2);
        return rc;
    }
    for (int i = first; i < first + n; i++) e->fresh_mask[(size_t)i] = 0;
    return e->order_end(stream, parts == 2);
}

int lc3gpu_encode_layout(lc3gpu_encoder *e, int layout, const int16_t *d_pcm, uint8_t *d_out, int nbytes, int n_frames, void *stream) {
    if (!e || e->mixed) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(e);
    return encode_launch(e, e->h, 0, e->num_channels, d_pcm, d_out, nbytes, n_frames, layout, (hipStream_t)stream, nullptr);
}

int lc3gpu_encode(lc3gpu_encoder *e, const int16_t *d_pcm, uint8_t *d_out, int nbytes, int n_frames, void *stream) {
    return lc3gpu_encode_layout(e, LC3GPU_LAYOUT_PLANAR, d_pcm, d_out, nbytes, n_frames, stream);
}

int lc3gpu_encode_range(lc3gpu_encoder *e, int first_channel, int n_channels, const int16_t *d_pcm, uint8_t *d_out,
                        int nbytes, int n_frames, void *stream) {
    if (!e || e->mixed) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(e);
    return encode_launch(e, e->h, first_channel, n_channels, d_pcm, d_out, nbytes, n_frames, LC3GPU_LAYOUT_PLANAR, (hipStream_t)stream,
                         nullptr);
}

// every stream of a mixed-configuration handle, ONE launch per kernel
int lc3gpu_encode_mixed(lc3gpu_encoder *e, const int16_t *d_pcm, uint8_t *d_out, int n_frames, void *stream_) {
    if (!e || !e->mixed || !d_pcm || !d_out) return LC3GPU_EINVAL;
    if (n_frames <= 0) return LC3GPU_ELENGTH;
    if (((uintptr_t)d_pcm & 3u) != 0) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(e);
    hipStream_t stream = (hipStream_t)stream_;
    int rc = e->order_begin(stream);
    if (rc) return rc;
    lc3_groups G;
    unsigned wg_stream, wg_frame;
    size_t frames;
    int max_nbytes;
    fill_groups(*e, n_frames, 256u, G, wg_stream, wg_frame, frames, max_nbytes);
    const unsigned fpb = lc3_frame_block_fit(LC3_PACK_LDS_FIXED, (size_t)max_nbytes);
    fill_groups(*e, n_frames, fpb, G, wg_stream, wg_frame, frames, max_nbytes);
    rc = encoder_reserve_planes(e, frames, stream);
    if (rc) return rc;
    int fresh = 1;
    for (uint8_t m : e->fresh_mask) fresh &= m;
    if (!fresh) {
        rc = encoder_materialise(e, 0, e->num_channels, stream);
        if (rc) return rc;
    }
    lc3_io io = {0, e->d_tab};
    const size_t t0 = e->timer.used;
    e->timer.begin(stream);
    hipLaunchKernelGGL(LC3_MIXED_LAUNCH(lc3_enc_front_mixed_kernel), dim3(wg_stream), dim3(64 * LC3_WG_WAVES), 0, stream, G, e->d_states, d_pcm, e->d_mid,
                       e->d_planes, n_frames, fresh, io, e->spec_flags);
    LC3_LAUNCH_CHECK(e, stream, t0);
    e->timer.mark(stream, 0);
    LC3_STAGE_RECORD(e, LC3GPU_ENC_STAGE_FRONT, stream, t0);
    lc3_groups G256;  // the vector quantiser runs 256 frames per workgroup
    {
        unsigned a, b;
        size_t f;
        int m;
        fill_groups(*e, n_frames, 256u, G256, a, b, f, m);
        hipLaunchKernelGGL(lc3_sns_vq_mixed_kernel, dim3(b), dim3(256), 0, stream, G256, e->d_mid, e->d_planes, n_frames, e->spec_flags);
        LC3_LAUNCH_CHECK(e, stream, t0);
    }
    e->timer.mark(stream, 1);
    LC3_STAGE_RECORD(e, LC3GPU_ENC_STAGE_VQ, stream, t0);
    hipLaunchKernelGGL(LC3_MIXED_LAUNCH(lc3_enc_back_mixed_kernel), dim3(wg_stream), dim3(64 * LC3_WG_WAVES), 0, stream, G, e->d_states,
                       (const float *)e->d_mid, e->d_planes, n_frames, e->spec_flags | lc3_prep_symbols_flag((size_t)e->num_channels * (size_t)n_frames, true));
    LC3_LAUNCH_CHECK(e, stream, t0);
    e->timer.mark(stream, 2);
    LC3_STAGE_RECORD(e, LC3GPU_ENC_STAGE_BACK, stream, t0);
    if (lc3_prep_symbols_mode((size_t)e->num_channels * (size_t)n_frames) == 0 && lc3_pack_pc_enabled()) {
        // full batches: the producer / consumer pairs (as lc3gpu_encode does), a group table of their own number of frames per workgroup
        if ((rc = lc3_pack_pc_optin()) != LC3GPU_OK) {
            e->timer.rollback(t0);
            (void)e->order_end(stream);
            return rc;
        }
        const unsigned pfpb = lc3_pack_pc_fpb(max_nbytes);
        lc3_groups Gp;
        unsigned a, b;
        size_t f;
        int m;
        fill_groups(*e, n_frames, pfpb, Gp, a, b, f, m);
        hipLaunchKernelGGL(lc3_pack_pc_mixed_kernel, dim3(b), dim3(2 * pfpb), lc3_pack_pc_lds(pfpb, max_nbytes), stream, Gp, (const int32_t *)e->d_planes,
                           d_out, n_frames, io, e->d_pc_timeouts);
    } else {
        const size_t lds = LC3_PACK_LDS_FIXED + (((size_t)fpb * (size_t)max_nbytes + 3) & ~(size_t)3) + 4;
        hipLaunchKernelGGL(lc3_pack_mixed_kernel, dim3(wg_frame), dim3(fpb), lds, stream, G, (const int32_t *)e->d_planes, d_out, n_frames, io);
    }
    LC3_LAUNCH_CHECK(e, stream, t0);
    e->timer.mark(stream, 3);
    e->fresh_mask.assign((size_t)e->num_channels, 0);
    return e->order_end(stream);
}

static int encode_frame_host(lc3gpu_encoder *e, int channel_index, const int16_t *samples_in, int n_samples,
                             uint8_t *buf_out, int nbytes, float *dbg) {
    if (!e || !samples_in || !buf_out) return LC3GPU_EINVAL;
    if (channel_index < 0 || channel_index >= e->num_channels) return LC3GPU_ECHANNEL;
    const HostCfg &h = e->cfg_of_channel(channel_index);
    if (n_samples != h.c.nf) return LC3GPU_ELENGTH;
    if (nbytes < 20 || nbytes > LC3_MAX_NE) return LC3GPU_ELENGTH;
    LC3_ON_DEVICE(e);
    int rc = e->quiesce();  // the staging buffers are reused
    if (rc) return rc;
    std::memcpy(e->d_pcm1, samples_in, sizeof(int16_t) * (size_t)n_samples);
    rc = encode_launch(e, h, e->internal_of_channel(channel_index), 1, e->d_pcm1, e->d_out1, nbytes, 1, LC3GPU_LAYOUT_PLANAR, nullptr,
                       dbg ? e->d_dbg : nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    std::memcpy(buf_out, e->d_out1, (size_t)nbytes);
    if (dbg) HIP_TRY(hipMemcpy(dbg, e->d_dbg, sizeof(float) * LC3_ENC_DBG_FLOATS, hipMemcpyDeviceToHost));
    return LC3GPU_OK;
}

int lc3gpu_encode_frame(lc3gpu_encoder *e, int channel_index, const int16_t *samples_in, int n_samples,
                        uint8_t *buf_out, int nbytes) {
    return encode_frame_host(e, channel_index, samples_in, n_samples, buf_out, nbytes, nullptr);
}

int lc3gpu_encode_frame_debug(lc3gpu_encoder *e, const int16_t *samples_in, int n_samples, uint8_t *buf_out, int nbytes,
                              float *dbg) {
    if (!dbg) return LC3GPU_EINVAL;
    return encode_frame_host(e, 0, samples_in, n_samples, buf_out, nbytes, dbg);
}

size_t lc3gpu_encoder_state_size(const lc3gpu_encoder *e) { return e ? sizeof(lc3_state_header) + sizeof(lc3_enc_state) : 0; }

int lc3gpu_encoder_state_save(lc3gpu_encoder *e, void *host_dst, size_t nbytes) {
    if (!e || !host_dst) return LC3GPU_EINVAL;
    if (nbytes != lc3gpu_encoder_state_size(e) * (size_t)e->num_channels) return LC3GPU_ELENGTH;
    LC3_ON_DEVICE(e);
    HIP_TRY(hipDeviceSynchronize());  // (a launch in flight stores its state at its end: the initialising launches below must come after it -- a reset does not wait)
    int rc = encoder_materialise(e, 0, e->num_channels, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return state_blobs_save(*e, e->d_states, LC3_STATE_MAGIC_ENC, e->spec_flags & LC3GPU_SPEC_ALL, host_dst, nbytes);
}

int lc3gpu_encoder_state_load(lc3gpu_encoder *e, const void *host_src, size_t nbytes) {
    if (!e || !host_src) return LC3GPU_EINVAL;
    if (nbytes != lc3gpu_encoder_state_size(e) * (size_t)e->num_channels) return LC3GPU_ELENGTH;
    LC3_ON_DEVICE(e);
    HIP_TRY(hipDeviceSynchronize());  // a launch in flight would store its state over the loaded one
    int rc = state_blobs_load(*e, e->d_states, LC3_STATE_MAGIC_ENC, e->spec_flags & LC3GPU_SPEC_ALL, host_src, nbytes);
    if (rc) return rc;
    e->fresh_mask.assign((size_t)e->num_channels, 0);
    return LC3GPU_OK;
}

// ---------------------------------------------------------------------------------------------
static int decoder_init_states(lc3gpu_decoder *d) {
    // decoder state is materialised eagerly (fresh = 1, zero frames): PLC counters must survive range launches
    HIP_TRY(hipDeviceSynchronize());  // nothing of this handle may still be in flight (its state is about to be rewritten)
    d->fresh_pending = false;
    lc3_io io = {0, nullptr};
    if (!d->mixed) {
        LC3_LAUNCH_CFG(lc3_decode_kernel, d->h, dim3((unsigned)((d->num_channels + LC3_WG_WAVES - 1) / LC3_WG_WAVES)),
                       dim3(64 * LC3_WG_WAVES), 0, nullptr, d->d_states, 0, d->num_channels, (const int32_t *)d->d_planes, d->d_pcm1, 20,
                       0, 1, io);
        HIP_TRY(hipGetLastError());
    } else {
        for (const GroupHost &g : d->groups) {
            LC3_LAUNCH_CFG(lc3_decode_kernel, g.h, dim3((unsigned)((g.n_streams + LC3_WG_WAVES - 1) / LC3_WG_WAVES)),
                           dim3(64 * LC3_WG_WAVES), 0, nullptr, d->d_states, g.first_stream, g.n_streams, (const int32_t *)d->d_planes,
                           d->d_pcm1, 20, 0, 1, io);
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    return LC3GPU_OK;
}

static int decoder_alloc(lc3gpu_decoder *d) {
    HIP_TRY(hipGetDevice(&d->device));
    HIP_TRY(hipEventCreateWithFlags(&d->done, hipEventDisableTiming));
    { const int rc = d->pc_health_alloc(); if (rc) return rc; }
    HIP_TRY(hipMalloc((void **)&d->d_states, sizeof(lc3_dec_state) * (size_t)d->num_channels));
    // A fresh launch stores the part of the post-filter's output ring its configuration uses (lc3_dec_state_store; 960 of the array's 1 080
    // floats at 48 kHz / 10 ms) and nothing ever writes the rest: zeroed once here, it stays zero -- state_save must not copy whatever
    // hipMalloc handed out to the host, and two handles that decoded the same frames must save the same blobs
    HIP_TRY(hipMemset(d->d_states, 0, sizeof(lc3_dec_state) * (size_t)d->num_channels));
    HIP_TRY(hipHostMalloc((void **)&d->d_in1, LC3_MAX_NE, hipHostMallocDefault));  // *_frame staging: pinned host memory, used in place
    HIP_TRY(hipHostMalloc((void **)&d->d_pcm1, sizeof(int16_t) * LC3_MAX_NF, hipHostMallocDefault));
    int rc = decoder_reserve_planes(d, (size_t)d->num_channels, nullptr);
    if (rc == LC3GPU_OK) rc = decoder_init_states(d);
    return rc;
}

int lc3gpu_decoder_create(lc3gpu_decoder **out, int num_channels, int frame_us, int fs_hz) {
    if (!out || num_channels <= 0) return LC3GPU_EINVAL;
    *out = nullptr;
    lc3_cfg c;
    int rc = make_config(c, frame_us, fs_hz);
    if (rc) return rc;
    if (lc3gpu_device_count() <= 0) return LC3GPU_ENODEVICE;
    lc3gpu_decoder *d = new (std::nothrow) lc3gpu_decoder();
    if (!d) return LC3GPU_EINVAL;
    d->num_channels = num_channels;
    rc = cfg_acquire(d->h, frame_us, fs_hz);
    if (rc == LC3GPU_OK) rc = decoder_alloc(d);
    if (rc) { lc3gpu_decoder_destroy(d); return rc; }
    *out = d;
    return LC3GPU_OK;
}

int lc3gpu_decoder_create_mixed(lc3gpu_decoder **out, int n_streams, const lc3gpu_stream_desc *descs) {
    if (!out || n_streams <= 0 || !descs) return LC3GPU_EINVAL;
    *out = nullptr;
    if (lc3gpu_device_count() <= 0) return LC3GPU_ENODEVICE;
    lc3gpu_decoder *d = new (std::nothrow) lc3gpu_decoder();
    if (!d) return LC3GPU_EINVAL;
    int rc = build_mixed(*d, n_streams, descs, false, 1);
    if (rc == LC3GPU_OK) rc = decoder_alloc(d);
    if (rc) { lc3gpu_decoder_destroy(d); return rc; }
    *out = d;
    return LC3GPU_OK;
}

int lc3gpu_decoder_destroy(lc3gpu_decoder *d) {
    if (!d) return LC3GPU_OK;
    {
        DeviceGuard g(d->device);
        (void)d->quiesce();
        if (d->d_states) (void)hipFree(d->d_states);
        if (d->d_in1) (void)hipHostFree(d->d_in1);
        if (d->d_pcm1) (void)hipHostFree(d->d_pcm1);
        if (d->d_planes) (void)hipFree(d->d_planes);
        if (d->d_dbg) (void)hipFree(d->d_dbg);
        d->release_common();
    }
    delete d;
    return LC3GPU_OK;
}

int lc3gpu_decoder_reset(lc3gpu_decoder *d) {
    if (!d) return LC3GPU_EINVAL;
    d->fresh_pending = true;  // (no wait, for the encoder's reasons: lc3gpu_encoder_reset; whoever reads the blobs first synchronises -- decoder_materialise)
    return LC3GPU_OK;
}
// the state blobs as a reader expects them: a reset that is still only noted is carried out
static int decoder_materialise(lc3gpu_decoder *d) { return d->fresh_pending ? decoder_init_states(d) : LC3GPU_OK; }

// the decoder kernels of channels [first, first + n) (internal order) on `stream`; buffers, flags and planes are those of this range
static int decode_kernels(lc3gpu_decoder *d, const HostCfg &h, int first, int n, const uint8_t *d_in, const uint8_t *d_bad, int16_t *d_pcm,
                          int32_t *planes, int nbytes, int n_frames, lc3_io io, int mode, hipStream_t stream, int chain, int fresh = 0) {
    // stage 1: parse all n * n_frames frames, one lane each (stateless); stage 2: synthesis, one wave per stream
    const size_t frames = (size_t)n * (size_t)n_frames;
    // frames per workgroup: as many as fit the default 64 KB of dynamic LDS (tables + 64 B of scale factors and nbytes of
    // frame data per frame)
    const unsigned fpb = lc3_frame_block_fit(LC3_PARSE_LDS_FIXED, (size_t)(64 + nbytes));
    const size_t lds = LC3_PARSE_LDS_FIXED + (size_t)fpb * (size_t)(64 + nbytes);
    d->timer.mark(stream, -1, chain);
    if (mode == LC3_RECON_LANE && lc3_parse_pc_enabled()) {
        if (!d->pc_optin_done) {  // (once per handle: the opt-in itself takes a process-wide lock)
            int rc = lc3_parse_pc_optin();
            if (rc) return rc;
            d->pc_optin_done = true;
        }
        const unsigned pfpb = lc3_parse_pc_fpb(nbytes);
        LC3_LAUNCH_CFG(lc3_parse_pc_kernel, h, dim3((unsigned)((frames + pfpb - 1) / pfpb)), dim3(2 * pfpb), lc3_parse_pc_lds(pfpb, nbytes), stream,
                       d_in, d_bad, planes, nbytes, (int)frames, n_frames, io, d->d_pc_timeouts);
    } else
        LC3_LAUNCH_CFG(lc3_parse_kernel, h, dim3((unsigned)((frames + fpb - 1) / fpb)), dim3(fpb), lds, stream, d_in, d_bad, planes, nbytes,
                       (int)frames, n_frames, io, mode);
    HIP_TRY(hipGetLastError());
    d->timer.mark(stream, 0, chain);
    if (mode == LC3_RECON_WAVE) {
        const size_t wgs = (frames + LC3_WG_WAVES - 1) / LC3_WG_WAVES;
        LC3_LAUNCH_CFG(lc3_recon_kernel, h, dim3((unsigned)(wgs < lc3_recon_grid() ? wgs : lc3_recon_grid())), dim3(64 * LC3_WG_WAVES), 0, stream,
                       planes, nbytes, (int)frames);
        HIP_TRY(hipGetLastError());
        d->timer.mark(stream, 1, chain);
        LC3_LAUNCH_CFG(lc3_tns_kernel, h, dim3((unsigned)((frames + LC3_TNS_FPB - 1) / LC3_TNS_FPB)), dim3(LC3_TNS_FPB), LC3_TNS_LDS, stream,
                       planes, (int)frames);
        HIP_TRY(hipGetLastError());
        d->timer.mark(stream, 2, chain);
    }
    if (chain == 0) {
        const int rc_stage = d->stage_record(LC3GPU_DEC_STAGE_PARSE, stream);
        if (rc_stage) return rc_stage;
    }
    if (mode == LC3_RECON_LATE)
        LC3_LAUNCH_CFG(lc3_decode_late_kernel, h, dim3((unsigned)((n + LC3_WG_WAVES - 1) / LC3_WG_WAVES)), dim3(64 * LC3_WG_WAVES), 0,
                       stream, d->d_states, first, n, (const int32_t *)planes, d_pcm, nbytes, n_frames, fresh, io);
    else
        LC3_LAUNCH_CFG(lc3_decode_kernel, h, dim3((unsigned)((n + LC3_WG_WAVES - 1) / LC3_WG_WAVES)), dim3(64 * LC3_WG_WAVES), lc3_lds_pad(2), stream,
                       d->d_states, first, n, (const int32_t *)planes, d_pcm, nbytes, n_frames, fresh, io);
    HIP_TRY(hipGetLastError());
    d->timer.mark(stream, 3, chain);
    return LC3GPU_OK;
}

static int decode_launch(lc3gpu_decoder *d, const HostCfg &h, int first, int n, const uint8_t *d_in, const uint8_t *d_bad,
                         int16_t *d_pcm, int nbytes, int n_frames, int layout, hipStream_t stream) {
    if (!d || !d_in || !d_pcm) return LC3GPU_EINVAL;
    if (first < 0 || n <= 0 || first + n > d->num_channels) return LC3GPU_ECHANNEL;
    if (nbytes < 1 || nbytes > LC3_MAX_NE || n_frames <= 0) return LC3GPU_ELENGTH;
    if (layout != LC3GPU_LAYOUT_PLANAR && layout != LC3GPU_LAYOUT_INTERLEAVED) return LC3GPU_EINVAL;
    if (layout == LC3GPU_LAYOUT_PLANAR ? ((uintptr_t)d_pcm & 3u) != 0 : ((uintptr_t)d_pcm & 1u) != 0) return LC3GPU_EINVAL;
    const size_t frames = (size_t)n * (size_t)n_frames;
    const int nf = h.c.nf;
    // a reset that is only noted so far: a launch over ALL channels carries it out itself (its synthesis kernel starts from the constructed
    // state and stores what it leaves); any other launch needs the other channels' blobs initialised first
    int fresh = 0;
    if (d->fresh_pending) {
        if (first == 0 && n == d->num_channels && !d->mixed) fresh = 1;
        else {
            const int rc0 = decoder_materialise(d);
            if (rc0) return rc0;
        }
    }
    int parts = (d->in_host_call || d->is_bound) ? 1 : lc3_split_parts(frames, n);
    const int na = parts == 2 ? lc3_split_point(n, n_frames) : n;
    if (parts == 2 && (na <= 0 || na >= n)) parts = 1;
    int rc = d->order_begin(stream);
    if (rc == LC3GPU_OK) rc = decoder_reserve_planes(d, frames, stream);
    if (rc == LC3GPU_OK && parts == 2) rc = d->ensure_split();
    if (rc) return rc;
    lc3_io io = {layout == LC3GPU_LAYOUT_INTERLEAVED ? n : 0, nullptr};
    const int mode = lc3_recon_mode(frames, n_frames);
    if (mode == LC3_RECON_WAVE && (rc = lc3_tns_lds_optin()) != LC3GPU_OK) return rc;
    const size_t t0 = d->timer.used;
    d->timer.arm();
    if (parts == 1) {
        rc = decode_kernels(d, h, first, n, d_in, d_bad, d_pcm, d->d_planes, nbytes, n_frames, io, mode, stream, 0, fresh);
    } else {
        rc = LC3GPU_OK;
        if (hipEventRecord(d->ev_fork, stream) != hipSuccess || hipStreamWaitEvent(d->sub[0], d->ev_fork, 0) != hipSuccess ||
            hipStreamWaitEvent(d->sub[1], d->ev_fork, 0) != hipSuccess) {
            g_last_hip = (int)hipGetLastError();
            rc = LC3GPU_EHIP;
        }
        const size_t fa = (size_t)na * (size_t)n_frames;
        const bool ilv = layout == LC3GPU_LAYOUT_INTERLEAVED;
        if (rc == LC3GPU_OK)
            rc = decode_kernels(d, h, first, na, d_in, d_bad, d_pcm, d->d_planes, nbytes, n_frames, io, mode, d->sub[0], 1, fresh);
        if (rc == LC3GPU_OK)
            rc = decode_kernels(d, h, first + na, n - na, ilv ? d_in + (size_t)na * (size_t)nbytes : d_in + fa * (size_t)nbytes,
                                d_bad ? (ilv ? d_bad + na : d_bad + fa) : nullptr, ilv ? d_pcm + na : d_pcm + fa * (size_t)nf,
                                d->d_planes + fa * (size_t)LC3_PLANE_WORDS, nbytes, n_frames, io, mode, d->sub[1], 2, fresh);
        for (int i = 0; i < 2; i++)
            if (hipEventRecord(d->ev_join[i], d->sub[i]) != hipSuccess || hipStreamWaitEvent(stream, d->ev_join[i], 0) != hipSuccess) {
                g_last_hip = (int)hipGetLastError();
                if (rc == LC3GPU_OK) rc = LC3GPU_EHIP;
            }
        if (rc == LC3GPU_OK) rc = d->stage_record_all(stream);
    }
    if (rc) {
        d->timer.rollback(t0);
        (void)d->order_end(stream, parts == 2);
        if (fresh) (void)decoder_init_states(d);  // (some of the kernels may have run: leave the handle in a defined state)
        return rc;
    }
    if (fresh) d->fresh_pending = false;
    return d->order_end(stream, parts == 2);
}

int lc3gpu_decode_layout(lc3gpu_decoder *d, int layout, const uint8_t *d_in, const uint8_t *d_bad, int16_t *d_pcm, int nbytes,
                         int n_frames, void *stream) {
    if (!d || d->mixed) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(d);
    return decode_launch(d, d->h, 0, d->num_channels, d_in, d_bad, d_pcm, nbytes, n_frames, layout, (hipStream_t)stream);
}

int lc3gpu_decode(lc3gpu_decoder *d, const uint8_t *d_in, const uint8_t *d_bad, int16_t *d_pcm, int nbytes, int n_frames,
                  void *stream) {
    return lc3gpu_decode_layout(d, LC3GPU_LAYOUT_PLANAR, d_in, d_bad, d_pcm, nbytes, n_frames, stream);
}

int lc3gpu_decode_range(lc3gpu_decoder *d, int first_channel, int n_channels, const uint8_t *d_in, const uint8_t *d_bad,
                        int16_t *d_pcm, int nbytes, int n_frames, void *stream) {
    if (!d || d->mixed) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(d);
    return decode_launch(d, d->h, first_channel, n_channels, d_in, d_bad, d_pcm, nbytes, n_frames, LC3GPU_LAYOUT_PLANAR,
                         (hipStream_t)stream);
}

int lc3gpu_decode_mixed(lc3gpu_decoder *d, const uint8_t *d_in, const uint8_t *d_bad, int16_t *d_pcm, int n_frames, void *stream_) {
    if (!d || !d->mixed || !d_in || !d_pcm) return LC3GPU_EINVAL;
    if (n_frames <= 0) return LC3GPU_ELENGTH;
    if (((uintptr_t)d_pcm & 3u) != 0) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(d);
    hipStream_t stream = (hipStream_t)stream_;
    int rc = decoder_materialise(d);
    if (rc == LC3GPU_OK) rc = d->order_begin(stream);
    if (rc) return rc;
    lc3_groups G;
    unsigned wg_stream, wg_frame;
    size_t frames;
    int max_nbytes;
    fill_groups(*d, n_frames, 256u, G, wg_stream, wg_frame, frames, max_nbytes);
    const unsigned fpb = lc3_frame_block_fit(LC3_PARSE_LDS_FIXED, (size_t)(64 + max_nbytes));
    fill_groups(*d, n_frames, fpb, G, wg_stream, wg_frame, frames, max_nbytes);
    rc = decoder_reserve_planes(d, frames, stream);
    if (rc) return rc;
    lc3_io io = {0, d->d_tab};
    const size_t lds = LC3_PARSE_LDS_FIXED + (size_t)fpb * (size_t)(64 + max_nbytes);
    const int mode = lc3_recon_mode((size_t)d->num_channels * (size_t)n_frames, n_frames);
    if (mode == LC3_RECON_WAVE && (rc = lc3_tns_lds_optin()) != LC3GPU_OK) return rc;
    const size_t t0 = d->timer.used;
    d->timer.begin(stream);
    if (mode == LC3_RECON_LANE && lc3_parse_pc_enabled()) {  // full batches: the producer / consumer pairs (as lc3gpu_decode does)
        if ((rc = lc3_parse_pc_optin()) != LC3GPU_OK) {
            d->timer.rollback(t0);
            (void)d->order_end(stream);
            return rc;
        }
        const unsigned pfpb = lc3_parse_pc_fpb(max_nbytes);
        lc3_groups Gp;
        unsigned a, b;
        size_t f;
        int m;
        fill_groups(*d, n_frames, pfpb, Gp, a, b, f, m);
        hipLaunchKernelGGL(LC3_MIXED_LAUNCH(lc3_parse_pc_mixed_kernel), dim3(b), dim3(2 * pfpb), lc3_parse_pc_lds(pfpb, max_nbytes), stream, Gp, d_in, d_bad, d->d_planes,
                           n_frames, io, d->d_pc_timeouts);
    } else
        hipLaunchKernelGGL(LC3_MIXED_LAUNCH(lc3_parse_mixed_kernel), dim3(wg_frame), dim3(fpb), lds, stream, G, d_in, d_bad, d->d_planes, n_frames, io, mode);
    LC3_LAUNCH_CHECK(d, stream, t0);
    d->timer.mark(stream, 0);
    lc3_groups Gx;  // the group table for other numbers of frames per workgroup
    unsigned gx_a, gx_b;
    size_t gx_f;
    int gx_m;
    if (mode == LC3_RECON_WAVE) {
        fill_groups(*d, n_frames, (unsigned)LC3_WG_WAVES, Gx, gx_a, gx_b, gx_f, gx_m);
        hipLaunchKernelGGL(LC3_MIXED_LAUNCH(lc3_recon_mixed_kernel), dim3(gx_b), dim3(64 * LC3_WG_WAVES), 0, stream, Gx, d->d_planes, n_frames);
        LC3_LAUNCH_CHECK(d, stream, t0);
        d->timer.mark(stream, 1);
    }
    if (mode == LC3_RECON_WAVE) {
        fill_groups(*d, n_frames, (unsigned)LC3_TNS_FPB, Gx, gx_a, gx_b, gx_f, gx_m);
        hipLaunchKernelGGL(LC3_MIXED_LAUNCH(lc3_tns_mixed_kernel), dim3(gx_b), dim3(LC3_TNS_FPB), LC3_TNS_LDS, stream, Gx, d->d_planes, n_frames);
        LC3_LAUNCH_CHECK(d, stream, t0);
        d->timer.mark(stream, 2);
    }
    LC3_STAGE_RECORD(d, LC3GPU_DEC_STAGE_PARSE, stream, t0);
    if (mode == LC3_RECON_LATE)
        hipLaunchKernelGGL(LC3_MIXED_LAUNCH(lc3_decode_mixed_late_kernel), dim3(wg_stream), dim3(64 * LC3_WG_WAVES), 0, stream, G, d->d_states,
                           (const int32_t *)d->d_planes, d_pcm, n_frames, 0, io);
    else
        hipLaunchKernelGGL(LC3_MIXED_LAUNCH(lc3_decode_mixed_kernel), dim3(wg_stream), dim3(64 * LC3_WG_WAVES), 0, stream, G, d->d_states,
                           (const int32_t *)d->d_planes, d_pcm, n_frames, 0, io);
    LC3_LAUNCH_CHECK(d, stream, t0);
    d->timer.mark(stream, 3);
    return d->order_end(stream);
}

int lc3gpu_decode_frame(lc3gpu_decoder *d, int num_bits_per_audio_sample, int channel_index, const uint8_t *buf_in,
                        int nbytes, int16_t *samples_out, int n_samples) {
    if (!d || !buf_in || !samples_out) return LC3GPU_EINVAL;
    if (num_bits_per_audio_sample != 16) return LC3GPU_EBITS;  // checked first, as in lc3_decoder.rs:80-82
    if (channel_index < 0 || channel_index >= d->num_channels) return LC3GPU_ECHANNEL;
    const HostCfg &h = d->cfg_of_channel(channel_index);
    if (n_samples != h.c.nf) return LC3GPU_ELENGTH;
    if (nbytes < 1 || nbytes > LC3_MAX_NE) return LC3GPU_ELENGTH;
    LC3_ON_DEVICE(d);
    int rc = d->quiesce();  // the staging buffers are reused
    if (rc) return rc;
    std::memcpy(d->d_in1, buf_in, (size_t)nbytes);
    rc = decode_launch(d, h, d->internal_of_channel(channel_index), 1, d->d_in1, nullptr, d->d_pcm1, nbytes, 1, LC3GPU_LAYOUT_PLANAR,
                       nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    std::memcpy(samples_out, d->d_pcm1, sizeof(int16_t) * (size_t)n_samples);
    return LC3GPU_OK;
}

// ---- diagnostic entry points: one frame of channel 0 of a uniform handle with stage dumps (LC3_DBG_*, lc3_dev_dec_parse.h) ----
static int decoder_debug_begin(lc3gpu_decoder *d, const float *fill) {
    if (!d->d_dbg) HIP_TRY(hipMalloc((void **)&d->d_dbg, sizeof(float) * LC3_DBG_FLOATS));
    std::vector<float> nanv((size_t)LC3_DBG_FLOATS, std::nanf(""));
    (void)fill;
    HIP_TRY(hipMemcpy(d->d_dbg, nanv.data(), sizeof(float) * LC3_DBG_FLOATS, hipMemcpyHostToDevice));
    return LC3GPU_OK;
}

int lc3gpu_decode_frame_debug(lc3gpu_decoder *d, int recon_form, const uint8_t *buf_in, int nbytes, int16_t *samples_out, int n_samples,
                              float *dbg) {
    if (!d || d->mixed || !buf_in || !samples_out || !dbg) return LC3GPU_EINVAL;
    if (recon_form != LC3_RECON_LANE && recon_form != LC3_RECON_LATE && recon_form != LC3_RECON_WAVE) return LC3GPU_EINVAL;
    const HostCfg &h = d->h;
    if (n_samples != h.c.nf) return LC3GPU_ELENGTH;
    if (nbytes < 1 || nbytes > LC3_MAX_NE) return LC3GPU_ELENGTH;
    LC3_ON_DEVICE(d);
    int rc = d->quiesce();
    if (rc == LC3GPU_OK) rc = decoder_materialise(d);
    if (rc == LC3GPU_OK) rc = decoder_reserve_planes(d, 1, nullptr);
    if (rc == LC3GPU_OK) rc = decoder_debug_begin(d, nullptr);
    if (rc == LC3GPU_OK && recon_form == LC3_RECON_WAVE) rc = lc3_tns_lds_optin();
    if (rc) return rc;
    std::memcpy(d->d_in1, buf_in, (size_t)nbytes);
    const unsigned fpb = lc3_frame_block_fit(LC3_PARSE_LDS_FIXED, (size_t)(64 + nbytes));
    const size_t lds = LC3_PARSE_LDS_FIXED + (size_t)fpb * (size_t)(64 + nbytes);
    std::vector<int32_t> col((size_t)LC3_PLANE_WORDS);
    // the integers: parsing is stateless, so the frame is first parsed in the form that leaves them in the plane
    LC3_LAUNCH_CFG(lc3_parse_debug_kernel, h, dim3(1), dim3(fpb), lds, nullptr, (const uint8_t *)d->d_in1, d->d_planes, nbytes, 1, (float *)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(col.data(), d->d_planes, sizeof(int32_t) * LC3_PLANE_WORDS, hipMemcpyDeviceToHost));
    const int parsed = col[AD_OK] != 0, lastnz = col[SI_LASTNZ];
    if (recon_form != LC3_RECON_LATE) {  // the form under test rebuilds the plane column
        LC3_LAUNCH_CFG(lc3_parse_debug_kernel, h, dim3(1), dim3(fpb), lds, nullptr, (const uint8_t *)d->d_in1, d->d_planes, nbytes, recon_form,
                       recon_form == LC3_RECON_LANE ? d->d_dbg : (float *)nullptr);
        HIP_TRY(hipGetLastError());
        if (recon_form == LC3_RECON_WAVE) {
            LC3_LAUNCH_CFG(lc3_recon_kernel, h, dim3(1), dim3(64 * LC3_WG_WAVES), 0, nullptr, d->d_planes, nbytes, 1);
            HIP_TRY(hipGetLastError());
            LC3_LAUNCH_CFG(lc3_tns_kernel, h, dim3(1), dim3(LC3_TNS_FPB), LC3_TNS_LDS, nullptr, d->d_planes, 1);
            HIP_TRY(hipGetLastError());
        }
    }
    LC3_LAUNCH_CFG(lc3_decode_debug_kernel, h, dim3(1), dim3(64 * LC3_WG_WAVES), 0, nullptr, d->d_states, 0, (const int32_t *)d->d_planes, d->d_pcm1,
                   nbytes, recon_form == LC3_RECON_LATE ? 1 : 0, d->d_dbg, (int)LC3_DBG_DUMP);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(nullptr));
    std::memcpy(samples_out, d->d_pcm1, sizeof(int16_t) * (size_t)n_samples);
    HIP_TRY(hipMemcpy(dbg, d->d_dbg, sizeof(float) * LC3_DBG_FLOATS, hipMemcpyDeviceToHost));
    if (parsed)  // (the late form dumps them itself; the others take them from the first parse)
        for (int k = 0; k < h.c.ne; k++) dbg[LC3_DBG_INT + k] = k < lastnz ? (float)col[LC3_PLANE_X + k] : 0.0f;
    return LC3GPU_OK;
}

int lc3gpu_decoder_synth_debug(lc3gpu_decoder *d, int time_in, const float *in, int n_in, int ltpf_active, int pitch_index, int nbytes,
                               int16_t *samples_out, int n_samples, float *dbg) {
    if (!d || d->mixed || !in || !samples_out || !dbg) return LC3GPU_EINVAL;
    const HostCfg &h = d->h;
    if (n_samples != h.c.nf || n_in != (time_in ? h.c.nf : h.c.ne)) return LC3GPU_ELENGTH;
    if (nbytes < 1 || nbytes > LC3_MAX_NE) return LC3GPU_ELENGTH;
    if (pitch_index < 0 || pitch_index > 511) return LC3GPU_EINVAL;  // nine bits in the bitstream (side_info_reader.rs:106-129): the post-filter derives lags and table rows from it
    LC3_ON_DEVICE(d);
    int rc = d->quiesce();
    if (rc == LC3GPU_OK) rc = decoder_materialise(d);
    if (rc == LC3GPU_OK) rc = decoder_reserve_planes(d, 1, nullptr);
    if (rc == LC3GPU_OK) rc = decoder_debug_begin(d, nullptr);
    if (rc) return rc;
    // a plane column that says: good frame, this long-term post-filter side information; the spectrum / the samples come from the buffer
    std::vector<int32_t> col((size_t)LC3_PLANE_WORDS, 0);
    col[AD_OK] = 1;
    col[SI_PITCH_PRESENT] = 1;
    col[SI_LTPF_ACTIVE] = ltpf_active != 0;
    col[SI_PITCH_INDEX] = pitch_index;
    HIP_TRY(hipMemcpy(d->d_planes, col.data(), sizeof(int32_t) * LC3_PLANE_WORDS, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->d_dbg + (time_in ? LC3_DBG_IMDCT : LC3_DBG_SPEC), in, sizeof(float) * (size_t)n_in, hipMemcpyHostToDevice));
    if (time_in) {  // the transform runs on silence
        std::vector<float> z((size_t)h.c.ne, 0.0f);
        HIP_TRY(hipMemcpy(d->d_dbg + LC3_DBG_SPEC, z.data(), sizeof(float) * z.size(), hipMemcpyHostToDevice));
    }
    LC3_LAUNCH_CFG(lc3_decode_debug_kernel, h, dim3(1), dim3(64 * LC3_WG_WAVES), 0, nullptr, d->d_states, 0, (const int32_t *)d->d_planes, d->d_pcm1,
                   nbytes, 0, d->d_dbg, (int)(LC3_DBG_SPEC_IN | LC3_DBG_DUMP | (time_in ? LC3_DBG_TIME_IN : 0)));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(nullptr));
    std::memcpy(samples_out, d->d_pcm1, sizeof(int16_t) * (size_t)n_samples);
    HIP_TRY(hipMemcpy(dbg, d->d_dbg, sizeof(float) * LC3_DBG_FLOATS, hipMemcpyDeviceToHost));
    return LC3GPU_OK;
}

size_t lc3gpu_decoder_state_size(const lc3gpu_decoder *d) { return d ? sizeof(lc3_state_header) + sizeof(lc3_dec_state) : 0; }

int lc3gpu_decoder_state_save(lc3gpu_decoder *d, void *host_dst, size_t nbytes) {
    if (!d || !host_dst) return LC3GPU_EINVAL;
    if (nbytes != lc3gpu_decoder_state_size(d) * (size_t)d->num_channels) return LC3GPU_ELENGTH;
    LC3_ON_DEVICE(d);
    HIP_TRY(hipDeviceSynchronize());
    { const int rc = decoder_materialise(d); if (rc) return rc; }
    return state_blobs_save(*d, d->d_states, LC3_STATE_MAGIC_DEC, 0, host_dst, nbytes);
}

int lc3gpu_decoder_state_load(lc3gpu_decoder *d, const void *host_src, size_t nbytes) {
    if (!d || !host_src) return LC3GPU_EINVAL;
    if (nbytes != lc3gpu_decoder_state_size(d) * (size_t)d->num_channels) return LC3GPU_ELENGTH;
    LC3_ON_DEVICE(d);
    HIP_TRY(hipDeviceSynchronize());  // a launch in flight would store its state over the loaded one
    const int rc = state_blobs_load(*d, d->d_states, LC3_STATE_MAGIC_DEC, 0, host_src, nbytes);
    if (rc == LC3GPU_OK) d->fresh_pending = false;  // every channel now has the loaded state
    return rc;
}

int lc3gpu_decoder_plc_events(lc3gpu_decoder *d, uint64_t *out) {
    if (!d || !out) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(d);
    HIP_TRY(hipDeviceSynchronize());
    { const int rc = decoder_materialise(d); if (rc) return rc; }
    std::vector<lc3_dec_state> st((size_t)d->num_channels);
    HIP_TRY(hipMemcpy(st.data(), d->d_states, sizeof(lc3_dec_state) * st.size(), hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (const auto &x : st) total += (uint64_t)x.core.plc_events;
    *out = total;
    return LC3GPU_OK;
}

// ---------------------------------------------------------------------------------------------
// Host-resident batches.  The reference's caller holds PCM and frame bytes in host memory and walks them frame by frame
// (examples/encode.rs:73-116, examples/decode.rs:60-112); lc3gpu_encode_host / lc3gpu_decode_host take such buffers whole: the handle's
// channels in ranges, each as  copy in -> the range's kernels -> copy out  on three internal HIP streams (one per step: host_lane_run), so
// that the copy-in of range c + 1, the kernels of range c and the copy-out of range c - 1 run at the same time (the kernels of the ranges
// themselves run one after the other: they share the handle's planes).
// Synchronous, like the loop they replace.  PCIe bounds them (a 48 kHz / 10 ms frame is 960 + 150 bytes each way); pinned buffers
// (lc3gpu_host_alloc) copy at the link's rate, pageable ones through the runtime's staging.
static int host_chunk_channels(int num_channels, int n_frames) {
    // Ranges of whole waves of the lane-per-frame kernels and at least 32 768 frames (LC3GPU_HOST_CHUNK_FRAMES overrides: tuning aid), at most
    // sixteen ranges.  Measured on the 65 536-frame batch, encode / decode in M frames/s: ranges of 8 192 frames 18.7 / 21.4, 16 384: 29.6 / 35.0,
    // 22 000: 32.2 / 41.4, 32 768: 34.4 / 41.4, one range (no overlap) 27.9 / 37.7 -- a lane-per-frame kernel takes as long for a quarter
    // of the frames as for all of them, so few large ranges beat many small ones; the link itself moves 56 GB/s each way
    static const long long target = [] {
        const char *e = std::getenv("LC3GPU_HOST_CHUNK_FRAMES");
        const long long v = e ? std::atoll(e) : 0;
        return v > 0 ? v : 32768ll;
    }();
    long long c = (target + n_frames - 1) / n_frames;
    const long long min_c = ((long long)num_channels + 15) / 16;
    if (c < min_c) c = min_c;
    c = (c + 63) / 64 * 64;
    return c >= num_channels ? num_channels : (int)c;
}
int lc3gpu_host_alloc(void **out, size_t nbytes) {
    if (!out || nbytes == 0) return LC3GPU_EINVAL;
    *out = nullptr;
    if (lc3gpu_device_count() <= 0) return LC3GPU_ENODEVICE;
    HIP_TRY(hipHostMalloc(out, nbytes, hipHostMallocDefault));
    return LC3GPU_OK;
}
int lc3gpu_host_free(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return LC3GPU_OK;
}
// The ranges of a host-resident call: range c uses staging buffers c & 1.  Three internal streams -- copies in, kernels, copies out -- so that
// the H2D engine, the compute queue and the D2H engine all run at once:
//   copy-in   waits for the kernels of range c - 2 (they read the buffer it refills), copies, records in_done
//   kernels   wait for in_done and for the copy-out of range c - 2 (it reads the buffer they refill), run, record k_done
//   copy-out  waits for k_done, copies, records out_done
// copy_in / launch / copy_out: what to queue for range (first, n) with buffers b on the given stream; -> LC3GPU_* code
extern "C++" {
template <class CopyIn, class Launch, class CopyOut>
static int host_lane_run(HandleCommon &hc, int C, int cc, CopyIn copy_in, Launch launch, CopyOut copy_out) {
    int rc = hc.ensure_host_lane();
    if (rc) return rc;
    bool k_rec[2] = {false, false}, out_rec[2] = {false, false};
    hc.in_host_call = true;
    int k = 0;
    for (int first = 0; first < C && rc == LC3GPU_OK; first += cc, k++) {
        const int n = first + cc <= C ? cc : C - first, b = k & 1;
        hipError_t e = hipSuccess;
        if (k_rec[b]) e = hipStreamWaitEvent(hc.host_s[0], hc.host_k_done[b], 0);
        if (e == hipSuccess) e = copy_in(first, n, b, hc.host_s[0]);
        if (e == hipSuccess) e = hipEventRecord(hc.host_in_done[b], hc.host_s[0]);
        if (e == hipSuccess) e = hipStreamWaitEvent(hc.host_s[1], hc.host_in_done[b], 0);
        if (e == hipSuccess && out_rec[b]) e = hipStreamWaitEvent(hc.host_s[1], hc.host_out_done[b], 0);
        if (e != hipSuccess) {
            g_last_hip = (int)e;
            (void)hipGetLastError();
            rc = LC3GPU_EHIP;
            break;
        }
        rc = launch(first, n, b, hc.host_s[1]);
        if (rc) break;
        e = hipEventRecord(hc.host_k_done[b], hc.host_s[1]);
        k_rec[b] = true;
        if (e == hipSuccess) e = hipStreamWaitEvent(hc.host_s[2], hc.host_k_done[b], 0);
        if (e == hipSuccess) e = copy_out(first, n, b, hc.host_s[2]);
        if (e == hipSuccess) e = hipEventRecord(hc.host_out_done[b], hc.host_s[2]);
        out_rec[b] = true;
        if (e != hipSuccess) {
            g_last_hip = (int)e;
            (void)hipGetLastError();
            rc = LC3GPU_EHIP;
        }
    }
    hc.in_host_call = false;
    for (int i = 0; i < 3; i++)
        if (hipStreamSynchronize(hc.host_s[i]) != hipSuccess && rc == LC3GPU_OK) {
            g_last_hip = (int)hipGetLastError();
            rc = LC3GPU_EHIP;
        }
    return rc;
}
}  // extern "C++"
int lc3gpu_encode_host(lc3gpu_encoder *e, const int16_t *pcm, uint8_t *out, int nbytes, int n_frames) {
    if (!e || e->mixed || !pcm || !out) return LC3GPU_EINVAL;
    if (nbytes < 20 || nbytes > LC3_MAX_NE || n_frames <= 0) return LC3GPU_ELENGTH;
    LC3_ON_DEVICE(e);
    int rc = e->quiesce();
    if (rc) return rc;
    const int nf = e->h.c.nf, C = e->num_channels, cc = host_chunk_channels(C, n_frames);
    const size_t in_row = (size_t)n_frames * (size_t)nf * sizeof(int16_t), out_row = (size_t)n_frames * (size_t)nbytes;
    rc = e->stage_reserve((size_t)cc * in_row, (size_t)cc * out_row, 0);
    if (rc) return rc;
    return host_lane_run(
        *e, C, cc,
        [&](int first, int n, int b, hipStream_t st) {
            return hipMemcpyAsync(e->d_stage_in[b], (const char *)pcm + (size_t)first * in_row, (size_t)n * in_row, hipMemcpyHostToDevice, st);
        },
        [&](int first, int n, int b, hipStream_t st) {
            return encode_launch(e, e->h, first, n, (const int16_t *)e->d_stage_in[b], (uint8_t *)e->d_stage_out[b], nbytes, n_frames,
                                 LC3GPU_LAYOUT_PLANAR, st, nullptr);
        },
        [&](int first, int n, int b, hipStream_t st) {
            return hipMemcpyAsync(out + (size_t)first * out_row, e->d_stage_out[b], (size_t)n * out_row, hipMemcpyDeviceToHost, st);
        });
}
int lc3gpu_decode_host(lc3gpu_decoder *d, const uint8_t *in, const uint8_t *bad_frame, int16_t *pcm, int nbytes, int n_frames) {
    if (!d || d->mixed || !in || !pcm) return LC3GPU_EINVAL;
    if (nbytes < 1 || nbytes > LC3_MAX_NE || n_frames <= 0) return LC3GPU_ELENGTH;
    LC3_ON_DEVICE(d);
    int rc = d->quiesce();
    if (rc) return rc;
    const int nf = d->h.c.nf, C = d->num_channels, cc = host_chunk_channels(C, n_frames);
    const size_t in_row = (size_t)n_frames * (size_t)nbytes, out_row = (size_t)n_frames * (size_t)nf * sizeof(int16_t);
    rc = d->stage_reserve((size_t)cc * in_row, (size_t)cc * out_row, bad_frame ? (size_t)cc * (size_t)n_frames : 0);
    if (rc) return rc;
    return host_lane_run(
        *d, C, cc,
        [&](int first, int n, int b, hipStream_t st) {
            hipError_t e2 = hipMemcpyAsync(d->d_stage_in[b], in + (size_t)first * in_row, (size_t)n * in_row, hipMemcpyHostToDevice, st);
            if (e2 == hipSuccess && bad_frame)
                e2 = hipMemcpyAsync(d->d_stage_flag[b], bad_frame + (size_t)first * (size_t)n_frames, (size_t)n * (size_t)n_frames, hipMemcpyHostToDevice, st);
            return e2;
        },
        [&](int first, int n, int b, hipStream_t st) {
            return decode_launch(d, d->h, first, n, (const uint8_t *)d->d_stage_in[b], bad_frame ? (const uint8_t *)d->d_stage_flag[b] : nullptr,
                                 (int16_t *)d->d_stage_out[b], nbytes, n_frames, LC3GPU_LAYOUT_PLANAR, st);
        },
        [&](int first, int n, int b, hipStream_t st) {
            return hipMemcpyAsync((char *)pcm + (size_t)first * out_row, d->d_stage_out[b], (size_t)n * out_row, hipMemcpyDeviceToHost, st);
        });
}

// lc3gpu_*_bind_stream (include/lc3gpu.h): see HandleCommon::bound_stream
static int bind_stream(HandleCommon &hc, void *hip_stream, int bind) {
    int rc = hc.quiesce();  // (whatever was launched under the other regime)
    if (rc) return rc;
    hc.is_bound = bind != 0;
    hc.bound_stream = bind ? (hipStream_t)hip_stream : nullptr;
    hc.has_work = false;
    return LC3GPU_OK;
}
int lc3gpu_encoder_bind_stream(lc3gpu_encoder *e, void *hip_stream, int bind) {
    if (!e) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(e);
    return bind_stream(*e, hip_stream, bind);
}
int lc3gpu_decoder_bind_stream(lc3gpu_decoder *d, void *hip_stream, int bind) {
    if (!d) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(d);
    return bind_stream(*d, hip_stream, bind);
}

// Producer / consumer pair kernels (full batches): how many pair halves ever gave up waiting for their partner (LC3_PC_SPIN_LIMIT
// polls -- a partner that died; never seen).  Sticky over the handle's life.  A parser pair that gives up conceals its frames (they
// also count as PLC events); a packer pair that gives up leaves its frames zero-filled.  Waits for the handle's work in flight.
static int pair_timeouts_read(HandleCommon &hc, uint64_t *out) {
    unsigned v = 0;
    int rc = hc.quiesce();
    if (rc) return rc;
    HIP_TRY(hipMemcpy(&v, hc.d_pc_timeouts, sizeof v, hipMemcpyDeviceToHost));
    *out = (uint64_t)v;
    return LC3GPU_OK;
}
// tests only: what a pair half that gives up does (lc3_pc_gave_up), from a one-lane kernel -- the path from the device to LC3GPU_EPAIR
// cannot be provoked otherwise (no pair has ever given up)
__global__ void lc3_pc_gave_up_kernel(unsigned *pc) { lc3_pc_gave_up(pc); }
static int pair_giveup_inject(HandleCommon &hc) {
    int rc = hc.quiesce();
    if (rc) return rc;
    hipLaunchKernelGGL(lc3_pc_gave_up_kernel, dim3(1), dim3(1), 0, nullptr, hc.d_pc_timeouts);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return LC3GPU_OK;
}
int lc3gpu_encoder_debug_pair_giveup(lc3gpu_encoder *e) {
    if (!e) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(e);
    return pair_giveup_inject(*e);
}
int lc3gpu_decoder_debug_pair_giveup(lc3gpu_decoder *d) {
    if (!d) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(d);
    return pair_giveup_inject(*d);
}
int lc3gpu_encoder_pair_timeouts(lc3gpu_encoder *e, uint64_t *out) {
    if (!e || !out) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(e);
    return pair_timeouts_read(*e, out);
}
int lc3gpu_decoder_pair_timeouts(lc3gpu_decoder *d, uint64_t *out) {
    if (!d || !out) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(d);
    return pair_timeouts_read(*d, out);
}

// per-kernel timing (HIP events on the launch stream).  enable = 1 starts recording every batch launch, reading
// synchronises and returns the per-kernel totals since the last read.
// encoder: out[5] = {front ms, vector-quantiser ms, back ms, pack ms, launches}
int lc3gpu_encoder_timing(lc3gpu_encoder *e, int enable, double out[5]) {
    if (!e) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(e);
    e->timer.collect();
    if (out) {
        for (int k = 0; k < 4; k++) out[k] = e->timer.ms[k];
        out[4] = (double)e->timer.launches;
    }
    for (int k = 0; k < 4; k++) e->timer.ms[k] = 0.0;
    e->timer.launches = 0;
    e->timer.set(enable);
    return LC3GPU_OK;
}
// decoder: out[3] = {parse + reconstruction ms, synthesis ms, launches}
static int decoder_timing_read(lc3gpu_decoder *d, int enable, double ms[4], double *launches) {
    if (!d) return LC3GPU_EINVAL;
    LC3_ON_DEVICE(d);
    d->timer.collect();
    for (int k = 0; k < 4; k++) {
        ms[k] = d->timer.ms[k];
        d->timer.ms[k] = 0.0;
    }
    *launches = (double)d->timer.launches;
    d->timer.launches = 0;
    d->timer.set(enable);
    return LC3GPU_OK;
}
int lc3gpu_decoder_timing(lc3gpu_decoder *d, int enable, double out[3]) {
    double ms[4], n;
    const int rc = decoder_timing_read(d, enable, ms, &n);
    if (rc) return rc;
    if (out) { out[0] = ms[0] + ms[1] + ms[2]; out[1] = ms[3]; out[2] = n; }
    return LC3GPU_OK;
}
// the same kernel by kernel: out[5] = {parse ms, reconstruction kernel ms, TNS kernel ms, synthesis ms, launches}
int lc3gpu_decoder_timing_kernels(lc3gpu_decoder *d, int enable, double out[5]) {
    double ms[4], n;
    const int rc = decoder_timing_read(d, enable, ms, &n);
    if (rc) return rc;
    if (out) { out[0] = ms[0]; out[1] = ms[1]; out[2] = ms[2]; out[3] = ms[3]; out[4] = n; }
    return LC3GPU_OK;
}

// Device arithmetic self-test (tests only): evaluates, ON THE DEVICE and as the kernels compile them, the float routines the codec's
// bit-exactness rests on.  which: 0 lc3_div_by(x, d) (the quantiser's division by a wave-uniform gain), 1 x / d (hipcc's IEEE division),
// 2 log2f, 3 log10f, 4 exp2f, 5 asinf, 6 exp2_raw, 7 10^x, 8 sinf on the TNS argument range, 9 the device-filled tables (out: 512
// gains 10^(k/28), 320 tilt factors, 17 encoder TNS sines, 17 decoder TNS sines = 866 values; x, d ignored).  Host pointers.
__global__ void lc3_math_test_kernel(int which, const float *x, const float *d, int n, float *out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float r = 0.0f;
        switch (which) {
        case 0: { const lc3_divisor v = lc3_divisor_make(d[i]); r = lc3_div_by(x[i], v); } break;
        case 1: r = x[i] / d[i]; break;
        case 2: r = lc3_log2f(x[i]); break;
        case 3: r = lc3_log10f(x[i]); break;
        case 4: r = lc3_exp2f(x[i]); break;
        case 5: r = lc3_asinf(x[i]); break;
        case 6: r = lc3_exp2_raw(x[i]); break;
        case 7: r = lc3_pow10f(x[i]); break;
        case 8: r = lc3_sinf_small(x[i]); break;
        default:
            if (i < 512) r = LC3_POW10_GG(i - 256);
            else if (i < 832) r = LC3_POW10_TILT((i - 512) / 64, (i - 512) % 64);
            else if (i < 849) r = LC3_TNS_SIN_ENC(i - 832);
            else r = LC3_TNS_SIN_DEC(i - 849);
        }
        out[i] = r;
    }
}
int lc3gpu_selftest_math(int which, const float *x, const float *d, int n, float *out) {
    if (which < 0 || which > 9 || n <= 0 || !out || (which < 9 && !x) || (which < 2 && !d)) return LC3GPU_EINVAL;
    if (which == 9 && n != 866) return LC3GPU_ELENGTH;
    if (lc3gpu_device_count() <= 0) return LC3GPU_ENODEVICE;
    HostCfg h;
    int rc = cfg_acquire(h, 10000, 48000);  // fills the device tables
    if (rc) return rc;
    float *dx = nullptr, *dd = nullptr, *dout = nullptr;
    const size_t bytes = sizeof(float) * (size_t)n;
    hipError_t e = hipMalloc((void **)&dx, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&dd, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&dout, bytes);
    if (e == hipSuccess && x) e = hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && d) e = hipMemcpy(dd, d, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lc3_math_test_kernel, dim3(1024), dim3(256), 0, nullptr, which, (const float *)dx, (const float *)dd, n, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(dx);
    (void)hipFree(dd);
    (void)hipFree(dout);
    if (e != hipSuccess) {
        g_last_hip = (int)e;
        return LC3GPU_EHIP;
    }
    return LC3GPU_OK;
}

// Measurement aid: the shader clock the chip is running at right now.  One wave stamps the shader-cycle counter (s_memtime) and the
// constant 100 MHz counter (s_memrealtime), spins through `spin` dependent vector additions and stamps both again; the clock is
// delta(s_memtime) / delta(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back, item 6).  Launched on a stream of its own
// beside the codec's kernels it reads the clock THEY run at (one wave of one SIMD: it takes nothing measurable from them).
__global__ void lc3_clock_probe_kernel(unsigned long long *out, int spin) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned v = threadIdx.x;
    for (int i = 0; i < spin; i++) {
        v += 0x9e3779b9u;
        asm volatile("" : "+v"(v));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
    if (v == 0x12345u && spin < 0) out[2] = v;  // (keeps the loop)
}
// asynchronous on `stream`: d_out (DEVICE, 2 x uint64) <- {shader cycles, 100 MHz ticks} over the probe's spin (~0.1 ms at spin = 50 000)
int lc3gpu_encoder_stage_event(lc3gpu_encoder *e, int stage, void *hip_event) {
    if (!e || stage > LC3GPU_ENC_STAGE_BACK) return LC3GPU_EINVAL;
    return e->stage_set(stage, hip_event);
}
int lc3gpu_decoder_stage_event(lc3gpu_decoder *d, int stage, void *hip_event) {
    if (!d || stage > LC3GPU_DEC_STAGE_PARSE) return LC3GPU_EINVAL;
    return d->stage_set(stage, hip_event);
}

int lc3gpu_clock_probe(void *stream, unsigned long long *d_out, int spin) {
    if (!d_out || spin <= 0) return LC3GPU_EINVAL;
    hipLaunchKernelGGL(lc3_clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_out, spin);
    HIP_TRY(hipGetLastError());
    return LC3GPU_OK;
}

// diagnostic build only: copy (and clear) the per-stage cycle accumulators; LC3GPU_EUNSUPPORTED otherwise
int lc3gpu_prof_read(unsigned long long out[64]) {
#ifdef LC3_PROFILE
    if (!out) return LC3GPU_EINVAL;
    unsigned long long zero[64] = {0};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(lc3_prof_acc), sizeof(zero)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(lc3_prof_acc), zero, sizeof(zero)));
    return LC3GPU_OK;
#else
    (void)out;
    return LC3GPU_EUNSUPPORTED;
#endif
}

int lc3gpu_kernel_info(int which, int out[5]) {
    if (!out) return LC3GPU_EINVAL;
    hipFuncAttributes a;
    // the six kernels a full batch of the headline configuration launches (48 kHz / 10 ms view; the pair forms of packer and parser).
    // 0 and 1 keep the meaning they had when there were two kernels (analysis back half, synthesis); the others are appended.
    const void *fn[6] = {(const void *)lc3_enc_back_kernel<lc3_cfg_48k10>, (const void *)lc3_decode_kernel<lc3_cfg_48k10>,
                         (const void *)lc3_enc_front_kernel<lc3_cfg_48k10>, (const void *)lc3_sns_vq_kernel,
                         (const void *)lc3_pack_pc_kernel, (const void *)lc3_parse_pc_kernel<lc3_cfg_48k10>};
    if (which < 0 || which >= 6) return LC3GPU_EINVAL;
    hipError_t e = hipFuncGetAttributes(&a, fn[which]);
    if (e != hipSuccess) {
        g_last_hip = (int)e;
        return LC3GPU_EHIP;
    }
    out[0] = (int)a.sharedSizeBytes;
    out[1] = a.numRegs;
    out[2] = 0;
    out[3] = (int)a.localSizeBytes;
    out[4] = a.maxThreadsPerBlock;
    return LC3GPU_OK;
}

}  // extern "C"
#endif  // LC3_IN_HOST_TU
