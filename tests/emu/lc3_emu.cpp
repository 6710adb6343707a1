// TEST INFRASTRUCTURE -- CPU wave emulator for the HIP device code.
//
// Compiles the product's device headers (lc3-codec_amd/csrc/lc3_dev_*.h) UNCHANGED with g++ and runs
// one "wavefront" as 64 host threads that meet at a pthread barrier wherever the kernel has a
// workgroup barrier.  It exists so that the lane-parallel decomposition (indexing, barrier
// placement, LDS aliasing) can be checked against the oracle in this GPU-less container before
// GPU minutes are spent; it is never shipped or timed.  Build: tests/emu_lib.py.
#include <pthread.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define __device__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))

// One emulated workgroup = LC3_WG_WAVES wavefronts of 64 host threads, one stream per wave -- the same shape as the
// HIP kernels (lc3gpu.hip).  LC3_SYNC is the wave-level barrier; the serial phases (LC3_SERIAL_BEGIN/END) meet at the
// workgroup barrier and run on the phase's leader wave with `L` rebound per stream, exactly as on the GPU, so a block
// that wrongly reads a per-stream value from a register (instead of LDS) reads the wrong stream's value here too.
#define LC3_WG_WAVES 4
static pthread_barrier_t g_wave_bar[LC3_WG_WAVES], g_wg_bar;
static thread_local int tl_wave = 0;
#define LC3_SYNC() pthread_barrier_wait(&g_wave_bar[tl_wave])
#define LC3_SERIAL_BEGIN(T, L, lane, phase, K)                                   \
    {                                                                            \
        T *lc3_wg_base_ = &(L) - tl_wave;                                        \
        pthread_barrier_wait(&g_wg_bar);                                         \
        if (tl_wave == ((phase) % LC3_WG_WAVES) && (lane) < LC3_WG_WAVES * (K)) { \
            T &L = lc3_wg_base_[(lane) / (K)];                                   \
            const int sub = (lane) % (K);                                        \
            (void)sub;
#define LC3_SERIAL_END                   \
        }                                \
        pthread_barrier_wait(&g_wg_bar); \
    }
// (K x streams beyond one wave: virtual lane v = lane + 64 j on the j-th wave after wave `phase`, as in lc3gpu.hip)
#define LC3_SERIAL_WIDE_BEGIN(T, L, lane, phase, K)                                                               \
    {                                                                                                             \
        T *lc3_wg_base_ = &(L) - tl_wave;                                                                         \
        pthread_barrier_wait(&g_wg_bar);                                                                          \
        const int lc3_v_ = (lane) + 64 * ((tl_wave + LC3_WG_WAVES - ((phase) % LC3_WG_WAVES)) % LC3_WG_WAVES);    \
        if (lc3_v_ < LC3_WG_WAVES * (K)) {                                                                        \
            T &L = lc3_wg_base_[lc3_v_ / (K)];                                                                    \
            const int sub = lc3_v_ % (K);                                                                         \
            (void)sub;
// every decision the device code takes from a guarded tree sum is checked against the sequential sum here
#define LC3_GUARD_SELFCHECK 1
#define LC3_GUARD_ASSERT(cond)                                                                         \
    do {                                                                                               \
        if (!(cond)) {                                                                                 \
            fprintf(stderr, "lc3_emu: guarded decision differs from the sequential sum (%s:%d)\n", __FILE__, __LINE__); \
            abort();                                                                                   \
        }                                                                                              \
    } while (0)
#include "../../lc3-codec_amd/csrc/lc3_dev_common.h"
// wave-level primitives: the GPU uses DPP / v_readlane; the emulator exchanges through memory
static int g_xi[LC3_WG_WAVES][64];
static inline int lc3_wave_max_i32(int v, int lane) {
    LC3_SYNC();
    g_xi[tl_wave][lane] = v;
    LC3_SYNC();
    int m = g_xi[tl_wave][0];
    for (int i = 1; i < 64; i++) m = g_xi[tl_wave][i] > m ? g_xi[tl_wave][i] : m;
    LC3_SYNC();
    return m;
}
static inline uint32_t lc3_wave_sum_u32(uint32_t v, int lane) {
    LC3_SYNC();
    g_xi[tl_wave][lane] = (int)v;
    LC3_SYNC();
    uint32_t m = 0;
    for (int i = 0; i < 64; i++) m += (uint32_t)g_xi[tl_wave][i];
    LC3_SYNC();
    return m;
}
static float g_xf[LC3_WG_WAVES][64];
static inline float lc3_wave_sum_f32_any(float v, int lane) {  // order unspecified on the GPU; here: a different tree on purpose
    LC3_SYNC();
    g_xf[tl_wave][lane] = v;
    LC3_SYNC();
    float m = 0.0f;
    for (int i = 63; i >= 0; i--) m += g_xf[tl_wave][i];
    LC3_SYNC();
    return m;
}
static inline float lc3_wave_shr1_f32(float v, int lane) {
    LC3_SYNC();
    g_xf[tl_wave][lane] = v;
    LC3_SYNC();
    const float r = lane > 0 ? g_xf[tl_wave][lane - 1] : v;
    LC3_SYNC();
    return r;
}
static inline int lc3_wave_shr1_i32(int v, int lane) {
    LC3_SYNC();
    g_xi[tl_wave][lane] = v;
    LC3_SYNC();
    const int r = lane > 0 ? g_xi[tl_wave][lane - 1] : 0;
    LC3_SYNC();
    return r;
}
static inline int lc3_wave_shl1_i32(int v, int lane) {
    LC3_SYNC();
    g_xi[tl_wave][lane] = v;
    LC3_SYNC();
    const int r = lane < 63 ? g_xi[tl_wave][lane + 1] : 0;
    LC3_SYNC();
    return r;
}
static inline float lc3_wave_read_f32(float v, int src, int lane) {
    LC3_SYNC();
    g_xf[tl_wave][lane] = v;
    LC3_SYNC();
    const float r = g_xf[tl_wave][src & 63];
    LC3_SYNC();
    return r;
}
static inline int lc3_wave_read_i32(int v, int src, int lane) {
    LC3_SYNC();
    g_xi[tl_wave][lane] = v;
    LC3_SYNC();
    const int r = g_xi[tl_wave][src & 63];
    LC3_SYNC();
    return r;
}
static inline int lc3_wave_bcast0_i32(int v, int lane) { return lc3_wave_read_i32(v, 0, lane); }
static inline float lc3_wave_bcast0_f32(float v, int lane) { return lc3_wave_read_f32(v, 0, lane); }
static inline unsigned long long lc3_wave_ballot(int pred, int lane) {
    LC3_SYNC();
    g_xi[tl_wave][lane] = pred != 0;
    LC3_SYNC();
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m |= (unsigned long long)(g_xi[tl_wave][i] != 0) << i;
    LC3_SYNC();
    return m;
}
static inline uint32_t lc3_wave_exscan_u32(uint32_t v, int lane) {
    LC3_SYNC();
    g_xi[tl_wave][lane] = (int)v;
    LC3_SYNC();
    uint32_t m = 0;
    for (int i = 0; i < lane; i++) m += (uint32_t)g_xi[tl_wave][i];
    LC3_SYNC();
    return m;
}

#include "../../lc3-codec_amd/csrc/lc3_dev_dec.h"
#include "../../lc3-codec_amd/csrc/lc3_dev_dec_recon.h"
#include "../../lc3-codec_amd/csrc/lc3_dev_enc.h"
#include "../../lc3-codec_amd/csrc/lc3_host_plan.h"

namespace {
struct Job {
    lc3_cfg cfg;
    int lane, wave, valid;
    int encode;   // 0 decode, 1 encoder front half, 2 encoder back half
    float *mid;   // mid planes of the whole batch (encoder)
    int n_frames, nbytes, fresh, spec_flags, late;
    lc3_enc_lds *EL;  // the workgroup's array of working sets
    lc3_dec_lds *DL;
    lc3_recon_tables *RT;  // wave-per-frame reconstruction: the workgroup's tables and the waves' scratch
    lc3_recon_wave *RW;
    lc3_enc_state *est;   // this stream's state blob
    lc3_dec_state *dst;
    const int16_t *pcm_in;
    int32_t *enc_planes;  // packer planes of the whole batch: frame t of this stream at column (frame0 + t)
    const uint8_t *bytes_in;
    const int32_t *planes;  // parsed frames of the whole batch
    size_t frame0;
    int16_t *pcm_out;
    float *dbg;
};

// body of lc3_encode_kernel / lc3_decode_kernel (lc3gpu.hip) for one lane
void *lane_main(void *arg) {
    Job *j = (Job *)arg;
    const int lane = j->lane;
    tl_wave = j->wave;
    if (j->encode == 1) {  // body of lc3_enc_front_kernel
        lc3_enc_lds &L = j->EL[j->wave];
        if (lane == 0) L.spec_flags = j->spec_flags;
        if (j->fresh) lc3_enc_state_init(L, lane, j->est, j->valid);
        else lc3_enc_state_load(L, lane, j->est);
        for (int t = 0; t < j->n_frames; t++) {
            const size_t f = j->frame0 + (size_t)t;
            int32_t *plane = j->valid ? LC3_PLANE_COL(j->enc_planes, f, EP_WORDS) : nullptr;
            float *mcol = j->valid ? j->mid + f * (size_t)MP_WORDS : nullptr;
            const int16_t *frame = j->pcm_in + (size_t)t * j->cfg.nf;
            const int16_t *hist = t > 0 ? frame - j->cfg.nf + j->cfg.z : (j->fresh ? nullptr : j->est->hist);
            lc3_encode_front_wave(j->cfg, L, lane, frame, hist, j->est, mcol, plane, LC3_PLANE_STRIDE, j->nbytes,
                                  j->valid ? j->dbg : nullptr, 1, 1, (t % LC3_WG_WAVES) + (t + 2 < j->n_frames ? 0x100 : 0));  // (as lc3_enc_front_body)
        }
        if (j->valid)
            lc3_enc_state_store(j->cfg, L, lane, j->est, j->n_frames > 0 ? j->pcm_in + (size_t)(j->n_frames - 1) * j->cfg.nf : nullptr);
    } else if (j->encode == 2) {  // body of lc3_enc_back_kernel
        lc3_enc_lds &L = j->EL[j->wave];
        if (lane == 0) L.spec_flags = j->spec_flags;
        lc3_enc_state_load(L, lane, j->est);
        lc3_encode_back_stream(j->cfg, L, lane, j->mid, j->enc_planes, j->frame0, j->n_frames, j->nbytes, j->valid, j->valid ? j->dbg : nullptr);
        if (j->valid) lc3_enc_state_store(j->cfg, L, lane, j->est, nullptr);
    } else if (j->encode == 4) {  // body of lc3_symbols_kernel: one wave per frame
        if (j->valid) lc3_enc_symbols_frame(j->cfg.ne, lane, LC3_PLANE_COL(j->enc_planes, j->frame0, EP_WORDS));
    } else if (j->encode == 3) {  // body of lc3_recon_kernel: one wave per frame
        if (j->valid)
            lc3_recon_frame_direct(j->cfg, *j->RT, j->RW[j->wave], lane, LC3_PLANE_COL((int32_t *)j->planes, j->frame0, LC3_PLANE_WORDS), j->nbytes);
    } else {
        lc3_dec_lds &L = j->DL[j->wave];
        if (j->fresh) lc3_dec_state_init(L, lane, j->dst, j->valid);
        else lc3_dec_state_load(L, lane, j->dst);
        lc3_decode_stream_wave(j->cfg, L, lane, j->nbytes, j->planes, j->frame0, j->n_frames, j->dst, j->valid, j->pcm_out, (size_t)j->cfg.nf, 1, j->late);
        if (j->valid) lc3_dec_state_store(j->cfg, L, lane, j->dst);
    }
    return 0;
}

// runs one workgroup: wave w works on protos[w]
void run_wg(const Job *protos) {
    static pthread_t th[LC3_WG_WAVES * LC3_WAVE];
    static Job jobs[LC3_WG_WAVES * LC3_WAVE];
    for (int w = 0; w < LC3_WG_WAVES; w++) pthread_barrier_init(&g_wave_bar[w], 0, LC3_WAVE);
    pthread_barrier_init(&g_wg_bar, 0, LC3_WG_WAVES * LC3_WAVE);
    for (int w = 0; w < LC3_WG_WAVES; w++)
        for (int i = 0; i < LC3_WAVE; i++) {
            Job &q = jobs[w * LC3_WAVE + i];
            q = protos[w];
            q.lane = i;
            q.wave = w;
            pthread_create(&th[w * LC3_WAVE + i], 0, lane_main, &q);
        }
    for (int i = 0; i < LC3_WG_WAVES * LC3_WAVE; i++) pthread_join(th[i], 0);
    for (int w = 0; w < LC3_WG_WAVES; w++) pthread_barrier_destroy(&g_wave_bar[w]);
    pthread_barrier_destroy(&g_wg_bar);
}
}  // namespace

extern "C" {
// pcm int16[S][T][nf] -> bytes uint8[S][T][nbytes]; every stream starts fresh; dbg optional float[LC3_ENC_DBG_FLOATS = 1600] (last frame)
int lc3emu_encode_spec(int fs_hz, int frame_us, int nbytes, int S, int T, const int16_t *pcm, uint8_t *bytes, float *dbg, int spec_flags);
int lc3emu_encode(int fs_hz, int frame_us, int nbytes, int S, int T, const int16_t *pcm, uint8_t *bytes, float *dbg) {
    return lc3emu_encode_spec(fs_hz, frame_us, nbytes, S, T, pcm, bytes, dbg, 0);
}
int lc3emu_encode_spec(int fs_hz, int frame_us, int nbytes, int S, int T, const int16_t *pcm, uint8_t *bytes, float *dbg, int spec_flags) {
    Job j;
    memset(&j, 0, sizeof(j));
    const int symbols_stage = (spec_flags & 1024) != 0;  // emulator only: the packer's symbols from lc3_symbols_kernel (LC3GPU_PREP_SYMBOLS=2)
    const int pack_pc = (spec_flags & 2048) != 0;        // emulator only: the packer's producer / consumer form (full batches on the GPU)
    spec_flags &= ~(1024 | 2048);
    j.spec_flags = spec_flags;
    lc3_host_plan pl;
    if (lc3_make_config(j.cfg, frame_us, fs_hz) || lc3_make_plan(j.cfg, pl)) return -1;
    j.cfg.fft_tw = pl.fft_tw.data();
    j.cfg.dct_tw = pl.dct_tw.data();
    j.cfg.perm = pl.perm.data();
    std::vector<float> poly((size_t)j.cfg.p_up * (size_t)j.cfg.resamp_stride);
    for (size_t i = 0; i < poly.size(); i++)
        poly[i] = lc3_resamp_poly_value(j.cfg.p_up, j.cfg.resamp_lim, j.cfg.resamp_stride, (int)i);
    j.cfg.resamp_poly = poly.data();
    std::vector<float> lw((size_t)j.cfg.ne);
    for (int k = 0; k < j.cfg.ne; k++) lw[(size_t)k] = lc3_line_width_value(j.cfg, k);
    j.cfg.line_width = lw.data();
    std::vector<uint8_t> lb((size_t)j.cfg.nf + 16);
    for (int k = 0; k < j.cfg.nf; k++) lb[(size_t)k] = (uint8_t)lc3_line_band_value(j.cfg, k);
    j.cfg.line_band = lb.data();
    j.encode = 1;
    j.n_frames = T;
    j.nbytes = nbytes;
    j.fresh = 1;
    j.dbg = dbg;
    // stage 1: analysis, one emulated wave per stream, leaves the packer planes
    const size_t frames = (size_t)S * (size_t)T;
    std::vector<int32_t> planes(((frames + 63) / 64) * 64 * EP_WORDS, 0);
    lc3_enc_lds *L = (lc3_enc_lds *)aligned_alloc(16, LC3_WG_WAVES * sizeof(lc3_enc_lds));
    lc3_enc_state *st = (lc3_enc_state *)aligned_alloc(16, (size_t)S * sizeof(lc3_enc_state));  // one blob per stream
    float *mid = (float *)aligned_alloc(16, frames * (size_t)MP_WORDS * sizeof(float));
    memset(st, 0, (size_t)S * sizeof(lc3_enc_state));
    memset(mid, 0, frames * (size_t)MP_WORDS * sizeof(float));
    j.EL = L;
    j.enc_planes = planes.data();
    j.mid = mid;
    for (int phase = 1; phase <= 2; phase++) {
        // phase 1: front halves of every workgroup; then the lane-per-frame vector quantiser; phase 2: back halves
        for (int s0 = 0; s0 < S; s0 += LC3_WG_WAVES) {
            Job protos[LC3_WG_WAVES];
            memset(L, 0xFF, LC3_WG_WAVES * sizeof(lc3_enc_lds));  // LDS is not zeroed on the GPU: NaN floats / -1 integers expose reads of stale words
            for (int w = 0; w < LC3_WG_WAVES; w++) {
                // waves past the end of the batch shadow the last stream and store nothing (as in the kernels)
                const int valid = s0 + w < S, s = valid ? s0 + w : S - 1;
                protos[w] = j;
                protos[w].encode = phase;
                protos[w].valid = valid;
                protos[w].est = st + s;
                protos[w].pcm_in = pcm + (size_t)s * T * j.cfg.nf;
                protos[w].frame0 = (size_t)s * T;
            }
            run_wg(protos);
        }
        if (phase == 1) {
            for (size_t f = 0; f < frames; f++) {  // lc3_sns_vq_kernel
                lc3_vq_ctx v;
                v.mid = mid + f * (size_t)MP_WORDS;
                v.gains = mid + f * (size_t)MP_WORDS + MP_G;
                v.plane = LC3_PLANE_COL(planes.data(), f, EP_WORDS);
                v.stride = LC3_PLANE_STRIDE;
                v.mpvq = &LC3T_MPVQ_OFFSETS[0][0];
                v.nb = j.cfg.nb;
                v.spec_flags = spec_flags;
                lc3_sns_vq_frame(v);
            }
        }
    }
    if (symbols_stage) {
        for (size_t f0 = 0; f0 < frames; f0 += LC3_WG_WAVES) {
            Job protos[LC3_WG_WAVES];
            for (int w = 0; w < LC3_WG_WAVES; w++) {
                protos[w] = j;
                protos[w].encode = 4;
                protos[w].valid = f0 + (size_t)w < frames;
                protos[w].frame0 = protos[w].valid ? f0 + (size_t)w : frames - 1;
            }
            run_wg(protos);
        }
    }
    free(L);
    free(st);
    free(mid);
    // stage 2: the lane-per-frame bitstream packer (lc3_dev_enc_pack.h) -- on the GPU 64 frames per wave
    std::vector<uint32_t> cf(64 * 17);
    for (int p = 0; p < 64; p++)
        for (int q = 0; q < 17; q++)
            cf[(size_t)p * 17 + q] = (uint32_t)(int)LC3T_AC_SPEC_CUMFREQ[p][q] | ((uint32_t)(int)LC3T_AC_SPEC_FREQ[p][q] << 16);
    static uint32_t tns_models[LC3_TNS_MODEL_WORDS];
    for (int i = 0; i < LC3_TNS_MODEL_WORDS; i++) tns_models[i] = lc3_tns_model_word(i);
    memset(bytes, 0, frames * (size_t)nbytes);
    for (size_t f = 0; f < frames; f++) {
        lc3_pack_ctx c;
        uint8_t sink = 0;
        c.buf = bytes + f * (size_t)nbytes;
        c.sink = &sink;
        c.tns = tns_models;
        c.nbytes = nbytes;
        c.lookup = LC3T_AC_SPEC_LOOKUP;
        c.cf = cf.data();
        c.plane = LC3_PLANE_COL(planes.data(), f, EP_WORDS);
        c.stride = LC3_PLANE_STRIDE;
        if (pack_pc) {
            // (emulator only) the producer / consumer form of full batches (lc3_pack_pc_kernel): the producer derives every symbol word of
            // the frame into a ring that holds them all, then the consumer codes them (one lane, no concurrency: the waits never wait)
            static uint32_t ring[4096];
            uint32_t fin[4] = {0, 0, 0, 0};
            int p_count = -1, c_count = 0;
            lc3_pc_link k;
            k.ring = ring;
            k.mask = 4095;
            k.stride = 1;
            k.fstride = 1;
            k.p_count = &p_count;
            k.c_count = &c_count;
            k.fin = fin;
            lc3_pack_produce(c, k, j.cfg.ne, 1);
            lc3_pack_consume(c, k, j.cfg.ne, 1);
        } else lc3_pack_frame(c, j.cfg.ne);
    }
    return 0;
}
int lc3emu_decode_late(int fs_hz, int frame_us, int nbytes, int S, int T, const uint8_t *bytes, const uint8_t *bad, int16_t *pcm,
                       int late);
int lc3emu_decode(int fs_hz, int frame_us, int nbytes, int S, int T, const uint8_t *bytes, const uint8_t *bad,
                  int16_t *pcm) {
    return lc3emu_decode_late(fs_hz, frame_us, nbytes, S, T, bytes, bad, pcm, 0);
}
// late = 1: the reconstruction runs in the synthesis stage (what the library does for launches of a few frames); late = 2: in the
// wave-per-frame reconstruction kernel between parser and synthesis (full batches)
int lc3emu_decode_late(int fs_hz, int frame_us, int nbytes, int S, int T, const uint8_t *bytes, const uint8_t *bad, int16_t *pcm,
                       int late) {
    Job j;
    memset(&j, 0, sizeof(j));
    lc3_host_plan pl;
    if (lc3_make_config(j.cfg, frame_us, fs_hz) || lc3_make_plan(j.cfg, pl)) return -1;
    j.cfg.fft_tw = pl.fft_tw.data();
    j.cfg.dct_tw = pl.dct_tw.data();
    j.cfg.perm = pl.perm.data();
    std::vector<uint8_t> lb((size_t)j.cfg.nf);
    for (int k = 0; k < j.cfg.nf; k++) lb[(size_t)k] = (uint8_t)lc3_line_band_value(j.cfg, k);
    j.cfg.line_band = lb.data();
    j.late = late == 3 ? 0 : late;  // (3: the parser's producer / consumer form; the synthesis stage finds f32 spectra as with 0)
    j.encode = 0;
    j.n_frames = T;
    j.nbytes = nbytes;
    j.fresh = 1;
    // stage 1: the lane-per-frame parser (lc3_dev_dec_parse.h) -- on the GPU 64 frames per wave, here a plain loop
    const size_t frames = (size_t)S * (size_t)T;
    std::vector<int32_t> planes(((frames + 63) / 64) * 64 * LC3_PLANE_WORDS, 0);
    alignas(16) static uint32_t cf[64 * LC3_DCF_ROW_WORDS];
    for (int i = 0; i < 64 * LC3_DCF_ROW_WORDS; i++) cf[i] = lc3_dcf_word(i);
    std::vector<uint32_t> tns(LC3_TNS_MODEL_WORDS);
    for (int i = 0; i < LC3_TNS_MODEL_WORDS; i++) tns[(size_t)i] = lc3_tns_model_word(i);
    for (size_t f = 0; f < frames; f++) {
        lc3_parse_ctx c;
        c.dbg = nullptr;
        c.tns = tns.data();
        c.bytes = bytes + f * (size_t)nbytes;
        c.len = nbytes;
        c.lookup = LC3T_AC_SPEC_LOOKUP;
        c.cf = cf;
        c.plane = LC3_PLANE_COL(planes.data(), f, LC3_PLANE_WORDS);
        c.stride = LC3_PLANE_STRIDE;
        c.head = 0;
        c.tail = 0;
        int rc;
        float pc_scf[16], pc_scf_regs[16];
        const float *pc_pre = nullptr;
        if (late == 3) {
            // the producer / consumer form of full batches (lc3_parse_pc_kernel): here the producer walks the frame to its end with a
            // ring that holds every symbol of a frame, then the consumer replays it (one lane, no concurrency: the link's waits never wait)
            static uint32_t ring[4096];
            uint32_t fin[4] = {0, 0, 0, 0};
            int p_count = -1, c_count = 0;
            lc3_pc_link k;
            k.ring = ring;
            k.mask = 4095;
            k.stride = 1;
            k.fstride = 1;
            k.p_count = &p_count;
            k.c_count = &c_count;
            k.fin = fin;
            const int rc_in = (bad && bad[f]) ? -100 : 0;
            lc3_parse_ctx cp = c;  // (its own cursors)
            lc3_pc_produce(cp, k, j.cfg.ne, j.cfg.fs_ind, j.cfg.n_ms_10, rc_in);
            lc3_recon_ctx rr;
            rr.scf = pc_scf;
            rr.sstride = 1;
            rr.mpvq = &LC3T_MPVQ_OFFSETS[0][0];
            rr.ifs = lc3_band_index(j.cfg);
            rc = lc3_pc_consume<1>(c, k, j.cfg.ne, j.cfg.fs_ind, rc_in, &rr, pc_scf_regs);
            pc_pre = pc_scf_regs;
        } else if (late == 2) rc = (bad && bad[f]) ? -100 : lc3_parse_frame<0>(c, j.cfg.ne, j.cfg.fs_ind, j.cfg.n_ms_10);
        else rc = (bad && bad[f]) ? -100 : lc3_parse_frame<1>(c, j.cfg.ne, j.cfg.fs_ind, j.cfg.n_ms_10);
        int ok = rc == 0;
        if (ok && late == 2) {  // the reconstruction kernels take the pulse vector de-enumerated and count the residual bits themselves
            lc3_recon_ctx r;
            r.scf = nullptr;
            r.sstride = 0;
            r.mpvq = &LC3T_MPVQ_OFFSETS[0][0];
            r.ifs = nullptr;
            lc3_reconstruct_prepare_wave(c);
            lc3_parse_pulses(c, r);
        } else if (ok && late && late != 3) {
            ok = lc3_reconstruct_prepare_late(c);
        } else if (ok) {  // the same lane rebuilds the spectrum (lc3_parse_kernel)
            float scf[16];
            lc3_recon_ctx r;
            r.scf = scf;
            r.sstride = 1;
            r.mpvq = &LC3T_MPVQ_OFFSETS[0][0];
            r.ifs = lc3_band_index(j.cfg);
            ok = lc3_reconstruct_frame(c, r, j.cfg, pc_pre);
        }
        lc3_px_set(c, AD_OK, ok);
    }
    // stage 2: synthesis, one emulated wave per stream
    lc3_dec_lds *L = (lc3_dec_lds *)aligned_alloc(16, LC3_WG_WAVES * sizeof(lc3_dec_lds));
    lc3_dec_state *st = (lc3_dec_state *)aligned_alloc(16, LC3_WG_WAVES * sizeof(lc3_dec_state));
    j.DL = L;
    j.planes = planes.data();
    if (late == 2) {  // stage 1b: the wave-per-frame reconstruction kernel (lc3_recon_kernel), LC3_WG_WAVES frames per workgroup
        static lc3_recon_tables rt;
        static lc3_recon_wave rw[LC3_WG_WAVES];
        lc3_recon_tables_stage(j.cfg, rt, 0, 1);
        j.RT = &rt;
        j.RW = rw;
        for (size_t f0 = 0; f0 < frames; f0 += LC3_WG_WAVES) {
            Job protos[LC3_WG_WAVES];
            memset(rw, 0xFF, sizeof(rw));  // LDS is not zeroed on the GPU
            for (int w = 0; w < LC3_WG_WAVES; w++) {
                protos[w] = j;
                protos[w].encode = 3;
                protos[w].valid = f0 + (size_t)w < frames;
                protos[w].frame0 = f0 + (size_t)w;
            }
            run_wg(protos);
        }
        // stage 1c: the lane-per-frame TNS pass (lc3_tns_kernel) -- on the GPU 64 frames per wave, here a plain loop
        static float gains[64], sin_tab[17];
        static uint32_t lb[LC3_MAX_NF / 4];
        for (int i = 0; i < 17; i++) sin_tab[i] = lc3_tns_sin_dec_value(i);
        for (int i = 0; i < LC3_MAX_NF / 4; i++) lb[i] = lc3_line_band_word(j.cfg, i);
        for (size_t f = 0; f < frames; f++) {
            lc3_tns_lane_ctx x;
            x.col = LC3_PLANE_COL(planes.data(), f, LC3_PLANE_WORDS);
            x.gains = gains;
            x.gstride = 1;
            x.sin_tab = sin_tab;
            x.line_band = lb;
            lc3_tns_lane_frame(j.cfg, x, 1);
        }
        j.late = 0;  // the synthesis stage finds f32 spectra
    }
    for (int s0 = 0; s0 < S; s0 += LC3_WG_WAVES) {
        Job protos[LC3_WG_WAVES];
        memset(L, 0xFF, LC3_WG_WAVES * sizeof(lc3_dec_lds));  // (see lc3emu_encode)
        memset(st, 0, LC3_WG_WAVES * sizeof(lc3_dec_state));
        for (int w = 0; w < LC3_WG_WAVES; w++) {
            const int valid = s0 + w < S, s = valid ? s0 + w : S - 1;
            protos[w] = j;
            protos[w].valid = valid;
            protos[w].dst = st + w;
            protos[w].bytes_in = bytes + (size_t)s * T * nbytes;
            protos[w].frame0 = (size_t)s * T;
            protos[w].pcm_out = pcm + (size_t)s * T * j.cfg.nf;
        }
        run_wg(protos);
    }
    free(L);
    free(st);
    return 0;
}
float lc3emu_pow10f(float y) { return lc3_pow10f(y); }
float lc3emu_log2f(float x) { return lc3_log2f(x); }
float lc3emu_log10f(float x) { return lc3_log10f(x); }
float lc3emu_exp2f(float x) { return lc3_exp2f(x); }
float lc3emu_asinf(float x) { return lc3_asinf(x); }
float lc3emu_sinf_small(float x) { return lc3_sinf_small(x); }
float lc3emu_exp2_raw(float x) { return lc3_exp2_raw(x); }
}
