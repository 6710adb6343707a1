// lc3gpu_pipeline -- the caller loop of the reference, as an object of the library (host code only, on top of the C ABI of lc3gpu.hip).
//
// The reference's callers hold the codec objects and walk the frames: examples/encode.rs:97-115 (for every frame, for every channel:
// encode_frame) and examples/decode.rs:93-112 (the mirror).  On the GPU the fastest arrangement of that loop is not one call after the
// other on one HIP stream: a lane-per-frame kernel (packer, parser, SNS vector quantiser) leaves most of the chip's workgroup slots
// free, a wave-per-stream kernel fills them, and kernels of DIFFERENT calls can share the chip.  What measured best (DESIGN section 6,
// `quad`): the channels in two groups, every group with an encoder handle and a decoder handle of its own, the encoder handle on a
// HIP stream of the higher priority, the decoder handle on a stream of the default priority, two byte buffers in flight per group, the
// groups never joining.  Until round 6 that arrangement lived in bench.py; this file makes it a property of the library:
//
//   lc3gpu_pipeline_create  -> the handles, streams and events of `n_groups` groups
//   lc3gpu_pipeline_submit  -> one step of every group: encode d_pcm -> d_bytes on the group's encoder stream, decode d_bytes -> d_pcm_out
//                              on its decoder stream behind an event; asynchronous
//   lc3gpu_pipeline_encode / _decode -> the halves alone, the groups side by side
//   lc3gpu_pipeline_wait / _join / _follow -> the host / a caller's HIP stream waits for the pipeline / the pipeline for a caller's stream
//
// Buffers are planar (LC3GPU_LAYOUT_PLANAR): a group's channels are a contiguous slice of every buffer.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/lc3gpu.h"

namespace {
struct Group {
    int first = 0, n = 0;
    lc3gpu_encoder *enc = nullptr;
    lc3gpu_decoder *dec = nullptr;
    hipStream_t s_enc = nullptr, s_dec = nullptr;
    // the two most recent submissions of this group: behind the encoder's / the decoder's work of each
    hipEvent_t enc_done[2] = {nullptr, nullptr}, dec_done[2] = {nullptr, nullptr};
    bool enc_rec[2] = {false, false}, dec_rec[2] = {false, false};
    // the byte range the decoder of that submission READS and the encoder of a later one may WRITE
    const uint8_t *bytes_lo[2] = {nullptr, nullptr}, *bytes_hi[2] = {nullptr, nullptr};
    // the byte range the encoder of that submission WRITES and a decoder of the same or a later submission may read
    const uint8_t *ebytes_lo[2] = {nullptr, nullptr}, *ebytes_hi[2] = {nullptr, nullptr};
    // the PCM range the decoder of that submission WRITES (a later encoder may read it: a transcoding chain)
    int last_enc = -1, last_dec = -1;  // slot of the latest encoder / decoder work (for wait / join)
    unsigned long long n_enc = 0, n_dec = 0;  // pieces of work queued per role (a role's slot = its count & 1)
    // mixed-configuration pipelines (ragged buffers, streams in descriptor order): per FRAME of a submission, the int16 samples and the
    // bytes of the streams before this group, and of this group's own streams
    size_t pcm_before = 0, bytes_before = 0, pcm_own = 0, bytes_own = 0;
};
}  // namespace

struct lc3gpu_pipeline {
    int device = 0;
    bool own_streams = false;  // the streams were created for this pipeline alone (LC3GPU_PIPELINE_OWN_STREAMS=1) and die with it
    bool mixed = false;        // lc3gpu_pipeline_create_mixed: per-stream configurations, ragged buffers
    int num_channels = 0, nf = 0;
    unsigned long long k = 0;  // submissions so far
    std::vector<Group> groups;
    hipEvent_t ev_follow = nullptr;
    bool follow_pending = false;
    int last_hip = 0;
};

namespace {
struct DeviceGuard {
    int prev = -1, dev;
    explicit DeviceGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};
#define PL_HIP(p, call)                      \
    do {                                     \
        hipError_t e_ = (call);              \
        if (e_ != hipSuccess) {              \
            (p)->last_hip = (int)e_;         \
            (void)hipGetLastError();         \
            return LC3GPU_EHIP;              \
        }                                    \
    } while (0)

// The pipeline streams of a device: created once per process and SHARED by every pipeline object on that device, never destroyed.
// Why (profiles/r06_pipeline_stream_order.txt): the runtime maps HIP streams onto a few hardware queues per priority class (4 by default),
// and a stream created when its class is full SHARES the least-used queue -- two chains on one queue run one behind the other.  Measured: the
// first pipeline of a process 58.9 M frames/s, a second one created beside it (or beside any four live streams) 49.4 M, and it stays there;
// a third that inherits the queues of a closed first one 59.2 M again.  With one set of streams per device every pipeline is "the first
// one".  (Pipelines that are busy at the same time then share the streams, i.e. take turns: a process normally has one per device.)
struct StreamSet {
    std::vector<hipStream_t> enc, dec;
};
std::mutex g_pool_mutex;
std::map<int, StreamSet> g_pool;

bool env_is(const char *name, const char *value) {
    const char *e = std::getenv(name);
    return e && !std::strcmp(e, value);
}
// priorities: the encoder streams the greatest the device has; the decoder streams the default one, or (LC3GPU_PIPELINE_DEC_PRIO=low) the
// least -- a class of hardware queues the application's own default-priority streams never touch
void stream_priorities(int &prio_enc, int &prio_dec) {
    int least = 0, greatest = 0;  // (numerically the greatest priority is the smallest number: -1 high, 0 default, 1 low on this platform)
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    prio_enc = greatest;
    prio_dec = env_is("LC3GPU_PIPELINE_DEC_PRIO", "low") ? least : (greatest < 0 ? 0 : least);
}
// stream `g` of a role on the current device, from the shared set (created on demand) or fresh (own = true)
int role_stream(int device, bool own, bool is_enc, int g, hipStream_t *out) {
    int prio_enc, prio_dec;
    stream_priorities(prio_enc, prio_dec);
    const int prio = is_enc ? prio_enc : prio_dec;
    if (own) return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio) == hipSuccess ? LC3GPU_OK : LC3GPU_EHIP;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    std::vector<hipStream_t> &v = is_enc ? g_pool[device].enc : g_pool[device].dec;
    while ((int)v.size() <= g) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess) return LC3GPU_EHIP;
        v.push_back(st);
    }
    *out = v[(size_t)g];
    return LC3GPU_OK;
}

// the pipeline's next work on `s` waits for what lc3gpu_pipeline_follow recorded
int follow_wait(lc3gpu_pipeline *p, hipStream_t s) {
    if (p->follow_pending) PL_HIP(p, hipStreamWaitEvent(s, p->ev_follow, 0));
    return LC3GPU_OK;
}
bool overlaps(const uint8_t *a_lo, const uint8_t *a_hi, const uint8_t *b_lo, const uint8_t *b_hi) { return a_lo < b_hi && b_lo < a_hi; }
}  // namespace

extern "C" {

int lc3gpu_pipeline_destroy(lc3gpu_pipeline *p) {
    if (!p) return LC3GPU_OK;
    {
        DeviceGuard g(p->device);
        for (Group &q : p->groups) {
            if (q.s_enc) (void)hipStreamSynchronize(q.s_enc);
            if (q.s_dec) (void)hipStreamSynchronize(q.s_dec);
            if (q.enc) (void)lc3gpu_encoder_destroy(q.enc);
            if (q.dec) (void)lc3gpu_decoder_destroy(q.dec);
            for (int i = 0; i < 2; i++) {
                if (q.enc_done[i]) (void)hipEventDestroy(q.enc_done[i]);
                if (q.dec_done[i]) (void)hipEventDestroy(q.dec_done[i]);
            }
            if (p->own_streams) {  // (the shared streams of the device stay for the next pipeline)
                if (q.s_enc) (void)hipStreamDestroy(q.s_enc);
                if (q.s_dec) (void)hipStreamDestroy(q.s_dec);
            }
        }
        if (p->ev_follow) (void)hipEventDestroy(p->ev_follow);
    }
    delete p;
    return LC3GPU_OK;
}

// streams, events and the binding of a group whose handles exist
static int group_finish(lc3gpu_pipeline *p, Group &q, int g) {
    // The encoder chain (front half -> vector quantiser -> back half -> packer) is each group's critical path: its stream gets the
    // higher HIP stream priority.  Streams of another priority also live on hardware queues of their own (see StreamSet).
    if (role_stream(p->device, p->own_streams, true, g, &q.s_enc) != LC3GPU_OK || role_stream(p->device, p->own_streams, false, g, &q.s_dec) != LC3GPU_OK)
        return LC3GPU_EHIP;
    for (int i = 0; i < 2; i++)
        if (hipEventCreateWithFlags(&q.enc_done[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q.dec_done[i], hipEventDisableTiming) != hipSuccess)
            return LC3GPU_EHIP;
    // the handles only ever run on the group's two streams, which outlive them: no event of their own per call
    int rc = lc3gpu_encoder_bind_stream(q.enc, q.s_enc, 1);
    if (rc == LC3GPU_OK) rc = lc3gpu_decoder_bind_stream(q.dec, q.s_dec, 1);
    return rc;
}

int lc3gpu_pipeline_create(lc3gpu_pipeline **out, int num_channels, int frame_us, int fs_hz, int n_groups) {
    if (!out || num_channels <= 0 || n_groups < 0 || n_groups > 8) return LC3GPU_EINVAL;
    *out = nullptr;
    int cfg[7];
    int rc = lc3gpu_config(frame_us, fs_hz, cfg);
    if (rc) return rc;
    if (lc3gpu_device_count() <= 0) return LC3GPU_ENODEVICE;
    if (n_groups == 0) n_groups = 2;  // what measured best on a full batch (more groups: every group adds its lane-per-frame kernels' latency)
    // groups of whole workgroups of the wave-per-stream kernels (four channels); fewer groups when there are not enough channels
    const int quads = (num_channels + 3) / 4;
    if (n_groups > quads) n_groups = quads;
    lc3gpu_pipeline *p = new (std::nothrow) lc3gpu_pipeline();
    if (!p) return LC3GPU_EINVAL;
    p->num_channels = num_channels;
    p->nf = cfg[5];
    if (hipGetDevice(&p->device) != hipSuccess) {
        delete p;
        return LC3GPU_EHIP;
    }
    p->own_streams = env_is("LC3GPU_PIPELINE_OWN_STREAMS", "1");
    p->groups.resize((size_t)n_groups);
    rc = LC3GPU_OK;
    for (int g = 0; g < n_groups && rc == LC3GPU_OK; g++) {
        Group &q = p->groups[(size_t)g];
        const int lo = (int)((long long)quads * g / n_groups) * 4, hi = g + 1 == n_groups ? num_channels : (int)((long long)quads * (g + 1) / n_groups) * 4;
        q.first = lo;
        q.n = hi - lo;
        rc = lc3gpu_encoder_create(&q.enc, q.n, frame_us, fs_hz);
        if (rc == LC3GPU_OK) rc = lc3gpu_decoder_create(&q.dec, q.n, frame_us, fs_hz);
        if (rc == LC3GPU_OK) rc = group_finish(p, q, g);
    }
    if (rc == LC3GPU_OK && hipEventCreateWithFlags(&p->ev_follow, hipEventDisableTiming) != hipSuccess) rc = LC3GPU_EHIP;
    if (rc) {
        if (rc == LC3GPU_EHIP) (void)hipGetLastError();
        lc3gpu_pipeline_destroy(p);
        return rc;
    }
    *out = p;
    return LC3GPU_OK;
}

// Streams of different configurations in one pipeline (BASELINE config 4 through the pipeline object): the reference would build one
// Lc3Encoder / Lc3Decoder per configuration (encoder/lc3_encoder.rs:117-124, common/config.rs:42-100) and loop over them; here group g
// takes the descriptors [group_first[g], group_first[g + 1]) -- the caller chooses the boundaries, e.g. so that every group holds a share
// of every configuration -- with a mixed encoder handle and a mixed decoder handle of its own.  8 kHz streams are refused as by
// lc3gpu_encoder_create_mixed (a pipeline encodes AND decodes; decode-only streams belong to a plain decoder handle).
int lc3gpu_pipeline_create_mixed(lc3gpu_pipeline **out, int n_streams, const lc3gpu_stream_desc *descs, int n_groups, const int *group_first) {
    if (!out || n_streams <= 0 || !descs || n_groups < 0 || n_groups > 8) return LC3GPU_EINVAL;
    *out = nullptr;
    if (lc3gpu_device_count() <= 0) return LC3GPU_ENODEVICE;
    if (n_groups == 0) n_groups = 2;
    if (n_groups > n_streams) n_groups = n_streams;
    std::vector<int> first((size_t)n_groups + 1);
    for (int g = 0; g <= n_groups; g++) first[(size_t)g] = g == n_groups ? n_streams : (group_first ? group_first[g] : (int)((long long)n_streams * g / n_groups));
    if (first[0] != 0) return LC3GPU_EINVAL;
    for (int g = 0; g < n_groups; g++)
        if (first[(size_t)g + 1] <= first[(size_t)g]) return LC3GPU_EINVAL;
    lc3gpu_pipeline *p = new (std::nothrow) lc3gpu_pipeline();
    if (!p) return LC3GPU_EINVAL;
    p->num_channels = n_streams;
    p->mixed = true;
    if (hipGetDevice(&p->device) != hipSuccess) {
        delete p;
        return LC3GPU_EHIP;
    }
    p->own_streams = env_is("LC3GPU_PIPELINE_OWN_STREAMS", "1");
    p->groups.resize((size_t)n_groups);
    int rc = LC3GPU_OK;
    size_t pcm_run = 0, bytes_run = 0;
    for (int g = 0; g < n_groups && rc == LC3GPU_OK; g++) {
        Group &q = p->groups[(size_t)g];
        q.first = first[(size_t)g];
        q.n = first[(size_t)g + 1] - q.first;
        q.pcm_before = pcm_run;
        q.bytes_before = bytes_run;
        for (int i = q.first; i < q.first + q.n && rc == LC3GPU_OK; i++) {
            int cfg[7];
            rc = lc3gpu_config(descs[i].frame_us, descs[i].fs_hz, cfg);
            if (rc == LC3GPU_OK && (descs[i].nbytes < 20 || descs[i].nbytes > 400)) rc = LC3GPU_ELENGTH;
            if (rc == LC3GPU_OK) {
                q.pcm_own += (size_t)cfg[5];
                q.bytes_own += (size_t)descs[i].nbytes;
            }
        }
        pcm_run += q.pcm_own;
        bytes_run += q.bytes_own;
        if (rc == LC3GPU_OK) rc = lc3gpu_encoder_create_mixed(&q.enc, q.n, descs + q.first);
        if (rc == LC3GPU_OK) rc = lc3gpu_decoder_create_mixed(&q.dec, q.n, descs + q.first);
        if (rc == LC3GPU_OK) rc = group_finish(p, q, g);
    }
    if (rc == LC3GPU_OK && hipEventCreateWithFlags(&p->ev_follow, hipEventDisableTiming) != hipSuccess) rc = LC3GPU_EHIP;
    if (rc) {
        if (rc == LC3GPU_EHIP) (void)hipGetLastError();
        lc3gpu_pipeline_destroy(p);
        return rc;
    }
    *out = p;
    return LC3GPU_OK;
}

int lc3gpu_pipeline_groups(const lc3gpu_pipeline *p) { return p ? (int)p->groups.size() : LC3GPU_EINVAL; }

int lc3gpu_pipeline_group(lc3gpu_pipeline *p, int group, int *first_channel, int *n_channels, lc3gpu_encoder **enc, lc3gpu_decoder **dec) {
    if (!p || group < 0 || group >= (int)p->groups.size()) return LC3GPU_EINVAL;
    const Group &q = p->groups[(size_t)group];
    if (first_channel) *first_channel = q.first;
    if (n_channels) *n_channels = q.n;
    if (enc) *enc = q.enc;
    if (dec) *dec = q.dec;
    return LC3GPU_OK;
}

int lc3gpu_pipeline_last_hip_error(const lc3gpu_pipeline *p) { return p ? p->last_hip : 0; }

// Hazards between the two streams of a group.  Each role keeps the records of its two most recent submissions (slot = the role's own
// count & 1): the byte range it read (decoder) or wrote (encoder) and an event behind its work.  A new piece of work waits for the
// other role's records it overlaps; records that have been dropped are all OLDER than both live ones, so when nothing live overlaps and
// something has been dropped, waiting for the older live record orders the work behind every dropped one as well (a caller rotating three
// byte buffers, halves submitted alone in any order).  In the round trip with two alternating byte buffers this is exactly one wait per
// role and submission: the encoder of submission k behind the decoder of k - 2, the decoder behind its own encoder.
static int wait_for_other_role(lc3gpu_pipeline *p, hipStream_t s, const uint8_t *lo, const uint8_t *hi, const hipEvent_t (&ev)[2], const bool (&rec)[2],
                               const uint8_t *const (&r_lo)[2], const uint8_t *const (&r_hi)[2], unsigned long long count) {
    if (count == 0) return LC3GPU_OK;
    const int newer = (int)((count - 1) & 1ull), older = newer ^ 1;
    // An event that has already completed orders nothing, and a wait for it would still cost the stream a barrier packet (~5 us of queue
    // time each; four of them per submission measured 1.7 % of the step): the host asks first.  A caller that stays a couple of submissions
    // ahead of the device -- a service ticking every 10 ms -- then queues none for the byte buffers; one that runs far ahead queues them all.
    auto wait = [&](int i) -> int {
        if (hipEventQuery(ev[i]) == hipSuccess) return LC3GPU_OK;
        (void)hipGetLastError();  // (hipErrorNotReady is not an error)
        PL_HIP(p, hipStreamWaitEvent(s, ev[i], 0));
        return LC3GPU_OK;
    };
    bool any = false;
    for (int i : {newer, older})
        if (rec[i] && overlaps(lo, hi, r_lo[i], r_hi[i])) {
            const int rc = wait(i);
            if (rc) return rc;
            any = true;
            if (i == newer) break;  // (the newer record's event is behind the older one's on the same stream)
        }
    if (!any && count > 2 && rec[older]) return wait(older);
    return LC3GPU_OK;
}

// what = 1 encode, 2 decode, 3 both
static int pipeline_step(lc3gpu_pipeline *p, int what, const int16_t *d_pcm, uint8_t *d_bytes, const uint8_t *d_bad, int16_t *d_pcm_out, int nbytes,
                         int n_frames) {
    if (!p || !d_bytes || ((what & 1) && !d_pcm) || ((what & 2) && !d_pcm_out)) return LC3GPU_EINVAL;
    if (n_frames <= 0) return LC3GPU_ELENGTH;
    if (p->mixed != (nbytes < 0)) return LC3GPU_EINVAL;  // (the mixed entry points pass nbytes = -1: frame sizes are the descriptors')
    DeviceGuard dg(p->device);
    int rc = LC3GPU_OK;
    for (Group &q : p->groups) {
        // the group's slice of every buffer: uniform [channel][frame][..] planar; mixed ragged, streams in descriptor order
        const size_t f0 = (size_t)q.first * (size_t)n_frames;
        const size_t pcm_off = p->mixed ? q.pcm_before * (size_t)n_frames : f0 * (size_t)p->nf;
        const size_t bytes_off = p->mixed ? q.bytes_before * (size_t)n_frames : f0 * (size_t)nbytes;
        const size_t bytes_len = p->mixed ? q.bytes_own * (size_t)n_frames : (size_t)q.n * (size_t)n_frames * (size_t)nbytes;
        uint8_t *bytes = d_bytes + bytes_off;
        const uint8_t *bytes_end = bytes + bytes_len;
        if (what & 1) {
            // the encoder may not overwrite bytes a decoder still reads
            if ((rc = wait_for_other_role(p, q.s_enc, bytes, bytes_end, q.dec_done, q.dec_rec, q.bytes_lo, q.bytes_hi, q.n_dec)) != 0) return rc;
            if ((rc = follow_wait(p, q.s_enc)) != 0) return rc;
            rc = p->mixed ? lc3gpu_encode_mixed(q.enc, d_pcm + pcm_off, bytes, n_frames, q.s_enc)
                          : lc3gpu_encode(q.enc, d_pcm + pcm_off, bytes, nbytes, n_frames, q.s_enc);
            if (rc) return rc;
            const int b = (int)(q.n_enc & 1ull);
            PL_HIP(p, hipEventRecord(q.enc_done[b], q.s_enc));
            q.enc_rec[b] = true;
            q.ebytes_lo[b] = bytes;
            q.ebytes_hi[b] = bytes_end;
            q.last_enc = b;
            q.n_enc++;
        }
        if (what & 2) {
            // the decoder reads what an encoder in flight writes: this submission's own (a round trip) or an earlier lc3gpu_pipeline_encode's
            if ((rc = wait_for_other_role(p, q.s_dec, bytes, bytes_end, q.enc_done, q.enc_rec, q.ebytes_lo, q.ebytes_hi, q.n_enc)) != 0) return rc;
            if ((rc = follow_wait(p, q.s_dec)) != 0) return rc;
            const uint8_t *bad = d_bad ? d_bad + f0 : nullptr;
            rc = p->mixed ? lc3gpu_decode_mixed(q.dec, bytes, bad, d_pcm_out + pcm_off, n_frames, q.s_dec)
                          : lc3gpu_decode(q.dec, bytes, bad, d_pcm_out + pcm_off, nbytes, n_frames, q.s_dec);
            if (rc) return rc;
            const int b = (int)(q.n_dec & 1ull);
            PL_HIP(p, hipEventRecord(q.dec_done[b], q.s_dec));
            q.dec_rec[b] = true;
            q.bytes_lo[b] = bytes;
            q.bytes_hi[b] = bytes_end;
            q.last_dec = b;
            q.n_dec++;
        }
    }
    p->follow_pending = false;
    p->k++;
    return LC3GPU_OK;
}

int lc3gpu_pipeline_submit(lc3gpu_pipeline *p, const int16_t *d_pcm, uint8_t *d_bytes, int16_t *d_pcm_out, int nbytes, int n_frames) {
    return pipeline_step(p, 3, d_pcm, d_bytes, nullptr, d_pcm_out, nbytes, n_frames);
}
int lc3gpu_pipeline_encode(lc3gpu_pipeline *p, const int16_t *d_pcm, uint8_t *d_bytes, int nbytes, int n_frames) {
    return pipeline_step(p, 1, d_pcm, d_bytes, nullptr, nullptr, nbytes, n_frames);
}
int lc3gpu_pipeline_decode(lc3gpu_pipeline *p, const uint8_t *d_bytes, const uint8_t *d_bad_frame, int16_t *d_pcm_out, int nbytes, int n_frames) {
    return pipeline_step(p, 2, nullptr, (uint8_t *)d_bytes, d_bad_frame, d_pcm_out, nbytes, n_frames);
}

int lc3gpu_pipeline_submit_mixed(lc3gpu_pipeline *p, const int16_t *d_pcm, uint8_t *d_bytes, int16_t *d_pcm_out, int n_frames) {
    return pipeline_step(p, 3, d_pcm, d_bytes, nullptr, d_pcm_out, -1, n_frames);
}
int lc3gpu_pipeline_encode_mixed(lc3gpu_pipeline *p, const int16_t *d_pcm, uint8_t *d_bytes, int n_frames) {
    return pipeline_step(p, 1, d_pcm, d_bytes, nullptr, nullptr, -1, n_frames);
}
int lc3gpu_pipeline_decode_mixed(lc3gpu_pipeline *p, const uint8_t *d_bytes, const uint8_t *d_bad_frame, int16_t *d_pcm_out, int n_frames) {
    return pipeline_step(p, 2, nullptr, (uint8_t *)d_bytes, d_bad_frame, d_pcm_out, -1, n_frames);
}

int lc3gpu_pipeline_wait(lc3gpu_pipeline *p) {
    if (!p) return LC3GPU_EINVAL;
    DeviceGuard dg(p->device);
    for (Group &q : p->groups) {
        if (q.last_enc >= 0) PL_HIP(p, hipEventSynchronize(q.enc_done[q.last_enc]));
        if (q.last_dec >= 0) PL_HIP(p, hipEventSynchronize(q.dec_done[q.last_dec]));
    }
    return LC3GPU_OK;
}

int lc3gpu_pipeline_join(lc3gpu_pipeline *p, void *hip_stream) {
    if (!p) return LC3GPU_EINVAL;
    DeviceGuard dg(p->device);
    for (Group &q : p->groups) {
        if (q.last_enc >= 0) PL_HIP(p, hipStreamWaitEvent((hipStream_t)hip_stream, q.enc_done[q.last_enc], 0));
        if (q.last_dec >= 0) PL_HIP(p, hipStreamWaitEvent((hipStream_t)hip_stream, q.dec_done[q.last_dec], 0));
    }
    return LC3GPU_OK;
}

int lc3gpu_pipeline_mark(lc3gpu_pipeline *p, void *hip_event) {
    if (!p || !hip_event) return LC3GPU_EINVAL;
    DeviceGuard dg(p->device);
    const Group &q = p->groups.back();
    PL_HIP(p, hipEventRecord((hipEvent_t)hip_event, q.last_dec >= 0 ? q.s_dec : q.s_enc));
    return LC3GPU_OK;
}

int lc3gpu_pipeline_follow(lc3gpu_pipeline *p, void *hip_stream) {
    if (!p) return LC3GPU_EINVAL;
    DeviceGuard dg(p->device);
    PL_HIP(p, hipEventRecord(p->ev_follow, (hipStream_t)hip_stream));
    p->follow_pending = true;
    return LC3GPU_OK;
}

int lc3gpu_pipeline_reset(lc3gpu_pipeline *p) {
    if (!p) return LC3GPU_EINVAL;
    int rc = LC3GPU_OK;  // (asynchronous, as the handles' resets: the next submission starts every channel from the constructed state)
    for (Group &q : p->groups) {
        if (rc == LC3GPU_OK) rc = lc3gpu_encoder_reset(q.enc);
        if (rc == LC3GPU_OK) rc = lc3gpu_decoder_reset(q.dec);
    }
    return rc;
}

}  // extern "C"
