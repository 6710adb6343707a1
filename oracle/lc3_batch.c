/* ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See lc3_oracle.h.
 * Batch drivers used by the parity tests and by bench.py's cpu_baseline leg:
 * every stream is an independent codec channel (its own encoder/decoder object,
 * exactly like one reference `Lc3Encoder` channel fed frame after frame,
 * examples/encode.rs:105-115).  Streams are distributed over host threads. */
#include "lc3_oracle.h"
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int fs_hz, frame_us, nbytes, n_frames, s_begin, s_end, nf, encode, rc, spec_flags;
    const int16_t *pcm_in;
    uint8_t *bytes_out;
    const uint8_t *bytes_in;
    int16_t *pcm_out;
} job;

static void *worker(void *arg) {
    job *j = (job *)arg;
    int s, t;
    if (j->encode) {
        lc3o_encoder *e = (lc3o_encoder *)malloc(sizeof(lc3o_encoder));
        if (!e) { j->rc = -2; return 0; }
        for (s = j->s_begin; s < j->s_end; s++) {
            if (lc3o_encoder_init_spec(e, j->fs_hz, j->frame_us, j->spec_flags)) { j->rc = -1; break; }
            for (t = 0; t < j->n_frames; t++) {
                size_t f = (size_t)s * (size_t)j->n_frames + (size_t)t;
                lc3o_encode_frame(e, j->pcm_in + f * (size_t)j->nf, j->bytes_out + f * (size_t)j->nbytes, j->nbytes);
            }
        }
        free(e);
    } else {
        lc3o_decoder *d = (lc3o_decoder *)malloc(sizeof(lc3o_decoder));
        if (!d) { j->rc = -2; return 0; }
        for (s = j->s_begin; s < j->s_end; s++) {
            if (lc3o_decoder_init(d, j->fs_hz, j->frame_us)) { j->rc = -1; break; }
            for (t = 0; t < j->n_frames; t++) {
                size_t f = (size_t)s * (size_t)j->n_frames + (size_t)t;
                lc3o_decode_frame(d, 16, j->bytes_in + f * (size_t)j->nbytes, j->nbytes, j->pcm_out + f * (size_t)j->nf);
            }
        }
        free(d);
    }
    return 0;
}

static int run(job proto, int n_streams, int n_threads) {
    int i, rc = 0;
    lc3o_config c;
    if (lc3o_config_new(&c, proto.fs_hz, proto.frame_us)) return -1;
    proto.nf = c.nf;
    if (n_threads <= 1) {
        proto.s_begin = 0;
        proto.s_end = n_streams;
        worker(&proto);
        return proto.rc;
    }
    {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
        job *jobs = (job *)malloc(sizeof(job) * (size_t)n_threads);
        for (i = 0; i < n_threads; i++) {
            jobs[i] = proto;
            jobs[i].s_begin = (int)((long long)n_streams * i / n_threads);
            jobs[i].s_end = (int)((long long)n_streams * (i + 1) / n_threads);
            pthread_create(&th[i], 0, worker, &jobs[i]);
        }
        for (i = 0; i < n_threads; i++) {
            pthread_join(th[i], 0);
            if (jobs[i].rc) rc = jobs[i].rc;
        }
        free(th);
        free(jobs);
    }
    return rc;
}

int lc3o_encode_batch(int fs_hz, int frame_us, int nbytes, int n_streams, int n_frames, const int16_t *pcm,
                      uint8_t *bytes, int n_threads) {
    return lc3o_encode_batch_spec(fs_hz, frame_us, nbytes, n_streams, n_frames, pcm, bytes, n_threads, 0);
}

int lc3o_encode_batch_spec(int fs_hz, int frame_us, int nbytes, int n_streams, int n_frames, const int16_t *pcm, uint8_t *bytes,
                           int n_threads, int spec_flags) {
    job j = {0};
    j.spec_flags = spec_flags;
    j.fs_hz = fs_hz; j.frame_us = frame_us; j.nbytes = nbytes; j.n_frames = n_frames; j.encode = 1;
    j.pcm_in = pcm; j.bytes_out = bytes;
    return run(j, n_streams, n_threads);
}

int lc3o_decode_batch(int fs_hz, int frame_us, int nbytes, int n_streams, int n_frames, const uint8_t *bytes,
                      int16_t *pcm, int n_threads) {
    job j = {0};
    j.fs_hz = fs_hz; j.frame_us = frame_us; j.nbytes = nbytes; j.n_frames = n_frames; j.encode = 0;
    j.bytes_in = bytes; j.pcm_out = pcm;
    return run(j, n_streams, n_threads);
}

/* ---- bench.py's cpu_baseline leg --------------------------------------------------------------------------------------------
 * n_threads host threads, each with ONE persistent encoder (and decoder, when roundtrip) coding its own stream of n_frames
 * consecutive frames, again and again, until `seconds` have passed.  Threads are created, allocate and initialise their codec
 * objects and buffers, and meet at a barrier BEFORE the clock starts: no allocation and no thread creation inside the timed
 * region.  pcm: int16[n_distinct][n_frames][nf]; thread t codes streams t, t + 1, ... (mod n_distinct), one per pass.
 * Out: total frames coded (an encode+decode pair counts once) and the time from the barrier to the last thread's finish. */
typedef struct {
    volatile int ready;  /* workers that have allocated and initialised */
    volatile int go;     /* 0: wait, 1: run, -1: a worker could not be created -- leave without working */
} start_gate;
typedef struct {
    int fs_hz, frame_us, nbytes, n_frames, nf, roundtrip, rc, first, n_distinct;
    const int16_t *pcm;
    double seconds, frames, t_start, t_end;
    start_gate *gate;
} tjob;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *timed_worker(void *arg) {
    tjob *j = (tjob *)arg;
    lc3o_encoder *e = (lc3o_encoder *)malloc(sizeof(lc3o_encoder));
    lc3o_decoder *d = (lc3o_decoder *)malloc(sizeof(lc3o_decoder));
    uint8_t *bytes = (uint8_t *)malloc((size_t)j->nbytes);
    int16_t *out = (int16_t *)malloc(sizeof(int16_t) * (size_t)j->nf);
    int t;
    double t0, deadline;
    if (!e || !d || !bytes || !out || lc3o_encoder_init_spec(e, j->fs_hz, j->frame_us, 0) || lc3o_decoder_init(d, j->fs_hz, j->frame_us))
        j->rc = -1;
    if (bytes) memset(bytes, 0, (size_t)j->nbytes);
    /* the start gate: counts the threads that really exist (a pthread barrier sized for the planned number would hang the others if one
     * could not be created) */
    __atomic_add_fetch(&j->gate->ready, 1, __ATOMIC_SEQ_CST);
    while (__atomic_load_n(&j->gate->go, __ATOMIC_SEQ_CST) == 0) sched_yield();
    t0 = now_s();
    j->t_start = t0;
    deadline = t0 + j->seconds;
    if (!j->rc && __atomic_load_n(&j->gate->go, __ATOMIC_SEQ_CST) > 0) {
        int pass = 0;
        do {  /* one pass = one stream from its first frame (fresh codec objects: ~1 frame's worth of work per n_frames) */
            const int16_t *x = j->pcm + (size_t)((j->first + pass) % j->n_distinct) * (size_t)j->n_frames * (size_t)j->nf;
            if (pass > 0 && (lc3o_encoder_init_spec(e, j->fs_hz, j->frame_us, 0) || lc3o_decoder_init(d, j->fs_hz, j->frame_us))) { j->rc = -1; break; }
            for (t = 0; t < j->n_frames; t++) {
                lc3o_encode_frame(e, x + (size_t)t * (size_t)j->nf, bytes, j->nbytes);
                if (j->roundtrip) lc3o_decode_frame(d, 16, bytes, j->nbytes, out);
            }
            j->frames += (double)j->n_frames;
            pass++;
        } while (now_s() < deadline);
    }
    j->t_end = now_s();
    free(e);
    free(d);
    free(bytes);
    free(out);
    return 0;
}

int lc3o_timed_run(int fs_hz, int frame_us, int nbytes, int n_frames, const int16_t *pcm, int n_distinct, int n_threads, int roundtrip,
                   double seconds, double *frames_out, double *elapsed_out) {
    lc3o_config c;
    start_gate gate = {0, 0};
    pthread_t *th;
    tjob *jobs;
    double t_start = 0.0, t_end = 0.0, frames = 0.0;
    int i, rc = 0;
    if (n_threads < 1 || n_distinct < 1 || n_frames < 1 || lc3o_config_new(&c, fs_hz, frame_us)) return -1;
    th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    jobs = (tjob *)calloc((size_t)n_threads, sizeof(tjob));
    if (!th || !jobs) {
        free(th);
        free(jobs);
        return -2;
    }
    for (i = 0; i < n_threads; i++) {
        tjob *j = &jobs[i];
        j->fs_hz = fs_hz; j->frame_us = frame_us; j->nbytes = nbytes; j->n_frames = n_frames; j->nf = c.nf; j->roundtrip = roundtrip;
        j->pcm = pcm;
        j->first = i % n_distinct;
        j->n_distinct = n_distinct;
        j->seconds = seconds;
        j->gate = &gate;
        if (pthread_create(&th[i], 0, timed_worker, j)) { rc = -3; n_threads = i; break; }
    }
    /* every thread that exists has allocated and initialised: open the gate (the clock starts in the workers) */
    while (__atomic_load_n(&gate.ready, __ATOMIC_SEQ_CST) < n_threads) sched_yield();
    __atomic_store_n(&gate.go, rc ? -1 : 1, __ATOMIC_SEQ_CST);
    for (i = 0; i < n_threads; i++) {
        pthread_join(th[i], 0);
        if (jobs[i].rc) rc = jobs[i].rc;
        frames += jobs[i].frames;
        if (i == 0 || jobs[i].t_start < t_start) t_start = jobs[i].t_start;
        if (jobs[i].t_end > t_end) t_end = jobs[i].t_end;
    }
    free(th);
    free(jobs);
    if (frames_out) *frames_out = frames;
    if (elapsed_out) *elapsed_out = t_end - t_start;
    return rc;
}
