/* ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See lc3_oracle.h.
 * Batch drivers used by the parity tests and by bench.py's cpu_baseline leg:
 * every stream is an independent codec channel (its own encoder/decoder object,
 * exactly like one reference `Lc3Encoder` channel fed frame after frame,
 * examples/encode.rs:105-115).  Streams are distributed over host threads. */
#include "lc3_oracle.h"
#include <pthread.h>
#include <stdlib.h>

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
