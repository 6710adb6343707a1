/* lc3gpu -- C ABI of the MI355X-native batched LC3 codec (liblc3gpu.so).
 *
 * Drop-in boundary for the per-frame hot path of ninjasource/lc3-codec v0.2.0:
 *   Lc3Encoder::{calc_working_buffer_lengths,new,encode_frame}   reference src/encoder/lc3_encoder.rs:117-209
 *   Lc3Decoder::{calc_working_buffer_lengths,new,decode_frame}   reference src/decoder/lc3_decoder.rs:181-244
 * A handle owns N independent codec channels ("streams") on one HIP device; each stream is one
 * reference EncoderChannel / DecoderChannel with its own carried state.  The batch calls process
 * `n_streams x n_frames` frames per launch: one CDNA4 wavefront per stream, frames of a stream in
 * time order.  The *_frame calls are the n_streams = 1, n_frames = 1 case with host buffers and
 * have the reference's argument meaning (slice lengths select the frame size / bitrate).
 *
 * Plain C types only (pointers + sizes); no torch / C++ types cross this boundary.
 * All functions return LC3GPU_OK (0) or a negative LC3GPU_E* code; nothing aborts across the ABI:
 * where the reference panics (bad channel index, wrong slice length: lc3_encoder.rs:181-190,
 * encoder/modified_dct.rs:109-111) an error code is returned instead.  Corrupt frames are NOT
 * errors: as in the reference (lc3_decoder.rs:138-141) they are concealed (PLC) and counted.
 * Handles are not thread-safe (mirrors `&mut self`); distinct handles may be used concurrently.
 * A handle is bound to the HIP device that was current when it was created; every call switches to that device for its
 * duration.  Batch calls are asynchronous on the HIP stream they are given; a handle's launches share its scratch planes, so
 * a call on another stream than the handle's previous one is ordered after it (an event wait on the GPU, no host
 * synchronisation).  State load / reset calls wait for the handle's work in flight.
 */
#ifndef LC3GPU_H_
#define LC3GPU_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC3GPU_OK 0
#define LC3GPU_EINVAL -1        /* bad argument (unsupported fs / duration, null pointer, size mismatch) */
#define LC3GPU_ECHANNEL -2      /* channel index out of range (reference: panic, lc3_encoder.rs:185-189) */
#define LC3GPU_ELENGTH -3       /* samples / buffer length mismatch (reference: assert_eq! panic); also: a frame size outside
                                   20 ... 400 bytes on the encoder (1 ... 400 on the decoder) -- a deliberate deviation: the reference's
                                   encode_frame takes any buf_out.len() (src/encoder/lc3_encoder.rs:65), LC3 defines this range */
#define LC3GPU_EBITS -4         /* Lc3DecoderError::Only16BitsPerAudioSampleSupported (lc3_decoder.rs:80-82) */
#define LC3GPU_EHIP -5          /* HIP runtime error; see lc3gpu_last_hip_error() */
#define LC3GPU_ENODEVICE -6     /* no usable HIP device */
#define LC3GPU_EUNSUPPORTED -7  /* configuration the reference cannot run (8 kHz encode: bandwidth_detector.rs:36-37) */
#define LC3GPU_EPAIR -8         /* a producer / consumer wave pair of an EARLIER batch call of this handle gave up on its partner (see
                                   lc3gpu_*_pair_timeouts): that call's frames are zero-filled (encoder) or concealed (decoder).  Returned
                                   once, by the first batch call that notices (read from pinned host memory: no synchronisation); the call
                                   that returns it has launched nothing and may be repeated */

/* Opt-in corrections of the reference's deviations from the LC3 specification, one bit each, for interop with
 * other LC3 codecs.  Default 0: every deviation is reproduced, and only then do the bit-exactness claims against the
 * reference hold (with a bit set there is no reference behaviour; the CPU oracle implements the same switches). */
#define LC3GPU_SPEC_8KHZ_ENCODE 1     /* 8 kHz encoders can be created (reference: constructor panics, bandwidth_detector.rs:36-37) */
#define LC3GPU_SPEC_TNS_SSWB_STOP 2   /* 10 ms, bandwidth index 2: TNS filters lines 12..240 (reference: ..200, temporal_noise_shaping.rs:134-140) */
#define LC3GPU_SPEC_BW_CUTOFF_DB 4    /* bandwidth cut-off test on 10 log10(eps + ratio) (reference: raw ratio, bandwidth_detector.rs:106-115) */
#define LC3GPU_SPEC_SNS_LAST_GAIN 8   /* SNS gain search includes the last gain of every shape (reference: spectral_noise_shaping.rs:495-503) */
#define LC3GPU_SPEC_NBITS_SPEC_OLD 16 /* nbits_spec_old is updated (reference: stays 0, spectral_quantization.rs:59,97-100) */
#define LC3GPU_SPEC_ALL 31

typedef struct lc3gpu_encoder lc3gpu_encoder;
typedef struct lc3gpu_decoder lc3gpu_decoder;

/* Buffer layouts of the batch calls (C = the handle's channel count, T = n_frames):
 *   PLANAR       int16[C][T][nf], uint8[C][T][nbytes], flags uint8[C][T]: what the reference's per-channel calls see after
 *                the caller has de-interleaved (examples/encode.rs:95-102)
 *   INTERLEAVED  int16[T][nf][C], uint8[T][C][nbytes], flags uint8[T][C]: the WAV sample order and the .lc3 file order
 *                (frames in time order, channels inside a frame: examples/encode.rs:105-115, examples/decode.rs:86-112); the
 *                kernels de-interleave on load and interleave on store */
#define LC3GPU_LAYOUT_PLANAR 0
#define LC3GPU_LAYOUT_INTERLEAVED 1

/* One stream of a mixed-configuration handle: the reference builds one Lc3Encoder / Lc3Decoder per configuration
 * (lc3_encoder.rs:117-124, common/config.rs:42-100) and takes the frame size from the slice length of every call
 * (lc3_encoder.rs:65); a mixed handle fixes all three per stream so that ONE launch per kernel serves every stream. */
typedef struct lc3gpu_stream_desc {
    int fs_hz;    /* 8000 (decoder only), 16000, 24000, 32000, 44100, 48000 */
    int frame_us; /* 7500 or 10000 */
    int nbytes;   /* bytes per frame */
} lc3gpu_stream_desc;

/* library / device */
int lc3gpu_version(void);
const char *lc3gpu_strerror(int code);
int lc3gpu_last_hip_error(void);
int lc3gpu_device_count(void);

/* common/config.rs:42-100 -- out[7] = {fs_ind, fs, ne, n_ms_is_10, nb, nf, z} */
int lc3gpu_config(int frame_us, int fs_hz, int out[7]);

/* Lc3Encoder::calc_working_buffer_lengths (lc3_encoder.rs:194-209): out = {i16_len, f32_len, complex_len}.
 * Kept for API parity; the GPU engine allocates its own device memory. */
int lc3gpu_encoder_working_buffer_lengths(int num_channels, int frame_us, int fs_hz, int64_t out[3]);
/* Lc3Decoder::calc_working_buffer_lengths (lc3_decoder.rs:236-244): out = {f32_len, complex_len} */
int lc3gpu_decoder_working_buffer_lengths(int num_channels, int frame_us, int fs_hz, int64_t out[2]);

/* ---- encoder ------------------------------------------------------------------------------------ */
/* Lc3Encoder::new (lc3_encoder.rs:117-173): num_channels fresh channels on the current HIP device. */
int lc3gpu_encoder_create(lc3gpu_encoder **out, int num_channels, int frame_us, int fs_hz);
/* the same with LC3GPU_SPEC_* corrections switched on (spec_flags = 0: identical to lc3gpu_encoder_create) */
int lc3gpu_encoder_create_spec(lc3gpu_encoder **out, int num_channels, int frame_us, int fs_hz, int spec_flags);
int lc3gpu_encoder_destroy(lc3gpu_encoder *enc);
/* back to the freshly constructed state (all channels).  Costs no synchronisation: work in flight completes as it is, the NEXT call starts
 * every channel from the constructed state (lc3gpu_decoder_reset likewise) */
int lc3gpu_encoder_reset(lc3gpu_encoder *enc);

/* Lc3Encoder::encode_frame (lc3_encoder.rs:175-191), host buffers.
 * samples_in: n_samples (must be nf) planar i16; buf_out: nbytes (= buf_out.len(), selects the bitrate). */
int lc3gpu_encode_frame(lc3gpu_encoder *enc, int channel_index, const int16_t *samples_in, int n_samples,
                        uint8_t *buf_out, int nbytes);

/* Batch: every channel encodes n_frames consecutive frames.  DEVICE pointers, stream-major:
 *   d_pcm  int16[num_channels][n_frames][nf]      (4-byte aligned)
 *   d_out  uint8[num_channels][n_frames][nbytes]
 * Asynchronous on hip_stream (a hipStream_t, may be NULL for the default stream). */
int lc3gpu_encode(lc3gpu_encoder *enc, const int16_t *d_pcm, uint8_t *d_out, int nbytes, int n_frames,
                  void *hip_stream);
/* same, restricted to channels [first_channel, first_channel + n_channels); buffers hold only those channels */
int lc3gpu_encode_range(lc3gpu_encoder *enc, int first_channel, int n_channels, const int16_t *d_pcm,
                        uint8_t *d_out, int nbytes, int n_frames, void *hip_stream);
/* same as lc3gpu_encode with the buffers in `layout` (LC3GPU_LAYOUT_*); interleaved PCM needs 2-byte alignment only */
int lc3gpu_encode_layout(lc3gpu_encoder *enc, int layout, const int16_t *d_pcm, uint8_t *d_out, int nbytes, int n_frames,
                         void *hip_stream);

/* Mixed-configuration encoder: n_streams streams, each with its own rate, frame duration and frame size; 8 kHz streams are
 * refused (LC3GPU_EUNSUPPORTED, as lc3gpu_encoder_create).  lc3gpu_encode_mixed encodes n_frames frames of every stream
 * with one launch per kernel.  Ragged DEVICE buffers, streams in descriptor order, each stream planar:
 *   d_pcm  stream i at element offset n_frames * sum_{j<i} nf_j      (int16[n_frames][nf_i]; 4-byte aligned base)
 *   d_out  stream i at byte offset    n_frames * sum_{j<i} nbytes_j  (uint8[n_frames][nbytes_i])
 * The per-frame calls, state blobs and timing work on a mixed handle as on a uniform one (channel index = descriptor
 * index); lc3gpu_encode / _range / _layout do not (LC3GPU_EINVAL). */
int lc3gpu_encoder_create_mixed(lc3gpu_encoder **out, int n_streams, const lc3gpu_stream_desc *descs);
int lc3gpu_encoder_create_mixed_spec(lc3gpu_encoder **out, int n_streams, const lc3gpu_stream_desc *descs, int spec_flags);
int lc3gpu_encode_mixed(lc3gpu_encoder *enc, const int16_t *d_pcm, uint8_t *d_out, int n_frames, void *hip_stream);

/* per-channel state blobs (checkpoint / CPU cross-checks): size per channel, device->host copy, host->device.
 * nbytes must equal state_size * num_channels (LC3GPU_ELENGTH otherwise); both calls synchronise the device.  A channel's blob
 * starts with a 16-byte header {magic "LC3E" / "LC3D", layout version, payload size, fs_hz, frame duration, spec_flags}: loading
 * a blob saved by a handle of another configuration, another stream-descriptor order, the other side or another library version
 * returns LC3GPU_EINVAL instead of decoding garbage. */
size_t lc3gpu_encoder_state_size(const lc3gpu_encoder *enc);
int lc3gpu_encoder_state_save(lc3gpu_encoder *enc, void *host_dst, size_t nbytes);
int lc3gpu_encoder_state_load(lc3gpu_encoder *enc, const void *host_src, size_t nbytes);

/* ---- decoder ------------------------------------------------------------------------------------ */
/* Lc3Decoder::new (lc3_decoder.rs:181-215) */
int lc3gpu_decoder_create(lc3gpu_decoder **out, int num_channels, int frame_us, int fs_hz);
int lc3gpu_decoder_destroy(lc3gpu_decoder *dec);
int lc3gpu_decoder_reset(lc3gpu_decoder *dec);

/* Lc3Decoder::decode_frame (lc3_decoder.rs:217-234), host buffers.  num_bits_per_audio_sample must be 16. */
int lc3gpu_decode_frame(lc3gpu_decoder *dec, int num_bits_per_audio_sample, int channel_index,
                        const uint8_t *buf_in, int nbytes, int16_t *samples_out, int n_samples);

/* Batch decode, DEVICE pointers, stream-major:
 *   d_in   uint8[num_channels][n_frames][nbytes]
 *   d_pcm  int16[num_channels][n_frames][nf]      (4-byte aligned)
 *   d_bad_frame  optional uint8[num_channels][n_frames]: non-zero marks a lost frame -> concealment
 *                (external bad-frame indicator; NULL = none) */
int lc3gpu_decode(lc3gpu_decoder *dec, const uint8_t *d_in, const uint8_t *d_bad_frame, int16_t *d_pcm, int nbytes,
                  int n_frames, void *hip_stream);
int lc3gpu_decode_range(lc3gpu_decoder *dec, int first_channel, int n_channels, const uint8_t *d_in,
                        const uint8_t *d_bad_frame, int16_t *d_pcm, int nbytes, int n_frames, void *hip_stream);
/* same as lc3gpu_decode with the buffers (and the flag array) in `layout` (LC3GPU_LAYOUT_*) */
int lc3gpu_decode_layout(lc3gpu_decoder *dec, int layout, const uint8_t *d_in, const uint8_t *d_bad_frame, int16_t *d_pcm,
                         int nbytes, int n_frames, void *hip_stream);

/* Mixed-configuration decoder (see lc3gpu_encoder_create_mixed; 8 kHz streams are allowed).  Ragged buffers as there:
 * d_in like the encoder's d_out, d_pcm like its d_pcm; d_bad_frame (optional) uint8[n_streams][n_frames] in descriptor order. */
int lc3gpu_decoder_create_mixed(lc3gpu_decoder **out, int n_streams, const lc3gpu_stream_desc *descs);
int lc3gpu_decode_mixed(lc3gpu_decoder *dec, const uint8_t *d_in, const uint8_t *d_bad_frame, int16_t *d_pcm, int n_frames,
                        void *hip_stream);

size_t lc3gpu_decoder_state_size(const lc3gpu_decoder *dec);
int lc3gpu_decoder_state_save(lc3gpu_decoder *dec, void *host_dst, size_t nbytes);
int lc3gpu_decoder_state_load(lc3gpu_decoder *dec, const void *host_src, size_t nbytes);
/* total number of frames concealed so far over all channels (synchronises the device) */
int lc3gpu_decoder_plc_events(lc3gpu_decoder *dec, uint64_t *out);
/* Full batches run the bit packer and the bitstream parser as producer / consumer pairs of wavefronts.  A half that waits 2^24 polls for
 * its partner gives up (a partner that died: never seen): a parser pair then conceals its frames as the reference conceals a frame whose
 * read_frame failed (decoder/lc3_decoder.rs:138-141; they count as PLC events too), a packer pair leaves its frames ZERO-FILLED -- the
 * reference has no such case (`Lc3EncoderError` is empty, encoder/lc3_encoder.rs:29-30), so it is made visible here: *out = the number of
 * pair halves that ever gave up on this handle (sticky; 0 in every run so far).  Waits for the handle's work in flight.
 * A caller need not poll: the handle's next batch call returns LC3GPU_EPAIR (once) when a pair half of an earlier call gave up. */
int lc3gpu_encoder_pair_timeouts(lc3gpu_encoder *enc, uint64_t *out);
int lc3gpu_decoder_pair_timeouts(lc3gpu_decoder *dec, uint64_t *out);
/* tests only: make the device do what a pair half that gives up does (count + host flag), so that the LC3GPU_EPAIR path can be exercised */
int lc3gpu_encoder_debug_pair_giveup(lc3gpu_encoder *enc);
int lc3gpu_decoder_debug_pair_giveup(lc3gpu_decoder *dec);

/* A caller that runs a handle on ONE HIP stream for the handle's whole life (the pipeline object does) can say so: bind = 1 binds the
 * handle to `hip_stream` -- which must outlive the handle or the binding --, bind = 0 releases it.  A bound handle takes batch calls on
 * that stream only (LC3GPU_EINVAL otherwise; the *_frame, *_host and diagnostic calls are not for bound handles) and records no event of
 * its own per call: "the handle's work in flight" is the stream.  Both calls wait for the handle's work in flight. */
int lc3gpu_encoder_bind_stream(lc3gpu_encoder *enc, void *hip_stream, int bind);
int lc3gpu_decoder_bind_stream(lc3gpu_decoder *dec, void *hip_stream, int bind);

/* ---- host-resident batches ----------------------------------------------------------------------- */
/* The reference's callers keep PCM and frame bytes in HOST memory and walk them frame by frame, channel by channel
 * (examples/encode.rs:73-116: read samples, de-interleave, encode_frame per channel, write; examples/decode.rs:60-112 the mirror).
 * These two calls take such buffers whole -- planar, as the batch calls: pcm int16[num_channels][n_frames][nf], bytes
 * uint8[num_channels][n_frames][nbytes], bad_frame (optional) uint8[num_channels][n_frames] -- and return when the results are in host
 * memory: the handle's channels go through the device in ranges of >= 32 768 frames, the copy-in of one range, the kernels of another and
 * the copy-out of a third at the same time (three internal HIP streams).  State is carried as by the batch calls.  The PCIe link bounds them (960 + 150 bytes per 48 kHz / 10 ms frame each way);
 * buffers from lc3gpu_host_alloc (pinned) copy at the link's rate, any other host memory through the runtime's staging copies. */
int lc3gpu_encode_host(lc3gpu_encoder *enc, const int16_t *pcm, uint8_t *out, int nbytes, int n_frames);
int lc3gpu_decode_host(lc3gpu_decoder *dec, const uint8_t *in, const uint8_t *bad_frame, int16_t *pcm, int nbytes, int n_frames);
int lc3gpu_host_alloc(void **out, size_t nbytes);
int lc3gpu_host_free(void *p);

/* ---- pipeline: the caller loop as an object -------------------------------------------------------- */
/* The reference's caller owns the loop "for every frame, for every channel: encode_frame" (examples/encode.rs:97-115) and its mirror
 * (examples/decode.rs:93-112).  On the GPU the arrangement of that loop decides a fifth of the throughput (kernels of different calls share
 * the chip): a pipeline owns the arrangement that measured best -- the channels in `n_groups` groups (0 = the default,
 * two), every group with an encoder handle on a HIP stream of the higher priority and a decoder handle on a stream of the default
 * priority, two submissions in flight per group, the groups never joining -- so that a C / Rust caller gets it from three calls.
 * Buffers: DEVICE pointers, planar like lc3gpu_encode / lc3gpu_decode (int16[num_channels][n_frames][nf], uint8[num_channels][n_frames][nbytes]).
 * Every call is asynchronous on the pipeline's own streams:
 *   lc3gpu_pipeline_submit   one round trip: encode d_pcm -> d_bytes, decode d_bytes -> d_pcm_out (the decoder of a group starts when its
 *                            encoder has finished, and the encoder of the NEXT submission runs beside it).  A submission's encoder waits
 *                            for the decoders of the two submissions before it only where it would overwrite bytes they still read:
 *                            alternate two byte buffers and nothing waits.
 *   lc3gpu_pipeline_encode / lc3gpu_pipeline_decode   the halves alone (d_bad_frame as for lc3gpu_decode, may be NULL)
 *   lc3gpu_pipeline_wait     the host waits for everything submitted
 *   lc3gpu_pipeline_join     `hip_stream` (the caller's) waits for everything submitted -- results may be consumed on that stream
 *   lc3gpu_pipeline_follow   the NEXT submission waits for what `hip_stream` holds now (e.g. the kernel that produces d_pcm)
 *   lc3gpu_pipeline_mark     records `hip_event` (a hipEvent_t of the caller's) on the LAST group's stream behind its latest work: a
 *                            progress mark / timing point that costs the pipeline nothing -- `join` puts event waits on a further stream, and
 *                            a waiting stream can hold up a pipeline stream that shares its hardware queue (measured: -10 % with a join
 *                            after every submission).  The groups are not tied to each other: the mark says nothing about the other groups
 *   lc3gpu_pipeline_group    the channel range and the handles of a group (borrowed: state blobs, PLC / health counters, timing,
 *                            stage events; never destroy them, never call their batch functions while the pipeline has work in flight)
 *   lc3gpu_pipeline_reset    every channel back to the freshly constructed state from the next submission on (no wait)
 * Errors as the batch calls (LC3GPU_EPAIR included).
 * Mixed configurations (lc3gpu_pipeline_create_mixed, BASELINE config 4 through the pipeline): streams of different rates, durations and frame
 * sizes, descriptors and ragged buffers as for lc3gpu_*_create_mixed / lc3gpu_encode_mixed (8 kHz streams are refused: a pipeline encodes and
 * decodes); group g takes the descriptors [group_first[g], group_first[g + 1]) (group_first[0] = 0, ascending, n_groups entries; NULL: equal
 * shares of the list) -- order the list so that every group holds a share of every configuration.  The *_mixed calls take the place of
 * submit / encode / decode (LC3GPU_EINVAL for the other kind); everything else is common. */
typedef struct lc3gpu_pipeline lc3gpu_pipeline;
int lc3gpu_pipeline_create(lc3gpu_pipeline **out, int num_channels, int frame_us, int fs_hz, int n_groups);
int lc3gpu_pipeline_create_mixed(lc3gpu_pipeline **out, int n_streams, const lc3gpu_stream_desc *descs, int n_groups, const int *group_first);
int lc3gpu_pipeline_submit_mixed(lc3gpu_pipeline *p, const int16_t *d_pcm, uint8_t *d_bytes, int16_t *d_pcm_out, int n_frames);
int lc3gpu_pipeline_encode_mixed(lc3gpu_pipeline *p, const int16_t *d_pcm, uint8_t *d_bytes, int n_frames);
int lc3gpu_pipeline_decode_mixed(lc3gpu_pipeline *p, const uint8_t *d_bytes, const uint8_t *d_bad_frame, int16_t *d_pcm_out, int n_frames);
int lc3gpu_pipeline_destroy(lc3gpu_pipeline *p);
int lc3gpu_pipeline_reset(lc3gpu_pipeline *p);
int lc3gpu_pipeline_submit(lc3gpu_pipeline *p, const int16_t *d_pcm, uint8_t *d_bytes, int16_t *d_pcm_out, int nbytes, int n_frames);
int lc3gpu_pipeline_encode(lc3gpu_pipeline *p, const int16_t *d_pcm, uint8_t *d_bytes, int nbytes, int n_frames);
int lc3gpu_pipeline_decode(lc3gpu_pipeline *p, const uint8_t *d_bytes, const uint8_t *d_bad_frame, int16_t *d_pcm_out, int nbytes, int n_frames);
int lc3gpu_pipeline_wait(lc3gpu_pipeline *p);
int lc3gpu_pipeline_join(lc3gpu_pipeline *p, void *hip_stream);
int lc3gpu_pipeline_follow(lc3gpu_pipeline *p, void *hip_stream);
int lc3gpu_pipeline_mark(lc3gpu_pipeline *p, void *hip_event);
int lc3gpu_pipeline_groups(const lc3gpu_pipeline *p);
int lc3gpu_pipeline_group(lc3gpu_pipeline *p, int group, int *first_channel, int *n_channels, lc3gpu_encoder **enc, lc3gpu_decoder **dec);
int lc3gpu_pipeline_last_hip_error(const lc3gpu_pipeline *p);

/* ---- diagnostics -------------------------------------------------------------------------------- */
/* encode one frame of channel 0 from host PCM (the frame IS a frame of that channel: its state advances) and also return stage dumps,
 * so that an encoder stage can be checked against the reference's own stage vectors (encoder/modified_dct.rs:191-337,
 * attack_detector.rs:138-180, spectral_noise_shaping.rs:658-801, temporal_noise_shaping.rs:359-471, long_term_post_filter.rs:479-843,
 * spectral_quantization.rs:404-480, noise_level_estimation.rs:65-137):
 * dbg float[LC3GPU_ENC_DBG_FLOATS] = spectrum after MDCT [0,480), after SNS [480,960), after TNS [960,1440); scalars from
 * LC3GPU_ENC_DBG_SCALARS: +0 bandwidth index, +1 attack flag, +2 ind_lf, +3 ind_hf, +4 shape_j, +5 gind, +6 / +7 TNS orders, +8 nbits_tns,
 * +9 pitch_index, +10 pitch_present, +11 ltpf_active, +12 gg_ind, +13 lastnz_trunc, +14 nbits_lsb, +15 lsb_mode, +16 residual bits,
 * +17 noise factor, +18 global gain, +19 nbits_spec, +20 nbits_trunc, +21 near-Nyquist flag, +22 ls_inda, +23 / +24 index_joint_j low /
 * high 16 bits; the 64 band energies from LC3GPU_ENC_DBG_EB; the attack detector's state after the frame from LC3GPU_ENC_DBG_ATTACK
 * (energy_last, max_energy_last, attack_pos_last, downsampled sample t-1, t-2) */
#define LC3GPU_ENC_DBG_SCALARS 1440
#define LC3GPU_ENC_DBG_EB 1472
#define LC3GPU_ENC_DBG_ATTACK 1536
#define LC3GPU_ENC_DBG_FLOATS 1600
int lc3gpu_encode_frame_debug(lc3gpu_encoder *enc, const int16_t *samples_in, int n_samples, uint8_t *buf_out,
                              int nbytes, float *dbg);
/* decode one frame of channel 0 (uniform handles) from host bytes like lc3gpu_decode_frame and also return stage dumps, so that a
 * decoder stage can be checked on its own (the reference tests every one: decoder/arithmetic_codec.rs:415-474,
 * residual_spectrum.rs:47-107, noise_filling.rs:65-146, temporal_noise_shaping.rs:147-238, spectral_noise_shaping.rs:244-350,
 * modified_dct.rs:174-329, long_term_post_filter.rs:504-1199).  dbg float[LC3GPU_DBG_FLOATS], NaN where a form has no such value:
 *   [LC3GPU_DBG_INT, +ne)    integers after the range decoder            [LC3GPU_DBG_GAIN, +ne)  after residual bits, noise filling, global gain
 *   [LC3GPU_DBG_TNS, +ne)    after the TNS filter                        [LC3GPU_DBG_SPEC, +ne)  after the SNS band gains (input of the IMDCT)
 *   [LC3GPU_DBG_IMDCT, +nf)  after IMDCT, window, overlap-add            [LC3GPU_DBG_LTPF, +nf)  after the long-term post-filter
 * recon_form selects which of the library's three forms of the spectrum reconstruction runs: LC3GPU_RECON_LANE (in the parse
 * kernel, what full batches use), LC3GPU_RECON_LATE (in the synthesis kernel, what lc3gpu_decode_frame and small launches use),
 * LC3GPU_RECON_WAVE (the wave-per-frame kernels; no GAIN / TNS dumps). */
#define LC3GPU_DBG_INT 0
#define LC3GPU_DBG_SPEC 400
#define LC3GPU_DBG_IMDCT 800
#define LC3GPU_DBG_LTPF 1280
#define LC3GPU_DBG_GAIN 1760
#define LC3GPU_DBG_TNS 2160
#define LC3GPU_DBG_FLOATS 2560
#define LC3GPU_RECON_LANE 0
#define LC3GPU_RECON_LATE 1
#define LC3GPU_RECON_WAVE 2
int lc3gpu_decode_frame_debug(lc3gpu_decoder *dec, int recon_form, const uint8_t *buf_in, int nbytes, int16_t *samples_out, int n_samples,
                              float *dbg);
/* the synthesis half alone on channel 0: `in` is a reconstructed spectrum (time_in = 0, n_in = ne: IMDCT -> LTPF -> PCM) or the
 * time samples that enter the long-term post-filter (time_in = 1, n_in = nf: LTPF -> PCM), with the frame's post-filter side
 * information and its size (nbits = 8 * nbytes selects the filter gain; pitch_index 0 .. 511 as in the bitstream, LC3GPU_EINVAL otherwise).
 * dbg as above (IMDCT and LTPF dumps).  Like lc3gpu_decode_frame_debug this ADVANCES channel 0's state (overlap memory, post-filter
 * memories): a diagnostic call is a frame of that channel. */
int lc3gpu_decoder_synth_debug(lc3gpu_decoder *dec, int time_in, const float *in, int n_in, int ltpf_active, int pitch_index, int nbytes,
                               int16_t *samples_out, int n_samples, float *dbg);
/* tests only: the float routines the codec's bit-exactness rests on, evaluated on the device as the kernels compile them.
 * which: 0 x / d by the quantiser's reciprocal sequence, 1 x / d by the compiler's IEEE division, 2 log2f, 3 log10f, 4 exp2f, 5 asinf,
 * 6 fast-math exp2_raw, 7 10^x, 8 sinf (|x| small), 9 the device-filled tables (n = 866: 512 gains 10^(k/28), k = -256..255, 320 tilt
 * factors, 17 + 17 TNS sines).  x, d, out: HOST arrays of n floats (d only for 0 and 1, neither for 9). */
int lc3gpu_selftest_math(int which, const float *x, const float *d, int n, float *out);
/* per-kernel timing of the batch calls with HIP events recorded on the launch stream.  An encoder batch call runs four
 * kernels: analysis front half (wave per stream), SNS vector quantiser (lane per frame), analysis back half (wave per
 * stream), bitstream packing (lane per frame); a decoder batch call two: frame parsing with the spectrum reconstruction on the
 * lane that parsed the frame (full batches; launches of a few frames reconstruct inside the synthesis kernel instead), synthesis
 * (wave per stream) -- or four when LC3GPU_RECON=wave selects the measured-but-not-default form with two reconstruction kernels of
 * their own (wave per frame and, for the TNS lattice, lane per frame).  With LC3GPU_SPLIT=1 (opt-in; measured slower) a call runs as
 * two halves of its streams on two internal HIP streams: a kernel's figure is then the sum over both launches.  `enable` = 0 switches recording off, 1 on for every batch call,
 * n > 1 on for every n-th batch call from now on (an event after every kernel costs the stream a few microseconds: sampling keeps a
 * long timed run undisturbed); the call synchronises and returns the per-kernel milliseconds accumulated since the previous call
 * followed by the number of batch calls that were timed:
 * encoder out[5] = {front, vq, back, pack, calls}, decoder out[3] = {parse + reconstruction, synthesis, calls},
 * lc3gpu_decoder_timing_kernels out[5] = {parse, reconstruction kernel, TNS kernel (both 0 where the launch has none), synthesis, calls}. */
int lc3gpu_encoder_timing(lc3gpu_encoder *enc, int enable, double out[5]);
int lc3gpu_decoder_timing(lc3gpu_decoder *dec, int enable, double out[3]);
int lc3gpu_decoder_timing_kernels(lc3gpu_decoder *dec, int enable, double out[5]);
/* Stage events (an extension without a counterpart in the reference: its caller loop, examples/encode.rs:97-115 / examples/decode.rs:93-112,
 * has no stages to observe).  `hip_event` is a hipEvent_t of the CALLER (NULL clears the slot); from then on every batch call of the handle
 * records it on the call's stream right behind the kernel(s) of the stage, so that a caller who runs several handles on several HIP streams
 * can make another stream wait (hipStreamWaitEvent) for a chosen point INSIDE this handle's call -- e.g. start a decoder's parser when the
 * encoder's back half has ended, beside the encoder's packer (both leave most of the chip's workgroup slots free) instead of beside its
 * front half (which fills them): bench.py's `staggered` arrangement, INTEGRATION.md section 4.  The event completes no earlier than the
 * stage; a call that runs as two halves (LC3GPU_SPLIT=1) records every stage event at its end.  The event must stay valid while it is
 * set.  LC3GPU_EINVAL for an unknown stage. */
#define LC3GPU_MAX_STAGES 4
#define LC3GPU_ENC_STAGE_FRONT 0 /* analysis front half done */
#define LC3GPU_ENC_STAGE_VQ 1    /* ... and the SNS vector quantiser */
#define LC3GPU_ENC_STAGE_BACK 2  /* ... and the back half: only the packer is left */
#define LC3GPU_DEC_STAGE_PARSE 0 /* frames parsed and spectra rebuilt: only the synthesis kernel is left */
int lc3gpu_encoder_stage_event(lc3gpu_encoder *enc, int stage, void *hip_event);
int lc3gpu_decoder_stage_event(lc3gpu_decoder *dec, int stage, void *hip_event);
/* diagnostic build (liblc3gpu_prof.so, -DLC3_PROFILE) only: per-stage shader-clock cycle sums since the last call.
 * slots 1..9 = encoder stages (mdct, bw+attack, sns, tns, ltpf, quant, residual+noise, bitstream, store),
 * slots 17..25 = decoder stages (names in tools/stage_profile.py); 32/33/34 = encoder whole-wave time sum / max / waves,
 * 35/36/37 the same for the decoder, 40..46 = sections of the parse kernel (side info, TNS data, spectral data,
 * zero fill + residual bits, reconstruction set-up, reconstruction pass, rest), 48..54 = sections of the pack kernel
 * (staging, side info, TNS data, spectral data, residual bits, finish, barrier wait + copy-out), 55 = its waves.
 * LC3GPU_EUNSUPPORTED in the normal build. */
int lc3gpu_prof_read(unsigned long long out[64]);
/* measurement aid (bench.py's sustained leg): one wave on `stream` stamps the shader-cycle counter and the constant 100 MHz counter around
 * `spin` dependent vector additions; asynchronous, d_out (DEVICE memory, 3 x uint64) <- {shader cycles, 100 MHz ticks, unused}.  The clock the
 * chip runs at while the probe is in flight = 100 MHz x cycles / ticks (launch it on a stream beside the codec's). */
int lc3gpu_clock_probe(void *stream, unsigned long long *d_out, int spin);
/* kernel resource report as the loaded code object has it: out = {static lds_bytes, vgprs, 0, scratch_bytes, max_threads} for the six kernels
 * a full batch of the headline configuration launches: which = 0 analysis back half, 1 synthesis (the numbers these two had when they
 * were the only ones), 2 analysis front half, 3 SNS vector quantiser, 4 packer (pair form), 5 parser (pair form); LC3GPU_EINVAL beyond.
 * (tests/test_kernel_resources.py reads the same, and the spill counts, from the built library's metadata without a GPU and holds them
 * to the register budgets the kernels are tuned for.) */
int lc3gpu_kernel_info(int which, int out[5]);

#ifdef __cplusplus
}
#endif
#endif /* LC3GPU_H_ */
