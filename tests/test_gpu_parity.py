"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, called through the C ABI,
against the CPU oracle and the reference's end-to-end known-answer vectors.

Bar (BASELINE.json north_star): arithmetic-coded bitstream BYTE-EXACT; decoded i16 PCM within
+-1 LSB (PCM_TOL below; the engine is expected to be exact, max|diff| is asserted == 0 where the
oracle and the engine share every arithmetic step)."""
import importlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
PCM_TOL = 1  # LSB, the tolerance the north star states for decode

pkg = importlib.import_module("lc3-codec_amd")
synth = importlib.import_module("lc3-codec_amd.synth")
FS, US = pkg.SamplingFrequency.Hz48000, pkg.FrameDuration.TenMs


def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU test needs a HIP device"
    return torch


def gpu_encode(pcm, nbytes, fs=48000, us=10000, enc=None):
    torch = torch_mod()
    S, T, nf = pcm.shape
    enc = enc or pkg.Lc3Encoder(S, us, fs)
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    d_out = torch.zeros((S, T, nbytes), dtype=torch.uint8, device="cuda")
    enc.encode(d_pcm, d_out, nbytes, T, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert enc.pair_timeouts() == 0, "a packer producer / consumer pair gave up on its partner"
    return d_out.cpu().numpy()


def gpu_decode(data, nf, fs=48000, us=10000, dec=None, bad=None):
    torch = torch_mod()
    S, T, nbytes = data.shape
    dec = dec or pkg.Lc3Decoder(S, us, fs)
    d_in = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    d_pcm = torch.zeros((S, T, nf), dtype=torch.int16, device="cuda")
    d_bad = torch.from_numpy(np.ascontiguousarray(bad)).cuda() if bad is not None else None
    dec.decode(d_in, d_pcm, nbytes, T, stream=torch.cuda.current_stream().cuda_stream, d_bad_frame=d_bad)
    torch.cuda.synchronize()
    assert dec.pair_timeouts() == 0, "a parser producer / consumer pair gave up on its partner"
    return d_pcm.cpu().numpy()


# ---------------------------------------------------------------- reference KATs through the drop-in API
def test_kat_encode_frame():  # encoder/lc3_encoder.rs:314-369
    t = "encoder/lc3_encoder.rs::lc3_encode_channel"
    enc = pkg.Lc3Encoder(1, US, FS)
    buf = np.zeros(150, np.uint8)
    enc.encode_frame(0, O.kat(t, "samples_in", 0, np.int16), buf)
    assert buf.tolist() == O.kat(t, "buf_out_expected")


def test_kat_decode_frame():  # decoder/lc3_decoder.rs:374-425
    t = "decoder/lc3_decoder.rs::lc3_decode_channel"
    dec = pkg.Lc3Decoder(1, US, FS)
    pcm = np.zeros(480, np.int16)
    dec.decode_frame(16, 0, np.array(O.kat(t, "buf_in"), np.uint8), pcm)
    assert pcm.tolist() == O.kat(t, "samples_out_expected")


def test_kat_stage_dumps():
    """stage-level localisation: spectrum after MDCT / SNS / TNS of the KAT frame vs the reference's stage goldens"""
    t = "encoder/lc3_encoder.rs::lc3_encode_channel"
    enc = pkg.Lc3Encoder(1, US, FS)
    out, dbg = enc.encode_frame_debug(O.kat(t, "samples_in", 0, np.int16), 150)
    sns_in = O.kat("encoder/spectral_noise_shaping.rs::sns_run", "x", 0, np.float32)
    sns_out = O.kat("encoder/spectral_noise_shaping.rs::sns_run", "x_s_expected", 0, np.float32)
    tns_out = O.kat("encoder/temporal_noise_shaping.rs::temporal_noise_shaping_run", "x_f_expected", 0, np.float32)
    assert np.array_equal(dbg[0:400], sns_in), "MDCT spectrum"
    assert np.array_equal(dbg[480:880], sns_out), "SNS-shaped spectrum"
    assert np.array_equal(dbg[960:1360], tns_out), "TNS-filtered spectrum"
    assert dbg[1452] == 193 and dbg[1453] == 350 and dbg[1454] == 107  # gg_ind, lastnz_trunc, nbits_lsb
    assert out.tolist() == O.kat(t, "buf_out_expected")


def test_encoder_stage_dumps_match_reference_goldens():
    """The encoder's stages ON THE DEVICE against the reference's own stage vectors (lc3gpu_encode_frame_debug), beyond the three
    spectra of test_kat_stage_dumps.  The end-to-end KAT frame is the frame several of the reference's per-stage tests were dumped
    from: its band energies are sns_run's / bandwidth_detector_run's `e_b`, it is attack_detector_run's and long_term_post_filter_run's
    `x_s`, its quantised spectrum is noise_level_estimation_run's `x_q`.  modified_dct_encode and long_term_post_filter_active bring
    their own frames (two, eight): they run through fresh encoders frame by frame."""
    import ctypes

    D = pkg.api
    S0 = D.ENC_DBG_SCALARS
    enc = pkg.Lc3Encoder(1, US, FS)
    t = "encoder/lc3_encoder.rs::lc3_encode_channel"
    out, dbg = enc.encode_frame_debug(O.kat(t, "samples_in", 0, np.int16), 150)
    assert out.tolist() == O.kat(t, "buf_out_expected")
    eb = dbg[D.ENC_DBG_EB:D.ENC_DBG_EB + 64]
    assert np.array_equal(eb, O.kat("encoder/spectral_noise_shaping.rs::sns_run", "e_b", 0, np.float32)), "band energies"
    assert np.array_equal(eb, O.kat("encoder/bandwidth_detector.rs::bandwidth_detector_run", "e_b", 0, np.float32))
    assert dbg[S0 + 0] == 4  # bandwidth_detector.rs:137-155 -> (4, 3)
    # attack_detector.rs:138-180: the flag and every state field
    att = dbg[D.ENC_DBG_ATTACK:D.ENC_DBG_ATTACK + 5]
    assert dbg[S0 + 1] == 1 and att[0] == np.float32(549861.5) and att[1] == np.float32(905588.875) and att[2:].tolist() == [0.0, 4846.0, 5210.0], att
    # spectral_noise_shaping.rs:780-801: ind_lf 8, ind_hf 17, shape_j 3, gind 0, index_joint_j 15253432 (ls_inda 0)
    joint = int(dbg[S0 + 23]) | (int(dbg[S0 + 24]) << 16)
    assert dbg[S0 + 2:S0 + 6].tolist() == [8.0, 17.0, 3.0, 0.0] and dbg[S0 + 22] == 0 and joint == 15253432, (dbg[S0 + 2:S0 + 6], joint)
    # long_term_post_filter.rs:479-520: nbits_ltpf 11 <=> pitch_present, ltpf_active false, pitch_index 0
    assert (dbg[S0 + 9], dbg[S0 + 10], dbg[S0 + 11]) == (0, 1, 0)
    # noise_level_estimation.rs:65-137 (p_bw 4, gg 24.709114) -> 6; spectral_quantization.rs:474: gg = 0x41C5AC44
    assert dbg[S0 + 17] == 6 and dbg[S0 + 18].view(np.uint32) == 0x41C5AC44
    # modified_dct.rs:191-337: the SECOND frame's spectrum and band energies (the time buffer carries the first), near-Nyquist flag false
    t = "encoder/modified_dct.rs::modified_dct_encode"
    enc = pkg.Lc3Encoder(1, US, FS)
    enc.encode_frame_debug(O.kat(t, "samples_in", 0, np.int16), 150)
    _, dbg = enc.encode_frame_debug(O.kat(t, "samples_in", 1, np.int16), 150)
    assert np.array_equal(dbg[0:480], O.kat(t, "output_expected", 0, np.float32)), "MDCT spectrum of the second frame"
    assert np.array_equal(dbg[D.ENC_DBG_EB:D.ENC_DBG_EB + 64], O.kat(t, "energy_bands_expected", 0, np.float32)), "its band energies"
    assert dbg[S0 + 21] == 0
    # long_term_post_filter.rs:523-843: eight consecutive frames at nbits = 400 (50-byte frames); (ltpf_active, pitch_present, pitch_index)
    # per frame :566-842 -- from the stage dump AND from the frame's bytes, read back with the oracle's side-information reader
    t = "encoder/long_term_post_filter.rs::long_term_post_filter_active"
    want = [(0, 0, 0), (0, 0, 0), (0, 1, 180), (0, 1, 184), (0, 1, 477), (0, 1, 478), (1, 1, 478), (1, 1, 478)]
    enc = pkg.Lc3Encoder(1, US, FS)
    oenc = O.Encoder()
    for i, w in enumerate(want):
        x = O.kat(t, "x_s", i, np.int16)
        out, dbg = enc.encode_frame_debug(x, 50)
        assert (int(dbg[S0 + 11]), int(dbg[S0 + 10]), int(dbg[S0 + 9])) == w, (i, dbg[S0 + 9:S0 + 12])
        si, tail = np.zeros(20, np.int64), ctypes.c_int(0)
        assert O.lib().lc3o_kat_side_info(O.P(out), 50, 4, 400, O.P(si), ctypes.byref(tail)) == 0
        assert (int(si[17]), int(si[16]), int(si[18])) == w, (i, si[16:19])
        assert np.array_equal(out, oenc.encode_frame(x, 50)), i


# ---------------------------------------------------------------- batch parity vs the oracle
def _roundtrip_check(fs, us, nbytes, S, T, seed=synth.SEED):
    cfg = pkg.Lc3Config(fs, us)
    pcm = synth.make_pcm(S, T, cfg.nf, fs, seed=seed)
    ref_bytes = O.encode_batch(pcm, nbytes, fs, us, threads=8)
    got_bytes = gpu_encode(pcm, nbytes, fs, us)
    bad = np.argwhere((got_bytes != ref_bytes).any(axis=2))
    assert len(bad) == 0, f"{len(bad)} frames differ, first (stream, frame) = {bad[0].tolist()}"
    ref_pcm = O.decode_batch(ref_bytes, cfg.nf, fs, us, threads=8)
    got_pcm = gpu_decode(ref_bytes, cfg.nf, fs, us)
    diff = np.abs(got_pcm.astype(np.int32) - ref_pcm.astype(np.int32)).max()
    assert diff <= PCM_TOL, f"PCM max |diff| = {diff}"
    assert diff == 0, f"PCM not exact: max |diff| = {diff}"



def _mixed_bandwidth_check(nbytes=150, S=80, T=6, seed=95):
    """A 48 kHz batch whose neighbouring streams stop at different cut-offs (3.5 / 7.5 / 11.5 / 15.5 kHz / full band), so that the lanes of
    one lane-per-frame wave (a wave holds 64 / T streams' frames) carry different bandwidth indices and TNS band layouts: the per-lane
    divergent paths of lc3_tns_lattice4 / lc3_tns_lane_frame.  Both directions against the oracle; returns the bandwidth indices met."""
    cuts = (3500.0, 7500.0, 11500.0, 15500.0, None)
    parts = [synth.make_bandlimited_pcm(S, T, 480, 48000, c, seed=seed + i) if c else synth.make_pcm(S, T, 480, 48000, seed=seed + i)
             for i, c in enumerate(cuts)]
    pcm = np.empty((S, T, 480), np.int16)
    for s in range(S):
        pcm[s] = parts[s % len(cuts)][s]
    ref_bytes = O.encode_batch(pcm, nbytes, 48000, 10000, threads=8)
    got_bytes = gpu_encode(pcm, nbytes, 48000, 10000)
    assert np.array_equal(got_bytes, ref_bytes)
    ref_pcm = O.decode_batch(ref_bytes, 480, 48000, 10000, threads=8)
    got_pcm = gpu_decode(ref_bytes, 480, 48000, 10000)
    assert np.array_equal(got_pcm, ref_pcm)
    # the bandwidth index is the last byte's low three bits at 48 kHz (write_bandwidth: bitstream_encoding.rs, read side side_info.rs)
    return sorted(set((ref_bytes[:, :, -1] & 7).reshape(-1).tolist()))


def test_lanes_of_a_wave_with_different_bandwidths():
    """default forms; the forced forms run the same check in test_late_reconstruction_on_and_off"""
    seen = _mixed_bandwidth_check()
    assert len(seen) >= 4, seen


def test_batch_48k_10ms_150B():  # BASELINE config shape, 2048 frames with carried state
    _roundtrip_check(48000, 10000, 150, 256, 8)


@pytest.mark.parametrize("fs,us,nbytes", [
    (48000, 10000, 60), (48000, 10000, 300), (48000, 10000, 400), (48000, 10000, 40), (48000, 7500, 113),
    (48000, 7500, 40), (44100, 10000, 110), (32000, 10000, 80), (32000, 10000, 120), (32000, 7500, 61),
    (24000, 10000, 60), (24000, 7500, 45), (16000, 10000, 40), (16000, 7500, 30), (16000, 10000, 20),
])
def test_batch_other_configs(fs, us, nbytes):  # BASELINE config 4's rate/duration matrix (8 kHz encode: no reference)
    _roundtrip_check(fs, us, nbytes, 24, 6)


@pytest.mark.parametrize("us,nbytes", [(10000, 30), (7500, 23)])
def test_decode_8khz(us, nbytes):
    """8 kHz has no reference encoder (bandwidth_detector.rs:36-37 panics); decode-only parity on a
    stream produced by the oracle's early-return encoder path."""
    cfg = pkg.Lc3Config(8000, us)
    pcm = synth.make_pcm(8, 6, cfg.nf, 8000)
    data = O.encode_batch(pcm, nbytes, 8000, us)
    ref = O.decode_batch(data, cfg.nf, 8000, us)
    got = gpu_decode(data, cfg.nf, 8000, us)
    assert np.array_equal(got, ref)
    with pytest.raises(pkg.Lc3EncoderError) as ei:
        pkg.Lc3Encoder(1, us, 8000)
    assert ei.value.code == -7


def test_cold_start_many_streams():  # Mode A: every frame from a fresh encoder/decoder
    pcm = synth.make_pcm(1024, 1, 480, 48000, seed=7)
    ref = O.encode_batch(pcm, 150, threads=8)
    got = gpu_encode(pcm, 150)
    assert np.array_equal(got, ref)
    assert np.array_equal(gpu_decode(ref, 480), O.decode_batch(ref, 480, threads=8))


def test_state_carry_across_launches():
    """T frames in one launch == the same frames split over several launches (state round-trips through HBM),
    and == frame-by-frame encode_frame / decode_frame."""
    S, T = 8, 6
    pcm = synth.make_pcm(S, T, 480, 48000, seed=11)
    ref = O.encode_batch(pcm, 150)
    enc = pkg.Lc3Encoder(S, US, FS)
    a = gpu_encode(pcm[:, :2], 150, enc=enc)
    b = gpu_encode(pcm[:, 2:], 150, enc=enc)
    assert np.array_equal(np.concatenate([a, b], axis=1), ref)
    ref_pcm = O.decode_batch(ref, 480)
    dec = pkg.Lc3Decoder(S, US, FS)
    pa = gpu_decode(ref[:, :3], 480, dec=dec)
    pb = gpu_decode(ref[:, 3:], 480, dec=dec)
    assert np.array_equal(np.concatenate([pa, pb], axis=1), ref_pcm)
    # frame API on one channel while the others idle
    enc2 = pkg.Lc3Encoder(3, US, FS)
    dec2 = pkg.Lc3Decoder(3, US, FS)
    for t in range(T):
        buf = np.zeros(150, np.uint8)
        enc2.encode_frame(1, pcm[5, t], buf)
        assert np.array_equal(buf, ref[5, t])
        out = np.zeros(480, np.int16)
        dec2.decode_frame(16, 2, buf, out)
        assert np.array_equal(out, ref_pcm[5, t])


def test_state_save_load_and_reset():
    S, T = 4, 4
    pcm = synth.make_pcm(S, T, 480, 48000, seed=13)
    ref = O.encode_batch(pcm, 150)
    enc = pkg.Lc3Encoder(S, US, FS)
    gpu_encode(pcm[:, :2], 150, enc=enc)
    blob = enc.state_save()
    enc_b = pkg.Lc3Encoder(S, US, FS)
    enc_b.state_load(blob)
    assert np.array_equal(gpu_encode(pcm[:, 2:], 150, enc=enc_b), ref[:, 2:])
    enc.reset()
    assert np.array_equal(gpu_encode(pcm, 150, enc=enc), ref)
    dec = pkg.Lc3Decoder(S, US, FS)
    ref_pcm = O.decode_batch(ref, 480)
    gpu_decode(ref[:, :2], 480, dec=dec)
    dblob = dec.state_save()
    dec_b = pkg.Lc3Decoder(S, US, FS)
    dec_b.state_load(dblob)
    assert np.array_equal(gpu_decode(ref[:, 2:], 480, dec=dec_b), ref_pcm[:, 2:])
    dec.reset()
    assert np.array_equal(gpu_decode(ref, 480, dec=dec), ref_pcm)


def test_variable_bitrate_per_call():  # nbits = 8 * buf_out.len() may change per call (lc3_encoder.rs:65)
    pcm = synth.make_pcm(1, 6, 480, 48000, seed=17)[0]
    sizes = [150, 100, 60, 300, 150, 40]
    oe, od = O.Encoder(), O.Decoder()
    enc, dec = pkg.Lc3Encoder(1, US, FS), pkg.Lc3Decoder(1, US, FS)
    for t, nb in enumerate(sizes):
        want = oe.encode_frame(pcm[t], nb)
        got = np.zeros(nb, np.uint8)
        enc.encode_frame(0, pcm[t], got)
        assert np.array_equal(got, want), f"frame {t} ({nb} bytes)"
        _, wp = od.decode_frame(want)
        gp = np.zeros(480, np.int16)
        dec.decode_frame(16, 0, want, gp)
        assert np.array_equal(gp, wp), f"frame {t} pcm"


def test_bit_rate_changes_between_launches_of_several_frames():
    """The frame size may change from call to call (nbits = 8 * buf_out.len(), lc3_encoder.rs:65), and with it whether the long-term
    post-filter may switch on (long_term_post_filter.rs:146).  A launch at a rate that keeps the filter off only computes the
    normalised correlation of its LAST TWO frames (lc3_enc_ltpf: nothing else can reach an output or the state): launches of 1 .. 5
    frames at alternating rates, state carried, against oracle encoders fed the same frames one by one.  The material keeps the filter
    switching (steady, gliding and interrupted tones) beside the usual synthetic streams."""
    lt = synth.make_ltpf_pcm(480, 48000, n_frames=20)
    pcm = np.concatenate([lt, synth.make_pcm(29, 20, 480, 48000, seed=23)], axis=0)
    S = pcm.shape[0]
    plan = [(4, 150), (3, 60), (5, 150), (1, 80), (2, 150), (3, 100), (2, 40)]  # (frames, bytes per frame): 880 bits = 110 bytes is the border
    assert sum(n for n, _ in plan) == 20
    enc = pkg.Lc3Encoder(S, US, FS)
    oracle = [O.Encoder() for _ in range(S)]
    t0, active = 0, 0
    for n, nb in plan:
        got = gpu_encode(pcm[:, t0:t0 + n], nb, enc=enc)
        for s_i in range(S):
            for t in range(n):
                want = oracle[s_i].encode_frame(pcm[s_i, t0 + t], nb)
                assert np.array_equal(got[s_i, t], want), f"stream {s_i} frame {t0 + t} ({nb} bytes)"
        if nb < 110:  # ltpf_active is the bit after pitch_present's group: count it through the decoder instead
            O.ltpf_transition_counts(reset=True)
            O.decode_batch(got, 480)
            c = O.ltpf_transition_counts()
            active += sum(c[2:])
        t0 += n
    assert active > 0, "the post-filter never switched on: the test material lost its point"


def test_long_launches_keep_the_post_filter_memories_for_a_later_rate():
    """Launches of MORE than 256 frames per stream at a rate that keeps the post-filter off, each followed by a launch at a rate that
    lets it switch on: the first low-rate frames decide `ltpf_active` from mem_nc / mem_mem_nc (long_term_post_filter.rs:365-409), which
    are the normalised correlations of the long launch's LAST TWO frames -- the only ones it computes.  (The "not one of the last two"
    mark once shared a word with the frame number and read frame 256's own bit 8 as the mark.)  Steady tones: the filter is on from the
    first low-rate frame only if both memories are right."""
    plan = [(258, 150), (3, 60), (300, 150), (3, 60), (2, 150), (2, 60)]
    T = sum(n for n, _ in plan)
    lt = synth.make_ltpf_pcm(480, 48000, n_frames=T)
    pcm = np.concatenate([lt, synth.make_pcm(5, T, 480, 48000, seed=31)], axis=0)
    S = pcm.shape[0]
    enc = pkg.Lc3Encoder(S, US, FS)
    oracle = [O.Encoder() for _ in range(S)]
    t0, first_frame_active = 0, 0
    for n, nb in plan:
        got = gpu_encode(pcm[:, t0:t0 + n], nb, enc=enc)
        for s_i in range(S):
            for t in range(n):
                want = oracle[s_i].encode_frame(pcm[s_i, t0 + t], nb)
                assert np.array_equal(got[s_i, t], want), f"stream {s_i} frame {t0 + t} ({nb} bytes)"
        if nb < 110:
            O.ltpf_transition_counts(reset=True)
            O.decode_batch(got[:, :1], 480)  # fresh decoders, one frame: a filter that is on shows as a transition out of "off"
            first_frame_active += sum(O.ltpf_transition_counts()[2:])
        t0 += n
    assert first_frame_active > 0, "no stream had the filter on in the first frame after a long launch: the test lost its point"


# ---------------------------------------------------------------- corrupt frames / PLC
def _loss_bursts_check():
    for fs, us, nbytes in ((48000, 10000, 150), (48000, 7500, 113)):  # (three bandwidth bits: `|= 7` is an out-of-range index)
        cfg = pkg.Lc3Config(fs, us)
        S, T = 12, 40
        pcm = synth.make_pcm(S, T, cfg.nf, fs, seed=37)
        data = O.encode_batch(pcm, nbytes, fs, us).copy()
        rng = np.random.default_rng(41)
        bad = np.zeros((S, T), np.uint8)
        marked = data.copy()  # what the oracle sees: an externally flagged frame is a frame it cannot parse
        for s_i in range(S):
            start = 2 + (s_i * 3) % 11
            run = 9 + s_i % 5  # 9 .. 13
            for k in range(run):
                t = start + k
                how = (s_i + k) % 3
                if how == 0:
                    data[s_i, t, -1] |= 7
                    marked[s_i, t, -1] |= 7  # invalid bandwidth index -> SideInfoError -> PLC
                elif how == 1:
                    bad[s_i, t] = 1
                    marked[s_i, t, -1] |= 7
                else:  # garbage that cannot parse either (garbage that happens to parse would end the run)
                    g = rng.integers(0, 256, nbytes, dtype=np.uint8)
                    g[-1] |= 7
                    data[s_i, t] = g
                    marked[s_i, t] = g
            # a second, short run later: the fade starts again from the first factor after a good frame
            for t in range(start + run + 4, min(T, start + run + 7)):
                bad[s_i, t] = 1
                marked[s_i, t, -1] |= 7
        ref = O.decode_batch(marked, cfg.nf, fs, us)
        cuts = [5, 3, 7, 1, 6, 4, 2, 8, 4]
        assert sum(cuts) == T
        dec = pkg.Lc3Decoder(S, us, fs)
        t0, parts = 0, []
        for n in cuts:
            parts.append(gpu_decode(data[:, t0:t0 + n], cfg.nf, fs, us, dec=dec, bad=bad[:, t0:t0 + n]))
            t0 += n
        got = np.concatenate(parts, axis=1)
        d = np.argwhere((got != ref).any(axis=2))
        assert len(d) == 0, f"{fs}/{us}: {len(d)} frames differ, first {d[0].tolist()}"
        assert dec.plc_events() >= S * 12
        assert np.array_equal(gpu_decode(data, cfg.nf, fs, us, bad=bad), ref), f"{fs}/{us}: one launch of {T} frames"
        # the same runs in a batch large enough for the producer / consumer parser (every stream repeated)
        rep = 16384 // (S * T) + 1
        big = gpu_decode(np.tile(data, (rep, 1, 1)), cfg.nf, fs, us, bad=np.tile(bad, (rep, 1)))
        assert np.array_equal(big, np.tile(ref, (rep, 1, 1))), f"{fs}/{us}: {rep * S} streams"
    print("bursts ok")


def test_loss_bursts_of_nine_to_thirteen_frames_across_launches():
    """packet_loss_concealment.rs:62-66: from the ninth lost frame of a run on the concealed spectrum fades by 0.85 per frame (0.9 for
    frames 4 .. 8 of the run).  Runs of 9 .. 13 lost frames -- unparsable side information, garbage and external flags mixed -- that
    cross launch boundaries (launches of 5, 3, 7, 1, 6, 4 ... frames: `num_lost_frames` and `alpha` travel in the state blob), in the
    form full batches use (reconstruction on the parser's lane) and the one small launches use (in the synthesis kernel), at 48 kHz /
    10 ms and at 48 kHz / 7.5 ms; the size rule's own choice as well."""
    import os
    import subprocess
    import sys

    _loss_bursts_check()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, 'tests')\nimport test_gpu_parity as t\nt._loss_bursts_check()\n"
    for v in ("lane", "late"):
        env = dict(os.environ, LC3GPU_RECON=v)
        env.pop("LC3GPU_LATE_RECON", None)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "bursts ok" in r.stdout, v + r.stdout[-2000:] + r.stderr[-2000:]


def test_corrupt_frames_are_concealed_like_the_reference():  # lc3_decoder.rs:138-141, packet_loss_concealment.rs
    S, T = 6, 12
    pcm = synth.make_pcm(S, T, 480, 48000, seed=19)
    data = O.encode_batch(pcm, 150).copy()
    rng = np.random.default_rng(5)
    # invalid bandwidth index (P_BW = 7 > fs_ind) -> SideInfoError::BandwidthIdxOutOfRange -> PLC
    for s, t in [(0, 3), (0, 4), (1, 0), (2, 5), (2, 6), (2, 7), (2, 8), (2, 9), (2, 10), (3, 11)]:
        data[s, t, -1] |= 7
    # random garbage frames: whatever the parser makes of them must match the oracle
    for s, t in [(4, 2), (4, 7), (5, 1), (5, 2)]:
        data[s, t] = rng.integers(0, 256, 150, dtype=np.uint8)
    ref = O.decode_batch(data, 480)
    dec = pkg.Lc3Decoder(S, US, FS)
    got = gpu_decode(data, 480, dec=dec)
    assert np.array_equal(got, ref)
    assert dec.plc_events() >= 10


def test_random_garbage_streams():
    rng = np.random.default_rng(23)
    for nbytes in (20, 40, 150, 400):
        data = rng.integers(0, 256, (32, 4, nbytes), dtype=np.uint8)
        assert np.array_equal(gpu_decode(data, 480), O.decode_batch(data, 480)), nbytes


def test_bad_frame_flag_forces_concealment():
    S, T = 2, 6
    pcm = synth.make_pcm(S, T, 480, 48000, seed=29)
    data = O.encode_batch(pcm, 150)
    bad = np.zeros((S, T), np.uint8)
    bad[0, 2] = bad[0, 3] = bad[1, 5] = 1
    corrupt = data.copy()
    corrupt[bad.astype(bool), -1] |= 7  # same effect in the oracle: unparsable side info
    ref = O.decode_batch(corrupt, 480)
    got = gpu_decode(data, 480, bad=bad)
    assert np.array_equal(got, ref)


def test_a_pair_that_gave_up_is_reported_by_the_next_batch_call():
    """A packer pair that gives up leaves its frames zero-filled, a parser pair conceals them (include/lc3gpu.h); the caller must not have
    to poll a counter to learn that: the device raises a flag in pinned host memory and the handle's next batch call returns LC3GPU_EPAIR
    once, launches nothing, and may be repeated.  No pair has ever given up, so the device side of the path is driven by the injection
    entry point (the same device function the kernels call)."""
    pcm = synth.make_pcm(8, 2, 480, 48000, seed=3)
    ref = O.encode_batch(pcm, 150)
    enc = pkg.Lc3Encoder(8, US, FS)
    assert np.array_equal(gpu_encode(pcm[:, :1], 150, enc=enc), ref[:, :1])
    enc.debug_pair_giveup()
    with pytest.raises(pkg.Lc3EncoderError) as ei:
        gpu_encode(pcm[:, 1:], 150, enc=enc)
    assert ei.value.code == -8
    torch_mod().cuda.synchronize()
    assert np.array_equal(gpu_encode_keep_count(pcm[:, 1:], 150, enc), ref[:, 1:])  # the refused call launched nothing: state intact
    assert enc.pair_timeouts() == 1
    dec = pkg.Lc3Decoder(8, US, FS)
    refp = O.decode_batch(ref, 480)
    dec.debug_pair_giveup()
    with pytest.raises(pkg.Lc3DecoderError) as ed:
        gpu_decode(ref, 480, dec=dec)
    assert ed.value.code == -8
    torch = torch_mod()
    d_in = torch.from_numpy(ref).cuda()
    d_pcm = torch.zeros((8, 2, 480), dtype=torch.int16, device="cuda")
    dec.decode(d_in, d_pcm, 150, 2, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_pcm.cpu().numpy(), refp) and dec.pair_timeouts() == 1


def gpu_encode_keep_count(pcm, nbytes, enc):
    """gpu_encode without its `pair_timeouts() == 0` assertion"""
    torch = torch_mod()
    S, T, nf = pcm.shape
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    d_out = torch.zeros((S, T, nbytes), dtype=torch.uint8, device="cuda")
    enc.encode(d_pcm, d_out, nbytes, T, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


def test_decoder_state_blobs_are_deterministic():
    """a decoder launch stores only the part of the post-filter's output ring it wrote: the rest of a blob is zero, not whatever the
    allocation held -- two handles that decoded the same frames save identical blobs, also after a reset"""
    pcm = synth.make_pcm(6, 3, 480, 48000, seed=43)
    data = O.encode_batch(pcm, 150)
    blobs = []
    for _ in range(2):
        junk = torch_mod().full((1 << 22,), 0x5A, dtype=torch_mod().uint8, device="cuda")  # dirty the allocator's memory
        del junk
        dec = pkg.Lc3Decoder(6, US, FS)
        gpu_decode(data[:, :1], 480, dec=dec)
        dec.reset()
        gpu_decode(data, 480, dec=dec)
        blobs.append(dec.state_save())
    assert np.array_equal(blobs[0], blobs[1])


def test_decoder_reset_paths():
    """lc3gpu_decoder_reset only notes the reset; whoever touches the states next carries it out: a batch launch over all channels (inside
    its synthesis kernel), or -- through the zero-frame initialising launch -- a range launch, a frame call, state_save, plc_events.  Each
    of them after a reset must behave like a fresh handle."""
    S, T = 12, 5
    pcm = synth.make_pcm(S, T, 480, 48000, seed=61)
    data = O.encode_batch(pcm, 150).copy()
    data[3, 1, -1] |= 7  # one concealed frame: the PLC counter is part of the state
    ref = O.decode_batch(data, 480)
    fresh_blob = pkg.Lc3Decoder(S, US, FS).state_save()
    dec = pkg.Lc3Decoder(S, US, FS)
    assert np.array_equal(gpu_decode(data, 480, dec=dec), ref) and dec.plc_events() == 1
    dec.reset()
    assert dec.plc_events() == 0  # (materialises)
    assert np.array_equal(gpu_decode(data, 480, dec=dec), ref)
    dec.reset()
    assert np.array_equal(dec.state_save(), fresh_blob)  # (materialises)
    dec.reset()
    assert np.array_equal(gpu_decode(data, 480, dec=dec), ref) and dec.plc_events() == 1  # the launch itself starts from the constructed state
    # a range launch right after a reset: the OTHER channels must be fresh afterwards, not stale
    dec.reset()
    torch = torch_mod()
    d_in = torch.from_numpy(np.ascontiguousarray(data[4:8])).cuda()
    d_pcm = torch.zeros((4, T, 480), dtype=torch.int16, device="cuda")
    dec.decode(d_in, d_pcm, 150, T, stream=torch.cuda.current_stream().cuda_stream, first_channel=4, n_channels=4)
    torch.cuda.synchronize()
    assert np.array_equal(d_pcm.cpu().numpy(), ref[4:8])
    rest = np.ascontiguousarray(data[:4])
    d_in2 = torch.from_numpy(rest).cuda()
    d_pcm2 = torch.zeros((4, T, 480), dtype=torch.int16, device="cuda")
    dec.decode(d_in2, d_pcm2, 150, T, stream=torch.cuda.current_stream().cuda_stream, first_channel=0, n_channels=4)
    torch.cuda.synchronize()
    assert np.array_equal(d_pcm2.cpu().numpy(), ref[:4])
    # a frame call right after a reset
    dec.reset()
    out = np.zeros(480, np.int16)
    dec.decode_frame(16, 9, data[9, 0], out)
    assert np.array_equal(out, ref[9, 0])
    # state_load cancels a pending reset
    dec2 = pkg.Lc3Decoder(S, US, FS)
    gpu_decode(data[:, :2], 480, dec=dec2)
    blob = dec2.state_save()
    dec.reset()
    dec.state_load(blob)
    assert np.array_equal(gpu_decode(data[:, 2:], 480, dec=dec), ref[:, 2:])


# ---------------------------------------------------------------- host-resident batches, the pipeline object
def test_host_resident_batch_path():
    """lc3gpu_encode_host / lc3gpu_decode_host: the caller loops of examples/encode.rs:73-116 / examples/decode.rs:60-112 over HOST buffers,
    the channels through the device in ranges on two internal HIP streams.  One range and several, pageable and pinned buffers, state
    carried over two calls, lost-frame flags: against the oracle (a sample) and against the device-pointer calls (everything)."""
    for S, T in ((300, 5), (20000, 2)):
        base = synth.make_pcm(min(S, 512), 2 * T, 480, 48000, seed=47)
        pcm = np.ascontiguousarray(np.tile(base, ((S + base.shape[0] - 1) // base.shape[0], 1, 1))[:S])
        enc_h, enc_d = pkg.Lc3Encoder(S, US, FS), pkg.Lc3Encoder(S, US, FS)
        dec_h, dec_d = pkg.Lc3Decoder(S, US, FS), pkg.Lc3Decoder(S, US, FS)
        pin_in = pkg.PinnedBuffer((S, T, 480), np.int16)
        pin_out = pkg.PinnedBuffer((S, T, 150), np.uint8)
        k = min(S, 96)
        ref_b = O.encode_batch(pcm[:k], 150, threads=8)
        ref_p = O.decode_batch(ref_b, 480, threads=8)
        rng = np.random.default_rng(53)
        for step in range(2):
            x = np.ascontiguousarray(pcm[:, step * T:(step + 1) * T])
            if step == 0:  # pinned buffers first, pageable numpy arrays second
                pin_in.array[...] = x
                enc_h.encode_host(pin_in.array, pin_out.array, 150, T)
                got = pin_out.array.copy()
            else:
                got = np.zeros((S, T, 150), np.uint8)
                enc_h.encode_host(x, got, 150, T)
            want = gpu_encode(x, 150, enc=enc_d)
            assert np.array_equal(got, want), (S, T, step)
            assert np.array_equal(got[:k], ref_b[:, step * T:(step + 1) * T])
            bad = (rng.random((S, T)) < 0.05).astype(np.uint8)
            out = np.zeros((S, T, 480), np.int16)
            dec_h.decode_host(got, out, 150, T, bad_frame=bad if step else None)
            wantp = gpu_decode(got, 480, dec=dec_d, bad=bad if step else None)
            assert np.array_equal(out, wantp), (S, T, step)
            if step == 0:
                assert np.array_equal(out[:k], ref_p[:, :T])
        pin_in.close()
        pin_out.close()
    with pytest.raises(pkg.Lc3EncoderError):
        enc_h.encode_host(x, got, 10, T)  # frame size out of range, as the batch calls


def test_handles_bound_to_one_stream():
    """lc3gpu_*_bind_stream: a handle whose batch calls all come on one stream that outlives it records no event of its own per call (the
    pipeline object binds its handles).  Bound: same bytes and samples on that stream, state carried, state blobs / counters / reset work
    (they wait on the stream), a call on another stream is refused; released: any stream again."""
    torch = torch_mod()
    S, T = 64, 3
    pcm = synth.make_pcm(S, 2 * T, 480, 48000, seed=67)
    ref = O.encode_batch(pcm, 150)
    refp = O.decode_batch(ref, 480)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    enc, dec = pkg.Lc3Encoder(S, US, FS), pkg.Lc3Decoder(S, US, FS)
    enc.bind_stream(s1.cuda_stream)
    dec.bind_stream(s1.cuda_stream)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_b = torch.zeros((S, 2 * T, 150), dtype=torch.uint8, device="cuda")
    d_o = torch.zeros((S, 2 * T, 480), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    for k in range(2):
        x = d_pcm[:, k * T:(k + 1) * T].contiguous()
        b = torch.zeros((S, T, 150), dtype=torch.uint8, device="cuda")
        o = torch.zeros((S, T, 480), dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        enc.encode(x, b, 150, T, stream=s1.cuda_stream)
        dec.decode(b, o, 150, T, stream=s1.cuda_stream)
        if k == 0:
            blob = enc.state_save()  # (waits for the stream)
            assert dec.plc_events() == 0
        s1.synchronize()
        d_b[:, k * T:(k + 1) * T] = b
        d_o[:, k * T:(k + 1) * T] = o
    torch.cuda.synchronize()
    assert np.array_equal(d_b.cpu().numpy(), ref) and np.array_equal(d_o.cpu().numpy(), refp)
    enc_ref = pkg.Lc3Encoder(S, US, FS)
    gpu_encode(pcm[:, :T], 150, enc=enc_ref)
    assert np.array_equal(blob, enc_ref.state_save())
    with pytest.raises(pkg.Lc3EncoderError) as ei:
        enc.encode(d_pcm[:, :T].contiguous(), torch.zeros((S, T, 150), dtype=torch.uint8, device="cuda"), 150, T, stream=s2.cuda_stream)
    assert ei.value.code == -1
    enc.bind_stream(None, bind=False)
    dec.bind_stream(None, bind=False)
    enc.reset()
    dec.reset()
    assert np.array_equal(gpu_encode(pcm, 150, enc=enc), ref) and np.array_equal(gpu_decode(ref, 480, dec=dec), refp)


def test_pipeline_object_equals_the_single_stream_calls():
    """lc3gpu_pipeline (the `quad` arrangement as a library object): three submissions with two byte buffers alternating, then with ONE
    byte buffer (every encoder then waits for the decoder before it), equal byte for byte and sample for sample to lc3gpu_encode +
    lc3gpu_decode of the same frames on one stream, and to the oracle on a sample; the halves alone; join / follow against a caller's
    stream; reset; the group table."""
    torch = torch_mod()
    S, T, steps = 8200, 4, 3  # two groups of 4 100 channels: 16 400 frames each, the pair kernels' size
    base = synth.make_pcm(1025, T * steps, 480, 48000, seed=59)
    pcm = np.ascontiguousarray(np.tile(base, (8, 1, 1)))
    assert pcm.shape[0] == S
    enc, dec = pkg.Lc3Encoder(S, US, FS), pkg.Lc3Decoder(S, US, FS)
    want_b, want_p = [], []
    for k in range(steps):
        b = gpu_encode(pcm[:, k * T:(k + 1) * T], 150, enc=enc)
        want_b.append(b)
        want_p.append(gpu_decode(b, 480, dec=dec))
    ref_b = O.encode_batch(pcm[:64], 150, threads=8)
    assert np.array_equal(np.concatenate(want_b, axis=1)[:64], ref_b)
    pl = pkg.Lc3Pipeline(S, US, FS)
    assert len(pl.groups) == 2 and [g["first"] for g in pl.groups] == [0, 4100] and sum(g["n"] for g in pl.groups) == S
    d_in = [torch.from_numpy(np.ascontiguousarray(pcm[:, k * T:(k + 1) * T])).cuda() for k in range(steps)]
    for n_buf in (2, 1):
        d_bytes = [torch.zeros((S, T, 150), dtype=torch.uint8, device="cuda") for _ in range(n_buf)]
        d_out = [torch.zeros((S, T, 480), dtype=torch.int16, device="cuda") for _ in range(steps)]
        keep = []
        torch.cuda.synchronize()
        for k in range(steps):
            pl.submit(d_in[k], d_bytes[k % n_buf], d_out[k], 150, T)
            if n_buf == 2 and k == 0:
                pl.wait()
                keep.append(d_bytes[0].cpu().numpy())
        pl.wait()
        for g in pl.groups:
            assert g["enc"].pair_timeouts() == 0 and g["dec"].pair_timeouts() == 0
        for k in range(steps):
            assert np.array_equal(d_out[k].cpu().numpy(), want_p[k]), (n_buf, k)
        assert np.array_equal(d_bytes[(steps - 1) % n_buf].cpu().numpy(), want_b[steps - 1])
        if keep:
            assert np.array_equal(keep[0], want_b[0])
        pl.reset()
    # the halves alone, a caller's stream on both sides: the PCM arrives on `side` (follow), the result is copied on `side` (join)
    side = torch.cuda.Stream()
    d_b = torch.zeros((S, T, 150), dtype=torch.uint8, device="cuda")
    d_o = torch.zeros((S, T, 480), dtype=torch.int16, device="cuda")
    host_in = torch.from_numpy(np.ascontiguousarray(pcm[:, :T])).pin_memory()
    d_x = torch.empty((S, T, 480), dtype=torch.int16, device="cuda")
    with torch.cuda.stream(side):
        d_x.copy_(host_in, non_blocking=True)
    pl.follow(side.cuda_stream)
    pl.encode(d_x, d_b, 150, T)
    pl.decode(d_b, d_o, 150, T)
    pl.join(side.cuda_stream)
    with torch.cuda.stream(side):
        got_p = d_o.to("cpu", non_blocking=True)
    side.synchronize()
    assert np.array_equal(got_p.numpy(), want_p[0]) and np.array_equal(d_b.cpu().numpy(), want_b[0])
    # hazards with submissions the pipeline's two slots no longer remember: three byte buffers in rotation over six round trips, then
    # three encode-only submissions into three buffers followed by their decodes in another order
    pl.reset()
    d_b3 = [torch.zeros((S, T, 150), dtype=torch.uint8, device="cuda") for _ in range(3)]
    d_o6 = [torch.zeros((S, T, 480), dtype=torch.int16, device="cuda") for _ in range(6)]
    for k in range(6):
        pl.submit(d_in[k % steps], d_b3[k % 3], d_o6[k], 150, T)
    pl.wait()
    enc3, dec3 = pkg.Lc3Encoder(S, US, FS), pkg.Lc3Decoder(S, US, FS)
    for k in range(6):
        b = gpu_encode(pcm[:, (k % steps) * T:(k % steps + 1) * T], 150, enc=enc3)
        assert np.array_equal(d_o6[k].cpu().numpy(), gpu_decode(b, 480, dec=dec3)), k
    pl.reset()
    for k in range(3):
        pl.encode(d_in[k], d_b3[k], 150, T)
    outs = {}
    for k in (0, 1, 2):  # (decoders carry state: the frames must be decoded in time order; the BUFFERS were filled in three submissions)
        outs[k] = torch.zeros((S, T, 480), dtype=torch.int16, device="cuda")
        pl.decode(d_b3[k], outs[k], 150, T)
    pl.wait()
    for k in range(3):
        assert np.array_equal(d_b3[k].cpu().numpy(), want_b[k]) and np.array_equal(outs[k].cpu().numpy(), want_p[k]), k
    pl.reset()
    pl.encode(d_in[0], d_b3[0], 150, T)
    pl.wait()
    # a group's handles are the caller's to read: state blobs of the pipeline's first group == those of the plain handles' first channels
    n0 = pl.groups[0]["n"]
    enc2 = pkg.Lc3Encoder(n0, US, FS)
    gpu_encode(pcm[:n0, :T], 150, enc=enc2)
    assert np.array_equal(pl.groups[0]["enc"].state_save(), enc2.state_save())
    pl.close()
    # few channels: fewer groups than asked for, odd counts
    for S2, G in ((3, 4), (9, 2), (64, 3)):
        p2 = pkg.Lc3Pipeline(S2, US, FS, n_groups=G)
        assert sum(g["n"] for g in p2.groups) == S2 and all(g["n"] > 0 for g in p2.groups)
        x = np.ascontiguousarray(pcm[:S2, :2])
        d_x2 = torch.from_numpy(x).cuda()
        d_b2 = torch.zeros((S2, 2, 150), dtype=torch.uint8, device="cuda")
        d_o2 = torch.zeros((S2, 2, 480), dtype=torch.int16, device="cuda")
        p2.submit(d_x2, d_b2, d_o2, 150, 2)
        p2.wait()
        rb = O.encode_batch(x, 150)
        assert np.array_equal(d_b2.cpu().numpy(), rb) and np.array_equal(d_o2.cpu().numpy(), O.decode_batch(rb, 480))
        p2.close()


# ---------------------------------------------------------------- API error behaviour
def test_error_codes():
    enc = pkg.Lc3Encoder(2, US, FS)
    dec = pkg.Lc3Decoder(2, US, FS)
    buf = np.zeros(150, np.uint8)
    with pytest.raises(pkg.Lc3EncoderError) as e:  # reference: panic "Cannot decode channel index"
        enc.encode_frame(2, np.zeros(480, np.int16), buf)
    assert e.value.code == -2
    with pytest.raises(pkg.Lc3EncoderError) as e:  # reference: assert_eq!(input.len(), nf) panic
        enc.encode_frame(0, np.zeros(479, np.int16), buf)
    assert e.value.code == -3
    with pytest.raises(pkg.Lc3DecoderError) as e:  # Only16BitsPerAudioSampleSupported
        dec.decode_frame(24, 0, buf, np.zeros(480, np.int16))
    assert e.value.code == -4
    with pytest.raises(pkg.Lc3DecoderError) as e:
        dec.decode_frame(16, 5, buf, np.zeros(480, np.int16))
    assert e.value.code == -2
    with pytest.raises(pkg.Lc3GpuError):
        pkg.Lc3Config(22050, 10000)


# ---------------------------------------------------------------- full-size batch (BASELINE configs[1])
def test_full_size_batch_properties():
    """65 536 frames (16 384 streams x 4 frames) on one GPU: determinism, sampled oracle parity,
    and a size-independent round-trip property (decode(encode(x)) tracks x)."""
    S, T, NB = 16384, 4, 150
    torch = torch_mod()
    pcm = synth.make_pcm(512, T, 480, 48000, seed=31)
    big = np.tile(pcm, (S // 512, 1, 1))  # 512 distinct streams repeated: identical streams must agree
    a = gpu_encode(big, NB)
    b = gpu_encode(big, NB)
    assert np.array_equal(a, b), "encode is not deterministic"
    assert np.array_equal(a[:512], a[512:1024]) and np.array_equal(a[:512], a[-512:])
    assert np.array_equal(a[:512], O.encode_batch(pcm, NB, threads=8))
    p = gpu_decode(a, 480)
    assert np.array_equal(p[:512], p[-512:])
    assert np.array_equal(p[:512], O.decode_batch(a[:512], 480, threads=8))
    # round trip: decode(encode(x)) must track x up to the codec's fixed algorithmic delay.  The delay is found
    # from the data (best lag within one frame) rather than assumed.
    x = big[:64].astype(np.float64).reshape(64, -1)
    y = p[:64].astype(np.float64).reshape(64, -1)
    live = x.std(axis=1) > 100
    n = x.shape[1] - 480
    best = -1e9
    best_lag = -1
    for lag in range(0, 480, 4):
        xs, ys = x[live, :n], y[live, lag: lag + n]
        err = ((xs - ys) ** 2).sum(axis=1)
        snr = float(np.median(10 * np.log10((xs ** 2).sum(axis=1) / np.maximum(err, 1e-9))))
        if snr > best:
            best, best_lag = snr, lag
    assert best > 15.0, f"median round-trip SNR {best:.1f} dB at lag {best_lag}"
    del torch


# ---------------------------------------------------------------- 1 M-frame encode batch (BASELINE configs[2], one GPU's worth of it)
@pytest.mark.parametrize("S,T", [(65536, 16), (1048576, 1)])
def test_one_million_frame_encode_batch(S, T):
    """1 048 576 frames through ONE lc3gpu_encode call on one GPU (the reference's caller loop, examples/encode.rs:97-115,
    over a million channel frames): mode B = 65 536 streams x 16 frames with carried state, mode A = 1 048 576 fresh streams x
    1 frame.  Parity: the distinct streams against the oracle, replicas of a stream against each other (every frame of the
    batch is compared with something), and a checksum of the whole bitstream against the one the replication implies."""
    torch = torch_mod()
    NB, D = 150, 1024
    pcm = synth.make_pcm(D, T, 480, 48000, seed=53)
    ref = O.encode_batch(pcm, NB, threads=8)
    d_small = torch.from_numpy(pcm).cuda()
    d_pcm = d_small.repeat(S // D, 1, 1).contiguous()  # replica r of stream i sits at r * D + i
    assert d_pcm.shape == (S, T, 480)
    d_out = torch.zeros((S, T, NB), dtype=torch.uint8, device="cuda")
    enc = pkg.Lc3Encoder(S, US, FS)
    enc.encode(d_pcm, d_out, NB, T, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d_ref = torch.from_numpy(ref).cuda()
    view = d_out.view(S // D, D, T, NB)
    assert bool((view == d_ref.unsqueeze(0)).all()), "a frame of the 1 M batch differs from the oracle's bitstream"
    assert int(d_out.to(torch.int64).sum().item()) == int(ref.astype(np.int64).sum()) * (S // D)
    enc.close()


@pytest.mark.parametrize("fs,us,nbytes", [(48000, 10000, 60), (32000, 10000, 40), (16000, 7500, 30), (48000, 7500, 60)])
def test_ltpf_transitions(fs, us, nbytes):
    """Decoder LTPF synthesis (block-parallel on the GPU) through all five transition cases of
    decoder/long_term_post_filter.rs:142-160; the oracle's counters prove the inputs reach every case."""
    nf = {48000: 480, 32000: 320, 16000: 160}[fs] * us // 10000
    pcm = synth.make_ltpf_pcm(nf, fs)
    ref = O.encode_batch(pcm, nbytes, fs, us)
    O.ltpf_transition_counts(reset=True)
    ref_pcm = O.decode_batch(ref, nf, fs, us)
    counts = O.ltpf_transition_counts()
    assert all(c > 0 for c in counts[1:]), counts
    assert np.array_equal(gpu_encode(pcm, nbytes, fs, us), ref)
    assert np.array_equal(gpu_decode(ref, nf, fs, us), ref_pcm)


@pytest.mark.parametrize("fs,us,nbytes,chunks", [(48000, 10000, 60, [1]), (48000, 10000, 60, [1, 2, 3, 5]), (48000, 7500, 60, [2, 1, 4]),
                                                 (16000, 7500, 30, [1, 3]), (32000, 10000, 40, [4])])
def test_ltpf_ring_is_fetched_lazily_and_stored_in_part(fs, us, nbytes, chunks):
    """The decoder's LTPF output ring (lc3_dec_core::x_hat_ltpf_mem, num_mem_blocks * nf samples) is not read at the start of a launch: the
    first frame whose filter is on -- or was on in the frame before -- pulls in what earlier launches left, minus the blocks this launch has
    written by then, and a launch stores only the blocks it wrote (lc3_dec_ring_fill, lc3_dec_state_store).  A stream that walks through
    all five filter transitions, decoded in launches of 1 .. 5 frames with the state carried in the handle (launches shorter than the ring,
    as long, longer), at 48 kHz with every seventh frame lost on the way (a concealed frame runs the filter with `off`), against the oracle's one pass."""
    torch = torch_mod()
    nf = {48000: 480, 32000: 320, 16000: 160}[fs] * us // 10000
    pcm = synth.make_ltpf_pcm(nf, fs)
    S, T, _ = pcm.shape
    data = O.encode_batch(pcm, nbytes, fs, us)
    bad = np.zeros((S, T), np.uint8)
    if fs == 48000:
        bad[:, 6::7] = 1
    corrupt = data.copy()
    corrupt[bad.astype(bool), -1] |= 7  # the oracle's way to a lost frame: an unparsable bandwidth field (3 bits at 48 kHz)
    O.ltpf_transition_counts(reset=True)
    ref_pcm = O.decode_batch(corrupt, nf, fs, us)
    assert all(c > 0 for c in O.ltpf_transition_counts()[1:])
    dec = pkg.Lc3Decoder(S, us, fs)
    got = np.zeros_like(ref_pcm)
    t, i = 0, 0
    while t < T:
        n = min(chunks[i % len(chunks)], T - t)
        i += 1
        got[:, t:t + n] = gpu_decode(np.ascontiguousarray(data[:, t:t + n]), nf, fs, us, dec=dec, bad=np.ascontiguousarray(bad[:, t:t + n]))
        t += n
    assert np.array_equal(got, ref_pcm)
    dec.close()


@pytest.mark.parametrize("nbytes", [20, 25, 50, 79, 80, 81, 99, 100, 139, 140, 141, 160, 200, 260, 320, 399])
def test_bitrate_sweep_48k(nbytes):
    """Bitrate boundaries at 48 kHz / 10 ms: rate_flag (nbits > 800), lsb mode (nbits >= 1120), attack detector on
    (>= 100 bytes), LTPF gain classes (nbits < 880), bit-budget steps (nbits <= 1280 / 2560)."""
    _roundtrip_check(48000, 10000, nbytes, 12, 5, seed=nbytes)


@pytest.mark.parametrize("S", [1, 2, 3, 5, 7])
def test_stream_counts_not_multiple_of_workgroup(S):  # workgroups hold four streams: partial last workgroup
    _roundtrip_check(48000, 10000, 150, S, 3, seed=100 + S)


def test_long_stream_state_carry():  # 6 s of two streams in ONE launch: state carried over 600 frames inside the kernel
    _roundtrip_check(48000, 10000, 150, 2, 600, seed=11)


def test_channel_ranges_on_separate_hip_streams():
    """lc3gpu_encode_range / lc3gpu_decode_range on sub-ranges of one handle's channels, different T per call."""
    t = torch_mod()
    S, T, nf, nb = 12, 6, 480, 150
    pcm = synth.make_pcm(S, T, nf, 48000, seed=21)
    ref_b = O.encode_batch(pcm, nb)
    ref_p = O.decode_batch(ref_b, nf)
    enc = pkg.Lc3Encoder(S, 10000, 48000)
    dec = pkg.Lc3Decoder(S, 10000, 48000)
    out_b = np.zeros((S, T, nb), np.uint8)
    out_p = np.zeros((S, T, nf), np.int16)
    for first, n in ((0, 5), (5, 4), (9, 3)):
        for t0, tn in ((0, 2), (2, 4)):  # two launches per range: 2 frames, then 4
            d_pcm = t.from_numpy(np.ascontiguousarray(pcm[first:first + n, t0:t0 + tn])).cuda()
            d_b = t.zeros((n, tn, nb), dtype=t.uint8, device="cuda")
            d_p = t.zeros((n, tn, nf), dtype=t.int16, device="cuda")
            enc.encode(d_pcm, d_b, nb, tn, first_channel=first, n_channels=n)
            dec.decode(d_b, d_p, nb, tn, first_channel=first, n_channels=n)
            t.cuda.synchronize()
            out_b[first:first + n, t0:t0 + tn] = d_b.cpu().numpy()
            out_p[first:first + n, t0:t0 + tn] = d_p.cpu().numpy()
    assert np.array_equal(out_b, ref_b)
    assert np.array_equal(out_p, ref_p)


def test_kernel_timer_counts_every_nth_call():
    """lc3gpu_*_timing: per-kernel HIP events on every batch call (enable = 1) or on every n-th one (enable = n), the count of timed
    calls and plausible durations; recording off leaves nothing behind."""
    t = torch_mod()
    S, T, nf, nb = 256, 2, 480, 150
    d_pcm = t.from_numpy(synth.make_pcm(S, T, nf, 48000, seed=5)).cuda()
    d_b = t.zeros((S, T, nb), dtype=t.uint8, device="cuda")
    d_p = t.zeros((S, T, nf), dtype=t.int16, device="cuda")
    enc = pkg.Lc3Encoder(S, 10000, 48000)
    dec = pkg.Lc3Decoder(S, 10000, 48000)
    for every, calls, want in ((1, 5, 5), (3, 7, 3), (4, 4, 1)):
        enc.timing(every)
        dec.timing_kernels(every)
        for _ in range(calls):
            enc.encode(d_pcm, d_b, nb, T)
            dec.decode(d_b, d_p, nb, T)
        e = enc.timing(0)
        d = dec.timing_kernels(0)
        assert e[4] == want and d[4] == want, (every, e, d)
        assert all(0.0 < x < 50.0 for x in e[:4]) and d[0] > 0.0 and d[3] > 0.0, (e, d)
    enc.encode(d_pcm, d_b, nb, T)
    assert enc.timing(0)[4] == 0


def test_pipelined_half_batches_on_four_hip_streams():
    """The stream arrangement INTEGRATION.md recommends for throughput (bench.py's `overlapped.pipelined_halves`): the batch as two
    handle pairs, each with its encoder on one HIP stream and its decoder on another, the decoder of step k under the encoder of
    step k + 1 (two byte buffers taking turns, events between producer and consumer), state carried over the steps.  Nothing is
    waited for on the host until the end; every frame of every step against the oracle's streaming run."""
    t = torch_mod()
    S, T, K, nf, nb = 2048, 4, 6, 480, 150
    pcm = synth.make_pcm(S, T * K, nf, 48000, seed=29)
    ref_b = O.encode_batch(pcm, nb)
    ref_p = O.decode_batch(ref_b, nf)
    H = S // 2
    d_pcm = t.from_numpy(pcm).cuda()
    encs = [pkg.Lc3Encoder(H, 10000, 48000) for _ in range(2)]
    decs = [pkg.Lc3Decoder(H, 10000, 48000) for _ in range(2)]
    es = [t.cuda.Stream() for _ in range(2)]
    ds = [t.cuda.Stream() for _ in range(2)]
    bufs = [t.zeros((S, T, nb), dtype=t.uint8, device="cuda") for _ in range(2)]
    keep_b = t.zeros((K, S, T, nb), dtype=t.uint8, device="cuda")  # a copy of every step's bytes, made on the decoder's stream
    out_p = t.zeros((K, S, T, nf), dtype=t.int16, device="cuda")
    enc_done = [[t.cuda.Event() for _ in range(2)] for _ in range(2)]
    dec_done = [[t.cuda.Event() for _ in range(2)] for _ in range(2)]
    t.cuda.synchronize()
    for k in range(K):
        b = k & 1
        step_pcm = d_pcm[:, k * T:(k + 1) * T].contiguous()  # (made on the current stream)
        ready = t.cuda.Event()
        ready.record()
        for p in range(2):
            lo, hi = p * H, (p + 1) * H
            es[p].wait_event(ready)
            if k >= 2:
                es[p].wait_event(dec_done[b][p])  # the buffer's previous reader
            encs[p].encode(step_pcm[lo:hi], bufs[b][lo:hi], nb, T, stream=es[p].cuda_stream)
            enc_done[b][p].record(es[p])
            ds[p].wait_event(enc_done[b][p])
            decs[p].decode(bufs[b][lo:hi], out_p[k, lo:hi], nb, T, stream=ds[p].cuda_stream)
            with t.cuda.stream(ds[p]):
                keep_b[k, lo:hi].copy_(bufs[b][lo:hi], non_blocking=True)
            dec_done[b][p].record(ds[p])
        step_pcm.record_stream(es[0])
        step_pcm.record_stream(es[1])
    t.cuda.synchronize()
    got_b = keep_b.cpu().numpy().transpose(1, 0, 2, 3).reshape(S, K * T, nb)
    got_p = out_p.cpu().numpy().transpose(1, 0, 2, 3).reshape(S, K * T, nf)
    assert np.array_equal(got_b, ref_b)
    assert np.array_equal(got_p, ref_p)


MIXED = [  # BASELINE config 4: (fs, frame_us, bytes per frame); 8 kHz is decode-only (no reference encoder)
    (16000, 10000, 40), (24000, 10000, 60), (32000, 10000, 80), (44100, 10000, 110), (48000, 10000, 150),
    (16000, 7500, 30), (24000, 7500, 45), (32000, 7500, 60), (44100, 7500, 83), (48000, 7500, 113),
    (8000, 10000, 30), (8000, 7500, 23),
]


def _mixed_setup(S, T2, seed=41):
    """S streams of each of the twelve configurations, interleaved in the caller's order (so that the handle has to sort them)"""
    per_cfg = []
    for fs, us, nb in MIXED:
        cfg = pkg.Lc3Config(fs, us)
        pcm = synth.make_pcm(S, T2, cfg.nf, fs, seed=seed)
        ref_b = O.encode_batch(pcm, nb, fs, us)
        ref_p = O.decode_batch(ref_b, cfg.nf, fs, us)
        per_cfg.append(dict(fs=fs, us=us, nb=nb, nf=cfg.nf, pcm=pcm, ref_b=ref_b, ref_p=ref_p))
    order = [(k, i) for i in range(S) for k in range(len(MIXED))]  # stream order: cfg0 s0, cfg1 s0, ..., cfg0 s1, ...
    return per_cfg, order


def test_mixed_configuration_batch():
    """BASELINE config 4: every (rate, duration) configuration in ONE handle pair with per-stream {fs, frame_us, nbytes}
    descriptors (lc3gpu_*_create_mixed): one launch per kernel for the whole mixed batch, ragged buffers in the caller's stream
    order.  8 kHz streams are decode-only (no reference encoder): the encoder handle holds the other ten configurations, the
    decoder handle all twelve.  Two calls per handle so that state is carried; byte-exact / 0 LSB against the per-configuration
    oracle."""
    t = torch_mod()
    S, T = 24, 3
    per_cfg, order = _mixed_setup(S, 2 * T)
    enc_order = [(k, i) for (k, i) in order if MIXED[k][0] != 8000]
    enc = pkg.Lc3Encoder.mixed([MIXED[k] for k, _ in enc_order])
    dec = pkg.Lc3Decoder.mixed([MIXED[k] for k, _ in order])
    with pytest.raises(pkg.Lc3EncoderError) as e8:
        pkg.Lc3Encoder.mixed([(8000, 10000, 30)])
    assert e8.value.code == -7
    st = t.cuda.current_stream().cuda_stream
    for t0 in (0, T):
        pcm_in = np.concatenate([per_cfg[k]["pcm"][i, t0:t0 + T].reshape(-1) for k, i in enc_order])
        d_pcm = t.from_numpy(pcm_in).cuda()
        d_b = t.zeros(sum(T * MIXED[k][2] for k, _ in enc_order), dtype=t.uint8, device="cuda")
        enc.encode_mixed(d_pcm, d_b, T, stream=st)
        got = d_b.cpu().numpy()
        off = 0
        for k, i in enc_order:
            n = T * MIXED[k][2]
            assert np.array_equal(got[off:off + n].reshape(T, -1), per_cfg[k]["ref_b"][i, t0:t0 + T]), (MIXED[k], i, t0)
            off += n
        # decoder: all twelve configurations, fed with the oracle's bitstreams (the 8 kHz ones have no GPU-encoded form)
        bytes_in = np.concatenate([per_cfg[k]["ref_b"][i, t0:t0 + T].reshape(-1) for k, i in order])
        d_in = t.from_numpy(bytes_in).cuda()
        d_out = t.zeros(sum(T * per_cfg[k]["nf"] for k, _ in order), dtype=t.int16, device="cuda")
        dec.decode_mixed(d_in, d_out, T, stream=st)
        gp = d_out.cpu().numpy()
        off = 0
        for k, i in order:
            n = T * per_cfg[k]["nf"]
            assert np.array_equal(gp[off:off + n].reshape(T, -1), per_cfg[k]["ref_p"][i, t0:t0 + T]), (MIXED[k], i, t0)
            off += n
    # the per-frame calls and the state blobs of a mixed handle use the caller's stream order
    k, i = enc_order[5]
    buf = np.zeros(MIXED[k][2], np.uint8)
    blob = enc.state_save()
    enc.encode_frame(5, per_cfg[k]["pcm"][i, 0], buf)  # continues stream 5 with an arbitrary next frame
    enc2 = pkg.Lc3Encoder.mixed([MIXED[kk] for kk, _ in enc_order])
    enc2.state_load(blob)
    buf2 = np.zeros(MIXED[k][2], np.uint8)
    enc2.encode_frame(5, per_cfg[k]["pcm"][i, 0], buf2)
    assert np.array_equal(buf, buf2)
    with pytest.raises(pkg.Lc3EncoderError):
        enc.encode(d_pcm, d_b, 40, T)  # the uniform batch call is refused on a mixed handle


def test_mixed_configuration_pipeline():
    """lc3gpu_pipeline_create_mixed: the ten encodable configurations of BASELINE config 4 through the pipeline object -- streams of every
    configuration in every group (the caller's order interleaves them), ragged buffers, two submissions with two byte buffers, state
    carried -- byte-exact / 0 LSB against the per-configuration oracle; uneven group boundaries given by the caller; 8 kHz refused; the
    uniform calls refused on a mixed pipeline and the other way round."""
    t = torch_mod()
    S, T = 24, 3
    per_cfg, order = _mixed_setup(S, 2 * T, seed=43)
    order = [(k, i) for (k, i) in order if MIXED[k][0] != 8000]
    descs = [MIXED[k] for k, _ in order]
    for kwargs in ({"n_groups": 2}, {"group_first": [0, 7, 100]}, {"n_groups": 1}):
        pl = pkg.Lc3Pipeline.mixed(descs, **kwargs)
        assert sum(g["n"] for g in pl.groups) == len(descs)
        if "group_first" in kwargs:
            assert [g["first"] for g in pl.groups] == kwargs["group_first"]
        d_b = [t.zeros(sum(T * d[2] for d in descs), dtype=t.uint8, device="cuda") for _ in range(2)]
        for step, t0 in enumerate((0, T)):
            d_pcm = t.from_numpy(np.concatenate([per_cfg[k]["pcm"][i, t0:t0 + T].reshape(-1) for k, i in order])).cuda()
            d_out = t.zeros(sum(T * per_cfg[k]["nf"] for k, _ in order), dtype=t.int16, device="cuda")
            pl.submit_mixed(d_pcm, d_b[step], d_out, T)
            pl.wait()
            gb, gp = d_b[step].cpu().numpy(), d_out.cpu().numpy()
            ob = op = 0
            for k, i in order:
                nb_, nf_ = T * MIXED[k][2], T * per_cfg[k]["nf"]
                assert np.array_equal(gb[ob:ob + nb_].reshape(T, -1), per_cfg[k]["ref_b"][i, t0:t0 + T]), (kwargs, MIXED[k], i, t0)
                assert np.array_equal(gp[op:op + nf_].reshape(T, -1), per_cfg[k]["ref_p"][i, t0:t0 + T]), (kwargs, MIXED[k], i, t0)
                ob += nb_
                op += nf_
        with pytest.raises(pkg.Lc3GpuError) as e1:
            pl.submit(d_pcm, d_b[0], d_out, 150, T)
        assert e1.value.code == -1
        pl.close()
    with pytest.raises(pkg.Lc3GpuError) as e8:
        pkg.Lc3Pipeline.mixed([(48000, 10000, 150), (8000, 10000, 30)])
    assert e8.value.code == -7
    with pytest.raises(pkg.Lc3GpuError):
        pkg.Lc3Pipeline.mixed(descs, group_first=[0, 50, 50])  # an empty group
    pu = pkg.Lc3Pipeline(8, US, FS)
    with pytest.raises(pkg.Lc3GpuError) as e2:
        pu.submit_mixed(d_pcm, d_b[0], d_out, T)
    assert e2.value.code == -1
    pu.close()


def test_mixed_batch_bad_frames_and_plc_counter():
    """external bad-frame flags on a mixed decoder: the flag array is in the caller's stream order; flagged frames are concealed
    like frames with unparsable side information in the oracle (48 kHz streams: bandwidth index 7 does not exist)"""
    t = torch_mod()
    S, T = 5, 4
    per_cfg, order = _mixed_setup(S, T, seed=43)
    dec = pkg.Lc3Decoder.mixed([MIXED[k] for k, _ in order])
    bytes_in = np.concatenate([per_cfg[k]["ref_b"][i].reshape(-1) for k, i in order])
    bad = np.zeros((len(order), T), np.uint8)
    flagged = [idx for idx, (k, i) in enumerate(order) if MIXED[k][0] == 48000][:3]
    for n, idx in enumerate(flagged):
        bad[idx, 1 + n % 3] = 1
    d_out = t.zeros(sum(T * per_cfg[k]["nf"] for k, _ in order), dtype=t.int16, device="cuda")
    dec.decode_mixed(t.from_numpy(bytes_in).cuda(), d_out, T, d_bad_frame=t.from_numpy(bad).cuda())
    gp = d_out.cpu().numpy()
    off = 0
    for idx, (k, i) in enumerate(order):
        n = T * per_cfg[k]["nf"]
        exp = per_cfg[k]["ref_p"][i]
        if bad[idx].any():
            corrupt = per_cfg[k]["ref_b"][i:i + 1].copy()
            corrupt[0, bad[idx].astype(bool), -1] |= 7
            exp = O.decode_batch(corrupt, per_cfg[k]["nf"], MIXED[k][0], MIXED[k][1])[0]
        assert np.array_equal(gp[off:off + n].reshape(T, -1), exp), (MIXED[k], i)
        off += n
    assert dec.plc_events() == len(flagged)


# ---------------------------------------------------------------- buffer layouts (BASELINE config 5: interleaved stereo)
@pytest.mark.parametrize("fs,us,nbytes,C,T", [(48000, 10000, 150, 2, 9), (48000, 7500, 113, 3, 5), (16000, 10000, 40, 5, 4),
                                              (32000, 10000, 81, 70, 3)])
def test_interleaved_layout(fs, us, nbytes, C, T):
    """lc3gpu_encode_layout / lc3gpu_decode_layout with LC3GPU_LAYOUT_INTERLEAVED: PCM int16[T][nf][C] as in a WAV file, frames
    uint8[T][C][nbytes] as in an .lc3 file -- the (de)interleave the reference's callers do on the host
    (examples/encode.rs:95-115, examples/decode.rs:86-112) happens in the kernels' loads and stores.  Two calls: state carries."""
    t = torch_mod()
    cfg = pkg.Lc3Config(fs, us)
    nf = cfg.nf
    pcm = synth.make_pcm(C, 2 * T, nf, fs, seed=47)            # planar [C][2T][nf]
    ref_b = O.encode_batch(pcm, nbytes, fs, us)                 # [C][2T][nbytes]
    ref_p = O.decode_batch(ref_b, nf, fs, us)
    enc, dec = pkg.Lc3Encoder(C, us, fs), pkg.Lc3Decoder(C, us, fs)
    st = t.cuda.current_stream().cuda_stream
    bad = np.zeros((2 * T, C), np.uint8)
    for t0 in (0, T):
        inter = np.ascontiguousarray(pcm[:, t0:t0 + T].transpose(1, 2, 0))   # [T][nf][C]
        d_pcm = t.from_numpy(inter).cuda()
        d_b = t.zeros((T, C, nbytes), dtype=t.uint8, device="cuda")
        enc.encode(d_pcm, d_b, nbytes, T, stream=st, layout="interleaved")
        d_o = t.zeros((T, nf, C), dtype=t.int16, device="cuda")
        dec.decode(d_b, d_o, nbytes, T, stream=st, layout="interleaved", d_bad_frame=t.from_numpy(bad[t0:t0 + T]).cuda())
        assert np.array_equal(d_b.cpu().numpy(), ref_b[:, t0:t0 + T].transpose(1, 0, 2))
        assert np.array_equal(d_o.cpu().numpy(), ref_p[:, t0:t0 + T].transpose(1, 2, 0))


def test_stereo_realtime_stream_latency():
    """BASELINE config 5: 2-channel interleaved 48 kHz / 10 ms stream, 6000 one-frame encode->decode steps (60 s of audio),
    host-observed submit -> complete latency per step with both PCIe copies; every frame exact against the oracle and the p99
    well inside the 10 ms real-time budget (bar: 1 ms).  The numbers go to gpurun_out/ for the record."""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import latency_stereo

    r = latency_stereo.run(6000)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "latency_stereo_test.json"), "w") as f:
        json.dump(r, f)
    assert r["bitstream_exact"] and r["pcm_max_abs_diff"] <= PCM_TOL and r["pcm_max_abs_diff"] == 0
    assert r["latency_us"]["p99"] < 1000.0, r["latency_us"]


def test_launches_on_different_hip_streams_are_ordered():
    """A handle's launches share its scratch planes.  Calls queued back to back on DIFFERENT HIP streams with no host
    synchronisation in between (ranges of one handle, and whole-handle calls) must not trample each other: the library orders
    them with an event."""
    t = torch_mod()
    S, T, nf, nb = 4096, 2, 480, 150
    base = synth.make_pcm(64, 3 * T, nf, 48000, seed=23)
    pcm = np.tile(base, (S // 64, 1, 1))
    ref_b = O.encode_batch(base, nb)
    ref_p = O.decode_batch(ref_b, nf)
    enc, dec = pkg.Lc3Encoder(S, 10000, 48000), pkg.Lc3Decoder(S, 10000, 48000)
    streams = [t.cuda.Stream() for _ in range(3)]
    d_pcm = t.from_numpy(pcm).cuda()
    outs = []
    t.cuda.synchronize()
    for j, st in enumerate(streams):  # three consecutive calls, three streams, nothing waited for
        x = d_pcm[:, j * T:(j + 1) * T].contiguous()
        d_b = t.zeros((S, T, nb), dtype=t.uint8, device="cuda")
        d_p = t.zeros((S, T, nf), dtype=t.int16, device="cuda")
        t.cuda.synchronize()  # the input copies above ran on the default stream
        with t.cuda.stream(st):
            enc.encode(x, d_b, nb, T, stream=st.cuda_stream)
            dec.decode(d_b, d_p, nb, T, stream=st.cuda_stream)
        outs.append((x, d_b, d_p))
    # and two half ranges of the next... not supported mid-stream here: the ranges variant below uses fresh handles
    t.cuda.synchronize()
    got_b = np.concatenate([b.cpu().numpy() for _, b, _ in outs], axis=1)
    got_p = np.concatenate([p.cpu().numpy() for _, _, p in outs], axis=1)
    for r in range(0, S, 64):
        assert np.array_equal(got_b[r:r + 64], ref_b), r
        assert np.array_equal(got_p[r:r + 64], ref_p), r
    # channel ranges of one handle on two streams, queued without a wait in between
    enc2 = pkg.Lc3Encoder(S, 10000, 48000)
    halves = []
    x_all = d_pcm[:, :T].contiguous()
    t.cuda.synchronize()
    for j, st in enumerate(streams[:2]):
        lo = j * (S // 2)
        d_b = t.zeros((S // 2, T, nb), dtype=t.uint8, device="cuda")
        t.cuda.synchronize()
        with t.cuda.stream(st):
            enc2.encode(x_all[lo:lo + S // 2], d_b, nb, T, stream=st.cuda_stream, first_channel=lo, n_channels=S // 2)
        halves.append(d_b)
    t.cuda.synchronize()
    got = np.concatenate([h.cpu().numpy() for h in halves], axis=0)
    for r in range(0, S, 64):
        assert np.array_equal(got[r:r + 64], ref_b[:, :T]), r


def test_state_blob_size_is_checked():
    enc, dec = pkg.Lc3Encoder(3, US, FS), pkg.Lc3Decoder(3, US, FS)
    small_e, small_d = pkg.Lc3Encoder(2, US, FS).state_save(), pkg.Lc3Decoder(2, US, FS).state_save()
    with pytest.raises(ValueError):
        enc.state_load(small_e)
    with pytest.raises(ValueError):
        dec.state_load(small_d)
    import ctypes
    L = pkg.load_library()
    assert L.lc3gpu_encoder_state_load(enc._h, small_e.ctypes.data_as(ctypes.c_void_p), small_e.size) == -3
    assert L.lc3gpu_decoder_state_load(dec._h, small_d.ctypes.data_as(ctypes.c_void_p), small_d.size) == -3


def test_state_blobs_say_what_they_are():
    """a channel's blob carries a header (side, layout version, payload size, rate, frame duration, switches): a blob of equal SIZE from a
    handle of another configuration, from a mixed handle with another descriptor order, from an encoder with other switches, or with a
    damaged header is refused (LC3GPU_EINVAL = -1) instead of being taken as state; a matching one loads and continues bit-exactly"""
    e48, e32 = pkg.Lc3Encoder(2, US, FS), pkg.Lc3Encoder(2, 10000, 32000)
    d48, d48_75 = pkg.Lc3Decoder(2, US, FS), pkg.Lc3Decoder(2, 7500, 48000)
    be, bd = e48.state_save(), d48.state_save()
    assert be[:4].tobytes() == b"LC3E" and bd[:4].tobytes() == b"LC3D"
    for handle, blob, err in ((e32, be, pkg.Lc3EncoderError), (d48_75, bd, pkg.Lc3DecoderError),
                              (pkg.Lc3Encoder(2, US, FS, spec_flags=2), be, pkg.Lc3EncoderError)):
        with pytest.raises(err) as ex:
            handle.state_load(blob)
        assert ex.value.code == -1
    bad = be.copy()
    bad[4] ^= 1  # layout version
    with pytest.raises(pkg.Lc3EncoderError):
        e48.state_load(bad)
    e48.state_load(be)  # the matching blob is fine
    a = pkg.Lc3Encoder.mixed([(48000, 10000, 100), (32000, 10000, 80)])
    b = pkg.Lc3Encoder.mixed([(32000, 10000, 80), (48000, 10000, 100)])
    with pytest.raises(pkg.Lc3EncoderError):
        b.state_load(a.state_save())
    a.state_load(a.state_save())


def test_runtime_configuration_view_of_the_headline_configuration():
    """Every standard configuration normally runs kernels instantiated with the configuration as compile-time constants
    (lc3_cfg_views.h: twelve views in the multi-unit library, four in a whole-source build); LC3GPU_GENERIC=1 keeps a process on the
    run-time view, the fall-back for whatever has no view.  Both must give the oracle's bytes and PCM (a fresh process: the switch is
    read once per process) -- here every frame length the run-time view's transforms pick a plan for, a mixed batch and the 8 kHz decoder."""
    import os
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, 'tests')\n"
        "import test_gpu_parity as t\n"
        "for nb in (150, 60, 300):\n"
        "    t._roundtrip_check(48000, 10000, nb, 96, 6, seed=61)\n"
        "for fs, us, nb in ((48000, 7500, 113), (32000, 10000, 80), (16000, 10000, 40), (44100, 10000, 110), (24000, 10000, 60),\n"
        "                   (32000, 7500, 61), (24000, 7500, 45), (16000, 7500, 30), (44100, 7500, 83)):\n"
        "    t._roundtrip_check(fs, us, nb, 64, 6, seed=62)\n"
        "t.test_decode_8khz(10000, 30)\n"
        "t.test_decode_8khz(7500, 23)\n"
        "t.test_mixed_configuration_batch()\n"
        "t.test_corrupt_frames_are_concealed_like_the_reference()\n"
        "t.test_ltpf_transitions(48000, 10000, 40)\n"
        "print('generic ok')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LC3GPU_GENERIC="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "generic ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_sequential_sum_path_of_guarded_decisions():
    """The gain bisection and the noise level only need the SIDE of a threshold their f32 sums fall on: the kernels decide from a
    tree sum when that is further from the threshold than the rounding of the reference's sequential sum can reach and repeat
    the sum in the reference's order otherwise (about one decision in 10^4).  LC3GPU_SEQ_SUMS=1 sends every decision down the
    sequential path; both must give the oracle's bytes (the default path is what every other test runs)."""
    import os
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, 'tests')\n"
        "import test_gpu_parity as t\n"
        "for nb in (150, 40, 300):\n"
        "    t._roundtrip_check(48000, 10000, nb, 96, 6, seed=71)\n"
        "for fs, us, nb in ((48000, 7500, 113), (24000, 10000, 60), (16000, 7500, 30)):\n"
        "    t._roundtrip_check(fs, us, nb, 64, 6, seed=72)\n"
        "t.test_mixed_configuration_batch()\n"
        "print('seq ok')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LC3GPU_SEQ_SUMS="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "seq ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_prepared_packer_symbols_on_and_off():
    """Launches of at most 16 384 frames have the analysis kernel prepare the packer's symbol words (lc3_enc_symbols), larger ones
    leave the derivation to the packer; the third form, the preparation as a wave-per-frame kernel of its own between back half and
    packer (lc3_symbols_kernel, LC3GPU_PREP_SYMBOLS=2), is selectable.  All three forms on the same launches, small and large
    (LC3GPU_PREP_SYMBOLS=0 / 1 / 2 overrides the size rule), against the oracle."""
    import os
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, 'tests')\n"
        "import test_gpu_parity as t\n"
        "for nb in (150, 40, 300):\n"
        "    t._roundtrip_check(48000, 10000, nb, 96, 6, seed=81)\n"
        "for fs, us, nb in ((48000, 7500, 113), (24000, 10000, 60), (16000, 7500, 30), (32000, 10000, 400)):\n"
        "    t._roundtrip_check(fs, us, nb, 64, 6, seed=82)\n"
        "t._roundtrip_check(48000, 10000, 150, 2048, 2, seed=83)\n"
        "t.test_mixed_configuration_batch()\n"
        "for nb in (20, 25, 50, 80, 100, 140, 141, 160, 260, 399):\n"
        "    t.test_bitrate_sweep_48k(nb)\n"
        "print('prep ok')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (form 0, the packer deriving its symbols itself: as producer / consumer wave pairs -- the default of full batches -- and as one wave
    # per 64 frames)
    for v, pc in (("0", "1"), ("0", "0"), ("1", "1"), ("2", "1")):
        env = dict(os.environ, LC3GPU_PREP_SYMBOLS=v, LC3GPU_PACK_PC=pc)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "prep ok" in r.stdout, v + pc + r.stdout[-2000:] + r.stderr[-2000:]


def test_producer_consumer_kernels_at_every_size():
    """Full batches parse and pack with producer / consumer wave pairs (lc3_parse_pc_kernel, lc3_pack_pc_kernel); launches of up to 16 384
    frames normally take the small-launch forms.  LC3GPU_RECON=lane + LC3GPU_PREP_SYMBOLS=0 put EVERY launch on the pair kernels: partial
    waves and workgroups, one-frame launches, the interleaved layout, channel ranges, state carry, damaged and flagged frames, garbage,
    every rate and duration, the smallest and the largest frames (fewer frames per workgroup), and 64 / 128 frames per workgroup
    (LC3GPU_FPB: a pair's waves then sit on different SIMDs)."""
    import os
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, 'tests')\n"
        "import test_gpu_parity as t\n"
        "for S in (1, 2, 3, 5, 63, 64, 65, 127, 129, 300):\n"
        "    t._roundtrip_check(48000, 10000, 150, S, 3, seed=401 + S)\n"
        "for nb in (20, 25, 40, 100, 200, 300, 400):\n"
        "    t._roundtrip_check(48000, 10000, nb, 70, 5, seed=402)\n"
        "for fs, us, nb in ((48000, 7500, 113), (44100, 10000, 110), (32000, 7500, 60), (24000, 10000, 60), (16000, 7500, 30), (16000, 10000, 40)):\n"
        "    t._roundtrip_check(fs, us, nb, 96, 5, seed=403)\n"
        "t.test_decode_8khz(10000, 30)\n"
        "t.test_decode_8khz(7500, 23)\n"
        "t.test_interleaved_layout(48000, 10000, 150, 2, 9)\n"
        "t.test_interleaved_layout(32000, 10000, 81, 70, 3)\n"
        "t.test_interleaved_layout(16000, 10000, 40, 5, 4)\n"
        "t.test_channel_ranges_on_separate_hip_streams()\n"
        "t.test_state_carry_across_launches()\n"
        "t.test_state_save_load_and_reset()\n"
        "t.test_variable_bitrate_per_call()\n"
        "t.test_corrupt_frames_are_concealed_like_the_reference()\n"
        "t.test_random_garbage_streams()\n"
        "t.test_bad_frame_flag_forces_concealment()\n"
        "t.test_cold_start_many_streams()\n"
        "t.test_long_stream_state_carry()\n"
        "t.test_ltpf_transitions(48000, 10000, 60)\n"
        "t.test_kat_encode_frame(); t.test_kat_decode_frame()\n"
        "t.test_mixed_configuration_batch()\n"          # the pair kernels of mixed handles (a body per configuration view)
        "t.test_mixed_batch_bad_frames_and_plc_counter()\n"
        "print('pairs ok')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for fpb in ("256", "128", "64"):
        env = dict(os.environ, LC3GPU_RECON="lane", LC3GPU_PREP_SYMBOLS="0", LC3GPU_PARSE_PC="1", LC3GPU_PACK_PC="1", LC3GPU_FPB=fpb)
        env.pop("LC3GPU_LATE_RECON", None)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0 and "pairs ok" in r.stdout, fpb + r.stdout[-2000:] + r.stderr[-2000:]


def _split_suite():
    """what test_split_calls_on_and_off runs in a child process with LC3GPU_SPLIT=1 (every batch call of at least 16 streams runs as two
    halves on the handle's internal HIP streams) and with LC3GPU_SPLIT=0 (never): every result against the oracle"""
    t = torch_mod()
    # odd stream counts (the halves are whole workgroups of four streams / whole waves of 64), several configurations
    for S in (16, 17, 19, 23, 130, 257):
        _roundtrip_check(48000, 10000, 150, S, 3, seed=301 + S)
    for fs, us, nb in ((48000, 7500, 113), (32000, 10000, 80), (16000, 7500, 30), (24000, 10000, 60)):
        _roundtrip_check(fs, us, nb, 37, 4, seed=302)
    # state carried across split launches, then across a launch of a size that does not split, through save / load and a reset
    S, T = 41, 6
    pcm = synth.make_pcm(S, T, 480, 48000, seed=303)
    ref = O.encode_batch(pcm, 150)
    ref_pcm = O.decode_batch(ref, 480)
    enc, dec = pkg.Lc3Encoder(S, US, FS), pkg.Lc3Decoder(S, US, FS)
    got = [gpu_encode(pcm[:, a:b], 150, enc=enc) for a, b in ((0, 2), (2, 3), (3, 6))]
    assert np.array_equal(np.concatenate(got, axis=1), ref)
    blob = enc.state_save()
    enc.reset()
    assert np.array_equal(gpu_encode(pcm[:, :2], 150, enc=enc), ref[:, :2])
    enc.state_load(blob)
    gotp = [gpu_decode(ref[:, a:b], 480, dec=dec) for a, b in ((0, 1), (1, 4), (4, 6))]
    assert np.array_equal(np.concatenate(gotp, axis=1), ref_pcm)
    # channel ranges of the same handles (ranges of at least 16 streams split again), state still the handle's
    enc2, dec2 = pkg.Lc3Encoder(S, US, FS), pkg.Lc3Decoder(S, US, FS)
    d_pcm = t.from_numpy(pcm).cuda()
    d_b = t.zeros((S, T, 150), dtype=t.uint8, device="cuda")
    d_p = t.zeros((S, T, 480), dtype=t.int16, device="cuda")
    for lo, n in ((0, 20), (20, 21)):
        enc2.encode(d_pcm[lo:lo + n], d_b[lo:lo + n], 150, T, first_channel=lo, n_channels=n)
        dec2.decode(d_b[lo:lo + n], d_p[lo:lo + n], 150, T, first_channel=lo, n_channels=n)
    t.cuda.synchronize()
    assert np.array_equal(d_b.cpu().numpy(), ref) and np.array_equal(d_p.cpu().numpy(), ref_pcm)
    # damaged frames and bad-frame flags on both sides of the split point, also at the edges of launches; the PLC counter
    S, T = 36, 8
    pcm = synth.make_pcm(S, T, 480, 48000, seed=304)
    data = O.encode_batch(pcm, 150).copy()
    rng = np.random.default_rng(304)
    bad = np.zeros((S, T), np.uint8)
    for s in range(S):
        for tt in rng.choice(T, 2, replace=False):
            if s % 3 == 0:
                data[s, tt] = rng.integers(0, 256, 150, dtype=np.uint8)
            elif s % 3 == 1:
                bad[s, tt] = 1
            else:
                data[s, tt, -1] |= 7
    corrupt = data.copy()
    corrupt[bad.astype(bool), -1] |= 7
    refp = O.decode_batch(corrupt, 480)
    dec3 = pkg.Lc3Decoder(S, US, FS)
    gp = [gpu_decode(data[:, a:b], 480, dec=dec3, bad=bad[:, a:b]) for a, b in ((0, 3), (3, 4), (4, 8))]
    assert np.array_equal(np.concatenate(gp, axis=1), refp)
    assert dec3.plc_events() >= 2 * (S // 3)
    # the interleaved layout (the halves are column ranges of the same buffers)
    test_interleaved_layout(32000, 10000, 81, 70, 3)
    test_interleaved_layout(48000, 10000, 150, 18, 4)
    # calls of one handle on different HIP streams, the per-kernel timer
    test_launches_on_different_hip_streams_are_ordered()
    test_kernel_timer_counts_every_nth_call()
    test_channel_ranges_on_separate_hip_streams()
    print("split ok")


def test_split_calls_on_and_off():
    """LC3GPU_SPLIT=1 (opt-in: measured slower than one launch per kernel, lc3_split_parts in lc3gpu.hip) runs every batch call of at
    least 16 streams as two halves of its streams on the handle's two internal HIP streams, forked from and joined to the caller's stream
    by events; LC3GPU_SPLIT=0 / unset never does.  Both ways through state carry, state blobs, ranges, damaged frames at launch edges, odd
    stream counts, the interleaved layout, cross-stream ordering and the timer, against the oracle; and the split and the unsplit form
    of the SAME full-size launches byte for byte."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, 'tests')\nimport test_gpu_parity as t\nt._split_suite()\n"
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, LC3GPU_SPLIT=v), capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0 and "split ok" in r.stdout, v + r.stdout[-2000:] + r.stderr[-2000:]
    # the same full-size launches (two consecutive calls: state, and a damaged frame per 97 in the second) in both forms
    code2 = (
        "import sys, hashlib; sys.path.insert(0, 'tests')\n"
        "import numpy as np, test_gpu_parity as t\n"
        "S, T = 8192 + 4, 4\n"
        "pcm = np.tile(t.synth.make_pcm(683, 2 * T, 480, 48000, seed=305), (13, 1, 1))[:S]\n"
        "enc, dec = t.pkg.Lc3Encoder(S, t.US, t.FS), t.pkg.Lc3Decoder(S, t.US, t.FS)\n"
        "h = hashlib.sha256()\n"
        "for k in range(2):\n"
        "    b = t.gpu_encode(pcm[:, k * T:(k + 1) * T], 150, enc=enc)\n"
        "    if k: b.reshape(-1, 150)[::97, -1] |= 7\n"
        "    p = t.gpu_decode(b, 480, dec=dec)\n"
        "    h.update(b.tobytes()); h.update(p.tobytes())\n"
        "print('digest', h.hexdigest(), dec.plc_events())\n"
    )
    digests = []
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code2], cwd=root, env=dict(os.environ, LC3GPU_SPLIT=v), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "digest" in r.stdout, v + r.stdout[-2000:] + r.stderr[-2000:]
        digests.append(r.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1], digests


def test_stress_parity_tool_one_million_frames():
    """tools/stress_parity.py at volume inside the suite: (14 configurations in both directions + 3 decode-only 8 kHz ones) x 2048 streams x
    18 frames x 2 rounds = 1 253 376 frames, every bitstream byte and every PCM sample compared with the oracle (which runs on the host threads the job is granted);
    one frame in 48 of the decode direction is damaged first (bit flips, random bytes, bad-frame flags), so concealment, the parser's
    error paths and the frames that follow a lost one are part of the volume.  (The guarded decisions of the quantiser fall back to
    their sequential sum about once in 10^4 decisions: it takes this many frames to see both sides of every guard on real data.)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_parity.py"), "--streams", "2048", "--frames", "18", "--rounds", "2"],
                       cwd=root, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["frames_differing"] == 0 and line["total_frames_each_direction"] == 17 * 2048 * 18 * 2 >= 1000000
    k8 = [c for c in line["cases"] if c["fs_hz"] == 8000]
    assert len(k8) == 3 and all(c["directions"].startswith("decode only") and c["decode_frames_damaged"] > 500 for c in k8)
    assert line["pair_timeouts"] == 0  # no producer / consumer pair ever gave up on its partner
    assert line["frames_damaged"] > 10000
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "stress_parity_suite.json"), "w") as f:
        f.write(r.stdout.strip().splitlines()[-1] + "\n")


def test_late_reconstruction_on_and_off():
    """Launches of at most 16 384 frames and four frames per stream rebuild the spectrum in the synthesis kernel with the wave's 64 lanes
    (lc3_dec_reconstruct_wave), larger ones in a wave-per-frame kernel of its own (lc3_recon_kernel); the round-1/2 form (one lane per
    frame inside the parse kernel, lc3_reconstruct_frame) stays selectable.  All three forms on the same launches
    (LC3GPU_RECON=lane|late|wave overrides the size rule): clean, corrupted, flagged and garbage streams."""
    import os
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, 'tests')\n"
        "import test_gpu_parity as t\n"
        "for nb in (150, 40, 300, 20, 400):\n"
        "    t._roundtrip_check(48000, 10000, nb, 96, 6, seed=91)\n"
        "for fs, us, nb in ((48000, 7500, 113), (24000, 10000, 60), (16000, 7500, 30), (32000, 10000, 80), (44100, 10000, 100)):\n"
        "    t._roundtrip_check(fs, us, nb, 64, 6, seed=92)\n"
        "t._roundtrip_check(48000, 10000, 150, 2048, 2, seed=93)\n"
        "t.test_corrupt_frames_are_concealed_like_the_reference()\n"
        "t.test_bad_frame_flag_forces_concealment()\n"
        "t.test_random_garbage_streams()\n"
        "t.test_mixed_configuration_batch()\n"
        "t.test_mixed_batch_bad_frames_and_plc_counter()\n"
        "t.test_ltpf_transitions(48000, 10000, 40)\n"
        "t.test_ltpf_transitions(16000, 10000, 40)\n"
        "assert len(t._mixed_bandwidth_check()) >= 4\n"
        "assert len(t._mixed_bandwidth_check(nbytes=60, S=70, T=3, seed=96)) >= 3\n"
        "print('late ok')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (lane form: the parser as producer / consumer wave pairs, the default of full batches, and as one wave per 64 frames)
    for v, pc in (("lane", "1"), ("lane", "0"), ("late", "1"), ("wave", "1")):
        env = dict(os.environ, LC3GPU_RECON=v, LC3GPU_PARSE_PC=pc)
        env.pop("LC3GPU_LATE_RECON", None)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "late ok" in r.stdout, v + pc + r.stdout[-2000:] + r.stderr[-2000:]


# ---------------------------------------------------------------- spec-conformant switches
def _gpu_encode_spec(pcm, nbytes, fs, us, flags):
    torch = torch_mod()
    S, T, nf = pcm.shape
    enc = pkg.Lc3Encoder(S, us, fs, spec_flags=flags)
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    d_out = torch.zeros((S, T, nbytes), dtype=torch.uint8, device="cuda")
    enc.encode(d_pcm, d_out, nbytes, T, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


@pytest.mark.parametrize("fs,nf,nbytes,cutoff", [(48000, 480, 100, 11800.0), (32000, 320, 80, 11500.0), (48000, 480, 150, 7500.0)])
def test_spec_conformant_switches(fs, nf, nbytes, cutoff):
    """Every LC3GPU_SPEC_* bit alone and all together against the oracle with the same switch; default (0) unchanged and equal
    to the reference's behaviour.  Where a switch bites on this input is asserted over the three parameter sets together by
    test_spec_switches_bite; here each must simply match."""
    pcm = synth.make_bandlimited_pcm(48, 8, nf, fs, cutoff)
    ref0 = O.encode_batch(pcm, nbytes, fs, 10000, threads=8)
    assert np.array_equal(_gpu_encode_spec(pcm, nbytes, fs, 10000, 0), ref0)
    for flag in (2, 4, 8, 16, 2 | 4 | 8 | 16, 31):
        ref = O.encode_batch(pcm, nbytes, fs, 10000, threads=8, spec_flags=flag)
        assert np.array_equal(_gpu_encode_spec(pcm, nbytes, fs, 10000, flag), ref), flag


def test_spec_switches_bite():
    """each switch changes the oracle's output somewhere on the test inputs (otherwise the test above proves nothing)"""
    hit = {2: 0, 4: 0, 8: 0, 16: 0}
    for fs, nf, nbytes, cutoff in [(48000, 480, 100, 11800.0), (32000, 320, 80, 11500.0), (48000, 480, 150, 7500.0)]:
        pcm = synth.make_bandlimited_pcm(48, 8, nf, fs, cutoff)
        ref0 = O.encode_batch(pcm, nbytes, fs, 10000, threads=8)
        for flag in hit:
            hit[flag] += int((O.encode_batch(pcm, nbytes, fs, 10000, threads=8, spec_flags=flag) != ref0).any(axis=2).sum())
    assert all(v > 0 for v in hit.values()), hit


def test_8khz_encode_switch_and_mixed_handle_with_switches():
    t = torch_mod()
    with pytest.raises(pkg.Lc3EncoderError) as e:
        pkg.Lc3Encoder(1, 10000, 8000)
    assert e.value.code == -7  # default: as the reference, no 8 kHz encoder
    with pytest.raises(pkg.Lc3EncoderError):
        pkg.Lc3Encoder(1, 10000, 48000, spec_flags=64)  # unknown bit
    for us, nb, nf in ((10000, 30, 80), (7500, 23, 60)):
        pcm = synth.make_pcm(40, 8, nf, 8000, seed=9)
        ref = O.encode_batch(pcm, nb, 8000, us, threads=8, spec_flags=1)
        assert np.array_equal(_gpu_encode_spec(pcm, nb, 8000, us, 1), ref)
        assert np.array_equal(gpu_decode(ref, nf, 8000, us), O.decode_batch(ref, nf, 8000, us, threads=8))
    # a mixed handle with all switches on, 8 kHz streams included
    descs = [(8000, 10000, 30), (48000, 10000, 100), (8000, 7500, 23), (32000, 10000, 80)] * 6
    enc = pkg.Lc3Encoder.mixed(descs, spec_flags=31)
    T = 5
    pcms = []
    for i, (fs, us, nb) in enumerate(descs):
        nf = pkg.Lc3Config(fs, us).nf
        pcms.append(synth.make_bandlimited_pcm(1, T, nf, fs, min(11800.0, fs * 0.45), seed=100 + i)[0] if fs > 8000
                    else synth.make_pcm(1, T, nf, fs, seed=100 + i)[0])
    d_pcm = t.from_numpy(np.concatenate([p.reshape(-1) for p in pcms])).cuda()
    d_b = t.zeros(sum(T * d[2] for d in descs), dtype=t.uint8, device="cuda")
    enc.encode_mixed(d_pcm, d_b, T, stream=t.cuda.current_stream().cuda_stream)
    got = d_b.cpu().numpy()
    off = 0
    for i, (fs, us, nb) in enumerate(descs):
        ref = O.encode_batch(pcms[i][None], nb, fs, us, spec_flags=31)[0]
        assert np.array_equal(got[off:off + T * nb].reshape(T, nb), ref), (i, fs, us)
        off += T * nb


# ---------------------------------------------------------------- the production launcher on the GPU
def _bench_line(extra, env_extra, timeout=900):
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", *extra], capture_output=True, text=True,
                       timeout=timeout, env=dict(os.environ, **env_extra))
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_bench_encode_only_mode():
    """`bench.py --mode encode` (BASELINE config 3 at a small size): the pipeline object's encode-only form (lc3gpu_pipeline_encode: the
    groups' encoder chains side by side, two byte buffers) by default, one handle on one caller stream with `--arrangement single`; the
    parity gate compares the bitstream only"""
    p, line = _bench_line(["--mode", "encode", "--frames-total", "16384", "--steps", "3", "--warmup", "1", "--sustain-seconds", "0.2", "--arrangement", "single"], {})
    assert p.returncode == 0 and line is not None, p.stderr[-2000:]
    assert "encode" in line["metric"] and line["config"]["mode"] == "encode" and line["config"]["arrangement"] == "single"
    assert line["parity"]["bitstream_exact"] and line["parity_mismatches_all_ranks"] == 0 and line["kernel_ms"]["lc3_pack_kernel"] > 0.0
    p, line = _bench_line(["--mode", "encode", "--frames-total", "65536", "--steps", "3", "--warmup", "1", "--sustain-seconds", "0.2"], {})
    assert p.returncode == 0 and line is not None, p.stderr[-2000:]
    assert "encode" in line["metric"] and line["config"]["mode"] == "encode" and line["config"]["arrangement"] == "pipeline"
    assert line["parity"]["bitstream_exact"] and line["parity_mismatches_all_ranks"] == 0 and line["other_arrangements"] == []
    assert line["kernel_ms"]["lc3_pack_kernel"] > 0.0 and line["sustained"]["steps"] > 0


def test_two_ranks_through_the_launcher_share_the_gpu():
    """`bench.py --gpus 2` on a one-GPU box: the production launcher (fresh child processes, rendezvous on 127.0.0.1), the stream
    sharding, the GPU engine and the final reduction together; the ranks share the visible device and reduce over gloo (RCCL needs
    one GPU per rank -- LC3_BENCH_BACKEND=gloo is the only thing that differs from the driver's multi-GPU invocation)."""
    p, line = _bench_line(["--gpus", "2", "--steps", "3", "--warmup", "1", "--streams", "2048"], {"LC3_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0 and line is not None, p.stderr[-2000:]
    assert line["n_gpus"] == 2 and "world size 2 (gloo)" in line["config"]["parallelism"] and line["config"]["engine"] == "gpu"
    assert line["scaling"] == "weak" and line["config"]["frames_per_step_per_gpu"] == 2048 * 4
    assert line["parity"]["bitstream_exact"] and line["parity"]["pcm_max_abs_diff"] == 0 and line["parity_mismatches_all_ranks"] == 0
    assert abs(line["value"] - 2 * 2048 * 4 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]  # frames of BOTH ranks / max time
    assert line["kernel_ms"]["lc3_enc_front_kernel"] > 0.0


def test_launcher_stops_the_gpu_ranks_when_one_dies():
    import time

    t0 = time.time()
    p, line = _bench_line(["--gpus", "2", "--steps", "2", "--warmup", "0", "--streams", "256"],
                          {"LC3_BENCH_BACKEND": "gloo", "LC3_BENCH_TEST_DIE_RANK": "1"})
    assert p.returncode != 0 and line is None and "rank 1 exited with code 17" in p.stderr
    assert time.time() - t0 < 300.0


def test_rccl_world_size_one():
    """RCCL itself on the hardware: the pool hands out one GPU per call, so the N-GPU path's library load, communicator creation, barrier
    and the all_reduce of the report counters on DEVICE tensors (lc3-codec_amd/dist.py::reduce_report) run here as a group of ONE rank --
    in a fresh child process that initialises the `nccl` backend before any other GPU call, exactly where a rank of the driver's 8-GPU
    run does (bench.py, LC3_BENCH_RCCL=1).  The line says which backend reduced it and the RCCL version."""
    p, line = _bench_line(["--steps", "3", "--warmup", "1", "--streams", "1024", "--no-overlap-probe", "--sustain-seconds", "0"],
                          {"LC3_BENCH_RCCL": "1"})
    assert p.returncode == 0 and line is not None, p.stderr[-2000:]
    par = line["config"]["parallelism"]
    assert "world size 1 (nccl" in par and "RCCL" in par, par
    assert line["n_gpus"] == 1 and line["parity"]["bitstream_exact"] and line["parity_mismatches_all_ranks"] == 0
    assert abs(line["value"] - 1024 * 4 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]


def test_default_arrangement_is_the_pipeline_object_and_a_mismatch_fails_the_run():
    """bench.py without flags times the library's pipeline object (lc3gpu_pipeline_submit: two groups, four HIP streams), says so in the
    line, walks through the resident frames from step to step, and gates parity on streams of both groups; a run whose gate finds a
    difference exits non-zero after printing the line (round-4 review: it used to exit 0)."""
    p, line = _bench_line(["--steps", "3", "--warmup", "1", "--streams", "4096", "--sustain-seconds", "0", "--no-overlap-probe"], {})
    assert p.returncode == 0 and line is not None, p.stderr[-2000:]
    assert line["config"]["arrangement"] == "pipeline" and line["config"]["hip_streams"] == 4
    assert line["parity"]["arrangement"] == "pipeline" and line["parity"]["bitstream_exact"] and line["parity"]["pcm_max_abs_diff"] == 0
    assert line["config"]["resident_frames_per_stream"] == 64 and line["config"]["resident_pcm_bytes_per_gpu"] == 4096 * 64 * 960
    assert line["roofline"]["bound"] == "valu-issue" and line["parity"]["frames_checked"] >= 1000
    # LC3_BENCH_TEST_CORRUPT_GATE=1: the gate's reference bytes are damaged on purpose -> the line reports mismatches, the exit code is 3
    p, line = _bench_line(["--steps", "2", "--warmup", "1", "--streams", "1024", "--sustain-seconds", "0", "--no-overlap-probe"],
                          {"LC3_BENCH_TEST_CORRUPT_GATE": "1"})
    assert line is not None and line["parity_mismatches_all_ranks"] > 0 and not line["parity"]["bitstream_exact"]
    assert p.returncode == 3, p.returncode


def test_every_arrangement_produces_the_single_stream_output():
    """bench.py's engine, full size (65 536 frames per step): three steps from fresh state in every caller arrangement -- one stream, two
    streams, three buffers + stage event, two and three groups of streams with handle pairs of their own -- leave the SAME bitstream and
    the SAME PCM for all 16 384 streams as the one-stream arrangement does (state carried across the steps, every buffer and event of the
    arrangement in play), and the first 64 streams of that agree with the oracle."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    torch_mod()
    S, T = 16384, 4
    args = bench.parse_args(["--no-cpu-baseline"])
    pcm = synth.make_pcm(S, T, bench.NF, bench.FS)
    eng = bench.GpuEngine(args, pcm, S, T, "roundtrip", 0)
    ref = None
    for arr in ("single", "pipeline", "pipelined", "staggered", "split:2+2", "split:1+1", "split:2+1+1"):
        eng.set_arrangement(arr)
        eng.reset()
        for _ in range(3):
            eng.step()
        eng.sync()
        got = (eng.last_bytes_all().cpu().numpy(), eng.d_out.cpu().numpy())
        if ref is None:
            ref = got
            three = np.ascontiguousarray(np.concatenate([pcm[:64]] * 3, axis=1))
            want_b = O.encode_batch(three, bench.NBYTES, threads=8)
            assert np.array_equal(got[0][:64], want_b[:, 2 * T:])
            assert np.array_equal(got[1][:64], O.decode_batch(want_b, bench.NF, threads=8)[:, 2 * T:])
        else:
            assert np.array_equal(got[0], ref[0]), arr + ": bitstream differs from the one-stream arrangement"
            assert np.array_equal(got[1], ref[1]), arr + ": PCM differs from the one-stream arrangement"
        for h in sum(eng._handles(), []):
            assert h.pair_timeouts() == 0


def test_every_caller_arrangement_passes_its_parity_gate():
    """bench.py times two ways of queueing the same steps -- encode then decode on ONE caller stream, and the recommended pattern
    (INTEGRATION.md): encoder handle on one stream, decoder handle on another, two byte buffers, events -- and runs its parity gate on
    each (two steps from fresh state, the second against the oracle).  Here on a 32 768-frame batch (producer / consumer pair kernels)."""
    p, line = _bench_line(["--steps", "4", "--warmup", "1", "--streams", "8192", "--sustain-seconds", "0.3", "--arrangement", "pipelined", "--also", "quad",
                           "--also", "duo", "--also", "pipeline"], {})
    assert p.returncode == 0 and line is not None, p.stderr[-2000:]
    assert line["config"]["arrangement"] == "pipelined" and line["config"]["hip_streams"] == 2
    assert line["parity"]["arrangement"] == "pipelined" and line["parity"]["bitstream_exact"] and line["parity"]["pcm_max_abs_diff"] == 0
    o = line["other_arrangement"]
    assert o["arrangement"] == "single" and o["parity"]["bitstream_exact"] and o["parity"]["pcm_max_abs_diff"] == 0
    # ... and the third one: like `pipelined`, the decoder call queued behind the encoder's LC3GPU_ENC_STAGE_BACK event of the NEXT step
    g = [a for a in line["other_arrangements"] if a["arrangement"] == "staggered"]
    assert len(g) == 1 and g[0]["parity"]["bitstream_exact"] and g[0]["parity"]["pcm_max_abs_diff"] == 0 and g[0]["hip_streams"] == 2
    # ... and the split arrangements (round 5): the streams in groups, every group with a handle pair of its own -- on two streams
    # (`quad` = split:2+2, what the library's pipeline object builds inside) or one (`duo` = split:1+1), and the pipeline object itself; their
    # gates look at streams of EVERY group
    for name, n_streams in (("split:2+2", 4), ("split:1+1", 2), ("pipeline", 4)):
        g = [a for a in line["other_arrangements"] if a["arrangement"] == name]
        assert len(g) == 1 and g[0]["parity"]["bitstream_exact"] and g[0]["parity"]["pcm_max_abs_diff"] == 0, (name, g)
        assert g[0]["hip_streams"] == n_streams and g[0]["parity"]["frames_checked"] >= 256 * 4 - 8
    assert line["parity_mismatches_all_ranks"] == 0
    s = line["sustained"]
    assert s["steps"] > 0 and s["shader_clock_MHz"]["probes"] > 0 and 500.0 < s["shader_clock_MHz"]["median"] < 3000.0, s


def test_stage_events_order_another_stream_inside_a_call():
    """lc3gpu_encoder_stage_event / lc3gpu_decoder_stage_event: the caller's event is recorded behind the stage's kernels of every batch
    call.  A second stream that waits for the encoder's BACK event and then overwrites the PCM input must not disturb the call (the
    front half -- the only reader of the PCM -- has ended by then); the decoder's PARSE event likewise guards its byte input.  Unknown
    stages are rejected; a cleared slot records nothing."""
    t = torch_mod()
    S, T = 256, 4
    pcm = synth.make_pcm(S, T, 480, 48000, seed=77)
    ref = O.encode_batch(pcm, 150)
    ref_pcm = O.decode_batch(ref, 480)
    enc, dec = pkg.Lc3Encoder(S, US, FS), pkg.Lc3Decoder(S, US, FS)
    s0, s1 = t.cuda.Stream(), t.cuda.Stream()
    ev_f, ev_b, ev_p = t.cuda.Event(), t.cuda.Event(), t.cuda.Event()
    with pytest.raises(ValueError):
        enc.stage_event(pkg.ENC_STAGE_BACK, ev_b)  # torch has not created the HIP event yet
    for e in (ev_f, ev_b, ev_p):
        e.record(s0)
    with pytest.raises(pkg.Lc3EncoderError):
        enc.stage_event(3, ev_b)
    with pytest.raises(pkg.Lc3DecoderError):
        dec.stage_event(1, ev_p)
    enc.stage_event(pkg.ENC_STAGE_FRONT, ev_f)
    enc.stage_event(pkg.ENC_STAGE_BACK, ev_b)
    dec.stage_event(pkg.DEC_STAGE_PARSE, ev_p)
    d_pcm = t.from_numpy(pcm).cuda()
    d_bytes = t.zeros((S, T, 150), dtype=t.uint8, device="cuda")
    d_out = t.zeros((S, T, 480), dtype=t.int16, device="cuda")
    t.cuda.synchronize()
    for rep in range(3):
        d_pcm.copy_(t.from_numpy(pcm))
        t.cuda.synchronize()
        enc.reset()
        dec.reset()
        enc.encode(d_pcm, d_bytes, 150, T, stream=s0.cuda_stream)
        s1.wait_event(ev_f if rep == 1 else ev_b)
        with t.cuda.stream(s1):
            d_pcm.zero_()                      # legal once the front half is through
        s0.synchronize()
        s1.synchronize()
        assert np.array_equal(d_bytes.cpu().numpy(), ref), rep
        d_in = d_bytes.clone()
        t.cuda.synchronize()
        dec.decode(d_in, d_out, 150, T, stream=s0.cuda_stream)
        s1.wait_event(ev_p)
        with t.cuda.stream(s1):
            d_in.fill_(255)                    # legal once the parser is through
        s0.synchronize()
        s1.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), ref_pcm), rep
    assert ev_f.query() and ev_b.query() and ev_p.query()
    # cleared: the events are not touched again
    enc.stage_event(pkg.ENC_STAGE_FRONT, None)
    enc.stage_event(pkg.ENC_STAGE_BACK, None)
    dec.stage_event(pkg.DEC_STAGE_PARSE, None)
    d_pcm.copy_(t.from_numpy(pcm))
    enc.reset()
    enc.encode(d_pcm, d_bytes, 150, T, stream=s0.cuda_stream)
    s0.synchronize()
    assert np.array_equal(d_bytes.cpu().numpy(), ref)


# ---------------------------------------------------------------- decoder stages on the device against the reference's stage goldens
def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("form", [0, 1, 2])  # lane (full batches), late (single frames / small launches), wave (wave-per-frame kernels)
def test_decoder_stage_dumps_match_reference_goldens(form):
    """lc3gpu_decode_frame_debug on the reference's end-to-end decode KAT frame (decoder/lc3_decoder.rs:374-425): the same frame
    the reference's per-stage tests were dumped from, so every stage of the HIP path is compared with the reference's OWN vector for
    it, bit for bit: integers after the range decoder (noise_filling.rs:65-146 `spec_lines_int`), after residual bits + noise
    filling + global gain (the input of temporal_noise_shaping.rs:147-238), after TNS (its `spec_lines_expected` = the input of
    spectral_noise_shaping.rs:244-350), after SNS (its `spec_lines_expected`), PCM (`samples_out_expected`); IMDCT / LTPF outputs
    against the oracle's stage functions on the same frame."""
    from test_oracle_kats import f32, kat

    t = "decoder/lc3_decoder.rs::lc3_decode_channel"
    buf = np.array(kat(t, "buf_in"), np.uint8)
    dec = pkg.Lc3Decoder(1, US, FS)
    pcm, dbg = dec.decode_frame_debug(buf, recon_form=form)
    assert pcm.tolist() == kat(t, "samples_out_expected")
    ints = np.array(kat("decoder/noise_filling.rs::decode_noise_filling", "spec_lines_int", 0, np.int32))
    assert np.array_equal(dbg[0:400], ints.astype(np.float32)), "integers after the range decoder"
    sns_out = f32("decoder/spectral_noise_shaping.rs::spectral_noise_shaping_decode", "spec_lines_expected")
    assert np.array_equal(_bits(dbg[400:800]), _bits(sns_out)), "spectrum after SNS"
    if form != 2:  # the wave-per-frame kernels hand the TNS range over between two kernels: no separate dumps
        tns_in = f32("decoder/temporal_noise_shaping.rs::decode_test", "spec_lines")
        tns_out = f32("decoder/temporal_noise_shaping.rs::decode_test", "spec_lines_expected")
        assert np.array_equal(_bits(dbg[1760:2160]), _bits(tns_in)), "after residual bits, noise filling, global gain"
        assert np.array_equal(_bits(dbg[2160:2560]), _bits(tns_out)), "after TNS"
    # IMDCT and LTPF of this frame: the oracle's stage functions fed with the reference's SNS output
    import ctypes

    od = O.Decoder()
    freq = np.zeros(480, np.float32)
    O.lib().lc3o_kat_imdct(od.h, O.P(sns_out), O.P(freq))
    assert np.array_equal(_bits(dbg[800:1280]), _bits(freq)), "IMDCT output"
    si = np.zeros(20, np.int64)
    tail = ctypes.c_int()
    assert O.lib().lc3o_kat_side_info(O.P(buf), 150, 4, 400, O.P(si), ctypes.byref(tail)) == 0
    O.lib().lc3o_kat_dec_ltpf(od.h, int(si[17]), int(si[16]), int(si[18]), 1200, O.P(freq))
    assert np.array_equal(_bits(dbg[1280:1760]), _bits(freq)), "LTPF output"


def test_synthesis_stages_match_reference_goldens():
    """lc3gpu_decoder_synth_debug: the IMDCT fed with the reference's two spectra (decoder/modified_dct.rs:174-329) and the
    long-term post-filter fed with the reference's six frames that walk its transition cases 1, 1, 2, 5, 4, 3
    (decoder/long_term_post_filter.rs:504-1199), each against the reference's expected output, bit for bit"""
    from test_oracle_kats import f32

    t = "decoder/modified_dct.rs::modified_dct_decode"
    dec = pkg.Lc3Decoder(1, US, FS)
    dec.synth_debug(f32(t, "x_hat", 0), 0, 0, 150)
    _, dbg = dec.synth_debug(f32(t, "x_hat", 1), 0, 0, 150)
    assert np.array_equal(_bits(dbg[800:1280]), _bits(f32(t, "freq_buf_expected"))), "IMDCT after two frames"
    t = "decoder/long_term_post_filter.rs::long_term_post_filter_full_cycle"
    dec = pkg.Lc3Decoder(1, US, FS)
    infos = [(0, 134), (0, 132), (1, 134), (1, 136), (1, 136), (0, 132)]
    for i, (active, idx) in enumerate(infos):
        pcm, dbg = dec.synth_debug(f32(t, "freq_samples", i), active, idx, 40, time_in=True)  # nbits = 320 as in the reference's test
        exp = f32(t, "freq_samples_expected", i)
        assert np.array_equal(_bits(dbg[1280:1760]), _bits(exp)), f"LTPF frame {i}"
        ref_pcm = np.zeros(480, np.int16)
        O.lib().lc3o_dec_output(O.P(exp), O.P(ref_pcm), 480)
        assert np.array_equal(pcm, ref_pcm), f"output scaling frame {i}"


# ---------------------------------------------------------------- the device's float routines, evaluated on the device
def test_device_math_on_the_device():
    """What frame parity only covers transitively, checked value by value ON the GPU (lc3gpu_selftest_math): the quantiser's division
    by a wave-uniform gain (`lc3_div_by`: the compiler's own f32 division sequence with the reciprocal refined once per gain -- the CPU
    emulator compiles the plain `x / d` instead) against IEEE division on 2^24 random (x, d) pairs over the quantiser's operating range
    and its stated edge (|x| < 2^60, d in [1.8e-9, 1.5e5]: bit-identical from |x| = 2^-103 up); hipcc's code for the restated msun routines (log2f, log10f, exp2f, asinf,
    powf(10, .), sinf) and fast-math's exp2_raw against the oracle's C routines, bit for bit; and the tables the device fills with
    those routines at start-up (gains 10^(k/28), tilt factors, TNS sines) against the oracle evaluating the same expressions."""
    import ctypes

    rng = np.random.default_rng(20240317)
    n = 1 << 24
    # x: signed, log-uniform magnitude over 2^-126 .. 2^60, with exact zeros, quantiser-typical values and huge ones mixed in
    mag = np.exp2(rng.uniform(-126.0, 60.0, n)).astype(np.float32)
    x = (mag * rng.choice(np.array([-1.0, 1.0], np.float32), n)).astype(np.float32)
    x[:: 1 << 10] = 0.0
    x[1:: 1 << 10] = (rng.uniform(-40000.0, 40000.0, x[1:: 1 << 10].size)).astype(np.float32)
    d = np.exp(rng.uniform(np.log(1.8e-9), np.log(1.5e5), n)).astype(np.float32)
    d[: 1 << 12] = np.float32(1.8e-9)
    d[1 << 12: 1 << 13] = np.float32(1.5e5)
    got = pkg.selftest_math(0, x, d)
    ieee_dev = pkg.selftest_math(1, x, d)
    with np.errstate(all="ignore"):
        ieee = (x / d).astype(np.float32)  # numpy's f32 division is the correctly rounded IEEE quotient
    assert np.array_equal(ieee_dev.view(np.uint32), ieee.view(np.uint32)), "hipcc's f32 division is not the IEEE quotient"
    big = np.abs(x) >= np.float32(2.0 ** -103)  # below: the hardware sequence would pre-scale the numerator; the quotient is < 2^-74
    assert np.array_equal(got.view(np.uint32)[big], ieee.view(np.uint32)[big]), "lc3_div_by differs from IEEE division on a numerator >= 2^-103"
    # there the callers (trunc(q + 0.375), spectral_quantization.rs:239-262) cannot tell the quotients apart: both are tiny and within an ulp
    assert np.all(np.abs(got[~big]) < np.float32(2.0 ** -73))
    assert np.abs(got.view(np.int32)[~big].astype(np.int64) - ieee.view(np.int32)[~big].astype(np.int64)).max() <= 1
    assert np.array_equal(np.trunc(got + np.float32(0.375)), np.trunc(ieee + np.float32(0.375)))

    def oracle(which, v):
        out = np.zeros(v.size, np.float32)
        O.lib().lc3o_kat_math(int(which), O.P(v), int(v.size), O.P(out))
        return out

    m = 1 << 20
    eps = np.float32(1.1920929e-7)
    cases = {
        2: (eps + np.exp(rng.uniform(np.log(1e-9), np.log(1e13), m))).astype(np.float32),           # log2f(eps + band energy)
        3: np.exp(rng.uniform(np.log(1e-9), np.log(1e13), m)).astype(np.float32),                   # log10f
        4: rng.uniform(-70.0, 70.0, m).astype(np.float32),                                          # exp2f(-scale factor)
        5: np.concatenate([rng.uniform(-1.0, 1.0, m - 5), [-1.0, 1.0, 0.0, 0.5, -0.5]]).astype(np.float32),  # asinf(reflection coefficient)
        6: rng.uniform(-40.0, 40.0, m).astype(np.float32),                                          # exp2_raw(scale factor)
        7: rng.uniform(-12.0, 12.0, m).astype(np.float32),                                          # 10^x
        8: rng.uniform(-1.6, 1.6, m).astype(np.float32),                                            # sin(k pi / 17), |k| <= 8
    }
    for which, v in cases.items():
        v[:4] = np.array([1.0, 2.0, 0.5, 1e-7], np.float32) if which in (2, 3) else v[:4]
        a, b = pkg.selftest_math(which, v), oracle(which, v)
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), (which, int((~same).sum()), v[~same][:4], a[~same][:4], b[~same][:4])
    # device-filled tables
    tab = pkg.selftest_math(9, n=866)
    k = np.arange(-256, 256, dtype=np.float32)
    assert np.array_equal(tab[:512].view(np.uint32), oracle(7, (k / np.float32(28.0)).astype(np.float32)).view(np.uint32)), "10^(k/28) table"
    g_tilt = np.array([14, 18, 22, 26, 30], np.float32)
    b = np.arange(64, dtype=np.float32)
    tilt_arg = (b[None, :] * (g_tilt[:, None] / np.float32(630.0)).astype(np.float32)).astype(np.float32).reshape(-1)
    assert np.array_equal(tab[512:832].view(np.uint32), oracle(7, tilt_arg).view(np.uint32)), "tilt table"
    ri = np.arange(17, dtype=np.float32)
    step_enc = np.float32(np.float32(np.pi) / np.float32(17.0))        # (PI as f32) / 17.0   encoder/temporal_noise_shaping.rs:268-273
    step_dec = np.float32(np.pi / 17.0)                                # (PI / 17.0) as f32   decoder/temporal_noise_shaping.rs:41-44
    assert np.array_equal(tab[832:849].view(np.uint32), oracle(8, (step_enc * (ri - np.float32(8.0))).astype(np.float32)).view(np.uint32)), "encoder TNS sines"
    assert np.array_equal(tab[849:866].view(np.uint32), oracle(8, (step_dec * (ri - np.float32(8.0))).astype(np.float32)).view(np.uint32)), "decoder TNS sines"


def test_one_handle_alternates_reconstruction_forms_with_lost_frames_at_launch_edges():
    """ONE decoder handle whose launches alternate between two frames per stream (spectrum rebuilt in the synthesis kernel, the last
    good spectrum written to the state blob by every good frame) and eight (rebuilt in the parse kernel, the last good spectrum taken
    from the launch's own plane columns and copied to the blob only at the launch's end), with corrupt and flagged frames as the
    first and the last frame of launches, so that concealment crosses every kind of hand-over: the PCM of the whole 30-frame run
    must be the oracle's."""
    torch = torch_mod()
    S, chunks = 48, [2, 8, 2, 8, 2, 8]
    T = sum(chunks)
    pcm = synth.make_pcm(S, T, 480, 48000, seed=1234)
    data = O.encode_batch(pcm, 150, threads=8).copy()
    flags = np.zeros((S, T), np.uint8)
    edges, t0 = [], 0
    for n in chunks:
        edges += [t0, t0 + n - 1]
        t0 += n
    rng = np.random.default_rng(5)
    for s in range(S):
        lost = rng.choice(edges, size=4, replace=False)
        for t in lost[:2]:
            data[s, t, -1] |= 7          # bandwidth index out of range: the parser rejects the frame
        for t in lost[2:]:
            flags[s, t] = 1              # external bad-frame indicator on an intact frame
    for_oracle = data.copy()
    for_oracle[flags.astype(bool), -1] |= 7
    ref = O.decode_batch(for_oracle, 480, threads=8)
    dec = pkg.Lc3Decoder(S, US, FS)
    st = torch.cuda.current_stream().cuda_stream
    got, t0 = [], 0
    for n in chunks:
        d_in = torch.from_numpy(np.ascontiguousarray(data[:, t0:t0 + n])).cuda()
        d_fl = torch.from_numpy(np.ascontiguousarray(flags[:, t0:t0 + n])).cuda()
        d_out = torch.zeros((S, n, 480), dtype=torch.int16, device="cuda")
        dec.decode(d_in, d_out, 150, n, stream=st, d_bad_frame=d_fl)
        torch.cuda.synchronize()
        got.append(d_out.cpu().numpy())
        t0 += n
    got = np.concatenate(got, axis=1)
    assert np.array_equal(got, ref), np.argwhere((got != ref).any(axis=2))[:8]
    assert dec.plc_events() == 4 * S
