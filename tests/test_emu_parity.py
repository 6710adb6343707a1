"""The HIP device code (lc3-codec_amd/csrc/lc3_dev_*.h) run under the CPU wave emulator vs the oracle.
Checks the lane-parallel decomposition and barrier placement in this GPU-less container; the -m gpu
tests repeat the same comparisons on the real hardware through the C ABI."""
import importlib

import numpy as np
import pytest

import emu_lib as E
import oracle_lib as O

synth = importlib.import_module("lc3-codec_amd.synth")


def test_emu_kats():
    t = "encoder/lc3_encoder.rs::lc3_encode_channel"
    out = E.encode(O.kat(t, "samples_in", 0, np.int16).reshape(1, 1, 480), 150)
    assert out[0, 0].tolist() == O.kat(t, "buf_out_expected")
    t = "decoder/lc3_decoder.rs::lc3_decode_channel"
    pcm = E.decode(np.array(O.kat(t, "buf_in"), np.uint8).reshape(1, 1, 150), 480)
    assert pcm[0, 0].tolist() == O.kat(t, "samples_out_expected")


@pytest.mark.parametrize("fs,us,nbytes", [
    (48000, 10000, 150), (48000, 10000, 60), (48000, 10000, 400), (48000, 7500, 113), (44100, 10000, 110),
    (32000, 10000, 80), (32000, 7500, 61), (24000, 10000, 60), (24000, 7500, 45), (16000, 10000, 40),
    (16000, 7500, 30), (48000, 10000, 40),
])
def test_emu_matches_oracle(fs, us, nbytes):
    cfg = np.zeros(7, np.int32)
    O.lib().lc3o_kat_config(fs, us, O.P(cfg))
    nf = int(cfg[5])
    pcm = synth.make_pcm(4, 6, nf, fs)
    ref = O.encode_batch(pcm, nbytes, fs, us)
    assert np.array_equal(E.encode(pcm, nbytes, fs, us), ref)
    assert np.array_equal(E.decode(ref, nf, fs, us), O.decode_batch(ref, nf, fs, us))


def test_emu_decode_8khz_and_corrupt():
    pcm = synth.make_pcm(3, 5, 80, 8000)
    data = O.encode_batch(pcm, 30, 8000, 10000)
    assert np.array_equal(E.decode(data, 80, 8000, 10000), O.decode_batch(data, 80, 8000, 10000))
    pcm = synth.make_pcm(3, 8, 480, 48000, seed=3)
    data = O.encode_batch(pcm, 150).copy()
    data[0, 2, -1] |= 7
    data[0, 3, -1] |= 7
    data[1, 4] = np.random.default_rng(1).integers(0, 256, 150, dtype=np.uint8)
    assert np.array_equal(E.decode(data, 480), O.decode_batch(data, 480))
    bad = np.zeros((3, 8), np.uint8)
    bad[2, 1] = 1
    corrupt = O.encode_batch(pcm, 150).copy()
    corrupt[2, 1, -1] |= 7
    assert np.array_equal(E.decode(O.encode_batch(pcm, 150), 480, bad=bad), O.decode_batch(corrupt, 480))


def test_device_math_matches_oracle_math():
    """the GPU float library (lc3_dev_common.h) against the oracle's restatement of the same msun routines"""
    import ctypes

    L, M = E.lib(), O.lib()
    for n in ("lc3m_log2f", "lc3m_log10f", "lc3m_exp2f", "lc3m_asinf", "lc3m_sinf", "lc3m_exp2_raw"):
        getattr(M, n).restype = ctypes.c_float
        getattr(M, n).argtypes = [ctypes.c_float]
    rng = np.random.default_rng(0)
    for k in range(-260, 261):
        y = float(np.float32(k) / np.float32(28))
        assert L.lc3emu_pow10f(y) == M.lc3o_kat_powf(10.0, y), k
    for b in range(64):
        for g in (14, 18, 22, 26, 30):
            y = float(np.float32(b) * (np.float32(g) / np.float32(630)))
            assert L.lc3emu_pow10f(y) == M.lc3o_kat_powf(10.0, y)
    xs = np.exp(rng.uniform(-30, 40, 4000)).astype(np.float32)
    for x in xs:
        assert L.lc3emu_log2f(float(x)) == M.lc3m_log2f(float(x))
        assert L.lc3emu_log10f(float(x)) == M.lc3m_log10f(float(x))
    for x in rng.uniform(-40, 40, 4000).astype(np.float32):
        assert L.lc3emu_exp2f(float(x)) == M.lc3m_exp2f(float(x))
        assert L.lc3emu_exp2_raw(float(x)) == M.lc3m_exp2_raw(float(x))
    for x in rng.uniform(-1, 1, 4000).astype(np.float32):
        assert L.lc3emu_asinf(float(x)) == M.lc3m_asinf(float(x))
    step = np.float32(np.float32(np.pi) / np.float32(17.0))
    for i in range(17):
        x = float(np.float32(step * np.float32(i - 8)))
        assert L.lc3emu_sinf_small(x) == M.lc3m_sinf(x)


@pytest.mark.parametrize("fs,us,nbytes", [(48000, 10000, 60), (16000, 7500, 30)])
def test_emu_ltpf_transitions(fs, us, nbytes):
    """Decoder LTPF synthesis through all five transition cases (off->off, off->on, on->off, same lag, new lag)."""
    cfg = np.zeros(7, np.int32)
    O.lib().lc3o_kat_config(fs, us, O.P(cfg))
    nf = int(cfg[5])
    pcm = synth.make_ltpf_pcm(nf, fs)
    ref = O.encode_batch(pcm, nbytes, fs, us)
    O.ltpf_transition_counts(reset=True)
    ref_pcm = O.decode_batch(ref, nf, fs, us)
    counts = O.ltpf_transition_counts()
    assert all(c > 0 for c in counts[1:]), counts
    assert np.array_equal(E.encode(pcm, nbytes, fs, us), ref)
    assert np.array_equal(E.decode(ref, nf, fs, us), ref_pcm)


SPEC_FLAGS = {"8khz": 1, "tns_sswb_stop": 2, "bw_cutoff_db": 4, "sns_last_gain": 8, "nbits_spec_old": 16}


@pytest.mark.parametrize("flag", [2, 4, 8, 16, 31])
def test_emu_spec_conformant_switches(flag):
    """the opt-in corrections of the reference's deviations (LC3_SPEC_* bits), one at a time and all
    together, device code vs the oracle with the same switch.  Band-limited input with clicks so that the bandwidth
    detector's cut-off stage and the SSWB TNS layout are reached; only streams on which the switch changes the oracle's
    output are run (the switch must bite), plus one on which it does not."""
    pcm = synth.make_bandlimited_pcm(48, 8, 480, 48000, 11800.0)
    ref0 = O.encode_batch(pcm, 100)
    ref1 = O.encode_batch(pcm, 100, spec_flags=flag)
    differs = np.flatnonzero((ref0 != ref1).any(axis=(1, 2)))
    assert len(differs) > 0, "the switch changes nothing on this input"
    same = np.flatnonzero(~(ref0 != ref1).any(axis=(1, 2)))
    pick = list(differs[:3]) + list(same[:1])
    got = E.encode(pcm[pick], 100, spec_flags=flag)
    assert np.array_equal(got, ref1[pick])
    assert np.array_equal(E.encode(pcm[pick[:1]], 100), ref0[pick[:1]])  # default: the reference's behaviour, unchanged


def test_emu_8khz_encode_switch():
    """LC3_SPEC_8KHZ_ENCODE: the reference cannot build an 8 kHz encoder (bandwidth_detector.rs:36-37); with the switch the
    encoder follows the early return of BandwidthDetector::run (:66-71).  Device code vs oracle, and the oracle's own decoder
    must make sense of the stream."""
    for us, nb, nf in ((10000, 30, 80), (7500, 23, 60)):
        pcm = synth.make_pcm(4, 6, nf, 8000, seed=9)
        ref = O.encode_batch(pcm, nb, 8000, us, spec_flags=1)
        assert np.array_equal(E.encode(pcm, nb, 8000, us, spec_flags=1), ref)
        dec = O.decode_batch(ref, nf, 8000, us)
        live = pcm.reshape(4, -1).std(axis=1) > 100
        assert (dec.reshape(4, -1)[live].std(axis=1) > 50).all()


@pytest.mark.parametrize("fs,us,nf,nbytes", [(48000, 10000, 480, 150), (48000, 7500, 360, 40), (16000, 10000, 160, 60)])
def test_emu_guarded_decisions_and_their_sequential_path(fs, us, nf, nbytes):
    """The quantiser's gain bisection and noise level are decided from tree sums where those are further from the threshold than
    the rounding of the reference's sequential sum can reach.  The emulator build runs the sequential sum
    next to EVERY such decision and aborts the process when the two disagree (LC3_GUARD_SELFCHECK), so simply running frames is
    the check; flag 256 (LC3_SPEC_TEST_SEQ_SUMS, what LC3GPU_SEQ_SUMS=1 sets on the GPU) forces the sequential path."""
    pcm = synth.make_pcm(10, 6, nf, fs, first_stream=40)
    ref = O.encode_batch(pcm, nbytes, fs, us)
    assert np.array_equal(E.encode(pcm, nbytes, fs, us), ref)
    assert np.array_equal(E.encode(pcm, nbytes, fs, us, spec_flags=256), ref)
    # 512 (LC3_LAUNCH_PREP_SYMBOLS, what the host sets for launches that do not fill the chip): the analysis kernel prepares the
    # packer's symbol words, the packer only runs the range coder over them
    assert np.array_equal(E.encode(pcm, nbytes, fs, us, spec_flags=512), ref)
    # 1024 (emulator only): the same symbol words from the wave-per-frame stage between back half and packer (lc3_symbols_kernel,
    # LC3GPU_PREP_SYMBOLS=2 on the GPU)
    assert np.array_equal(E.encode(pcm, nbytes, fs, us, spec_flags=1024), ref)
    # 2048 (emulator only): the packer of full batches as a producer / consumer pair (lc3_pack_produce / lc3_pack_consume): symbol words
    # derived by one wave, range coder and writers on another
    assert np.array_equal(E.encode(pcm, nbytes, fs, us, spec_flags=2048), ref)
    hard = synth.make_pcm(6, 4, nf, fs, seed=77) // 2 + synth.make_bandlimited_pcm(6, 4, nf, fs, fs / 6.0, seed=5) // 2
    for nb in (20, nbytes, 400 if us == 10000 else 300):  # LSB mode at the low end, long escape chains at the high end
        assert np.array_equal(E.encode(hard, nb, fs, us, spec_flags=2048), O.encode_batch(hard, nb, fs, us)), nb


@pytest.mark.parametrize("late", [1, 2, 3])
def test_emu_late_reconstruction(late):
    """Launches of a few frames reconstruct the spectrum (residual bits, noise filling, gain, TNS, band gains) with the 64 lanes of
    the stream's wave in the synthesis stage instead of one lane of the parser (lc3_dec_reconstruct_wave): the same PCM as the
    oracle on clean streams of every kind, on corrupted and flagged frames (concealment then reads the state blob's copy of the
    last good spectrum), on garbage, and through the LTPF transitions.  late = 2: the same wave-parallel reconstruction as the
    wave-per-FRAME kernel of full batches (lc3_recon_kernel) between parser and synthesis.  late = 3: the parser's producer / consumer
    form of full batches (lc3_pc_produce / lc3_pc_consume: the range decoder's recurrence on one wave, everything that only consumes
    symbols on another), with the reconstruction on the consumer's lane."""
    for fs, us, nf, nb in [(48000, 10000, 480, 150), (48000, 7500, 360, 113), (32000, 10000, 320, 40), (24000, 7500, 180, 60),
                           (16000, 10000, 160, 120), (8000, 10000, 80, 30), (48000, 10000, 480, 20), (48000, 10000, 480, 400)]:
        pcm = synth.make_pcm(8, 6, nf, fs, first_stream=300)
        data = O.encode_batch(pcm, nb, fs, us, spec_flags=1 if fs == 8000 else 0)
        assert np.array_equal(E.decode(data, nf, fs, us, late=late), O.decode_batch(data, nf, fs, us)), (fs, us, nb)
    pcm = synth.make_pcm(3, 8, 480, 48000, seed=3)
    data = O.encode_batch(pcm, 150).copy()
    data[0, 2, -1] |= 7
    data[0, 3, -1] |= 7
    data[1, 4] = np.random.default_rng(1).integers(0, 256, 150, dtype=np.uint8)
    data[2, 7, -1] |= 7  # a launch that ends in a lost frame
    assert np.array_equal(E.decode(data, 480, late=late), O.decode_batch(data, 480))
    bad = np.zeros((3, 8), np.uint8)
    bad[2, 1] = bad[0, 0] = 1
    corrupt = O.encode_batch(pcm, 150).copy()
    corrupt[2, 1, -1] |= 7
    corrupt[0, 0, -1] |= 7
    assert np.array_equal(E.decode(O.encode_batch(pcm, 150), 480, bad=bad, late=late), O.decode_batch(corrupt, 480))
    rng = np.random.default_rng(23)
    for nbytes in (20, 40, 150, 400):
        g = rng.integers(0, 256, (12, 4, nbytes), dtype=np.uint8)
        assert np.array_equal(E.decode(g, 480, late=late), O.decode_batch(g, 480)), nbytes
    for fs, nf, nb in ((48000, 480, 40), (16000, 160, 40)):
        lp = synth.make_ltpf_pcm(nf, fs)
        d = O.encode_batch(lp, nb, fs, 10000)
        assert np.array_equal(E.decode(d, nf, fs, 10000, late=late), O.decode_batch(d, nf, fs, 10000))
    bl = synth.make_bandlimited_pcm(8, 6, 480, 48000, 7000.0)  # lower bandwidth indices: other TNS / noise-filling limits
    d = O.encode_batch(bl, 100)
    assert np.array_equal(E.decode(d, 480, late=late), O.decode_batch(d, 480))


def test_gain_limitation_bound_covers_the_reference_expression():
    """lc3_enc_quant decides `gg_ind < gg_min` from an integer bound on ceil(28 log10(x_f_max / 32767.625)) taken from the exponent of
    x_f_max (lc3_dev_enc.h) and evaluates the reference's expression (spectral_quantization.rs:218-221) only when the bound cannot decide.
    The bound has to be >= the expression for EVERY x_f_max of its binade: checked here with the device's own log10f (compiled for
    the CPU) on the largest value, the smallest value and 4 000 random values of every binade of normal floats."""
    rng = np.random.default_rng(5)
    worst = 1000
    for n in range(-125, 129):  # 2^(n-1) <= x < 2^n
        lo = np.float32(2.0) ** np.float32(n - 1)
        top = np.nextafter(np.float32(2.0) ** np.float32(n) if n < 128 else np.float32(np.inf), np.float32(0), dtype=np.float32)
        xs = np.concatenate([[lo, top], (lo * (np.float32(1) + rng.random(4000, dtype=np.float32))).astype(np.float32)])
        xs = xs[(xs >= lo) & (xs <= top)]
        a = (8632 * n + 1023) >> 10 if n >= 0 else -((8631 * -n) >> 10)
        bound = a - 125
        for x in xs:
            y = np.float32(x) / np.float32(32768.0 - 0.375)
            v = np.ceil(np.float32(28.0) * np.float32(E.lib().lc3emu_log10f(float(y))))
            v = min(32767.0, max(-32768.0, float(v)))  # lc3_f2i16
            assert bound >= v, (n, float(x), bound, v)
            worst = min(worst, bound - int(v))
    assert 0 <= worst <= 2, worst  # the bound is tight: it does not give the fast path away


@pytest.mark.parametrize("T", [1, 2, 3, 5, 9])
def test_emu_frame_counts_around_the_tns_chunk(T):
    """The encoder's back half analyses the frames of a launch in chunks of four (lc3_encode_back_stream: TNS autocorrelations of the
    chunk, one pass of the Levinson recursions for all of its frames, then frame by frame); a chunk of ONE frame takes the short path.
    Launches of 1, 2, 3, 5 (= 4 + 1) and 9 (= 4 + 4 + 1) frames on material with clicks (active TNS filters) against the oracle."""
    pcm = synth.make_pcm(6, T, 480, 48000, seed=12)
    for nbytes in (150, 60):
        assert np.array_equal(E.encode(pcm, nbytes), O.encode_batch(pcm, nbytes))



@pytest.mark.parametrize("late", [0, 1])
def test_emu_loss_bursts_reach_the_second_fade(late):
    """packet_loss_concealment.rs:62-66: from the ninth lost frame of a run the concealed spectrum fades by 0.85 per frame.  Runs of 13 and 9
    lost frames (unparsable side information, garbage and external flags mixed) through the device headers under the emulator, in both
    places the decoder rebuilds the spectrum (parser lane / synthesis wave); the -m gpu suite repeats this across launch boundaries."""
    S, T = 3, 26
    pcm = synth.make_pcm(S, T, 480, 48000, seed=71)
    data = O.encode_batch(pcm, 150).copy()
    marked = data.copy()
    bad = np.zeros((S, T), np.uint8)
    rng = np.random.default_rng(73)
    for s_i, (start, run) in enumerate(((3, 13), (5, 9), (2, 11))):
        for k in range(run):
            t = start + k
            how = (s_i + k) % 3
            if how == 0:
                data[s_i, t, -1] |= 7
                marked[s_i, t, -1] |= 7
            elif how == 1:
                bad[s_i, t] = 1
                marked[s_i, t, -1] |= 7
            else:
                g = rng.integers(0, 256, 150, dtype=np.uint8)
                g[-1] |= 7
                data[s_i, t] = g
                marked[s_i, t] = g
    ref = O.decode_batch(marked, 480)
    assert np.array_equal(E.decode(data, 480, bad=bad, late=late), ref)
