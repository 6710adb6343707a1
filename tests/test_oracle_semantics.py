"""What the oracle MEANS on the configurations the reference never tests.

All 36 of the reference's known-answer vectors are 48 kHz / 10 ms: at the other eleven rate x duration pairs the oracle is
a line-by-line restatement with nothing of the reference's to compare against.  This is the semantic guard for them: whatever the
restatement got wrong in a table index, a band layout or a window would show up as a codec that no longer reproduces its input.  An
encode -> decode round trip of tonal / noisy synthetic streams must come back (a) at the LC3 algorithmic delay -- 2.5 ms at 10 ms frames
(nf / 4 samples), 4 ms at 7.5 ms frames (8 nf / 15), the 44.1 kHz configurations counted in the 48 kHz samples they run as
(/root/reference/src/common/config.rs:64-65,81-82) -- and (b) with a signal-to-noise ratio above a floor per configuration (measured
medians minus ~3 dB).  8 kHz has no reference encoder (encoder/bandwidth_detector.rs:36-37 panics); the oracle's early-return path
(:66-71) feeds the decoder there, and at 8 kHz / 7.5 ms encoder and decoder disagree about the scale-factor bands:
the floor is lower."""
import importlib

import numpy as np
import pytest

import oracle_lib as O

synth = importlib.import_module("lc3-codec_amd.synth")

NB10 = {8000: 30, 16000: 40, 24000: 60, 32000: 80, 44100: 110, 48000: 150}  # SURVEY 8d Config 4: bytes per 10 ms frame
FLOOR_DB = {  # median SNR floor per (fs, frame_us)
    (8000, 10000): 12.0, (8000, 7500): 7.0, (16000, 10000): 11.0, (16000, 7500): 12.0, (24000, 10000): 13.0, (24000, 7500): 11.0,
    (32000, 10000): 15.0, (32000, 7500): 13.0, (44100, 10000): 17.0, (44100, 7500): 14.0, (48000, 10000): 20.0, (48000, 7500): 17.0,
}


@pytest.mark.parametrize("fs,us", sorted(FLOOR_DB))
def test_round_trip_tracks_the_input_at_the_lc3_delay(fs, us):
    cfg = np.zeros(7, np.int32)
    O.lib().lc3o_kat_config(fs, us, O.P(cfg))
    nf = int(cfg[5])
    nbytes = NB10[fs] if us == 10000 else int(round(NB10[fs] * 0.75))
    S, T = 24, 24
    pcm = synth.make_pcm(S, T, nf, fs, seed=101)
    out = O.decode_batch(O.encode_batch(pcm, nbytes, fs, us), nf, fs, us)
    x, y = pcm.reshape(S, -1).astype(np.float64), out.reshape(S, -1).astype(np.float64)
    want = nf // 4 if us == 10000 else nf * 8 // 15
    n, a0 = x.shape[1], 3 * nf  # (the first frames are the codec's start-up)
    snrs, delays = [], []
    for s in range(S):
        peak = np.abs(x[s]).max()
        if peak == 0 or peak >= 32767:  # the generator's silent and clipping-noise streams say nothing about tracking
            continue
        best = (-1e9, -1)
        for d in range(0, nf):
            e = y[s, a0 + d:n] - x[s, a0:n - d]
            snr = 10.0 * np.log10((x[s, a0:n - d] ** 2).sum() / max(1e-9, (e ** 2).sum()))
            if snr > best[0]:
                best = (snr, d)
        snrs.append(best[0])
        delays.append(best[1])
    assert len(snrs) >= 10
    assert int(np.median(delays)) == want, (delays, want)
    assert sum(d == want for d in delays) >= 0.8 * len(delays), delays
    assert np.median(snrs) >= FLOOR_DB[(fs, us)], (np.median(snrs), sorted(snrs))
    # silence in, silence out
    z = O.decode_batch(O.encode_batch(np.zeros((1, 4, nf), np.int16), nbytes, fs, us), nf, fs, us)
    assert np.abs(z.astype(np.int32)).max() <= 1
