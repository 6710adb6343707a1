"""Pin the CPU oracle against every hot-path known-answer test the reference carries
(SURVEY.md section 8c).  Vectors: tests/golden/ref_kats.json, lifted from the reference's
in-file `#[test]` functions by tools/extract_goldens.py.  All comparisons are exact
(the reference asserts f32 equality), except where a comment says otherwise.

CPU only (no GPU needed)."""
import ctypes

import numpy as np
import pytest

from oracle_lib import Decoder, Encoder, P, kat, lib

F32 = np.float32
FS, US = 48000, 10000


def f32(test, name=None, index=None):
    return kat(test, name, index, dtype=np.float32)


def assert_same_f32(got, want, what=""):
    got = np.asarray(got, np.float32)
    want = np.asarray(want, np.float32)
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    # +0.0 and -0.0 compare equal under the reference's assert_eq!
    bad = [i for i in bad if not (got[i] == 0.0 and want[i] == 0.0)]
    assert not bad, f"{what}: {len(bad)} mismatches, first at {bad[0]}: got {got[bad[0]]!r} want {want[bad[0]]!r}"


# ---------------------------------------------------------------- common/
def test_config_simple():  # common/config.rs:109-119
    cfg = np.zeros(7, np.int32)
    lib().lc3o_kat_config(48000, 10000, P(cfg))
    assert cfg.tolist() == [4, 48000, 400, 1, 64, 480, 180]


@pytest.mark.parametrize(
    "fs,us,nf,ne,nb,z",
    [(8000, 10000, 80, 80, 64, 30), (16000, 10000, 160, 160, 64, 60), (24000, 10000, 240, 240, 64, 90),
     (32000, 10000, 320, 320, 64, 120), (44100, 10000, 480, 400, 64, 180), (8000, 7500, 60, 60, 60, 14),
     (16000, 7500, 120, 120, 64, 28), (24000, 7500, 180, 180, 64, 42), (32000, 7500, 240, 240, 64, 56),
     (48000, 7500, 360, 300, 64, 84)])
def test_config_table(fs, us, nf, ne, nb, z):  # common/config.rs:57-88
    cfg = np.zeros(7, np.int32)
    lib().lc3o_kat_config(fs, us, P(cfg))
    assert (cfg[5], cfg[2], cfg[4], cfg[6]) == (nf, ne, nb, z)


def test_working_buffer_lengths():  # lc3_encoder.rs:194-209, lc3_decoder.rs:236-244, README.md:130
    e = np.zeros(3, np.int64)
    d = np.zeros(2, np.int64)
    lib().lc3o_encoder_working_buffer_lengths(1, FS, US, P(e))
    lib().lc3o_decoder_working_buffer_lengths(1, FS, US, P(d))
    assert e.tolist() == [1900, 1106, 960]
    assert d.tolist() == [4971, 960]
    assert d[0] * 4 + d[1] * 8 == 27564


def test_kissfft_non_inverse():  # common/kissfft.rs:298-425
    t = "common/kissfft.rs::kissfft_non_inverse"
    im, re = f32(t, "i"), f32(t, "r")
    ore, oim = np.zeros(240, F32), np.zeros(240, F32)
    lib().lc3o_kat_fft(240, P(re), P(im), P(ore), P(oim))
    assert_same_f32(oim, f32(t, "i_expected"), "fft imag")
    assert_same_f32(ore, f32(t, "r_expected"), "fft real")


def test_dct_iv_run():  # common/dct_iv.rs:80-194
    t = "common/dct_iv.rs::mdct_iv_run"
    buf = f32(t, "buf").copy()
    lib().lc3o_kat_dct4(480, P(buf))
    assert_same_f32(buf, f32(t, "output_expected"), "dct-iv")


# ---------------------------------------------------------------- encoder/
def test_modified_dct_encode():  # encoder/modified_dct.rs:191-337
    t = "encoder/modified_dct.rs::modified_dct_encode"
    enc = Encoder()
    out, eb = np.zeros(480, F32), np.zeros(64, F32)
    lib().lc3o_kat_enc_mdct(enc.h, P(kat(t, "samples_in", 0, np.int16)), P(out), P(eb))
    nn = lib().lc3o_kat_enc_mdct(enc.h, P(kat(t, "samples_in", 1, np.int16)), P(out), P(eb))
    assert_same_f32(out, f32(t, "output_expected"), "mdct spectrum")
    assert_same_f32(eb, f32(t, "energy_bands_expected"), "band energies")
    assert nn == 0


def test_bandwidth_detector_run():  # encoder/bandwidth_detector.rs:137-155
    out = np.zeros(2, np.int32)
    lib().lc3o_kat_bandwidth(FS, US, P(f32("encoder/bandwidth_detector.rs::bandwidth_detector_run", "e_b")), P(out))
    assert out.tolist() == [4, 3]


def test_attack_detector_run():  # encoder/attack_detector.rs:138-180
    enc = Encoder()
    fo, io = np.zeros(2, F32), np.zeros(3, np.int32)
    x_s = kat("encoder/attack_detector.rs::attack_detector_run", "x_s", 0, np.int16)
    r = lib().lc3o_kat_attack(enc.h, P(x_s), 150, P(fo), P(io))
    assert r == 1
    assert fo[1] == F32(905588.875) and fo[0] == F32(549861.5)
    assert io.tolist() == [0, 4846, 5210]


def test_sns_run():  # encoder/spectral_noise_shaping.rs:658-776
    t = "encoder/spectral_noise_shaping.rs::sns_run"
    x = f32(t, "x").copy()
    out = np.zeros(7, np.int64)
    lib().lc3o_kat_sns(FS, US, P(x), P(f32(t, "e_b")), 1, P(out))
    assert_same_f32(x, f32(t, "x_s_expected"), "sns shaped spectrum")


def test_sns_quant_run():  # encoder/spectral_noise_shaping.rs:780-801
    t = "encoder/spectral_noise_shaping.rs::sns_quant_run"
    scfq = np.zeros(16, F32)
    out = np.zeros(7, np.int64)
    lib().lc3o_kat_sns_quant(P(f32(t, "scf")), P(scfq), P(out))
    assert_same_f32(scfq, f32(t, "scfq_expected"), "scfq")
    ind_lf, ind_hf, shape_j, gind, ls_inda, ls_indb, joint = out.tolist()
    assert (gind, ind_hf, ind_lf, joint, shape_j, ls_inda, ls_indb) == (0, 17, 8, 15253432, 3, 0, 0)


def test_temporal_noise_shaping_run():  # encoder/temporal_noise_shaping.rs:359-471
    t = "encoder/temporal_noise_shaping.rs::temporal_noise_shaping_run"
    x = f32(t, "x_s").copy()
    io, rc_q = np.zeros(21, np.int32), np.zeros(16, F32)
    lib().lc3o_kat_tns(FS, US, P(x), 4, 1200, 0, P(io), P(rc_q))
    assert_same_f32(x, f32(t, "x_f_expected"), "tns filtered spectrum")
    assert io[5:21].tolist() == kat(t, index=2)  # rc_i
    assert_same_f32(rc_q, np.array(kat(t, index=3), F32), "rc_q")
    assert (io[1], io[2], io[3], io[4], io[0]) == (0, 2, 8, 6, 42)


def test_long_term_post_filter_run():  # encoder/long_term_post_filter.rs:479-520
    enc = Encoder()
    out = np.zeros(4, np.int32)
    x_s = kat("encoder/long_term_post_filter.rs::long_term_post_filter_run", "x_s", 0, np.int16)
    lib().lc3o_kat_ltpf_enc(enc.h, P(x_s), 0, 1200, P(out))
    pitch_index, pitch_present, ltpf_active, nbits_ltpf = out.tolist()
    assert (nbits_ltpf, pitch_present, ltpf_active, pitch_index) == (11, 1, 0, 0)


def test_long_term_post_filter_active():  # encoder/long_term_post_filter.rs:523-843
    t = "encoder/long_term_post_filter.rs::long_term_post_filter_active"
    enc = Encoder()
    # (ltpf_active, pitch_present, pitch_index, nbits_ltpf) per frame, :566-842
    want = [(0, 0, 0, 1), (0, 0, 0, 1), (0, 1, 180, 11), (0, 1, 184, 11), (0, 1, 477, 11), (0, 1, 478, 11),
            (1, 1, 478, 11), (1, 1, 478, 11)]
    for i, w in enumerate(want):
        out = np.zeros(4, np.int32)
        lib().lc3o_kat_ltpf_enc(enc.h, P(kat(t, "x_s", i, np.int16)), 0, 400, P(out))
        pitch_index, pitch_present, ltpf_active, nbits_ltpf = out.tolist()
        assert (ltpf_active, pitch_present, pitch_index, nbits_ltpf) == w, f"frame {i}"


def test_spectral_quantization_run():  # encoder/spectral_quantization.rs:404-480
    t = "encoder/spectral_quantization.rs::spectral_quantization_run"
    enc = Encoder()
    x_q = np.zeros(400, np.int16)
    io, gg = np.zeros(7, np.int32), np.zeros(1, F32)
    lib().lc3o_kat_quant(enc.h, P(f32(t, "x_f")), P(x_q), 1200, 3, 42, 11, P(io), P(gg))
    assert x_q.tolist() == kat(t, "x_q_expected")
    # golden 24.7091141 == 0x41C5AC44: NOT the correctly rounded 10^(39/28); pins the libm-crate powf
    assert gg.view(np.uint32)[0] == 0x41C5AC44 and gg[0] == F32(24.7091141)
    gg_ind, nbits_spec, nbits_lsb, nbits_trunc, lsb_mode, rate_flag, lastnz_trunc = io.tolist()
    assert (lastnz_trunc, lsb_mode, gg_ind, rate_flag, nbits_lsb) == (350, 0, 193, 512, 107)


def test_noise_level_estimation_run():  # encoder/noise_level_estimation.rs:65-137
    t = "encoder/noise_level_estimation.rs::noise_level_estimation_run"
    lib().lc3o_kat_noise_factor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_float]
    r = lib().lc3o_kat_noise_factor(FS, US, P(f32(t, "x_f")), P(kat(t, "x_q", 0, np.int16)), 4, 24.709114)
    assert r == 6


def test_bitstream_encoding_run():  # encoder/bitstream_encoding.rs:457-531
    t = "encoder/bitstream_encoding.rs::bitstream_encoding_run"
    rc_order = np.array(kat(t, "rc_order"), np.int32)
    rc_i = np.array(kat(t, "rc_i"), np.int32)
    x_q = kat(t, "x_q", 0, np.int16)
    res = np.array(kat(t, "res_bits"), np.uint8)
    sns = np.array([8, 17, 3, 0, 0, 15253432], np.int64)
    out = np.zeros(150, np.uint8)
    lib().lc3o_kat_bitstream(FS, US, 4, 3, 350, 0, 193, 2, P(rc_order), P(rc_i), 0, 1, 0, 0, P(sns), 6, 512, 107,
                             P(x_q), P(res), int(res.size), P(out), 150)
    assert out.tolist() == kat(t, "buf_out_expected")


def test_buffer_writer_semantics():  # encoder/buffer_writer.rs:75-98 (through the frame writer)
    # write_bool_backward is LSB-first from the last byte: a frame whose only non-zero side field
    # is noise_factor shows the bit positions.  nbits_side_written(1200) with bp_side=140, mask=4 -> 74:
    assert 1200 - (8 * 140 + 8 - 2) == 74


def test_lc3_encode_channel():  # encoder/lc3_encoder.rs:314-369 -- end-to-end encode KAT
    t = "encoder/lc3_encoder.rs::lc3_encode_channel"
    out = Encoder().encode_frame(kat(t, "samples_in", 0, np.int16), 150)
    assert out.tolist() == kat(t, "buf_out_expected")


def test_stage_chain_matches_end_to_end():
    # the reference's stage KATs chain into the same 150 bytes
    a = kat("encoder/bitstream_encoding.rs::bitstream_encoding_run", "buf_out_expected")
    b = kat("encoder/lc3_encoder.rs::lc3_encode_channel", "buf_out_expected")
    assert a == b


# ---------------------------------------------------------------- decoder/
def test_buffer_reader():  # decoder/buffer_reader.rs:123-169
    L = lib()
    val, tail = ctypes.c_uint32(), ctypes.c_int()
    buf = np.array([248, 52, 26, 166, 60], np.uint8)
    assert L.lc3o_kat_read_tail_usize(P(buf), 5, 0, 23, 5, ctypes.byref(val), ctypes.byref(tail)) == 0
    assert val.value == 8
    buf = np.array([0b00011011, 0b00001100], np.uint8)
    assert L.lc3o_kat_read_tail_usize(P(buf), 2, 0, 0, 3, ctypes.byref(val), ctypes.byref(tail)) == 0
    assert val.value == 4
    assert L.lc3o_kat_read_tail_usize(P(buf), 2, 0, tail.value, 8, ctypes.byref(val), ctypes.byref(tail)) == 0
    assert val.value == 97
    buf = np.array([0b01001000], np.uint8)
    bits = []
    for i in range(8):
        b = ctypes.c_int()
        assert L.lc3o_kat_read_tail_bool(P(buf), 1, 0, i, ctypes.byref(b)) == 0
        bits.append(b.value)
    assert bits == [0, 0, 0, 1, 0, 0, 1, 0]


def test_read_side_info():  # decoder/side_info_reader.rs:208-237
    buf = np.array(kat("decoder/side_info_reader.rs::read_side_info_test", "buf"), np.uint8)
    out, tail = np.zeros(20, np.int64), ctypes.c_int()
    assert lib().lc3o_kat_side_info(P(buf), 8, 4, 400, P(out), ctypes.byref(tail)) == 0
    assert out.tolist() == [4, 398, 0, 184, 2, 1, 1, 25, 1, 0, 0, 307189, 0, 1, 0, 0, 0, 0, 0, 6]


SI_ARITH = [4, 400, 0, 204, 2, 1, 0, 13, 4, 1, 0, 1718290, 2, 0, 0, 0, 0, 0, 0, 3]  # arithmetic_codec.rs:420-444


def test_arithmetic_decode():  # decoder/arithmetic_codec.rs:415-474
    t = "decoder/arithmetic_codec.rs::arithmetic_decode"
    buf = np.array(kat(t, "buf"), np.uint8)
    x = np.zeros(400, np.int32)
    io, res = np.zeros(22, np.int32), np.zeros(480, np.uint8)
    si = np.array(SI_ARITH, np.int64)
    assert lib().lc3o_kat_arith(P(buf), 150, 0, 64, 4, 400, P(si), 1, P(x), P(io), P(res)) == 0
    assert io[20] == 0 and io[21] == 1200 and io[19] == 56909
    assert io[2:18].tolist() == [6, 10, 7, 8, 7, 9, 7, 7, 0, 0, 0, 0, 0, 0, 0, 0]
    assert res[: io[18]].astype(bool).tolist() == kat(t, index=3)
    assert io[0:2].tolist() == [8, 0]


def test_residual_spectrum_decode():  # decoder/residual_spectrum.rs:47-107
    t = "decoder/residual_spectrum.rs::residual_spectrum_decode"
    bits = np.array(kat(t, "residual_bits"), np.uint8)
    x = f32(t, "x_hat").copy()
    lib().lc3o_dec_residual(0, P(bits), int(bits.size), P(x), 400)
    assert_same_f32(x, f32(t, "x_hat_expected"), "residual")


def test_decode_noise_filling():  # decoder/noise_filling.rs:65-146
    t = "decoder/noise_filling.rs::decode_noise_filling"
    x = f32(t, "spec_lines_float").copy()
    xi = kat(t, "spec_lines_int", 0, np.int32)
    lib().lc3o_dec_noise_filling(0, 56909, 4, 1, 3, P(xi), P(x), 400)
    assert_same_f32(x, f32(t, "x_hat_expected"), "noise filling")


def test_global_gain_decode():  # decoder/global_gain.rs:33-39
    x = np.array([1.0, 10.0, 100.0], F32)
    lib().lc3o_dec_global_gain(1200, 4, 204, P(x), 3)
    assert_same_f32(x, np.array([61.0540199, 610.540199, 6105.40199], F32), "global gain")


def test_tns_decode():  # decoder/temporal_noise_shaping.rs:147-238
    t = "decoder/temporal_noise_shaping.rs::decode_test"
    x = f32(t, "spec_lines").copy()
    order = np.array([8, 0], np.int32)
    rc_i = np.array(kat(t, "reflect_coef_ints") + [0] * 8, np.int32)
    lib().lc3o_dec_tns(1, 4, 2, P(order), P(rc_i), P(x))
    assert_same_f32(x, f32(t, "spec_lines_expected"), "tns decode")


def test_sns_decode():  # decoder/spectral_noise_shaping.rs:244-350 -- pins fast_math::exp2_raw
    t = "decoder/spectral_noise_shaping.rs::spectral_noise_shaping_decode"
    x = f32(t, "spec_lines").copy()
    si = np.array(SI_ARITH, np.int64)
    lib().lc3o_kat_dec_sns(FS, US, P(si), P(x))
    assert_same_f32(x, f32(t, "spec_lines_expected"), "sns decode")


def test_mpvq_deenum():  # decoder/spectral_noise_shaping.rs:353-368
    v = np.zeros(16, np.int32)
    lib().lc3o_mpvq_deenum(10, 10, 1, 1718290, P(v))
    assert v.tolist() == [0, -2, 0, 0, 1, 1, 3, -2, 1, 0, 0, 0, 0, 0, 0, 0]
    v[:] = 0
    lib().lc3o_mpvq_deenum(6, 1, 0, 2, P(v))
    assert v.tolist() == [0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]


def test_plc_save_and_load():  # decoder/packet_loss_concealment.rs:94-106
    t = "decoder/packet_loss_concealment.rs::save_and_load"
    out = np.zeros(4, F32)
    lib().lc3o_kat_plc(4, P(f32(t, "spec_lines")), 3, P(out))
    assert_same_f32(out, f32(t, "spec_lines_expected"), "plc")


def test_modified_dct_decode():  # decoder/modified_dct.rs:174-329
    t = "decoder/modified_dct.rs::modified_dct_decode"
    dec = Decoder()
    freq = np.zeros(480, F32)
    lib().lc3o_kat_imdct(dec.h, P(f32(t, "x_hat", 0)), P(freq))
    lib().lc3o_kat_imdct(dec.h, P(f32(t, "x_hat", 1)), P(freq))
    assert_same_f32(freq, f32(t, "freq_buf_expected"), "imdct")


def test_ltpf_decode_full_cycle():  # decoder/long_term_post_filter.rs:504-1199 (transition cases 1,1,2,5,4,3)
    t = "decoder/long_term_post_filter.rs::long_term_post_filter_full_cycle"
    dec = Decoder()
    infos = [(0, 1, 134), (0, 1, 132), (1, 1, 134), (1, 1, 136), (1, 1, 136), (0, 1, 132)]
    for i, (active, present, idx) in enumerate(infos):
        x = f32(t, "freq_samples", i).copy()
        lib().lc3o_kat_dec_ltpf(dec.h, active, present, idx, 320, P(x))
        assert_same_f32(x, f32(t, "freq_samples_expected", i), f"ltpf frame {i}")


def test_ltpf_decode_activated_runs():  # decoder/long_term_post_filter.rs:434-501 (no assertion in the reference)
    t = "decoder/long_term_post_filter.rs::long_term_post_filter_activated"
    dec = Decoder()
    x = f32(t, "freq_samples").copy()
    lib().lc3o_kat_dec_ltpf(dec.h, 1, 1, 473, 600, P(x))
    assert np.all(np.isfinite(x))


def test_scale_and_round():  # decoder/output_scaling.rs:34-41
    t = "decoder/output_scaling.rs::scale_and_round_test"
    out = np.zeros(9, np.int16)
    lib().lc3o_dec_output(P(f32(t, "x_hat_ltpf")), P(out), 9)
    assert out.tolist() == [0, 0, -1, -1, 0, 1, 1, 32767, -32768]


def test_lc3_decode_channel():  # decoder/lc3_decoder.rs:374-425 -- end-to-end decode KAT
    t = "decoder/lc3_decoder.rs::lc3_decode_channel"
    rc, out = Decoder().decode_frame(np.array(kat(t, "buf_in"), np.uint8))
    assert rc == 0
    assert out.tolist() == kat(t, "samples_out_expected")


def test_decode_rejects_non_16_bit():  # decoder/lc3_decoder.rs:80-82
    t = "decoder/lc3_decoder.rs::lc3_decode_channel"
    rc, _ = Decoder().decode_frame(np.array(kat(t, "buf_in"), np.uint8), bits_per_sample=24)
    assert rc == 1
