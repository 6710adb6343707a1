"""Differential stress run of the GPU engine against the CPU oracle (test infrastructure: the oracle is only the checker).

    python tools/stress_parity.py [--streams 2048] [--frames 16] [--rounds 2] [--shapes 16384x4,2048x18] [--split 1,2,3,5] > gpurun_out/stress_parity.json

For every (sampling rate, frame duration, frame size) of the list below and every round: synthetic PCM of mixed character
(tonal / noisy / clicks from lc3-codec_amd.synth, plus band-limited, very quiet, clipping and silent streams), encoded by the
GPU engine and by the oracle (all host threads), the bitstreams compared byte for byte; then the oracle's bytes decoded by
both -- a share of the frames damaged first (bit flips, random bytes, bad-frame flags: concealment and error paths) -- and the PCM
compared sample for sample.  Exit code 1 on any difference.  The point of the volume: decisions the kernels
take from guarded tree sums fall back to the sequential sum about once in 10^4 decisions, so millions of
frames are needed to exercise both sides of every guard on real data."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [  # fs_hz, frame_us, nbytes
    (48000, 10000, 150), (48000, 10000, 40), (48000, 10000, 80), (48000, 10000, 300), (48000, 7500, 113), (48000, 7500, 30),
    (44100, 10000, 100), (32000, 10000, 80), (32000, 7500, 60), (24000, 10000, 60), (24000, 7500, 45), (16000, 10000, 40),
    (16000, 7500, 30), (16000, 10000, 120),
]
# 8 kHz: the reference has a decoder and no encoder (the constructor panics, encoder/bandwidth_detector.rs:36-37) -- decode direction only, on
# streams the ORACLE's encoder produced (its early-return path, bandwidth_detector.rs:66-71), damage included
DECODE_ONLY = [(8000, 10000, 30), (8000, 7500, 23), (8000, 10000, 60)]


def make_mixed(synth, S, T, nf, fs, rnd):
    pcm = synth.make_pcm(S, T, nf, fs, first_stream=100000 * (rnd + 1))
    rng = np.random.default_rng([77, rnd, fs, nf])
    # a quarter of the streams band-limited (lower bandwidth indices, other TNS layouts)
    nb = S // 4
    cut = float(rng.choice([3500.0, 7000.0, 11000.0])) if fs >= 32000 else float(fs) / 4.5
    pcm[:nb] = synth.make_bandlimited_pcm(nb, T, nf, fs, min(cut, fs / 2.2), seed=synth.SEED + rnd + 1)
    # very quiet streams (a few LSB: global-gain floor, zero frames) and streams scaled into clipping
    q = slice(nb, nb + S // 16)
    pcm[q] = (pcm[q].astype(np.int32) // int(rng.integers(500, 4000))).astype(np.int16)
    c = slice(nb + S // 16, nb + S // 8)
    pcm[c] = np.clip(pcm[c].astype(np.int32) * int(rng.integers(3, 9)), -32768, 32767).astype(np.int16)
    return np.ascontiguousarray(pcm)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--corrupt", type=float, default=1.0 / 48.0,
                    help="share of the frames the decode direction damages (random bit flips, whole frames of random bytes and, at 48 kHz, "
                         "frames flagged bad through the external indicator): concealment, error paths and the frames after them")
    ap.add_argument("--shapes", default="",
                    help="comma-separated launch shapes STREAMSxFRAMES that the rounds cycle through (round r uses shape r mod n), e.g. "
                         "16384x4,2048x18,4096x16: the 65 536-frame launch the benchmark times among them; default: --streams x --frames")
    ap.add_argument("--only-8khz", action="store_true", help="run the 8 kHz decode-only cases alone")
    ap.add_argument("--split", default="",
                    help="comma-separated launch lengths (frames) a round's frames are cut into, cycled, the state carried in the handles from "
                         "launch to launch, e.g. 1,2,3,5: launches shorter than, as long as and longer than the decoder's filter ring")
    a = ap.parse_args()
    import torch

    import oracle_lib as O

    pkg = importlib.import_module("lc3-codec_amd")
    synth = importlib.import_module("lc3-codec_amd.synth")
    try:
        from bench import granted_cpus
        threads = granted_cpus()[0]  # what the job may run on (cgroup quota), not the machine's core count
    except Exception:
        threads = os.cpu_count() or 8
    shapes = [tuple(int(v) for v in sh.split("x")) for sh in a.shapes.split(",") if sh] or [(a.streams, a.frames)]
    split = [int(v) for v in a.split.split(",") if v]  # launch lengths a round's frames are cut into (cycled); empty: one launch per round

    def launches(T):
        if not split:
            return [(0, T)]
        out, t, i = [], 0, 0
        while t < T:
            n = min(split[i % len(split)], T - t)
            out.append((t, n))
            t += n
            i += 1
        return out

    st = torch.cuda.current_stream().cuda_stream
    rows, bad, timeouts = [], 0, 0
    t0 = time.time()
    for fs, us, nbytes in ([] if a.only_8khz else CASES) + DECODE_ONLY:
        nf = (fs if fs != 44100 else 48000) * us // 1000000
        both = (fs, us, nbytes) not in DECODE_ONLY
        handles = {}
        enc_bad = dec_bad = damaged = frames = 0
        for rnd in range(a.rounds):
            S, T = shapes[rnd % len(shapes)]
            if S not in handles:
                handles[S] = (pkg.Lc3Encoder(S, us, fs) if both else None, pkg.Lc3Decoder(S, us, fs))
            enc, dec = handles[S]
            frames += S * T
            pcm = make_mixed(synth, S, T, nf, fs, rnd)
            dec.reset()
            ref = O.encode_batch(pcm, nbytes, fs, us, threads=threads)
            if both:
                enc.reset()
                d_pcm = torch.from_numpy(pcm).cuda()
                d_out = torch.zeros((S, T, nbytes), dtype=torch.uint8, device="cuda")
                for t_i, n_i in launches(T):  # state carried in the handle from launch to launch
                    if n_i == T:
                        enc.encode(d_pcm, d_out, nbytes, T, stream=st)
                    else:
                        part = torch.zeros((S, n_i, nbytes), dtype=torch.uint8, device="cuda")
                        enc.encode(d_pcm[:, t_i:t_i + n_i].contiguous(), part, nbytes, n_i, stream=st)
                        d_out[:, t_i:t_i + n_i] = part
                torch.cuda.synchronize()
                got = d_out.cpu().numpy()
                enc_bad += int((got != ref).any(axis=2).sum())
            # decode direction: the oracle's bytes, a share of them damaged -- the same bytes for both decoders, except for frames
            # handed to the GPU intact but FLAGGED bad, which the oracle receives with an out-of-range bandwidth index instead
            # (48 kHz: three bandwidth bits, 7 > 4; side_info_reader.rs:43-50), so that both conceal them
            crng = np.random.default_rng([91, rnd, fs, us, nbytes])
            data, flags = ref.copy(), np.zeros((S, T), np.uint8)
            kind = crng.random((S, T))
            flips = kind < a.corrupt / 3.0
            for s_i, t_i in zip(*np.nonzero(flips)):
                for _ in range(int(crng.integers(1, 4))):
                    data[s_i, t_i, int(crng.integers(0, nbytes))] ^= np.uint8(1 << int(crng.integers(0, 8)))
            garbage = (kind >= a.corrupt / 3.0) & (kind < 2.0 * a.corrupt / 3.0)
            data[garbage] = crng.integers(0, 256, (int(garbage.sum()), nbytes), dtype=np.uint8)
            for_oracle = data
            if fs == 48000:
                flags = ((kind >= 2.0 * a.corrupt / 3.0) & (kind < a.corrupt)).astype(np.uint8)
                for_oracle = data.copy()
                for_oracle[flags.astype(bool), -1] |= np.uint8(7)
            damaged += int(flips.sum() + garbage.sum() + flags.sum())
            d_in = torch.from_numpy(data).cuda()
            d_flags = torch.from_numpy(flags).cuda()
            d_dec = torch.zeros((S, T, nf), dtype=torch.int16, device="cuda")
            for t_i, n_i in launches(T):
                if n_i == T:
                    dec.decode(d_in, d_dec, nbytes, T, stream=st, d_bad_frame=d_flags)
                else:
                    part = torch.zeros((S, n_i, nf), dtype=torch.int16, device="cuda")
                    dec.decode(d_in[:, t_i:t_i + n_i].contiguous(), part, nbytes, n_i, stream=st, d_bad_frame=d_flags[:, t_i:t_i + n_i].contiguous())
                    d_dec[:, t_i:t_i + n_i] = part
            torch.cuda.synchronize()
            ref_pcm = O.decode_batch(for_oracle, nf, fs, us, threads=threads)
            dec_bad += int((d_dec.cpu().numpy() != ref_pcm).any(axis=2).sum())
        # the producer / consumer pair kernels' give-up counters (include/lc3gpu.h): a pair whose partner never answered would have
        # produced concealed / zero-filled frames -- which the comparison above also catches -- but the counter says WHY
        case_timeouts = sum((e.pair_timeouts() if e is not None else 0) + d.pair_timeouts() for e, d in handles.values())
        timeouts += case_timeouts
        rows.append({"fs_hz": fs, "frame_us": us, "nbytes": nbytes, "frames": frames, "directions": "encode + decode" if both else "decode only (oracle-encoded)",
                     "encode_frames_differing": enc_bad,
                     "decode_frames_differing": dec_bad, "decode_frames_damaged": damaged, "pair_timeouts": case_timeouts})
        bad += enc_bad + dec_bad
        print(f"{fs} {us} {nbytes}: {frames} frames, enc diff {enc_bad}, dec diff {dec_bad}, pair time-outs {case_timeouts}", file=sys.stderr)
        del handles
    total = sum(r["frames"] for r in rows)
    print(json.dumps({"what": "GPU engine vs CPU oracle, byte-exact bitstreams and sample-exact PCM (tools/stress_parity.py)",
                      "launch_shapes_streams_x_frames": ["%dx%d" % sh for sh in shapes], "launch_lengths_within_a_round": split or "one launch per round",
                      "rounds": a.rounds, "total_frames_each_direction": total,
                      "frames_differing": bad, "pair_timeouts": timeouts, "corrupt_share": a.corrupt, "frames_damaged": sum(r["decode_frames_damaged"] for r in rows), "host_threads": threads, "seconds": round(time.time() - t0, 1), "env_seq_sums": os.environ.get("LC3GPU_SEQ_SUMS"),
                      "cases": rows}))
    return 1 if (bad or timeouts) else 0


if __name__ == "__main__":
    sys.exit(main())
