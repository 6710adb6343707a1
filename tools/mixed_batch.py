#!/usr/bin/env python3
"""BASELINE configs[3]: a mixed batch -- equal shares of {16, 24, 32, 44.1, 48 kHz} x {7.5, 10 ms} encode+decode and
8 kHz x {7.5, 10 ms} decode-only -- on one MI355X, through ONE encoder handle and ONE decoder handle with per-stream
{fs, frame_us, nbytes} descriptors (lc3gpu_*_create_mixed): six kernel launches per step for the whole mixed batch.
Prints one JSON line (frames/s over all configurations; parity of every configuration against the CPU oracle).

usage: python tools/mixed_batch.py [--streams-per-config 2048] [--frames 4] [--steps 10]
"""
import argparse, importlib, json, os, sys, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MIXED = [(16000, 10000, 40), (24000, 10000, 60), (32000, 10000, 80), (44100, 10000, 110), (48000, 10000, 150),
         (16000, 7500, 30), (24000, 7500, 45), (32000, 7500, 60), (44100, 7500, 83), (48000, 7500, 113),
         (8000, 10000, 30), (8000, 7500, 23)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams-per-config", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import torch

    import oracle_lib as O
    pkg = importlib.import_module("lc3-codec_amd")
    synth = importlib.import_module("lc3-codec_amd.synth")
    S, T = args.streams_per_config, args.frames
    k = min(S, 64)  # distinct streams per configuration, tiled to S
    cfgs = []
    for fs, us, nb in MIXED:
        c = pkg.Lc3Config(fs, us)
        base = synth.make_pcm(k, T, c.nf, fs, seed=51)
        ref_b = O.encode_batch(base, nb, fs, us)
        cfgs.append(dict(fs=fs, us=us, nb=nb, nf=c.nf, base=base, ref_b=ref_b, ref_p=O.decode_batch(ref_b, c.nf, fs, us)))
    # stream order: all streams of configuration 0, then 1, ...; the 8 kHz (decode-only) configurations last, so that the
    # encoder's output buffer is the head of the decoder's input buffer
    enc_desc = [MIXED[q] for q in range(10) for _ in range(S)]
    dec_desc = [MIXED[q] for q in range(12) for _ in range(S)]
    enc, dec = pkg.Lc3Encoder.mixed(enc_desc), pkg.Lc3Decoder.mixed(dec_desc)
    rep = (S + k - 1) // k
    pcm_in = np.concatenate([np.tile(c["base"], (rep, 1, 1))[:S].reshape(-1) for c in cfgs[:10]])
    d_pcm = torch.from_numpy(pcm_in).cuda()
    n_bytes = [S * T * c["nb"] for c in cfgs]
    d_bytes = torch.zeros(sum(n_bytes), dtype=torch.uint8, device="cuda")
    off8 = sum(n_bytes[:10])
    tail = np.concatenate([np.tile(c["ref_b"], (rep, 1, 1))[:S].reshape(-1) for c in cfgs[10:]])
    d_bytes[off8:].copy_(torch.from_numpy(tail))
    d_out = torch.zeros(sum(S * T * c["nf"] for c in cfgs), dtype=torch.int16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def step():
        enc.encode_mixed(d_pcm, d_bytes, T, stream=st)
        dec.decode_mixed(d_bytes, d_out, T, stream=st)

    torch.cuda.synchronize()
    step()
    torch.cuda.synchronize()
    ok = True
    gb, gp = d_bytes.cpu().numpy(), d_out.cpu().numpy()
    ob = op = 0
    for q, c in enumerate(cfgs):  # first step: fresh state, every stream of every configuration against the oracle
        b = gb[ob:ob + n_bytes[q]].reshape(S, T, c["nb"])
        p = gp[op:op + S * T * c["nf"]].reshape(S, T, c["nf"])
        for r in range(0, S, k):
            m = min(k, S - r)
            ok = ok and np.array_equal(b[r:r + m], c["ref_b"][:m]) and np.array_equal(p[r:r + m], c["ref_p"][:m])
        ob += n_bytes[q]
        op += S * T * c["nf"]
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = (10 + 12) * S * T * args.steps  # encoded frames + decoded frames ...
    roundtrips = 10 * S * T * args.steps     # ... of which these many went through encode AND decode
    print(json.dumps({
        "config": "mixed batch: 10 rate x duration configurations encode+decode, 8 kHz x {7.5, 10 ms} decode-only; ONE encoder handle and "
                  "ONE decoder handle with per-stream descriptors, 6 kernel launches per step (BASELINE configs[3])",
        "streams_per_config": S, "frames_per_stream_per_step": T, "steps": args.steps,
        "frames_per_s": (12 * S * T * args.steps) / dt,
        "note": "frames_per_s counts every stream's frame once per step (ten configurations round-trip, two decode only)",
        "encode_frames": 10 * S * T * args.steps, "decode_frames": 12 * S * T * args.steps, "ms_per_step": dt / args.steps * 1e3,
        "parity_all_streams_first_step": bool(ok)}))
    del frames, roundtrips


if __name__ == "__main__":
    main()
