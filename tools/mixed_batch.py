#!/usr/bin/env python3
"""BASELINE configs[3]: a mixed batch -- equal shares of {16, 24, 32, 44.1, 48 kHz} x {7.5, 10 ms} encode+decode and
8 kHz x {7.5, 10 ms} decode-only -- on one MI355X.  One handle pair per configuration, each on its own HIP stream; all
twelve are queued before the device is waited for.  Prints one JSON line (frames/s over all configurations; parity of a
sample of every configuration against the CPU oracle).

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
    jobs = []
    for fs, us, nb in MIXED:
        cfg = pkg.Lc3Config(fs, us)
        k = min(S, 64)
        base = synth.make_pcm(k, T, cfg.nf, fs, seed=51)
        pcm = np.tile(base, ((S + k - 1) // k, 1, 1))[:S]
        ref_b = O.encode_batch(base, nb, fs, us)
        ref_p = O.decode_batch(ref_b, cfg.nf, fs, us)
        j = dict(fs=fs, us=us, nb=nb, k=k, ref_b=ref_b, ref_p=ref_p, st=torch.cuda.Stream(),
                 enc=pkg.Lc3Encoder(S, us, fs) if fs != 8000 else None, dec=pkg.Lc3Decoder(S, us, fs),
                 d_pcm=torch.from_numpy(pcm).cuda(), d_b=torch.zeros((S, T, nb), dtype=torch.uint8, device="cuda"),
                 d_p=torch.zeros((S, T, cfg.nf), dtype=torch.int16, device="cuda"))
        if j["enc"] is None:  # decode-only share: the oracle's bitstream
            j["d_b"].copy_(torch.from_numpy(np.tile(ref_b, ((S + k - 1) // k, 1, 1))[:S]))
        jobs.append(j)

    def step():
        for j in jobs:
            if j["enc"] is not None:
                j["enc"].encode(j["d_pcm"], j["d_b"], j["nb"], T, stream=j["st"].cuda_stream)
            j["dec"].decode(j["d_b"], j["d_p"], j["nb"], T, stream=j["st"].cuda_stream)

    torch.cuda.synchronize()
    step()
    torch.cuda.synchronize()
    ok = True
    for j in jobs:
        ok = ok and np.array_equal(j["d_b"][:j["k"]].cpu().numpy(), j["ref_b"]) and np.array_equal(j["d_p"][:j["k"]].cpu().numpy(), j["ref_p"])
    for j in jobs:  # the timed steps re-encode the same frames from fresh state
        if j["enc"] is not None:
            j["enc"].reset()
        j["dec"].reset()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    frames = len(jobs) * S * T * args.steps
    print(json.dumps({"workload": "mixed batch: 10 configurations encode+decode, 2 (8 kHz) decode-only, one handle pair and HIP stream each",
                      "configurations": len(jobs), "streams_per_configuration": S, "frames_per_stream_per_step": T,
                      "steps": args.steps, "frames_per_s": frames / el, "ms_per_step": el / args.steps * 1e3,
                      "parity_first_step_all_configurations": bool(ok)}))


if __name__ == "__main__":
    main()
