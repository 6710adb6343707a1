#!/usr/bin/env python3
"""BASELINE configs[3]: a mixed batch -- equal shares of {16, 24, 32, 44.1, 48 kHz} x {7.5, 10 ms} encode+decode and
8 kHz x {7.5, 10 ms} decode-only -- on one MI355X, through ONE encoder handle and ONE decoder handle with per-stream
{fs, frame_us, nbytes} descriptors (lc3gpu_*_create_mixed): six kernel launches per step for the whole mixed batch.
Prints one JSON line (frames/s over all configurations; parity of every configuration against the CPU oracle).

usage: python tools/mixed_batch.py [--streams-per-config 2048] [--frames 4] [--steps 10] [--pipeline]

--pipeline (round 6): the ten encodable configurations through the library's MIXED PIPELINE OBJECT (lc3gpu_pipeline_create_mixed: two groups,
each holding half of the streams of every configuration, encoder and decoder handles on the pipeline's own streams, two byte buffers), the two
8 kHz decode-only configurations through a plain mixed decoder handle on the caller's stream beside it.
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
    ap.add_argument("--pipeline", action="store_true")
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

    if args.pipeline:
        # stream order for the pipeline: first half of every configuration's streams, then the second halves -- two groups with a share of
        # every configuration each; the ragged buffers follow that order, so the per-configuration views below are rebuilt from it
        H = S // 2
        pl_order = [(q, lo) for lo in (0, H) for q in range(10)]  # (configuration, first stream) blocks of H streams
        pl_desc = [MIXED[q] for q, _ in pl_order for _ in range(H)]
        first1 = 10 * H
        pl = pkg.Lc3Pipeline.mixed(pl_desc, group_first=[0, first1])
        dec8 = pkg.Lc3Decoder.mixed([MIXED[q] for q in (10, 11) for _ in range(S)])
        pcm_pl = np.concatenate([np.tile(cfgs[q]["base"], (rep, 1, 1))[lo:lo + H].reshape(-1) for q, lo in pl_order])
        d_pcm_pl = torch.from_numpy(pcm_pl).cuda()
        d_b_pl = [torch.zeros(sum(H * T * cfgs[q]["nb"] for q, _ in pl_order), dtype=torch.uint8, device="cuda") for _ in range(2)]
        d_out_pl = torch.zeros(sum(H * T * cfgs[q]["nf"] for q, _ in pl_order), dtype=torch.int16, device="cuda")
        d_in8 = d_bytes[off8:].clone()
        d_out8 = torch.zeros(sum(S * T * c["nf"] for c in cfgs[10:]), dtype=torch.int16, device="cuda")
        kk = [0]

        def step():
            pl.submit_mixed(d_pcm_pl, d_b_pl[kk[0] & 1], d_out_pl, T)
            dec8.decode_mixed(d_in8, d_out8, T, stream=st)
            kk[0] += 1

        torch.cuda.synchronize()
        step()
        pl.wait()
        torch.cuda.synchronize()
        ok = True
        gb, gp = d_b_pl[0].cpu().numpy(), d_out_pl.cpu().numpy()
        ob = op = 0
        for q, lo in pl_order:
            c = cfgs[q]
            b = gb[ob:ob + H * T * c["nb"]].reshape(H, T, c["nb"])
            p = gp[op:op + H * T * c["nf"]].reshape(H, T, c["nf"])
            want_b, want_p = np.tile(c["ref_b"], (rep, 1, 1))[lo:lo + H], np.tile(c["ref_p"], (rep, 1, 1))[lo:lo + H]
            ok = ok and np.array_equal(b, want_b) and np.array_equal(p, want_p)
            ob += H * T * c["nb"]
            op += H * T * c["nf"]
        g8, o8 = d_out8.cpu().numpy(), 0
        for c in cfgs[10:]:
            p = g8[o8:o8 + S * T * c["nf"]].reshape(S, T, c["nf"])
            ok = ok and np.array_equal(p, np.tile(c["ref_p"], (rep, 1, 1))[:S])
            o8 += S * T * c["nf"]
        for _ in range(args.warmup):
            step()
        pl.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        pl.wait()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({
            "config": "mixed batch through the library's mixed PIPELINE object (lc3gpu_pipeline_create_mixed: two groups, half of every configuration's "
                      "streams each): 10 rate x duration configurations encode+decode; 8 kHz x {7.5, 10 ms} decode-only through a plain mixed decoder "
                      "handle on the caller's stream beside it (BASELINE configs[3])",
            "streams_per_config": S, "frames_per_stream_per_step": T, "steps": args.steps, "frames_per_s": (12 * S * T * args.steps) / dt,
            "note": "frames_per_s counts every stream's frame once per step (ten configurations round-trip, two decode only)",
            "ms_per_step": dt / args.steps * 1e3, "parity_all_streams_first_step": bool(ok)}))
        return
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
