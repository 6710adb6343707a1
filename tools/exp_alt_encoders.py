#!/usr/bin/env python3
"""TIMING-ONLY proxy (its output is not checked and its stream state is wrong by construction): how much would it be worth to let the
FRONT HALF of a group's step k + 1 run beside the quantiser / back half / packer of its step k?

Today a handle's four encoder kernels run in order on one stream, and under `quad` each group's encoder chain (front -> quantiser -> back
-> packer, ~1.1 of the 1.2 ms) is its critical path.  Letting front(k + 1) overlap the rest of step k would need double-buffered planes and
a split of the stream state inside the library.  Before building that, this script imitates its schedule with what exists: every group
gets TWO encoder handles that take the steps alternately (even steps on A, odd steps on B, each on a stream of its own), the front half of
step k + 1 ordered behind the front half of step k through the handles' stage events.  The kernels do the same amount of work as the real
thing would; the states are nonsense (each handle sees every other step), so nothing is compared.

  GPU_MAX_HW_QUEUES=8 python tools/exp_alt_encoders.py [--groups 2] [--steps 40]
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--alternate", type=int, default=1, help="0: one encoder handle per group (= bench.py's split:2+2.. on fresh streams)")
    a = ap.parse_args()
    import torch

    pkg = importlib.import_module("lc3-codec_amd")
    synth = importlib.import_module("lc3-codec_amd.synth")
    S, T, NF, NB = 16384, 4, 480, 150
    pcm = torch.from_numpy(synth.make_pcm(S, T, NF, 48000)).cuda()
    out = torch.zeros((S, T, NF), dtype=torch.int16, device="cuda")
    bufs = [torch.zeros((S, T, NB), dtype=torch.uint8, device="cuda") for _ in range(2)]
    G = a.groups
    groups = []
    for g in range(G):
        lo, hi = S * g // G, S * (g + 1) // G
        n_enc = 2 if a.alternate else 1
        d = {"lo": lo, "hi": hi, "encs": [pkg.Lc3Encoder(hi - lo, 10000, 48000) for _ in range(n_enc)], "dec": pkg.Lc3Decoder(hi - lo, 10000, 48000),
             "s_enc": [torch.cuda.Stream() for _ in range(n_enc)], "s_dec": torch.cuda.Stream(),
             "enc_done": [torch.cuda.Event() for _ in range(2)], "dec_done": [torch.cuda.Event() for _ in range(2)],
             "front_done": [torch.cuda.Event() for _ in range(n_enc)]}
        for i, e in enumerate(d["encs"]):
            d["front_done"][i].record(d["s_enc"][i])
            e.stage_event(pkg.ENC_STAGE_FRONT, d["front_done"][i])
        groups.append(d)

    def step(k):
        b = k & 1
        for d in groups:
            lo, hi = d["lo"], d["hi"]
            i = k % len(d["encs"])
            se = d["s_enc"][i]
            if k >= 2:
                se.wait_event(d["dec_done"][b])
            if len(d["encs"]) > 1 and k >= 1:
                se.wait_event(d["front_done"][1 - i])  # front(k) behind front(k - 1), as the real thing would be
            d["encs"][i].encode(pcm[lo:hi], bufs[b][lo:hi], NB, T, stream=se.cuda_stream)
            d["enc_done"][b].record(se)
            d["s_dec"].wait_event(d["enc_done"][b])
            d["dec"].decode(bufs[b][lo:hi], out[lo:hi], NB, T, stream=d["s_dec"].cuda_stream)
            d["dec_done"][b].record(d["s_dec"])

    for k in range(a.warmup):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.warmup, a.warmup + a.steps):
        step(k)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"what": "timing-only proxy: front(k+1) beside the rest of step k (two encoder handles per group taking the steps alternately)"
                      if a.alternate else "one encoder handle per group on fresh streams", "groups": G, "hip_streams": sum(len(d["s_enc"]) + 1 for d in groups),
                      "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "steps": a.steps, "ms_per_step": el / a.steps * 1e3,
                      "M_frames_per_s": S * T * a.steps / el / 1e6}))


if __name__ == "__main__":
    main()
