#!/usr/bin/env python3
"""BASELINE config 5: a stereo pair (2 channels), one 10 ms frame per step, 150 bytes per channel, INTERLEAVED PCM as it
arrives from a WAV stream; host-observed submit -> complete latency of one encode+decode step (pinned host buffers, both PCIe
copies included) and parity of everything that was produced against the CPU oracle.
Usage: python tools/latency_stereo.py [n_steps=6000]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def run(n_steps=6000):
    import torch

    import oracle_lib as O

    pkg = importlib.import_module("lc3-codec_amd")
    synth = importlib.import_module("lc3-codec_amd.synth")
    N, NF, NB = n_steps, 480, 150
    pcm = synth.make_pcm(2, N, NF, 48000, seed=5)              # [2][N][480]
    inter = np.ascontiguousarray(pcm.transpose(1, 2, 0))        # [N][480][2]: interleaved L/R, the order of a WAV stream
    enc = pkg.Lc3Encoder(2, 10000, 48000)
    dec = pkg.Lc3Decoder(2, 10000, 48000)
    d_in = torch.zeros((1, NF, 2), dtype=torch.int16, device="cuda")
    d_b = torch.zeros((1, 2, NB), dtype=torch.uint8, device="cuda")
    d_o = torch.zeros((1, NF, 2), dtype=torch.int16, device="cuda")
    h_in = torch.zeros((1, NF, 2), dtype=torch.int16).pin_memory()
    h_b = torch.zeros((1, 2, NB), dtype=torch.uint8).pin_memory()
    h_o = torch.zeros((1, NF, 2), dtype=torch.int16).pin_memory()
    st = torch.cuda.current_stream().cuda_stream
    out_b = np.zeros((N, 2, NB), np.uint8)
    out_p = np.zeros((N, NF, 2), np.int16)
    lat = np.zeros(N)
    t_in = torch.from_numpy(inter)
    for i in range(N):
        t0 = time.perf_counter()
        h_in[0] = t_in[i]                                       # the step's 480 stereo samples, as they arrive
        d_in.copy_(h_in, non_blocking=True)
        enc.encode(d_in, d_b, NB, 1, stream=st, layout="interleaved")
        dec.decode(d_b, d_o, NB, 1, stream=st, layout="interleaved")
        h_b.copy_(d_b, non_blocking=True)
        h_o.copy_(d_o, non_blocking=True)
        torch.cuda.synchronize()
        lat[i] = time.perf_counter() - t0
        out_b[i] = h_b[0].numpy()
        out_p[i] = h_o[0].numpy()
    ref_b = O.encode_batch(pcm, NB)
    ref_p = O.decode_batch(ref_b, NF)
    w = lat[50:] * 1e6
    return {"config": "stereo pair, 48 kHz / 10 ms / 150 B per channel, interleaved PCM, one frame per step (BASELINE config 5)",
            "steps": N, "latency_us": {"p50": float(np.percentile(w, 50)), "p99": float(np.percentile(w, 99)),
                                       "mean": float(w.mean()), "max": float(w.max())},
            "realtime_budget_us": 10000,
            "bitstream_exact": bool(np.array_equal(out_b, ref_b.transpose(1, 0, 2))),
            "pcm_max_abs_diff": int(np.abs(out_p.astype(np.int32) - ref_p.transpose(1, 2, 0).astype(np.int32)).max())}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 6000)))
