#!/usr/bin/env python3
"""A few steps of ONE shape of the 65 536-frame batch (S streams x T frames, 48 kHz / 10 ms / 150 B) on one caller stream, for the
counter passes of tools/run_profiles_modes.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE serialises kernels anyway).
usage: python tools/shape_run.py S T [cold] [--steps 6]   cold = every step from fresh state (SURVEY 8d Mode A)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    S, T, cold = int(a[0]), int(a[1]), len(a) > 2 and a[2] == "cold"
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 6
    import torch

    pkg = importlib.import_module("lc3-codec_amd")
    synth = importlib.import_module("lc3-codec_amd.synth")
    base = synth.make_pcm(min(S, 2048), T, 480, 48000)
    pcm = np.ascontiguousarray(np.tile(base, ((S + base.shape[0] - 1) // base.shape[0], 1, 1))[:S])
    d_pcm = torch.from_numpy(pcm).cuda()
    d_b = torch.zeros((S, T, 150), dtype=torch.uint8, device="cuda")
    d_o = torch.zeros((S, T, 480), dtype=torch.int16, device="cuda")
    enc, dec = pkg.Lc3Encoder(S, 10000, 48000), pkg.Lc3Decoder(S, 10000, 48000)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(steps):
        if cold:
            torch.cuda.synchronize()
            enc.reset()
            dec.reset()
        enc.encode(d_pcm, d_b, 150, T, stream=st)
        dec.decode(d_b, d_o, 150, T, stream=st)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
