#!/usr/bin/env python3
"""Per-stage share of wave time inside the two kernels, from the diagnostic build with in-kernel stamps
(LC3GPU_PROFILE=1 -> liblc3gpu_prof.so).  Reads SHARES only; never quote this build's run time
(cdna_hip_programming.md section 7).  Usage on the GPU box:
    LC3GPU_PROFILE=1 python tools/stage_profile.py [streams frames] > profiles/rNN_stage_shares.txt"""
import importlib
import os
import sys

os.environ["LC3GPU_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

pkg = importlib.import_module("lc3-codec_amd")
api = importlib.import_module("lc3-codec_amd.api")
synth = importlib.import_module("lc3-codec_amd.synth")
pkg.build_native()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
base = synth.make_pcm(min(S, 1024), T, 480, 48000)
pcm = np.tile(base, ((S + len(base) - 1) // len(base), 1, 1))[:S]
d_pcm = torch.from_numpy(pcm).cuda()
d_b = torch.zeros((S, T, 150), dtype=torch.uint8, device="cuda")
d_o = torch.zeros((S, T, 480), dtype=torch.int16, device="cuda")
enc = pkg.Lc3Encoder(S, 10000, 48000)
dec = pkg.Lc3Decoder(S, 10000, 48000)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    enc.encode(d_pcm, d_b, 150, T, stream=st)
    dec.decode(d_b, d_o, 150, T, stream=st)
torch.cuda.synchronize()
api.prof_read()
for _ in range(5):
    enc.encode(d_pcm, d_b, 150, T, stream=st)
    dec.decode(d_b, d_o, 150, T, stream=st)
torch.cuda.synchronize()
acc = api.prof_read()
names = {26: "enc sns: pad, smooth, pre-emph, floor, log2", 27: "enc sns: grouping, mean, attack", 28: "enc sns: stage-1 codebooks",
         29: "enc sns: stage-2 target", 30: "enc sns: pulse search", 31: "enc sns: normalise, gain search, enumeration",
         9: "enc quant: energies+max", 10: "enc quant: gain bisection", 11: "enc quant: first quantise+bit count",
         12: "enc ltpf: shift+resample", 13: "enc ltpf: 50 Hz high-pass", 14: "enc ltpf: pitch detection",
         15: "enc ltpf: lag refinement", 22: "dec spectrum: residual+noise fill", 24: "dec spectrum: gain, serial phase (tns lattice, sns scale factors)", 25: "dec imdct: dct-iv",
         1: "enc mdct+energy", 2: "enc bandwidth+attack", 3: "enc sns (rest: synthesis, interpolation, shaping)", 4: "enc tns", 5: "enc ltpf (rest: activation)", 6: "enc quant (rest: adjust + 2nd pass)",
         7: "enc residual+noise", 8: "enc plane store", 17: "dec load parsed frame (planes) + epilogue",
         18: "dec spectrum (rest: band scaling, plc save)", 19: "dec imdct (rest: window+ola)", 20: "dec ltpf", 21: "dec output"}
frames = 5 * S * T
for lo, hi, label in ((1, 16, "encoder analysis kernel"), (26, 32, "  (encoder, continued)"), (17, 26, "decoder synthesis kernel")):
    tot = sum(acc[lo:hi])
    print(f"{label}: {tot / frames:.0f} wave-cycles per frame (sum over stages, S={S} T={T})")
    for i in range(lo, hi):
        if i in names: print(f"  {names[i]:48s} {acc[i] / frames:10.0f} cyc/frame  {100.0 * acc[i] / max(tot, 1):5.1f} %")
for base, label in ((32, "encoder"), (35, "decoder")):
    tot, mx, n = acc[base], acc[base + 1], max(acc[base + 2], 1)
    print(f"{label} waves: {n} launches-waves, mean whole-wave time {tot / n:.0f} cyc ({tot / n / T:.0f} per frame), max {mx} cyc "
          f"(x{mx / (tot / n):.2f} of the mean)")
print(f"decoder: state load {acc[38] / max(acc[37], 1):.0f} cyc per wave, frame loop {acc[39] / max(acc[37], 1):.0f} cyc per wave")
