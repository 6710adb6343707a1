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
ENC_ONLY = os.environ.get("LC3_PROF_ENC_ONLY", "0") == "1"  # the encoder then also uses the decoder's stamp ids (finer sections)
for _ in range(2):
    enc.encode(d_pcm, d_b, 150, T, stream=st)
    if not ENC_ONLY:
        dec.decode(d_b, d_o, 150, T, stream=st)
torch.cuda.synchronize()
api.prof_read()
for _ in range(5):
    enc.encode(d_pcm, d_b, 150, T, stream=st)
    if not ENC_ONLY:
        dec.decode(d_b, d_o, 150, T, stream=st)
torch.cuda.synchronize()
acc = api.prof_read()
names = {22: "front: mdct load, window, fold", 23: "front: mdct dct-iv", 1: "front: mdct scale + band energies", 24: "front: bandwidth", 2: "front: attack",
         28: "back: load mid plane, shaping, tns autocorrelation", 29: "back: tns levinson + lpc->rc", 30: "back: tns quantisation, orders, bits", 26: "front: sns pad, smooth, pre-emph, floor, log2",
         27: "front: sns grouping, mean, attack smoothing", 3: "front: targets + spectrum -> mid plane",
         12: "front: ltpf shift+resample", 13: "front: ltpf 50 Hz high-pass", 14: "front: ltpf pitch detection",
         15: "front: ltpf lag refinement", 5: "front: ltpf activation, ring store",
         4: "back: tns lattice", 9: "back: quant energies+max", 10: "back: quant gain bisection",
         11: "back: quant first quantise+bit count", 6: "back: quant adjust + 2nd pass", 7: "back: residual+noise",
         8: "back: plane store",
         17: "dec load reconstructed frame (plane)", 18: "dec plc save/load", 25: "dec imdct: dct-iv",
         19: "dec imdct: window+ola", 20: "dec ltpf", 21: "dec output"}
frames = 5 * S * T
if ENC_ONLY:
    names.update({20: "front: mdct scale", 21: "front: mdct x*x/width", 1: "front: mdct band sums", 17: "front: ltpf 17-lag correlations",
                  15: "front: ltpf lag scans", 18: "back: pick up mid plane", 19: "back: sns shaping", 28: "back: tns autocorrelation"})
enc_ids = [22, 23, 1, 24, 2, 26, 27, 3, 12, 13, 14, 15, 5, 28, 29, 30, 4, 9, 10, 11, 6, 7, 8]
if ENC_ONLY:
    enc_ids = [22, 23, 20, 21, 1, 24, 2, 26, 27, 3, 12, 13, 14, 17, 15, 5, 18, 19, 28, 29, 30, 4, 9, 10, 11, 6, 7, 8]
dec_ids = [] if ENC_ONLY else [17, 18, 25, 19, 20, 21]  # (with LC3_PROF_ENC_ONLY the decoder's ids are encoder sub-stages, listed above)
for ids, label in ((enc_ids, "encoder analysis kernels (front + back)"), (dec_ids, "decoder synthesis kernel")):
    if not ids:
        continue
    tot = sum(acc[i] for i in ids)
    print(f"{label}: {tot / frames:.0f} wave-cycles per frame (sum over stages, S={S} T={T})")
    for i in ids:
        print(f"  {names[i]:48s} {acc[i] / frames:10.0f} cyc/frame  {100.0 * acc[i] / max(tot, 1):5.1f} %")
for base, label in ((32, "encoder front+back"), (35, "decoder")):
    tot, mx, n = acc[base], acc[base + 1], max(acc[base + 2], 1)
    print(f"{label} waves: {n} launches-waves, mean whole-wave time {tot / n:.0f} cyc ({tot / n / T:.0f} per frame), max {mx} cyc "
          f"(x{mx / (tot / n):.2f} of the mean)")
print(f"decoder: state load {acc[38] / max(acc[37], 1):.0f} cyc per wave, frame loop {acc[39] / max(acc[37], 1):.0f} cyc per wave")
pn = ["side information", "TNS data", "spectral data (range decoder)", "zero fill, LSB refinement", "reconstruction set-up (SNS scale factors, gain, TNS coefficients)",
      "reconstruction pass over the lines", "rest"]
waves = 5 * S * T / 64
ptot = sum(acc[40:47])
print(f"parse kernel: {ptot / waves:.0f} cycles per wave of 64 frames")
for i, nme in enumerate(pn):
    print(f"  {nme:70s} {acc[40 + i] / waves:10.0f} cyc/wave  {100.0 * acc[40 + i] / max(ptot, 1):5.1f} %")
kn = ["table + frame staging (to the first barrier)", "side information", "TNS data", "spectral data (range coder)", "residual bits / LSBs",
      "range coder finish", "wait for the workgroup's slowest wave + copy-out"]
kw = max(acc[55], 1)
ktot = sum(acc[48:55])
print(f"pack kernel: {ktot / kw:.0f} cycles per wave of 64 frames")
for i, nme in enumerate(kn):
    print(f"  {nme:70s} {acc[48 + i] / kw:10.0f} cyc/wave  {100.0 * acc[48 + i] / max(ktot, 1):5.1f} %")
