import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
pkg = importlib.import_module("lc3-codec_amd"); synth = importlib.import_module("lc3-codec_amd.synth")
S, T, NB, NF = 16384, 4, 150, 480
base = synth.make_pcm(1024, T, NF, 48000)
pcm = torch.from_numpy(np.tile(base, (S // 1024, 1, 1))).cuda()
def run(nsplit, steps=20):
    n = S // nsplit
    encs = [pkg.Lc3Encoder(n, 10000, 48000) for _ in range(nsplit)]
    decs = [pkg.Lc3Decoder(n, 10000, 48000) for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    d_b = [torch.zeros((n, T, NB), dtype=torch.uint8, device="cuda") for _ in range(nsplit)]
    d_o = [torch.zeros((n, T, NF), dtype=torch.int16, device="cuda") for _ in range(nsplit)]
    parts = [pcm[i * n:(i + 1) * n].contiguous() for i in range(nsplit)]
    def step():
        for i in range(nsplit):
            st = streams[i].cuda_stream
            encs[i].encode(parts[i], d_b[i], NB, T, stream=st)
            decs[i].decode(d_b[i], d_o[i], NB, T, stream=st)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return S * T * steps / dt / 1e6, dt / steps * 1e3
for ns in (1, 2, 4, 8):
    print(ns, "splits: %.2f M frames/s, %.3f ms/step" % run(ns))
