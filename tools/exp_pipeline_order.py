"""Experiment (round 6): does the pipeline object's speed depend on what was created before it in the process?
python tools/exp_pipeline_order.py [scenario]   -> JSON lines on stdout.  Scenarios:
  plain     P1, then P2 beside it, then P1 again, then (P1 closed) P3
  torchpool the same after torch's stream pools (default and high priority) have been created
  closed    P1 created, timed, closed; P2 created, timed (same sizes: the allocator hands out the same memory)"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

pkg = importlib.import_module("lc3-codec_amd")
synth = importlib.import_module("lc3-codec_amd.synth")
S, T, NB, NF = 16384, 4, 150, 480
scenario = sys.argv[1] if len(sys.argv) > 1 else "plain"
pcm = np.tile(synth.make_pcm(2048, T, NF, 48000), (8, 1, 1))
d_pcm = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
bufs = [torch.zeros((S, T, NB), dtype=torch.uint8, device="cuda") for _ in range(2)]
d_out = torch.zeros((S, T, NF), dtype=torch.int16, device="cuda")


def timed(pl, steps=60, warm=10):
    for k in range(warm):
        pl.submit(d_pcm, bufs[k & 1], d_out, NB, T)
    pl.wait()
    t0 = time.perf_counter()
    for k in range(steps):
        pl.submit(d_pcm, bufs[k & 1], d_out, NB, T)
    pl.wait()
    return S * T * steps / (time.perf_counter() - t0) / 1e6


def mk():
    return pkg.Lc3Pipeline(S, 10000, 48000)


out = {"scenario": scenario}
if scenario == "torchpool":
    keep = [torch.cuda.Stream(), torch.cuda.Stream(priority=-1)]
    out["note"] = "torch's default- and high-priority stream pools exist before the first pipeline"
if scenario in ("plain", "torchpool"):
    p1 = mk()
    out["p1_first"] = [timed(p1) for _ in range(3)]
    p2 = mk()
    out["p2_beside_p1"] = [timed(p2) for _ in range(3)]
    out["p1_again"] = [timed(p1) for _ in range(2)]
    p1.close()
    out["p2_after_p1_closed"] = [timed(p2) for _ in range(2)]
    p3 = mk()
    out["p3_beside_p2"] = [timed(p3) for _ in range(3)]
elif scenario == "streams_first":
    tiny = pkg.Lc3Pipeline(8, 10000, 48000)  # four HIP streams and next to no memory, kept alive
    out["note"] = "an 8-channel pipeline (4 HIP streams, a few hundred KB) exists before the big one"
    p1 = mk()
    out["p1_beside_tiny"] = [timed(p1) for _ in range(3)]
    tiny.close()
    out["p1_after_tiny_closed"] = [timed(p1) for _ in range(2)]
    p2 = mk()
    out["p2_beside_p1"] = [timed(p2) for _ in range(2)]
elif scenario == "memory_first":
    hold = [torch.zeros(64 << 20, dtype=torch.uint8, device="cuda") for _ in range(16)]  # 1 GB in 64 MB blocks, kept alive
    out["note"] = "1 GB of device memory in 64 MB blocks allocated (torch) and held before the first pipeline"
    p1 = mk()
    out["p1_after_memory"] = [timed(p1) for _ in range(3)]
    p2 = mk()
    out["p2_beside_p1"] = [timed(p2) for _ in range(2)]
elif scenario == "host_cost":
    # host time of one submission (what a rank's CPU core spends per step: 12 kernel launches, 8 event records / waits behind the C ABI)
    p1 = mk()
    timed(p1, 20, 5)
    import statistics
    per = []
    for rep in range(5):
        p1.wait()
        t0 = time.perf_counter()
        for k in range(24):  # (short enough that the launch queue never fills and the host never blocks)
            p1.submit(d_pcm, bufs[k & 1], d_out, NB, T)
        per.append((time.perf_counter() - t0) / 24 * 1e6)
        p1.wait()
    out["host_us_per_submit"] = per
    out["host_us_per_submit_median"] = statistics.median(per)
elif scenario == "closed":
    for i in range(4):
        p = mk()
        out["p%d" % (i + 1)] = [timed(p) for _ in range(3)]
        p.close()
print(json.dumps(out))
