#!/usr/bin/env python3
"""bench.py -- LC3 frames/sec (encode+decode) @ 48 kHz / 10 ms on 1..8 MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic PCM already resident in HBM:
encode S streams x T frames (150-byte frames) and decode the bitstream that was just produced,
S*T = 65 536 frames per GPU (BASELINE.json configs[1]).  Stream state carries from step to step
(streaming operation).  With --gpus N each rank owns its own S streams (weak scaling: frames are
independent across streams, no data-path collective); torch.distributed (RCCL) is used only for the
barrier and the final max-time / frame-count reduction.

Prints ONE JSON line on rank 0 (see README/DESIGN.md section "Measurement").
PyTorch is used for device memory, the HIP stream, events and torch.distributed only; the codec is
liblc3gpu.so (hand-written HIP) called through its C ABI.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS, US, NBYTES, NF = 48000, 10000, 150, 480
ALG_BYTES_ENC = 2 * NF + NBYTES   # i16 PCM read + frame bytes written   
ALG_BYTES_DEC = NBYTES + 2 * NF   # frame bytes read + i16 PCM written
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(pcm_sample, budget_s=12.0):
    """Time the CPU oracle (a C port of the reference, the reference itself is Rust and cannot be built on
    the box) on a bounded sample of the same workload, using every host core."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cores = os.cpu_count() or 1
    S, T, _ = pcm_sample.shape
    # probe to size the sample for ~budget_s of wall time
    t0 = time.perf_counter()
    probe = pcm_sample[: max(cores, 8)]
    b = O.encode_batch(probe, NBYTES, FS, US, threads=cores)
    O.decode_batch(b, NF, FS, US, threads=cores)
    dt = time.perf_counter() - t0
    rate = probe.shape[0] * T / max(dt, 1e-6)
    n_streams = int(min(S, max(cores, rate * budget_s / T)))
    sample = pcm_sample[:n_streams]
    t0 = time.perf_counter()
    b = O.encode_batch(sample, NBYTES, FS, US, threads=cores)
    O.decode_batch(b, NF, FS, US, threads=cores)
    dt = time.perf_counter() - t0
    return {
        "value": n_streams * T / dt,
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_streams} streams x {T} frames of the bench workload, encode+decode, {cores} host threads, {dt:.1f} s",
    }, b[: min(n_streams, 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=16384, help="streams per GPU")
    ap.add_argument("--frames", type=int, default=4, help="consecutive frames per stream per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (the engine has no CPU path)"

    pkg = importlib.import_module("lc3-codec_amd")
    synth = importlib.import_module("lc3-codec_amd.synth")
    S, T = args.streams, args.frames
    frames_per_step = S * T

    # synthetic input: 1024 distinct streams per rank, tiled to S (generation is host-side numpy)
    n_distinct = min(S, 1024)
    base = synth.make_pcm(n_distinct, T, NF, FS, first_stream=rank * S)
    pcm_host = np.tile(base, ((S + n_distinct - 1) // n_distinct, 1, 1))[:S]
    d_pcm = torch.from_numpy(pcm_host).cuda()
    d_bytes = torch.zeros((S, T, NBYTES), dtype=torch.uint8, device="cuda")
    d_out = torch.zeros((S, T, NF), dtype=torch.int16, device="cuda")
    enc = pkg.Lc3Encoder(S, pkg.FrameDuration.TenMs, pkg.SamplingFrequency.Hz48000)
    dec = pkg.Lc3Decoder(S, pkg.FrameDuration.TenMs, pkg.SamplingFrequency.Hz48000)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        enc.encode(d_pcm, d_bytes, NBYTES, T, stream=stream)
        dec.decode(d_bytes, d_out, NBYTES, T, stream=stream)

    # parity gate on the first step (fresh state): GPU bitstream / PCM vs the CPU oracle on a sample
    parity = None
    cpu = None
    if rank == 0 and not args.no_parity:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O

        step()
        torch.cuda.synchronize()
        k = min(n_distinct, 256)
        ref_b = O.encode_batch(pcm_host[:k], NBYTES, FS, US, threads=os.cpu_count() or 1)
        ref_p = O.decode_batch(ref_b, NF, FS, US, threads=os.cpu_count() or 1)
        got_b = d_bytes[:k].cpu().numpy()
        got_p = d_out[:k].cpu().numpy()
        parity = {
            "frames_checked": int(k * T),
            "bitstream_exact": bool(np.array_equal(got_b, ref_b)),
            "pcm_max_abs_diff": int(np.abs(got_p.astype(np.int32) - ref_p.astype(np.int32)).max()),
        }
        enc.reset()
        dec.reset()
    elif not args.no_parity:
        step()
        torch.cuda.synchronize()
        enc.reset()
        dec.reset()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()

    # timed region: exactly K steps; per-kernel durations from events on the launch stream
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        enc.encode(d_pcm, d_bytes, NBYTES, T, stream=stream)
        ev[i][1].record()
        dec.decode(d_bytes, d_out, NBYTES, T, stream=stream)
        ev[i][2].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    enc_ms = float(np.mean([ev[i][0].elapsed_time(ev[i][1]) for i in range(args.steps)]))
    dec_ms = float(np.mean([ev[i][1].elapsed_time(ev[i][2]) for i in range(args.steps)]))

    total_frames = frames_per_step * args.steps
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        fr = torch.tensor([total_frames], dtype=torch.int64, device="cuda")
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        total_frames = int(fr.item())

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            cpu, _ = cpu_baseline(pcm_host)
        # roofline of the dominant kernel (the longer of the two launches of a step)
        if enc_ms >= dec_ms:
            dom, dom_ms, alg = "lc3_encode_kernel", enc_ms, ALG_BYTES_ENC
        else:
            dom, dom_ms, alg = "lc3_decode_kernel", dec_ms, ALG_BYTES_DEC
        achieved = frames_per_step * alg / (dom_ms * 1e-3) / 1e9
        value = total_frames / elapsed
        line = {
            "metric": "LC3 frames/sec (encode+decode) @48kHz/10ms",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "65536-frame batch mono 48 kHz / 10 ms / 150-byte frames, encode+decode (BASELINE configs[1])",
                "streams_per_gpu": S,
                "frames_per_stream_per_step": T,
                "frames_per_step_per_gpu": frames_per_step,
                "nbytes": NBYTES,
                "state": "carried across steps (streaming)",
                "parallelism": f"streams sharded over {world} GPU(s), no data-path collective",
            },
            "kernel_ms": {"lc3_encode_kernel": enc_ms, "lc3_decode_kernel": dec_ms},
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_frame": alg,
                "note": "path is VALU/latency-bound, not HBM-bound (SURVEY 8d honesty note); PMC traffic in profiles/",
            },
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
