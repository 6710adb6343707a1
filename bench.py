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
    """Time the CPU oracle (a C port of the reference; the reference itself is Rust and cannot be built on the box)
    on a bounded sample of the same workload, one oracle channel per stream, spread over every host core."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cores = os.cpu_count() or 1
    S, T, _ = pcm_sample.shape
    n_streams = min(S, max(4 * cores, 1024))
    sample = pcm_sample[:n_streams]
    done, t0 = 0, time.perf_counter()
    while True:  # repeat the sample until ~budget_s of wall time has been spent
        b = O.encode_batch(sample, NBYTES, FS, US, threads=cores)
        O.decode_batch(b, NF, FS, US, threads=cores)
        done += n_streams * T
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {
        "value": done / dt,
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{done} frames ({n_streams} streams x {T} frames, repeated) of the bench workload, encode+decode, "
                  f"{cores} host threads, {dt:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=16384, help="streams per GPU")
    ap.add_argument("--frames", type=int, default=4, help="consecutive frames per stream per step")
    ap.add_argument("--hip-streams", type=int, default=1,
                    help="split the rank's streams over this many codec handle pairs, each on its own HIP stream, so that "
                         "the low-occupancy lane-per-frame kernels of one part overlap the wave kernels of another "
                         "(default 1: one stream, clean per-kernel timing)")
    ap.add_argument("--no-overlap-probe", action="store_true",
                    help="skip the extra, untimed-for-`value` leg that repeats the steps with the batch split over four handle "
                         "pairs on four HIP streams (reported as `overlapped`)")
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
    # one encoder/decoder handle pair per HIP stream (handles are independent; a handle's launches are ordered)
    NP = max(1, args.hip_streams)
    assert S % NP == 0, "--streams must be a multiple of --hip-streams"
    SP = S // NP
    encs = [pkg.Lc3Encoder(SP, pkg.FrameDuration.TenMs, pkg.SamplingFrequency.Hz48000) for _ in range(NP)]
    decs = [pkg.Lc3Decoder(SP, pkg.FrameDuration.TenMs, pkg.SamplingFrequency.Hz48000) for _ in range(NP)]
    hip_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(NP - 1)]
    enc, dec = encs[0], decs[0]

    def step():
        for p in range(NP):
            st = hip_streams[p].cuda_stream
            lo, hi = p * SP, (p + 1) * SP
            encs[p].encode(d_pcm[lo:hi], d_bytes[lo:hi], NBYTES, T, stream=st)
            decs[p].decode(d_bytes[lo:hi], d_out[lo:hi], NBYTES, T, stream=st)

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
        for h in encs + decs:
            h.reset()
    elif not args.no_parity:
        step()
        torch.cuda.synchronize()
        for h in encs + decs:
            h.reset()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()

    # timed region: exactly K steps; per-kernel durations from events on the launch stream
    # the C ABI records HIP events around each of its kernels on the launch stream
    for h in encs + decs:
        h.timing(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    ef = ev = eb = ep = en = dp = ds = dn = 0.0
    for h in encs:
        a, v, b, p_, n = h.timing(False)
        ef, ev, eb, ep, en = ef + a, ev + v, eb + b, ep + p_, en + n
    for h in decs:
        a, b, n = h.timing(False)
        dp, ds, dn = dp + a, ds + b, dn + n
    # launches per step = NP per kernel; scale to "per step" so that the numbers stay comparable across --hip-streams
    en, dn = en / NP, dn / NP
    kernel_ms = {
        "lc3_enc_front_kernel": ef / max(en, 1),  # analysis front half, wave per stream
        "lc3_sns_vq_kernel": ev / max(en, 1),     # SNS vector quantiser, lane per frame
        "lc3_enc_back_kernel": eb / max(en, 1),   # analysis back half, wave per stream
        "lc3_pack_kernel": ep / max(en, 1),       # bitstream packing, lane per frame
        "lc3_parse_kernel": dp / max(dn, 1),      # frame parsing + spectrum reconstruction, lane per frame
        "lc3_decode_kernel": ds / max(dn, 1),     # synthesis, wave per stream
    }

    # max time over ranks, frames summed over ranks (the only collective of the job)
    D = importlib.import_module("lc3-codec_amd.dist")
    elapsed, total_frames, _, _ = D.reduce_report(dist, "cuda", elapsed, frames_per_step * args.steps)

    # Informational leg, outside the timed region and not part of `value`: the same batch split over four handle pairs, each
    # on its own HIP stream, so that one pair's lane-per-frame kernels (one wave per SIMD) run under another pair's
    # wave-per-stream kernels.  What a caller who pipelines independent batches gets; per-kernel event timings are not taken
    # here (under overlap every launch's duration includes the others').
    overlapped = None
    if world == 1 and NP == 1 and not args.no_overlap_probe and S % 4 == 0:
        Q, SQ = 4, S // 4
        qe = [pkg.Lc3Encoder(SQ, pkg.FrameDuration.TenMs, pkg.SamplingFrequency.Hz48000) for _ in range(Q)]
        qd = [pkg.Lc3Decoder(SQ, pkg.FrameDuration.TenMs, pkg.SamplingFrequency.Hz48000) for _ in range(Q)]
        qs = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(Q - 1)]

        def qstep():
            for p in range(Q):
                lo, hi = p * SQ, (p + 1) * SQ
                qe[p].encode(d_pcm[lo:hi], d_bytes[lo:hi], NBYTES, T, stream=qs[p].cuda_stream)
                qd[p].decode(d_bytes[lo:hi], d_out[lo:hi], NBYTES, T, stream=qs[p].cuda_stream)

        for _ in range(args.warmup):
            qstep()
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        for _ in range(args.steps):
            qstep()
        torch.cuda.synchronize()
        qel = time.perf_counter() - q0
        overlapped = {"hip_streams": Q, "value": frames_per_step * args.steps / qel, "unit": "frames/s",
                      "ms_per_step": qel / args.steps * 1e3,
                      "note": "same batch as four independent quarter batches on four HIP streams; informational, not `value`"}
        del qe, qd

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(pcm_host)
        # roofline of the dominant kernel of a step.  Algorithmic bytes per frame: the analysis kernel
        # reads 2*nf of PCM, the packer writes nbytes, the parser reads nbytes, the synthesis kernel writes 2*nf.
        # front half reads 2*nf of PCM, the packer writes nbytes, the parser reads nbytes, the synthesis kernel writes 2*nf;
        # the vector quantiser and the back half touch no algorithmic bytes (their traffic is the planes between kernels)
        alg_bytes = {"lc3_enc_front_kernel": 2 * NF, "lc3_sns_vq_kernel": 0, "lc3_enc_back_kernel": 0, "lc3_pack_kernel": NBYTES,
                     "lc3_parse_kernel": NBYTES, "lc3_decode_kernel": 2 * NF}
        # The analysis of a frame is three kernels since the SNS vector quantiser moved to its own lane-per-frame stage
        # (front half, quantiser, back half): they are reported as ONE unit for the roofline, with the frame's PCM as its
        # algorithmic bytes -- otherwise the longest single kernel (the back half) would have no algorithmic bytes at all.
        groups = {"analysis (lc3_enc_front_kernel + lc3_sns_vq_kernel + lc3_enc_back_kernel)":
                  ["lc3_enc_front_kernel", "lc3_sns_vq_kernel", "lc3_enc_back_kernel"],
                  "lc3_pack_kernel": ["lc3_pack_kernel"], "lc3_parse_kernel": ["lc3_parse_kernel"],
                  "lc3_decode_kernel": ["lc3_decode_kernel"]}
        group_ms = {g: sum(kernel_ms[k] for k in ks) for g, ks in groups.items()}
        dom = max(group_ms, key=group_ms.get)
        dom_ms, alg = group_ms[dom], sum(alg_bytes[k] for k in groups[dom])
        value = total_frames / elapsed
        achieved = frames_per_step * alg / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch of the dominant unit from the committed PMC passes (rocprofv3 cannot run inside this
        # process); scaled to this run's frames per launch
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")) as f:
                tj = json.load(f)
            traffic = sum(tj["kernels"][k]["fetch_size_kb_per_launch"] + tj["kernels"][k]["write_size_kb_per_launch"]
                          for k in groups[dom]) * 1024.0 * frames_per_step / tj["frames_per_launch"]
        except (OSError, KeyError, ValueError):
            traffic = None
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
                "hip_streams": NP,
            },
            "kernel_ms": kernel_ms,
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_frame": alg,
                "roundtrip_algorithmic_bytes_per_frame": ALG_BYTES_ENC + ALG_BYTES_DEC,
                "roundtrip_achieved_GBs": value * (ALG_BYTES_ENC + ALG_BYTES_DEC) / 1e9 / world,
                "note": "instruction/latency-bound, not HBM-bound (SURVEY 8d honesty note): ~60 flop per algorithmic byte; "
                        "traffic = FETCH_SIZE + WRITE_SIZE of this kernel from profiles/hbm_traffic_latest.json (bytes per "
                        "launch); instruction mix and wait counters in profiles/r01_v20_pmc_summary.csv, DESIGN.md section 5",
            },
            "cpu_baseline": cpu,
            "overlapped": overlapped,
            "parity": parity,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
