"""Differential volume run of the library's PIPELINE OBJECT (lc3gpu_pipeline_submit, bench.py's default arrangement) against the CPU oracle
(test infrastructure: the oracle is only the checker).

    python tools/stress_pipeline.py [--streams 16384] [--frames 64] [--per-step 4] [--rounds 4] > gpurun_out/stress_pipeline.json

Every round: `streams` synthetic streams of `frames` consecutive frames (48 kHz / 10 ms / 150 B), submitted `per-step` frames at a time with
two alternating byte buffers -- exactly what the benchmark times, state carried over frames / per-step submissions -- and EVERY bitstream
byte and PCM sample of the round compared with oracle encoders / decoders that walked the same streams frame by frame.  Exit code 1 on a
difference."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=16384)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--per-step", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    import torch

    import oracle_lib as O
    from bench import granted_cpus

    pkg = importlib.import_module("lc3-codec_amd")
    synth = importlib.import_module("lc3-codec_amd.synth")
    S, R, T = a.streams, a.frames, a.per_step
    assert R % T == 0
    thr = granted_cpus()[0]
    pl = pkg.Lc3Pipeline(S, 10000, 48000)
    rows, bad, t0 = [], 0, time.time()
    for rnd in range(a.rounds):
        pcm = synth.make_pcm_parallel(S, R, 480, 48000, first_stream=1000003 * (rnd + 1), workers=thr)
        d_rot = torch.from_numpy(pcm).cuda().reshape(S, R // T, T, 480).permute(1, 0, 2, 3).contiguous()
        d_bytes = [torch.zeros((S, T, 150), dtype=torch.uint8, device="cuda") for _ in range(2)]
        d_all_b = torch.zeros((R // T, S, T, 150), dtype=torch.uint8, device="cuda")
        d_all_p = torch.zeros((R // T, S, T, 480), dtype=torch.int16, device="cuda")
        side = torch.cuda.Stream()
        pl.reset()
        for k in range(R // T):
            pl.submit(d_rot[k], d_bytes[k & 1], d_all_p[k], 150, T)
            # keep the step's bytes: copied on a stream of ours that joins the pipeline; the pipeline's next use of the buffer follows the copy
            pl.join(side.cuda_stream)
            with torch.cuda.stream(side):
                d_all_b[k].copy_(d_bytes[k & 1], non_blocking=True)
            pl.follow(side.cuda_stream)
        pl.wait()
        torch.cuda.synchronize()
        got_b = d_all_b.permute(1, 0, 2, 3).reshape(S, R, 150).cpu().numpy()
        got_p = d_all_p.permute(1, 0, 2, 3).reshape(S, R, 480).cpu().numpy()
        ref_b = O.encode_batch(pcm, 150, 48000, 10000, threads=thr)
        eb = int((got_b != ref_b).any(axis=2).sum())
        ref_p = O.decode_batch(ref_b, 480, 48000, 10000, threads=thr)
        dp = int((got_p != ref_p).any(axis=2).sum())
        timeouts = sum(g["enc"].pair_timeouts() + g["dec"].pair_timeouts() for g in pl.groups)
        rows.append({"round": rnd, "frames": S * R, "encode_frames_differing": eb, "decode_frames_differing": dp, "pair_timeouts": timeouts})
        bad += eb + dp + timeouts
        print(f"round {rnd}: {S * R} frames, enc diff {eb}, dec diff {dp}, pair time-outs {timeouts}", file=sys.stderr)
    print(json.dumps({"what": "lc3gpu_pipeline_submit (two groups, four HIP streams, two alternating byte buffers, state carried over %d submissions of %d frames per "
                              "stream) vs the CPU oracle: every bitstream byte and PCM sample (tools/stress_pipeline.py)" % (R // T, T),
                      "streams": S, "frames_per_stream": R, "rounds": a.rounds, "total_frames_each_direction": S * R * a.rounds, "frames_differing": bad,
                      "host_threads": thr, "seconds": round(time.time() - t0, 1), "rounds_detail": rows}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
