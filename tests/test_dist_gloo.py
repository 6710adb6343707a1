"""N > 1 path on CPU: world_size-2 gloo job that shards streams exactly as bench.py does on GPUs
(contiguous stream ranges, no data-path collective, counters reduced at the end).  The per-rank
"engine" here is the CPU wave emulator of the device code (tests only)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_streams, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import emu_lib as E

    synth = importlib.import_module("lc3-codec_amd.synth")
    D = importlib.import_module("lc3-codec_amd.dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = D.shard_range(total_streams, world, rank)
    pcm = synth.make_pcm(hi - lo, T, 480, 48000, first_stream=lo)
    data = E.encode(pcm, 150)
    out = E.decode(data, 480)
    dist.barrier()
    elapsed, frames, mism, plc = D.reduce_report(dist, "cpu", 1.0 + rank, (hi - lo) * T, mismatches=rank, plc_events=0)
    # gather the shards' bitstreams to rank 0 for comparison with the single-process result
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, data, out))
    if rank == 0:
        q.put((elapsed, frames, mism, plc, gathered))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    synth = importlib.import_module("lc3-codec_amd.synth")
    total, T, world = 6, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    elapsed, frames, mism, plc, gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert elapsed == 2.0 and frames == total * T and mism == 1 and plc == 0  # MAX time, SUM counters
    pcm = synth.make_pcm(total, T, 480, 48000)
    ref = O.encode_batch(pcm, 150)
    ref_pcm = O.decode_batch(ref, 480)
    covered = np.zeros(total, bool)
    for lo, hi, data, out in gathered:
        assert np.array_equal(data, ref[lo:hi])
        assert np.array_equal(out, ref_pcm[lo:hi])
        covered[lo:hi] = True
    assert covered.all()


def test_shard_range_partitions():
    D = importlib.import_module("lc3-codec_amd.dist")
    for total in (1, 7, 8, 65536, 1048576):
        for world in (1, 2, 4, 8):
            spans = [D.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_synth_shards_are_consistent():
    synth = importlib.import_module("lc3-codec_amd.synth")
    a = synth.make_pcm(8, 2, 480, 48000)
    b = synth.make_pcm(4, 2, 480, 48000, first_stream=4)
    assert np.array_equal(a[4:], b)
    assert np.array_equal(a, synth.make_pcm(8, 2, 480, 48000))
    assert a.dtype == np.int16 and np.abs(a).max() > 1000


def _run_bench(*extra):
    import json
    import subprocess

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--engine", "emu", "--steps", "1", "--warmup", "0",
                        *extra], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None)


def test_bench_launcher_starts_two_ranks():
    """`bench.py --gpus 2` itself starts the ranks (the driver's invocation): the same launcher, sharding and reduction path
    as on the GPU box, with CPU ranks (wave emulator of the device code, gloo)."""
    rc, line = _run_bench("--gpus", "2")
    assert rc == 0 and line is not None
    assert line["n_gpus"] == 2 and "world size 2 (gloo)" in line["config"]["parallelism"]
    assert line["scaling"] == "weak" and line["config"]["frames_per_step_per_gpu"] == 8
    assert line["parity"]["bitstream_exact"] and line["parity"]["pcm_max_abs_diff"] == 0
    assert line["parity_mismatches_all_ranks"] == 0
    # value = frames of ALL ranks / max elapsed
    assert abs(line["value"] - 2 * 8 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]


def test_bench_encode_mode_shards_the_batch():
    """BASELINE configs[2] shape (encode only, the batch's streams sharded over the ranks: strong scaling), tiny on CPU"""
    rc, line = _run_bench("--gpus", "2", "--mode", "encode", "--frames-total", "12", "--frames", "2")
    assert rc == 0 and line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["streams_per_gpu"] == 3 and line["config"]["mode"] == "encode"
    assert line["parity"]["bitstream_exact"] and line["parity_mismatches_all_ranks"] == 0
    assert abs(line["value"] - 12 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]


def test_bench_launcher_reports_a_failing_rank():
    rc, line = _run_bench("--gpus", "2", "--mode", "encode", "--frames-total", "3", "--frames", "2")
    assert rc != 0 and line is None


def test_bench_launcher_stops_the_others_when_one_rank_dies_after_the_rendezvous():
    """exactly one rank exits (code 17) after init_process_group + a barrier; rank 0 would wait in the next collective until the
    backend's timeout: the launcher sees the dead child, terminates the children it started and returns non-zero quickly"""
    import subprocess
    import time

    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--engine", "emu", "--steps", "1", "--warmup", "0", "--gpus", "2"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, LC3_BENCH_TEST_DIE_RANK="1"))
    assert p.returncode != 0
    assert "rank 1 exited with code 17" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 120.0


def test_eight_ranks_through_the_launcher():
    """the shape of the driver's 8-GPU run without an 8-GPU box: `bench.py --gpus 8` with CPU ranks (the device code under the wave emulator,
    gloo): eight fresh children, contiguous shards, one reduction; every rank gates the same number of its own streams (the sample does
    not shrink with N), the oracle's share of a rank's time is reported, and the N > 1 line points at the N = 1 line for `cpu_baseline`"""
    rc, line = _run_bench("--gpus", "8")
    assert rc == 0 and line is not None
    assert line["n_gpus"] == 8 and "world size 8 (gloo)" in line["config"]["parallelism"]
    assert line["config"]["frames_per_step_per_gpu"] == 8 and line["scaling"] == "weak"
    assert line["parity"]["bitstream_exact"] and line["parity"]["frames_checked"] == 8 and line["parity_mismatches_all_ranks"] == 0
    assert line["parity"]["oracle_seconds"] < 30.0
    assert abs(line["value"] - 8 * 8 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert line["cpu_baseline"]["value"] is None and "--gpus 1" in line["cpu_baseline"]["sample"]
    # strong scaling: 16 streams of 2 frames over 8 ranks, two each
    rc, line = _run_bench("--gpus", "8", "--mode", "encode", "--frames-total", "32", "--frames", "2")
    assert rc == 0 and line["n_gpus"] == 8 and line["config"]["streams_per_gpu"] == 2 and line["parity_mismatches_all_ranks"] == 0


def test_eight_ranks_one_dies():
    import subprocess
    import time

    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--engine", "emu", "--steps", "1", "--warmup", "0", "--gpus", "8"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, LC3_BENCH_TEST_DIE_RANK="5"))
    assert p.returncode != 0 and "rank 5 exited with code 17" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 300.0
