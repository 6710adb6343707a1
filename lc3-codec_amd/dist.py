"""Multi-GPU layer: streams are independent codec channels, so a job shards by stream range with NO
data-path collective.  One process per GPU; torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests) is used only for barriers and for the final
reduction of the report counters (frames done, max elapsed time, parity mismatches, PLC events)."""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total_streams, world, rank):
    """contiguous stream range [lo, hi) owned by `rank` (stereo pairs / consecutive ids stay together)"""
    lo = total_streams * rank // world
    hi = total_streams * (rank + 1) // world
    return lo, hi


def reduce_report(dist, device, elapsed_s, frames, mismatches=0, plc_events=0, force=False):
    """-> (max elapsed over ranks, total frames, total mismatches, total plc events); dist may be None (1 rank).
    force: run the two all_reduce calls even in a group of one rank (the RCCL bring-up check: library load, communicator, reduction
    on device tensors -- tests/test_gpu_parity.py::test_rccl_world_size_one, bench.py with LC3_BENCH_RCCL=1)"""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return float(elapsed_s), int(frames), int(mismatches), int(plc_events)
    import torch

    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([frames, mismatches, plc_events], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c[0].item()), int(c[1].item()), int(c[2].item())
