"""GPU box: what a cross-stream event hop costs (fork / join between HIP streams), with torch streams and trivial kernels"""
import time
import torch

x = torch.zeros(1 << 20, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N = 2000


def run(name, fn):
    fn(50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(N)
    torch.cuda.synchronize()
    print(f"{name:50s} {(time.perf_counter() - t0) / N * 1e6:8.2f} us per round", flush=True)


def one_stream(n):
    with torch.cuda.stream(sa):
        for _ in range(n):
            x.add_(1.0)
            x.add_(1.0)


def ping_pong(n):
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    for _ in range(n):
        with torch.cuda.stream(sa):
            x.add_(1.0)
            ea.record(sa)
        sb.wait_event(ea)
        with torch.cuda.stream(sb):
            x.add_(1.0)
            eb.record(sb)
        sa.wait_event(eb)


def fork_join(n):
    sc = torch.cuda.Stream()
    ef, e1, e2 = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
    y = torch.zeros(1 << 20, device="cuda")
    for _ in range(n):
        ef.record(sc)
        sa.wait_event(ef)
        sb.wait_event(ef)
        with torch.cuda.stream(sa):
            x.add_(1.0)
            e1.record(sa)
        with torch.cuda.stream(sb):
            y.add_(1.0)
            e2.record(sb)
        sc.wait_event(e1)
        sc.wait_event(e2)


run("2 kernels per round, one stream", one_stream)
run("2 kernels per round, ping-pong over two streams", ping_pong)
run("fork to two streams, 1 kernel each, join", fork_join)
