"""Does a single-branch graph replay slower once the process has replayed a graph with parallel branches?  Host launch time and device
time of a 53-node chain: before, after a forked graph was replayed, after that graph was destroyed.
    python tools/graph_fork_after.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

from m2h import graphs, ops  # noqa: E402

dev = torch.device("cuda", 0)
a = [torch.zeros(14, 512, device=dev) for _ in range(4)]
w = torch.zeros(3, 512, device=dev)
b = torch.zeros(3, device=dev)
w1, b1 = w[:1].contiguous(), b[:1].contiguous()


def heads(i):
    ops.policy_heads(a[i], w, b, w1, b1)


def chain():
    for _ in range(53):
        heads(0)


def capture(fn):
    g = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with graphs.capture(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    return g


def measure(g, label, reps=200):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-70s device %7.1f us per replay, host enqueue %7.1f us per replay" % (label, 1e3 * e0.elapsed_time(e1) / reps, 1e6 * (t1 - t0) / reps), flush=True)


g = capture(chain)
measure(g, "53-node chain, fresh process")
measure(g, "again")


def forked(nside):
    sides = [torch.cuda.Stream() for _ in range(nside)]

    def fn():
        cur = torch.cuda.current_stream()
        heads(0)
        for i, s in enumerate(sides):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                for _ in range(8):
                    heads(1 + i)
        for _ in range(8):
            heads(0)
        for s in sides:
            cur.wait_stream(s)
        heads(0)
    return fn


for nside in (1, 2, 3):
    gf = capture(forked(nside))
    measure(gf, "forked graph: main 8 || %d side chains of 8" % nside, reps=50)
    measure(g, "53-node chain after a %d-side forked graph was replayed" % nside)
    # interleaved, free-running: 1 forked replay then 20 chain replays
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    t0 = time.perf_counter()
    evs = []
    for _ in range(10):
        gf.replay()
        x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x0.record()
        for _ in range(20):
            g.replay()
        x1.record()
        evs.append((x0, x1))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-70s device %7.1f us per replay, host enqueue %7.1f us per (21 replays)/21" % ("  interleaved free-running (1 forked + 20 chains) x 10: chain",
          1e3 * sum(x.elapsed_time(y) for x, y in evs) / 200, 1e6 * (t1 - t0) / 210), flush=True)
    del gf
    torch.cuda.synchronize()
    measure(g, "53-node chain after that graph was destroyed")
