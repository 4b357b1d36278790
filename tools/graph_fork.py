"""Does a fork / join inside a captured HIP graph buy concurrency for small-grid kernels, and what does the join cost?
A block = [pre] -> (main chain of NA nodes || side chain of NB nodes) -> [post]; 20 blocks per graph; serial capture vs forked capture.
    python tools/graph_fork.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

from m2h import graphs, ops  # noqa: E402

dev = torch.device("cuda", 0)
BLOCKS = 20
a = [torch.zeros(14, 512, device=dev) for _ in range(2)]
w = torch.zeros(3, 512, device=dev)
b = torch.zeros(3, device=dev)
w1, b1 = w[:1].contiguous(), b[:1].contiguous()
x = [torch.zeros(64, device=dev) for _ in range(2)]


def node(kind, i):
    if kind == "heads":
        ops.policy_heads(a[i], w, b, w1, b1)       # ~6.4 us, 4 blocks
    else:
        x[i].add_(1.0)                             # ~1.5 us


def build(kind, na, nb, fork, side):
    cur = torch.cuda.current_stream()
    for _ in range(BLOCKS):
        node(kind, 0)
        if fork:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(nb):
                    node(kind, 1)
        else:
            for _ in range(nb):
                node(kind, 1)
        for _ in range(na):
            node(kind, 0)
        if fork:
            cur.wait_stream(side)
        node(kind, 0)


def timeit(kind, na, nb, fork, reps=30):
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    build(kind, na, nb, False, side)
    torch.cuda.synchronize()
    with graphs.capture(g):
        build(kind, na, nb, fork, side)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps / BLOCKS


for kind in ("heads", "add"):
    for na, nb in ((10, 7), (10, 1), (4, 4), (1, 1)):
        s, f = timeit(kind, na, nb, False), timeit(kind, na, nb, True)
        print("%-6s main %2d || side %2d (+2): serial %7.2f us per block, forked %7.2f us (%+.2f)" % (kind, na, nb, s, f, f - s))
