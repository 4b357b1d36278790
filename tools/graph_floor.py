"""Per-node cost of a replayed HIP graph of dependent trivial kernels (what bounds the RL loop's launch chains).
    python tools/graph_floor.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

from m2h import graphs, ops  # noqa: E402

dev = torch.device("cuda", 0)
N = 400


def timeit(name, build, reps=20):
    g = torch.cuda.CUDAGraph()
    build()  # warm
    torch.cuda.synchronize()
    with graphs.capture(g):
        build()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%-60s %6.2f us per node (graph of %d nodes)" % (name, 1e3 * e0.elapsed_time(e1) / reps / N, N))
    # eager, from python (host-bound?)
    t0 = time.perf_counter()
    e0.record()
    build()
    e1.record()
    torch.cuda.synchronize()
    print("%-60s %6.2f us per node eager (device span), host %.2f us" % ("", 1e3 * e0.elapsed_time(e1) / N, 1e6 * (time.perf_counter() - t0) / N))


x = torch.zeros(64, device=dev)
timeit("torch add_ on 64 floats (1 block)", lambda: [x.add_(1.0) for _ in range(N)])
y = torch.zeros(256 * 256 * 4, device=dev)
timeit("torch add_ on 256K floats (256 blocks)", lambda: [y.add_(1.0) for _ in range(N)])
z = torch.zeros(64 * 1024 * 1024, device=dev)
idx = torch.zeros(3, dtype=torch.int64, device=dev)
timeit("m2h step_index_advance (1 block of 64)", lambda: [ops.step_index_advance(idx, 20, 120) for _ in range(N)])
a = torch.zeros(14, 512, device=dev)
w = torch.zeros(3, 512, device=dev)
b = torch.zeros(3, device=dev)
timeit("m2h policy_heads (14 rows)", lambda: [ops.policy_heads(a, w, b, w[:1].contiguous(), b[:1].contiguous()) for _ in range(N)])
# a chain with a LARGE kernel's worth of dirty data between trivial ones
big = torch.zeros(8 * 1024 * 1024, device=dev)
timeit("torch add_ on 8M floats (32 MB written per node)", lambda: [big.add_(1.0) for _ in range(N)])

# distinct kernels in a cycle: is the floor a cold-start (instruction fetch) cost?
xs = [torch.zeros(64, device=dev) for _ in range(8)]
xi = torch.zeros(64, device=dev, dtype=torch.int64)
xh = torch.zeros(64, device=dev, dtype=torch.float16)
xd = torch.zeros(64, device=dev, dtype=torch.float64)
fns = [lambda: xs[0].add_(1.0), lambda: xs[1].mul_(1.5), lambda: xs[2].exp_(), lambda: xs[3].neg_(), lambda: xs[4].abs_(), lambda: xs[5].sigmoid_(),
       lambda: xs[6].tanh_(), lambda: xs[7].fill_(2.0), lambda: xi.add_(1), lambda: xi.mul_(3), lambda: xh.add_(1.0), lambda: xh.mul_(2.0),
       lambda: xd.add_(1.0), lambda: xd.exp_(), lambda: xs[0].clamp_(0, 1), lambda: xs[1].sqrt_(), lambda: xs[2].log1p_(), lambda: xs[3].relu_(),
       lambda: xs[4].floor_(), lambda: xs[5].cos_(), lambda: xs[6].sin_(), lambda: xi.fill_(3), lambda: xh.neg_(), lambda: xd.neg_(),
       lambda: torch.cumsum(xs[0], 0, out=xs[1]), lambda: torch.argmax(xs[2], 0), lambda: xs[3].div_(2.0), lambda: xs[4].sub_(1.0),
       lambda: xs[5].pow_(2.0), lambda: xs[6].reciprocal_(), lambda: xs[7].sign_(), lambda: xs[0].erf_()]
timeit("32 distinct small torch kernels in a cycle", lambda: [fns[i % len(fns)]() for i in range(N)])
thr = torch.zeros(48 * 1024 * 1024, device=dev)   # 192 MB: evicts L2 (and its code lines) when touched


def cyc():
    for i in range(N):
        if i % 8 == 7:
            thr.add_(1.0)
        else:
            fns[i % len(fns)]()


g = torch.cuda.CUDAGraph()
cyc()
torch.cuda.synchronize()
with graphs.capture(g):
    cyc()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record()
torch.cuda.synchronize()
tot = 1e3 * e0.elapsed_time(e1) / 10
# cost of the 50 big kernels alone
g2 = torch.cuda.CUDAGraph()
with graphs.capture(g2):
    for _ in range(N // 8):
        thr.add_(1.0)
g2.replay()
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    g2.replay()
e1.record()
torch.cuda.synchronize()
big_t = 1e3 * e0.elapsed_time(e1) / 10
print("cycle of distinct small kernels with a 192 MB read-modify-write every 8th node: %.2f us per small node (total %.0f us, the %d big nodes alone %.0f us)"
      % ((tot - big_t) / (N - N // 8), tot, N // 8, big_t))
