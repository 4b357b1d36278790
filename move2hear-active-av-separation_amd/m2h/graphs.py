"""HIP-graph replay of the eval-mode separator pair (MI355X: launch-bound inner loops are captured once and replayed).

``GraphedSeparatorPair(policy, observations)`` captures ``policy.get_binSepMasks(observations)`` followed by
``policy.convert_bin2mono(masks, mixed_audio=...)`` (pretrain/passive/policy.py:61-71, rl/ppo/policy.py:183-193) -- the 22
kernels of the two U-Nets -- for the given input tensors BY ADDRESS: new data is fed by copying into those tensors, results are
read from the two static outputs.  Same kernels, same values as the direct calls; what goes away is the per-kernel launch
gap.  The arithmetic mode (ops.set_math_mode) and the weights' packed buffers are fixed at capture: re-capture after changing
either (``stale()`` tells).
"""
import os

import torch

from . import _lib, ops


class capture:
    """torch.cuda.graph with capture_error_mode="thread_local": other threads of the process (RCCL's watchdog polling its
    events, a data feeder) may keep calling into HIP while this thread captures; work the autograd engine's thread enqueues
    on the capturing stream is recorded all the same.  The graph remembers how many libm2h kernels it holds (``replay`` below)."""

    def __init__(self, graph, pool=None):
        from . import functional
        functional.unit_grad(torch.device("cuda", torch.cuda.current_device()))   # the cached root gradient must not be born inside a capture
        self.graph = graph
        self.ctx = torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local")

    def __enter__(self):
        self.n0 = _lib.load().m2h_launch_count()
        return self.ctx.__enter__()

    def __exit__(self, *exc):
        import warnings
        # torch only WARNS when a capture recorded nothing ("The CUDA Graph is empty"): for a body that ran to its end that is how a mis-captured
        # graph first shows (work enqueued on another stream than the capturing one), so here it is an error.  A body that raised leaves an empty
        # graph by design (the exception is what the caller gets): that warning is dropped.
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            out = self.ctx.__exit__(*exc)
        empty = [w for w in caught if "Graph is empty" in str(w.message)]
        for w in caught:
            if w not in empty:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if empty and exc[0] is None:
            raise RuntimeError("m2h.graphs.capture: the capture recorded no work on the capturing stream (launches went to another stream, or nothing was launched)")
        n = _lib.load().m2h_launch_count() - self.n0
        self.graph._m2h_kernels = n
        _counts["captured"] += n
        return out


_counts = {"captured": 0, "replayed": 0}


def replay(graph):
    """graph.replay(), counted: the graph's libm2h kernels run once more (launch_total)."""
    graph.replay()
    _counts["replayed"] += getattr(graph, "_m2h_kernels", 0)


def launch_total():
    """libm2h kernels executed by this process so far: enqueued one by one + replayed from graphs (captures execute nothing).
    A diagnostic for bench.py's per-phase launch counts; torch's own kernels are not in it."""
    return _lib.load().m2h_launch_count() - _counts["captured"] + _counts["replayed"]


# ------------------------------------------------------------------------------------------------------------------
# Fork / join of independent kernel chains onto side HIP streams WHILE A HIP GRAPH IS BEING CAPTURED: they become parallel
# branches of the graph.  Same kernels, same values.  Used for the policy's three encoders at update batches (forward and,
# through autograd's stream bookkeeping, backward): their kernels each occupy a few of the chip's 256 CUs for 5-35 us.
#
# What was measured on this stack (ROCm 7.2, tools/graph_fork.py, tools/graph_fork_after.py, profiles/r04_graph_fork*.txt):
#   * a fork / join inside a replayed graph costs ~22 us (a 7-node side chain of 45 us beside a 10-node main chain returns 23 us;
#     a 1-node side chain costs 25 us) and launching a multi-branch graph is HOST-bound (83-128 us for 25 nodes on 2-4 branches);
#   * a multi-branch graph enqueued BEHIND other work parks one barrier packet per side queue until its turn comes, and while such
#     packets wait every kernel boundary of the running queue is dearer: a 53-node chain replays in 345 us alone, in 413 / 533 /
#     611 us while a 2 / 3 / 4-branch graph waits behind it; in the free-running DD-PPO cycle that turned -6 ms of update_pol into
#     +19 ms of rollout (GPU_MAX_HW_QUEUES=8: +110 ms).  Hence the rule in ppo.py: a graph with parallel branches is launched
#     onto a DRAINED stream (one host synchronize per epoch: 24 per cycle), never queued behind the rollout's replays;
#   * at the rollout width (14 rows) the fork / join costs more than the 55 us side chain returns, and every step's graph would
#     need the drained stream: the rollout step stays one chain.
# With the rule: update_pol 54.6 -> 46.5 ms per cycle, rollout unchanged, 10 800 -> 11 450 env-steps/s.
# M2H_PARALLEL_BRANCHES=0 switches the fork off (A/B runs).
# ------------------------------------------------------------------------------------------------------------------
_side_streams = {}
parallel_branches = os.environ.get("M2H_PARALLEL_BRANCHES", "1") != "0"     # module switch (tests / A-B measurements)


def side_stream(device, index=0):
    """A side HIP stream of `device` for a hand-made fork inside a graph capture (kept per device: graphs captured later re-use it)."""
    key = (torch.device(device).index, "side", index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device)
    return _side_streams[key]


def _tensors(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (tuple, list)):
        for y in x:
            yield from _tensors(y)


def run_parallel(device, fns):
    """Runs fns[0] on the current stream and fns[1:] on side streams that start after the work enqueued so far; the current
    stream waits for all of them before this returns.  Tensors produced on a side stream are handed to the current stream
    (allocator bookkeeping: ``record_stream``).  device None, or no graph capture in progress (eager launches are host-bound: a
    second stream buys nothing there): plain sequential calls."""
    if device is None or not parallel_branches or len(fns) == 1 or ops.timing_enabled() or not torch.cuda.is_current_stream_capturing():
        return [f() for f in fns]
    main = torch.cuda.current_stream(device)
    key = (device.index, len(fns) - 1)
    if key not in _side_streams:
        _side_streams[key] = [torch.cuda.Stream(device) for _ in range(len(fns) - 1)]
    fork = torch.cuda.Event()
    fork.record(main)
    outs, joins = [None] * len(fns), []
    for i, s in enumerate(_side_streams[key], 1):
        s.wait_event(fork)
        with torch.cuda.stream(s):
            outs[i] = fns[i]()
            ev = torch.cuda.Event()
            ev.record(s)
        joins.append(ev)
    outs[0] = fns[0]()
    for ev in joins:
        main.wait_event(ev)
    for o in outs[1:]:
        for t in _tensors(o):
            t.record_stream(main)
    return outs


class GraphedSeparatorPair:
    def __init__(self, policy, observations):
        self.policy = policy
        self.observations = observations
        self._graph = None
        self._sig = None
        self.masks = None
        self.mono = None

    def _signature(self):
        sep = [self.policy.binSep_enc, self.policy.binSep_dec, self.policy.bin2mono_enc, self.policy.bin2mono_dec]
        return (ops.math_mode(), tuple((p.data_ptr(), p._version) for m in sep for p in list(m.parameters()) + list(m.buffers())),
                tuple((k, v.data_ptr(), tuple(v.shape)) for k, v in sorted(self.observations.items())))

    def stale(self):
        return self._graph is None or self._sig != self._signature()

    def _run(self):
        obs = self.observations
        masks = self.policy.get_binSepMasks(obs)
        return masks, self.policy.convert_bin2mono(masks, mixed_audio=obs["mixed_bin_audio_mag"])

    def capture(self):
        with torch.no_grad():
            self._run()  # warm-up outside the capture: packs the weights, loads the kernels
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with capture(g):
                self.masks, self.mono = self._run()
        self._graph, self._sig = g, self._signature()

    def __call__(self):
        if self.stale():
            self.capture()
        replay(self._graph)
        return self.masks, self.mono
