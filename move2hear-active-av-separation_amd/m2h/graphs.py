"""HIP-graph replay of the eval-mode separator pair (MI355X: launch-bound inner loops are captured once and replayed).

``GraphedSeparatorPair(policy, observations)`` captures ``policy.get_binSepMasks(observations)`` followed by
``policy.convert_bin2mono(masks, mixed_audio=...)`` (pretrain/passive/policy.py:61-71, rl/ppo/policy.py:183-193) -- the 22
kernels of the two U-Nets -- for the given input tensors BY ADDRESS: new data is fed by copying into those tensors, results are
read from the two static outputs.  Same kernels, same values as the direct calls; what goes away is the per-kernel launch
gap.  The arithmetic mode (ops.set_math_mode) and the weights' packed buffers are fixed at capture: re-capture after changing
either (``stale()`` tells).
"""
import torch

from . import ops


def capture(graph, pool=None):
    """torch.cuda.graph with capture_error_mode="thread_local": other threads of the process (RCCL's watchdog polling its
    events, a data feeder) may keep calling into HIP while this thread captures; work the autograd engine's thread enqueues
    on the capturing stream is recorded all the same."""
    return torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local")


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
        self._graph.replay()
        return self.masks, self.mono
