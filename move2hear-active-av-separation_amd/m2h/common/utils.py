"""Distribution heads and small helpers: drop-in for the model-facing part of audio_separation/common/utils.py.

``CategoricalNet`` / ``CustomFixedCategorical`` (:16-50) keep their names and methods (sample, log_probs, mode, get_probs,
get_log_probs, entropy).  logits, log-softmax, softmax and entropy come from one wave-per-row HIP kernel.  Sampling is the
single-draw path of ``torch.multinomial(probs, 1, True)`` -- what the reference's ``Categorical.sample`` runs -- in one of three
modes (``Policy.set_action_sampling``):
  * ``"fused"`` (the trainers' default since round 5): the Exp(1) noise is made inside the heads kernel by Philox4x32-10 (the
    generator family torch's device generator uses) keyed by the config's seed, with its counter on the device: the draw costs no
    launch of its own (inside a HIP graph torch's generator costs three: two state fills and the exponential kernel);
  * ``"device"``: the Exp(1) noise comes from torch's device generator (Philox), as the reference does when its policy lives on a GPU;
  * ``"cpu_generator"``: the noise comes from the CPU default generator (mt19937) at the same stream position as on the
    reference's CPU path, crosses to the device through a pinned ring, and ``m2h_sample_actions`` takes the argmax: same seed,
    same actions as the reference PyTorch-CPU run (north-star contract; tests/test_gpu_trainer_golden.py samples this way).
"""
import torch
import torch.nn as nn

from .. import ops


class HostNoise:
    """Exp(1) noise of torch.multinomial's single-draw path, drawn on the CPU default generator and shipped to the device.

    ``torch.empty(M, A).exponential_(1)`` is the call ATen's multinomial makes on a CPU ``probs`` tensor (``empty_like(probs)
    .exponential_(1)``); issuing it here, once per ``sample()``, keeps the process-wide mt19937 stream aligned with a reference
    run that interleaves these draws with ``torch.randperm`` (rollout_storage.py:197, :406).  The device copy is ONE static
    buffer per row count -- a captured HIP graph reads it by address; ``stage()`` refills it before every replay.  Host slots
    form a ring of pinned buffers, each guarded by the event of the copy that last read it, so the host never waits for the
    device unless it runs a full ring ahead."""
    RING = 16

    def __init__(self, device):
        self.device = device
        self._bufs = {}

    def _state(self, M, A):
        st = self._bufs.get((M, A))
        if st is None:
            st = {"host": torch.empty(self.RING, M, A, dtype=torch.float32).pin_memory(), "events": [None] * self.RING, "next": 0,
                  "dev": torch.empty(M, A, dtype=torch.float32, device=self.device)}
            self._bufs[(M, A)] = st
        return st

    def buffer(self, M, A):
        """The static device buffer [M, A] (what a captured graph reads)."""
        return self._state(M, A)["dev"]

    def stage(self, M, A):
        """Draws the next [M, A] block of noise on the CPU default generator and enqueues its copy on the current stream."""
        st = self._state(M, A)
        i = st["next"]
        st["next"] = (i + 1) % self.RING
        if st["events"][i] is not None:
            st["events"][i].synchronize()
        slot = st["host"][i]
        slot.exponential_(1)
        st["dev"].copy_(slot, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["events"][i] = ev
        return st["dev"]


class CustomFixedCategorical:
    """Holds the kernel's outputs; mirrors the methods of the reference class (common/utils.py:16-39)."""

    def __init__(self, logp_all, probs, entropy, host_noise=None):
        self._logp_all = logp_all
        self.probs = probs
        self._entropy = entropy
        self._host_noise = host_noise

    def sample(self, sample_shape=torch.Size()):
        """[M,1] int64, the draw of ``Categorical.sample().unsqueeze(-1)`` == ``torch.multinomial(probs, 1, True)``.
        For one sample per row torch.multinomial IS ``argmax(probs / Exp(1))`` with the exponential noise drawn from the tensor's
        generator (ATen multinomial_out: q = empty_like(probs).exponential_(1); q = probs / q; argmax(q, -1, keepdim)),
        preceded by four validity checks of ``probs`` (eight tiny launches and two device asserts per call on a GPU); the
        probabilities here come from this build's own softmax kernel, finite and normalised by construction.
        device mode: the same three ops on the device generator: same generator state in, same noise, same actions out
        (tests/test_gpu_rl.py::test_sampling_is_torch_multinomial_bit_for_bit).
        cpu_generator mode: the noise is drawn on the CPU default generator (HostNoise) and m2h_sample_actions divides and takes
        the argmax.  While a HIP graph is being captured nothing is drawn: the graph reads the static noise buffer, which its
        owner refills (``HostNoise.stage``) before each replay."""
        if self._host_noise is not None:
            M, A = self.probs.shape
            if torch.cuda.is_current_stream_capturing():
                noise = self._host_noise.buffer(M, A)
            else:
                noise = self._host_noise.stage(M, A)
            return ops.sample_actions(self.probs, noise)
        q = torch.empty_like(self.probs).exponential_(1)
        torch.div(self.probs, q, out=q)
        return torch.argmax(q, dim=-1, keepdim=True)

    def log_probs(self, actions):
        return ops.gather_logp(self._logp_all, actions.reshape(-1, 1).contiguous())

    def mode(self):
        return self.probs.argmax(dim=-1, keepdim=True)

    def get_probs(self):
        return self.probs

    def get_log_probs(self):
        return torch.log(self.probs + 1e-7)

    def entropy(self):
        return self._entropy


class CategoricalNet(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.linear = nn.Linear(num_inputs, num_outputs)
        nn.init.orthogonal_(self.linear.weight, gain=0.01)
        nn.init.constant_(self.linear.bias, 0)

    def forward(self, x):
        w, b = self.linear.weight.detach(), self.linear.bias.detach()
        _, logp_all, probs, ent, _ = ops.policy_heads(x.contiguous(), w, b, w[:1].contiguous(), b[:1].contiguous())
        return CustomFixedCategorical(logp_all, probs, ent)


def linear_decay(epoch: int, total_num_updates: int) -> float:
    """Multiplicative factor for linear value decay (common/utils.py:53-63)."""
    return 1 - (epoch / float(total_num_updates))


def batch_obs(observations, device=None):
    """List of per-env observation dicts -> dict of batched float tensors (common/utils.py:75-97)."""
    from collections import defaultdict
    import numpy as np
    batch = defaultdict(list)
    for obs in observations:
        for sensor in obs:
            v = obs[sensor]
            if torch.is_tensor(v):
                t = v.to(device=device, dtype=torch.float)
            elif isinstance(v, np.ndarray):
                t = torch.from_numpy(v).to(device=device, dtype=torch.float)
            else:
                t = torch.tensor(v, dtype=torch.float, device=device)
            batch[sensor].append(t)
    for sensor in batch:
        batch[sensor] = torch.stack(batch[sensor], dim=0)
    return batch
