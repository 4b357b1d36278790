"""Distribution heads and small helpers: drop-in for the model-facing part of audio_separation/common/utils.py.

``CategoricalNet`` / ``CustomFixedCategorical`` (:16-50) keep their names and methods (sample, log_probs, mode, get_probs,
get_log_probs, entropy).  logits, log-softmax, softmax and entropy come from one wave-per-row HIP kernel; sampling is
``torch.multinomial(probs, 1, True)`` on the probabilities' device generator -- exactly what the reference's
``Categorical.sample`` does -- so equal probabilities and seed give equal actions.
"""
import torch
import torch.nn as nn

from .. import ops


class CustomFixedCategorical:
    """Holds the kernel's outputs; mirrors the methods of the reference class (common/utils.py:16-39)."""

    def __init__(self, logp_all, probs, entropy):
        self._logp_all = logp_all
        self.probs = probs
        self._entropy = entropy

    def sample(self, sample_shape=torch.Size()):
        """[M,1] int64, the draw of ``Categorical.sample().unsqueeze(-1)`` == ``torch.multinomial(probs, 1, True)``.
        For one sample per row torch.multinomial IS ``argmax(probs / Exp(1))`` with the exponential noise drawn from the tensor's
        device generator (ATen multinomial_out: q = empty_like(probs).exponential_(1); q = probs / q; argmax(q, -1, keepdim)),
        preceded by four validity checks of ``probs`` (eight tiny launches and two device asserts per call).  The same three
        ops are issued here directly: same generator state in, same noise, same actions out
        (tests/test_gpu_rl.py::test_sampling_is_torch_multinomial_bit_for_bit); the probabilities come from this build's own
        softmax kernel, finite and normalised by construction."""
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
