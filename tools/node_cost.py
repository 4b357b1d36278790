"""Marginal cost of a trivial node inside the REAL replayed rollout-step graph: the step as it is, with K extra fill kernels at
its end, and with K extra fills spread behind its slice kernels (between heavy layers).
    python tools/node_cost.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import ops, synthetic  # noqa: E402
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config  # noqa: E402

dev = torch.device("cuda", 0)
scratch = torch.zeros(64, device=dev)
PAD = {"end": 0, "mid": 0}
_stats, _slice = ops.rollout_step_stats, ops.slice_concat_input


def stats(*a, **k):
    out = _stats(*a, **k)
    for _ in range(PAD["end"]):
        scratch.add_(1.0)
    return out


def slc(*a, **k):
    out = _slice(*a, **k)
    for _ in range(PAD["mid"]):
        scratch.add_(1.0)
    return out


ops.rollout_step_stats, ops.slice_concat_input = stats, slc


def measure(tag):
    tr = PPOTrainer(near_target_config(), dev)
    tr.setup()
    tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
    for _ in range(40):
        tr._collect_rollout_step()
    tr._update_pol()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record()
        for _s in range(20):
            tr._collect_rollout_step()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / 20)
        tr._update_pol()
    print("%-40s %.1f us per step" % (tag, best))
    return best


base = measure("as is")
PAD["end"] = 20
t = measure("+20 trivial nodes at the end")
print("   -> %.2f us per extra node" % ((t - base) / 20))
PAD["end"] = 0
PAD["mid"] = 5
t = measure("+5 trivial nodes behind each slice kernel")
print("   (slice_concat_input is called several times per step: read the per-node cost off the total)")
