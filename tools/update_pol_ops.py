#!/usr/bin/env python3
"""Per-op times of ONE update_pol epoch (tuning tool): every libm2h launch of PPO._pol_epoch at the 280-sample update batch, kernel by
kernel (no graph), bracketed by HIP events through ops.set_timing, with its GEMM shape and the kernel the dispatch took.
usage: python tools/update_pol_ops.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import ops, synthetic  # noqa: E402
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config  # noqa: E402

dev = torch.device("cuda", 0)
tr = PPOTrainer(near_target_config(use_hip_graphs=False), dev)
tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
for _ in range(20):
    tr._collect_rollout_step()
tr._update_pol()
for _ in range(20):
    tr._collect_rollout_step()
ag, ro = tr.agent, tr.rollouts_pol
adv = ag.get_advantages(ro)
sample = next(iter(ro.recurrent_generator(adv, 1)))
acc = torch.zeros(4, device=dev)
ag._pol_epoch(sample, 0.1, acc)
torch.cuda.synchronize()
sink = []
ops.set_timing(sink)
ag._pol_epoch(sample, 0.1, acc)
ops.set_timing(None)
torch.cuda.synchronize()
tot = 0.0
for name, meta, e0, e1 in sink:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    shape = "M=%s N=%s K=%s" % (meta.get("M"), meta.get("N"), meta.get("K")) if meta and "M" in meta else ""
    print("%-30s %-36s %8.1f us  %s" % (name, shape, us, (meta or {}).get("kernel", "")))
print("timed ops: %d, %.1f us" % (len(sink), tot))
