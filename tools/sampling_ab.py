#!/usr/bin/env python3
"""Rollout time per cycle under the two action-sampling modes (tuning tool): "device" (torch's Philox exponential + div + argmax: the
default) and "cpu_generator" (Exp(1) noise from the CPU default generator through a pinned ring + one sampling kernel: bit-exact
against the reference PyTorch-CPU run).  usage: python tools/sampling_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import synthetic  # noqa: E402
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config  # noqa: E402

dev = torch.device("cuda", 0)
for mode in ("device", "cpu_generator", "device", "cpu_generator"):
    cfg = near_target_config()
    cfg.action_sampling = mode
    tr = PPOTrainer(cfg, dev)
    tr.setup()
    tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
    tr.train_cycle()
    for _ in range(40):
        tr._collect_rollout_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(240):
        tr._collect_rollout_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 240
    print("%-14s %.1f us per rollout step (%.1f ms per 120-step cycle)" % (mode, dt * 1e6, dt * 120e3))
    del tr
