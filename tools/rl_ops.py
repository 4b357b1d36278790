#!/usr/bin/env python3
"""Per-op times of the DD-PPO loop's three phases (tuning tool): every libm2h launch of one rollout step, one update_pol and one
update_sep (kernel by kernel, no graphs) bracketed by HIP events through ops.set_timing, with GEMM shapes.
usage: python tools/rl_ops.py [rollout|pol|sep]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import ops, synthetic  # noqa: E402
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "rollout"
dev = torch.device("cuda", 0)
cfg = near_target_config()
cfg.use_hip_graphs = False
tr = PPOTrainer(cfg, dev)
tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
tr.train_cycle()
for _ in range(20):
    tr._collect_rollout_step()
torch.cuda.synchronize()
sink = []
ops.set_timing(sink)
if which == "rollout":
    tr._collect_rollout_step()
elif which == "pol":
    tr._update_pol()
else:
    tr._update_sep()
ops.set_timing(None)
torch.cuda.synchronize()
tot = 0.0
agg = {}
for name, meta, e0, e1 in sink:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    shape = "M=%s N=%s K=%s" % (meta.get("M"), meta.get("N"), meta.get("K")) if meta and "M" in meta else ""
    key = (name, shape + "  " + (meta or {}).get("label", ""))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += us
for (name, shape), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-22s %-78s x%-4d %9.1f us  (%.1f each)" % (name, shape, n, us, us / n))
print("timed ops: %d, sum %.1f us" % (len(sink), tot))
