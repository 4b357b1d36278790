"""Per-launch table of one update_pol epoch (280 samples), kernel by kernel with HIP events around every libm2h conv-engine call
(the eager path: ops timing switches the graphs off): name, label of the kernel the dispatch took, M x N x K, microseconds."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "move2hear-active-av-separation_amd"))
import numpy as np
import torch
from m2h import ops, synthetic as syn
from m2h import functional as MF
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config

dev = torch.device("cuda", 0)
MF.carry_tuning(True)
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else []):
    k, v = kv.split("=")
    ops.debug_set(int(k), int(v))
tr = PPOTrainer(near_target_config(), dev, world_rank=0, world_size=1)
tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_state_dict(syn.policy_shapes(), 1).items()})
tr.train_cycle(); tr.train_cycle()
torch.cuda.synchronize()
ag = tr.agent
ag.ppo_epoch = 1
sink = []
ops.set_timing(sink)
ag.update_pol(tr.rollouts_pol)
torch.cuda.synchronize()
ops.set_timing(None)
tot = 0.0
for name, meta, e0, e1 in sink:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    m = meta or {}
    fl = m.get("flops")
    print("%-22s %-52s M %6s N %5s K %6s  %7.1f us  %s" % (name, (m.get("label") or "")[:52], m.get("M", ""), m.get("N", ""), m.get("K", ""), us,
                                                        ("%5.1f TFLOP/s" % (fl / us / 1e6)) if fl else ""))
print("timed launches: %d, sum %.1f us" % (len(sink), tot))
