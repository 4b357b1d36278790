import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np, torch
from m2h import synthetic as syn
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
tr = PPOTrainer(near_target_config(sep_update_math="bf16x3"), dev); tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_state_dict(syn.policy_shapes(), 1).items()})
tr.train_cycle(); tr.train_cycle(); torch.cuda.synchronize()
for rep in range(3):
    tr._next_cache = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr._collect_rollout_step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    tr._collect_rollout_step()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for _ in range(10): tr._collect_rollout_step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("first step without the cached separator outputs (eager): %.0f us; next step (graph, alone): %.0f us; 10 more: %.0f us each" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e5))
