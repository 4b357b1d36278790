import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np, torch
from m2h import synthetic
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
dev = torch.device("cuda", 0)
tr = PPOTrainer(near_target_config(), dev); tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
for _ in range(20): tr._collect_rollout_step()
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(20): tr._collect_rollout_step()
t1=time.perf_counter()-t; torch.cuda.synchronize(); t2=time.perf_counter()-t
print("20 steps: host-issue %.1f ms, incl. GPU drain %.1f ms" % (t1*1e3, t2*1e3))
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): tr._collect_rollout_step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
