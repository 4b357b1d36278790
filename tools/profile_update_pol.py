import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np, torch
from m2h import synthetic
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
dev = torch.device("cuda", 0)
tr = PPOTrainer(near_target_config(), dev); tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
tr.train_cycle()
for _ in range(20): tr._collect_rollout_step()
torch.cuda.synchronize()
t=time.perf_counter(); tr._update_pol(); torch.cuda.synchronize(); print("update_pol wall", time.perf_counter()-t)
for _ in range(20): tr._collect_rollout_step()
pr = cProfile.Profile(); pr.enable(); tr._update_pol(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
