"""Which python line launches each kernel of the RL loop: one eager rollout step, one update_pol epoch and one update_sep epoch under
torch.profiler with stacks; prints, in launch order, kernel name + the innermost m2h / trainer frame.  usage (GPU box):
    python tools/kernel_sources.py > gpurun_out/kernel_sources.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from m2h import synthetic  # noqa: E402
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config  # noqa: E402

dev = torch.device("cuda", 0)
tr = PPOTrainer(near_target_config(use_hip_graphs=False), dev)
tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
tr.train_cycle()
torch.cuda.synchronize()


def trace(title, fn):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        fn()
        torch.cuda.synchronize()
    print("== %s" % title)
    fa = prof.events()
    for fe in sorted([f for f in fa if getattr(f, "kernels", [])], key=lambda f: f.time_range.start):
        st = [s for s in (fe.stack or []) if "m2h" in s or "ppo" in s]
        for k in fe.kernels:
            print("  %-60s %6.1f us  <- %s | %s" % (k.name[:60], k.duration, fe.name[:30], st[0][-110:] if st else "-"))


tr._next_cache = tr._next_cache  # (keep the cache: the steady-state step)
trace("rollout step (eager)", lambda: tr._collect_rollout_step())


def graph_variant():
    from m2h import ops
    ro, rs = tr.rollouts_pol, tr.rollouts_sep
    cache = tuple(t.clone() for t in tr._next_cache)
    idx = torch.tensor([ro.step, ro.step + 1, rs.step + 1], dtype=torch.int64, device=dev)
    with torch.no_grad():
        tr._rollout_step_device(cache, idx, False, False)
        ops.step_index_advance(idx, ro.num_steps, rs.num_steps)
    ro.advance()
    rs.advance()
    tr.envs.t += 1
    tr._episode_step_host += 1
    tr._next_cache = cache


trace("rollout step (the captured variant, run eagerly)", graph_variant)
for _ in range(18):
    tr._collect_rollout_step()
trace("update_pol (eager: get_value + returns + 4 epochs)", lambda: tr._update_pol())
trace("update_sep (4 epochs)", lambda: tr._update_sep())
