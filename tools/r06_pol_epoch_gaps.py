"""Where an update_pol epoch's wall time goes: HIP events right before / behind every replay of the epoch's graph (the GPU time of the graph
itself) against the distance between consecutive epochs (graph + optimizer step + packs + host synchronisation + graph launch)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "move2hear-active-av-separation_amd"))
import numpy as np
import torch
from m2h import graphs, ops, synthetic as syn
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config

dev = torch.device("cuda", 0)
tr = PPOTrainer(near_target_config(), dev, world_rank=0, world_size=1)
tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_state_dict(syn.policy_shapes(), 1).items()})
tr.train_cycle(); tr.train_cycle()
torch.cuda.synchronize()
rec = []
orig = graphs.replay


def replay(g):
    n = getattr(g, "_m2h_kernels", 0)
    if 100 <= n <= 220:     # the policy epoch's graph (~150 libm2h kernels; the rollout step has ~50, the separator epoch fewer)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        orig(g)
        e1.record()
        rec.append((n, e0, e1, t0, time.perf_counter()))
    else:
        orig(g)


graphs.replay = replay
import m2h.rl.ppo.ppo as P
P.graphs.replay = replay
for _ in range(2):
    tr.train_cycle()
torch.cuda.synchronize()
g_us = [e0.elapsed_time(e1) * 1e3 for _n, e0, e1, _a, _b in rec]
host_us = [(b - a) * 1e6 for _n, _e0, _e1, a, b in rec]
ep = [rec[i][1].elapsed_time(rec[i + 1][1]) * 1e3 for i in range(len(rec) - 1)]
ep4 = [x for i, x in enumerate(ep) if (i % 4) != 3]     # distances inside one update (4 epochs); every 4th spans other phases
print("epoch graphs replayed: %d (kernels per graph: %s)" % (len(rec), sorted({r[0] for r in rec})))
print("GPU time of the graph, start event -> end event: median %.1f us (min %.1f, max %.1f)" % (np.median(g_us), min(g_us), max(g_us)))
print("host time of the hipGraphLaunch call:            median %.1f us" % np.median(host_us))
print("start of one epoch's graph -> start of the next (same update): median %.1f us" % np.median(ep4))
print("=> outside the graph per epoch (optimizer step, packs, host synchronisation, launch): %.1f us" % (np.median(ep4) - np.median(g_us)))
