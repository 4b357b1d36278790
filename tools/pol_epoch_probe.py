"""Where an update_pol epoch's time goes (no profiler): HIP events around the epoch's graph replay and around the optimizer step that follows
it, over the 24 epochs of a DD-PPO cycle.   python tools/pol_epoch_probe.py"""
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
tr = PPOTrainer(near_target_config(sep_update_math="bf16x3"), dev)
tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
tr.train_cycle()
tr.train_cycle()
torch.cuda.synchronize()
marks = []
host = []


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


class Wrap:
    def __init__(self, g):
        self.g = g

    def replay(self):
        a = ev()
        t0 = time.perf_counter()
        self.g.replay()
        host.append(time.perf_counter() - t0)
        marks.append(("graph", a, ev()))


gs = tr.agent._pol_graph
gs.graph = Wrap(gs.graph)
orig = tr.agent._reduce_and_step


def step(group, opt, last):
    a = ev()
    orig(group, opt, last)
    marks.append(("step:" + group, a, ev()))


tr.agent._reduce_and_step = step
pe = []
tr.train_cycle(phase_events=pe)
torch.cuda.synchronize()
tot = {}
for name, a, b in marks:
    tot.setdefault(name, []).append(a.elapsed_time(b) * 1e3)
for k, v in tot.items():
    print("%-10s n=%3d  mean %8.1f us  sum %8.2f ms" % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
ph = {}
for n, a, b in pe:
    ph[n] = ph.get(n, 0.0) + a.elapsed_time(b)
print({k: round(v, 2) for k, v in ph.items()})
# gaps between consecutive marks inside update_pol: end of one -> start of next
g = [marks[i][2].elapsed_time(marks[i + 1][1]) * 1e3 for i in range(len(marks) - 1) if marks[i][0] == "graph" and marks[i + 1][0] == "step:pol"]
g2 = [marks[i][2].elapsed_time(marks[i + 1][1]) * 1e3 for i in range(len(marks) - 1) if marks[i][0] == "step:pol" and marks[i + 1][0] == "graph"]
print("gap graph -> step: mean %.1f us; gap step -> next graph (pack refresh + drained-stream sync + launch): mean %.1f us (n=%d)" % (sum(g) / max(1, len(g)), sum(g2) / max(1, len(g2)), len(g2)))
print("host time of the graph launch call: mean %.1f us" % (1e6 * sum(host) / len(host)))
