"""Structure and replay time of the rollout step's HIP graph: DOT dump (is it one linear chain?), node count, replay time
unprofiled, and the same kernels re-timed after removing nothing -- the reference point for fusion work.
    python tools/graph_dump.py"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

made = []
_Base = torch.cuda.CUDAGraph


class DebugGraph(_Base):
    def __new__(cls, *a, **k):
        g = super().__new__(cls, *a, **k)
        return g

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.enable_debug_mode()
        made.append(self)


torch.cuda.CUDAGraph = DebugGraph
from m2h import synthetic  # noqa: E402
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config  # noqa: E402

dev = torch.device("cuda", 0)
tr = PPOTrainer(near_target_config(), dev)
tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
tr.train_cycle()
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for i, g in enumerate(made):
    path = os.path.join(ROOT, "gpurun_out", "graph_%d.dot" % i)
    try:
        g.debug_dump(path)
    except Exception as e:  # noqa: BLE001
        print("dump %d failed: %s" % (i, e))
        continue
    txt = open(path).read()
    nodes = len(re.findall(r"^\s*\"?[\w]+\"?\s*\[", txt, re.M))
    edges = re.findall(r"(\S+)\s*->\s*(\S+)", txt)
    outdeg, indeg = {}, {}
    for a, b in edges:
        outdeg[a] = outdeg.get(a, 0) + 1
        indeg[b] = indeg.get(b, 0) + 1
    print("graph %d: %d bytes of DOT, ~%d node lines, %d edges, max out-degree %d, max in-degree %d" %
          (i, len(txt), nodes, len(edges), max(outdeg.values() or [0]), max(indeg.values() or [0])))
# replay timing of the steady-state rollout step
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record()
    for _s in range(20):
        tr._collect_rollout_step()
    e1.record()
    torch.cuda.synchronize()
    print("20 rollout steps: %.1f us per step" % (1e3 * e0.elapsed_time(e1) / 20))
    tr._update_pol()
