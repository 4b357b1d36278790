"""do the rollout step's weights (157 MB < the 256 MB Infinity Cache) stay resident between steps, and does it matter?
FETCH_SIZE cannot tell (the guide: Infinity-Cache hits are counted in it), so the probe is by time: the 14-environment separator pair (two
U-Net passes, 134 MB of fp32 weights, one HIP graph of 22 kernels) replayed back to back -- nothing else touches memory between two passes --
against the same graph with (a) a 160 MB read-modify-write between passes (what a step's other traffic and the storages' inserts amount to),
(b) a 700 MB one (what update_sep's 1680-sample kernels leave behind: nothing of the weights can survive).  HIP events around the pair only.
usage: gpurun -- python3 tools/r06_mall_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
from m2h import synthetic  # noqa: E402
from m2h.common.spaces import move2hear_observation_space  # noqa: E402
from m2h.graphs import GraphedSeparatorPair  # noqa: E402
from m2h.pretrain.passive.policy import Move2HearPassiveWoMemoryPolicy  # noqa: E402

dev = torch.device("cuda", 0)
pol = Move2HearPassiveWoMemoryPolicy(move2hear_observation_space())
pol.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), 1).items()})
pol = pol.to(dev).eval()
mixed, tc = synthetic.make_passive_inputs(14, 32, 3)
obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}
pair = GraphedSeparatorPair(pol, obs)
pair()
torch.cuda.synchronize()
def run(n, fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


N = 300
for label, mb in (("back to back", 0), ("160 MB touched between passes", 160), ("320 MB", 320), ("700 MB touched between passes", 700)):
    buf = torch.zeros(max(1, mb) * (1 << 20) // 4 // 2, device=dev)
    touch = (lambda: buf.add_(1.0)) if mb else (lambda: None)
    both, alone = [], []
    for rep in range(3):       # no host synchronisation inside a run: the device never idles (as in the rollout loop)
        alone.append(run(N, touch) if mb else 0.0)
        both.append(run(N, lambda: (touch(), pair())))
    print("%-34s [touch + pair] %.1f / %.1f / %.1f us, touch alone %.1f / %.1f / %.1f us -> pair %.1f us" % (
        (label,) + tuple(both) + tuple(alone) + (min(both) - min(alone),)))
