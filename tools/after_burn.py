"""Does the DD-PPO cycle run slower right after the headline's matrix-bound burn (power / clock state)?  Per-cycle times of 12 cycles after a
3 s burn of the separator pair at batch 256, against 12 cycles measured before it.   python tools/after_burn.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from m2h import ops, synthetic  # noqa: E402
from m2h.graphs import GraphedSeparatorPair  # noqa: E402
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config  # noqa: E402

dev = torch.device("cuda", 0)
tr = PPOTrainer(near_target_config(sep_update_math="bf16x3"), dev)
tr.setup()
tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
for _ in range(3):
    tr.train_cycle()


def cycles(n):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_cycle()
        torch.cuda.synchronize()
        out.append(round(1e3 * (time.perf_counter() - t0), 1))
    return out


print("before the burn (ms per cycle):", cycles(12))
pol, _sd = bench.make_policy(dev)
mix, tc = bench.make_inputs(dev, 256, 256, 1000)
ops.set_math_mode(ops.MATH_BF16X3)
pair = GraphedSeparatorPair(pol, {"mixed_bin_audio_mag": mix, "target_class": tc})
pair()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 3.0:
    for _ in range(20):
        pair()
    torch.cuda.synchronize()
    n += 20
print("burn: %d steps of the pair in %.2f s (%.3f ms per step)" % (n, time.perf_counter() - t0, 1e3 * (time.perf_counter() - t0) / n))
ops.set_math_mode(ops.MATH_FP32)
print("right after the burn:", cycles(12))
time.sleep(2.0)
print("after 2 s of idle:", cycles(6))
