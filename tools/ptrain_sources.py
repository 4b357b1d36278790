"""Torch-side kernels (copies, fills, adds) of one eager passive training step, with the op that launched each, in launch order
among the libm2h kernels.     python tools/ptrain_sources.py > gpurun_out/ptrain_sources.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config  # noqa: E402

tr = PassiveTrainer(passive_config(BATCH_SIZE=64, TM=32, use_hip_graphs=False), torch.device("cuda", 0))
tr.setup()
batch = tr.feeders["train"].batch()
for _ in range(3):
    tr.train_batch(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.train_batch(*batch)
    torch.cuda.synchronize()
n = 0
for fe in sorted([f for f in prof.events() if getattr(f, "kernels", [])], key=lambda f: f.time_range.start):
    for k in fe.kernels:
        n += 1
        print("%4d  %-70s %7.1f us  <- %s" % (n, k.name[:70], k.duration, fe.name[:40]))
