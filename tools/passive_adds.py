#!/usr/bin/env python3
"""Tuning tool: where do the aten::add launches of a passive pre-training step come from?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config
cfg = passive_config(BATCH_SIZE=64, TM=32, SEED=3, use_hip_graphs=False)
tr = PassiveTrainer(cfg, torch.device("cuda", 0)); tr.setup()
batch = tr.feeders["train"].batch()
for _ in range(2): tr.train_batch(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.train_batch(*batch)
torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.name in ("aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::contiguous", "aten::clone", "aten::fill_", "aten::zero_"):
        st = [s for s in (ev.stack or []) if "m2h" in s or "torch/autograd" in s or "optim" in s][:3]
        c[(ev.name, str(ev.input_shapes)[:60], tuple(st))] += 1
for k, v in c.most_common(40):
    print(v, k)
