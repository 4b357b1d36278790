#!/usr/bin/env python3
"""Per-op times of ONE passive training step (tuning tool): every libm2h launch of PassiveTrainer.train_batch (kernel by kernel,
no graph) bracketed by HIP events through ops.set_timing, with its GEMM shape and the label of the kernel the dispatch took.
usage: python tools/train_step_ops.py [--batch 64] [--tm 32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

from m2h import ops  # noqa: E402
from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tm", type=int, default=32)
    a = ap.parse_args()
    cfg = passive_config(BATCH_SIZE=a.batch, TM=a.tm, use_hip_graphs=False)
    tr = PassiveTrainer(cfg)
    tr.setup()
    tr.actor_critic.train()
    for _ in range(2):
        tr.train_batch(*tr.feeders["train"].batch())
    torch.cuda.synchronize()
    sink = []
    ops.set_timing(sink)
    tr.train_batch(*tr.feeders["train"].batch())
    ops.set_timing(None)
    torch.cuda.synchronize()
    tot = 0.0
    for name, meta, e0, e1 in sink:
        us = e0.elapsed_time(e1) * 1e3
        tot += us
        shape = "M=%s N=%s K=%s" % (meta.get("M"), meta.get("N"), meta.get("K")) if meta and "M" in meta else ""
        print("%-28s %-34s %8.1f us  %s" % (name, shape, us, (meta or {}).get("kernel", "")))
    print("timed ops: %d, sum %.1f us" % (len(sink), tot))


if __name__ == "__main__":
    main()
