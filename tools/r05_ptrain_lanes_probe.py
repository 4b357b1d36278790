#!/usr/bin/env python3
"""Would passive pre-training gain from running the two networks' step chains as two independent lanes (net 1 of step k + 1 beside net 2 of
step k)?  Probe without any product change: TWO trainers, each replaying its whole step as ONE chain (M2H_PARALLEL_BRANCHES=0), alternately
on two HIP streams, against one trainer alone.  usage: M2H_PARALLEL_BRANCHES=0 python tools/r05_ptrain_lanes_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import torch  # noqa: E402

from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    trs = []
    for seed in (3, 4):
        tr = PassiveTrainer(passive_config(BATCH_SIZE=64, TM=32, SEED=seed), dev)
        tr.setup()
        trs.append(tr)
    batch = trs[0].feeders["train"].batch()
    for tr in trs:
        for _ in range(3):
            tr.train_batch(*batch)
    torch.cuda.synchronize()
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    N = 20

    def timed(fn):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        return best / N

    one = timed(lambda: [trs[0].train_batch(*batch) for _ in range(N)])

    def serial():
        for _ in range(N):
            trs[0].train_batch(*batch)
            trs[1].train_batch(*batch)

    def lanes():
        for _ in range(N):
            for i in (0, 1):
                with torch.cuda.stream(s[i]):
                    trs[i].train_batch(*batch)

    ser, lan = timed(serial), timed(lanes)
    print("parallel branches inside a graph: %s" % os.environ.get("M2H_PARALLEL_BRANCHES", "1"))
    print("one trainer:                    %.0f us per step" % (one * 1e6))
    print("two trainers, one stream:       %.0f us per step pair" % (ser * 1e6))
    print("two trainers, two streams:      %.0f us per step pair" % (lan * 1e6))


if __name__ == "__main__":
    main()
