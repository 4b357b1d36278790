#!/usr/bin/env python3
"""Would the rollout phase gain from two INDEPENDENT half-width step chains on two HIP streams?  A rollout step at 14 environments is
a chain of ~50 small kernels bound by their dependent round trips, not by the chip.  Probe (no product code): one trainer with 14
environments against two trainers with 7 each, stepped alternately on two streams (each step one single-chain HIP graph replay).
usage: python tools/r05_two_lane_rollout_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import synthetic as syn  # noqa: E402
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config  # noqa: E402


def make(dev, n, seed):
    tr = PPOTrainer(near_target_config(NUM_PROCESSES=n, SEED=seed), dev)
    tr.setup()
    tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_state_dict(syn.policy_shapes(), 1).items()})
    return tr


def steps(tr, n):
    for _ in range(n):
        tr._collect_rollout_step()


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    T = 19   # steps per measurement: stays inside one rollout of 20 (no update in between)
    t14 = make(dev, 14, 0)
    a, b = make(dev, 7, 100), make(dev, 7, 200)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for tr in (t14, a, b):      # first steps eager + graph capture
        steps(tr, 3)
    torch.cuda.synchronize()

    def reset(tr):
        # back to the start of a rollout so that every measurement walks the same rows
        pass

    one = timed(lambda: steps(t14, T)) / T
    half = timed(lambda: steps(a, T)) / T

    def serial():
        for _ in range(T):
            a._collect_rollout_step()
            b._collect_rollout_step()

    def lanes():
        for _ in range(T):
            with torch.cuda.stream(s1):
                a._collect_rollout_step()
            with torch.cuda.stream(s2):
                b._collect_rollout_step()

    ser = timed(serial) / T
    # graphs captured on the default stream replay on whatever stream is current: first lane steps re-use them
    lan = timed(lanes) / T
    print("one trainer, 14 envs:            %.1f us per step" % (one * 1e6))
    print("one trainer, 7 envs:             %.1f us per step" % (half * 1e6))
    print("two trainers x 7, one stream:    %.1f us per step pair" % (ser * 1e6))
    print("two trainers x 7, two streams:   %.1f us per step pair" % (lan * 1e6))


if __name__ == "__main__":
    main()
