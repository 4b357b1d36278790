#!/usr/bin/env python3
"""Tuning probe: the DD-PPO rollout as ONE chain over 14 envs vs TWO / FOUR independent chains over 7 / 4+3 envs each on their own
HIP streams.  A replayed rollout step is ~70 dependent kernels of 3-20 us that each occupy a fraction of the chip: independent
env groups are independent chains, so they should overlap almost perfectly.  Separate trainers stand in for the lanes here."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import synthetic  # noqa: E402
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config  # noqa: E402


def make(n, dev):
    tr = PPOTrainer(near_target_config(NUM_PROCESSES=n), dev)
    tr.setup()
    tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
    return tr


def main():
    dev = torch.device("cuda", 0)
    steps = 60
    for lanes in ((14,), (7, 7), (4, 4, 3, 3)):
        trs = [make(n, dev) for n in lanes]
        streams = [torch.cuda.Stream() for _ in lanes]
        for tr, s in zip(trs, streams):
            with torch.cuda.stream(s):
                for _ in range(25):
                    tr._collect_rollout_step()   # warm-up, graph capture (one per flag pair)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for tr, s in zip(trs, streams):
                with torch.cuda.stream(s):
                    tr._collect_rollout_step()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("lanes %-14s  %7.1f us per step of all %d envs (host issue %.1f us)" % (lanes, 1e6 * el / steps, sum(lanes), 1e6 * t_issue / steps))
        del trs


if __name__ == "__main__":
    main()


def forked():
    """The same lanes as parallel branches of ONE HIP graph (fork / join inside the capture): no reliance on how separately created
    streams map to hardware queues."""
    from m2h import graphs, ops
    dev = torch.device("cuda", 0)
    for lanes in ((14,), (7, 7), (5, 5, 4), (4, 4, 3, 3)):
        trs = [make(n, dev) for n in lanes]
        for tr in trs:
            for _ in range(25):
                tr._collect_rollout_step()
        torch.cuda.synchronize()
        side = [torch.cuda.Stream() for _ in lanes[1:]]
        g = torch.cuda.CUDAGraph()

        def body():
            cur = torch.cuda.current_stream()
            for tr, s in zip(trs[1:], side):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    gs = tr._graph_state
                    tr._rollout_step_device(gs.cache, gs.idx, False, False)
                    ops.step_index_advance(gs.idx, tr.rollouts_pol.num_steps, tr.rollouts_sep.num_steps)
            gs = trs[0]._graph_state
            trs[0]._rollout_step_device(gs.cache, gs.idx, False, False)
            ops.step_index_advance(gs.idx, trs[0].rollouts_pol.num_steps, trs[0].rollouts_sep.num_steps)
            for s in side:
                cur.wait_stream(s)
        with torch.no_grad(), graphs.capture(g):
            body()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        steps = 60
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        print("one graph, branches %-14s %7.1f us per step of all %d envs" % (lanes, 1e6 * (time.perf_counter() - t0) / steps, sum(lanes)))
        del trs, g


if __name__ == "__main__" and os.environ.get("M2H_FORKED", "1") == "1":
    forked()
