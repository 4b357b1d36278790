#!/usr/bin/env python3
"""Phase breakdown of one DD-PPO cycle on the GPU (tuning tool): rollout / update_pol / update_sep wall time with a device
sync at phase boundaries, plus launches per phase from the ops timing hook."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m2h import ops, synthetic  # noqa: E402
from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for kv in os.environ.get("M2H_KNOBS", "").split(","):   # e.g. M2H_KNOBS=24=4096 (m2h_tuning_set knob=value)
        if kv:
            k, v = kv.split("=")
            ops.debug_set(int(k), int(v))
    tr = PPOTrainer(near_target_config(), dev)
    tr.setup()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()}
    tr.actor_critic.load_state_dict(sd)
    tr.train_cycle()
    cfg = tr.config
    phases_only = "--phases" in sys.argv  # wall time per phase only: the op timing hook switches the HIP-graph rollout off
    for rep in range(2):
        ph = {"rollout": 0.0, "update_pol": 0.0, "update_sep": 0.0}
        sink = []
        if not phases_only:
            ops.set_timing(sink)
        marks = {}
        for _sub in range(cfg.num_updates_per_cycle):
            torch.cuda.synchronize(); t = time.perf_counter(); n0 = len(sink)
            for _s in range(cfg.num_steps):
                tr._collect_rollout_step()
            torch.cuda.synchronize(); ph["rollout"] += time.perf_counter() - t; marks["rollout"] = marks.get("rollout", 0) + len(sink) - n0
            t = time.perf_counter(); n0 = len(sink)
            tr._update_pol()
            torch.cuda.synchronize(); ph["update_pol"] += time.perf_counter() - t; marks["update_pol"] = marks.get("update_pol", 0) + len(sink) - n0
        for _sub in range(cfg.num_updates_per_cycle):
            t = time.perf_counter(); n0 = len(sink)
            tr._update_sep()
            torch.cuda.synchronize(); ph["update_sep"] += time.perf_counter() - t; marks["update_sep"] = marks.get("update_sep", 0) + len(sink) - n0
        ops.set_timing(None)
        gpu_ms = {}
        shapes = {}
        for name, meta, e0, e1 in sink:
            ms = e0.elapsed_time(e1)
            gpu_ms[name] = gpu_ms.get(name, 0.0) + ms
            key = (name, meta.get("M"), meta.get("N"), meta.get("K"))
            c = shapes.setdefault(key, [0, 0.0])
            c[0] += 1
            c[1] += ms
        tot = sum(ph.values())
        print("cycle %d: %.3f s  -> %.0f env-steps/s" % (rep, tot, 1680 / tot))
        for k, v in ph.items():
            print("  %-11s %.3f s   timed-op launches %d" % (k, v, marks[k]))
        print("  GPU time of timed ops (ms):", {k: round(v, 1) for k, v in sorted(gpu_ms.items(), key=lambda kv: -kv[1])[:14]})
        if rep == 1:
            print("  per-shape (event time includes the split-K reduce):")
            for (name, M, N, K), (n, ms) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:40]:
                print("    %-22s M=%-8s N=%-6s K=%-6s calls=%5d total=%7.2f ms avg=%7.1f us" % (name, M, N, K, n, ms, 1e3 * ms / n))


if __name__ == "__main__":
    main()
