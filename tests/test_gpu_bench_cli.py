"""GPU: ``python bench.py --gpus 2`` with no launcher starts its own two ranks (rehearsed on the box's one card: both ranks pinned
to GPU 0, collectives over gloo -- the driver's runs use RCCL and one card per rank) and rank 0 prints one JSON line that says so."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_spawns_two_ranks_and_reports_them():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(M2H_BENCH_DEVICE="0", M2H_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--tm", "32",
           "--ddppo-cycles", "1", "--no-far-target", "--train-steps", "0", "--feeder-steps", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_observed"] == 2 and line["scaling"] == "weak"
    assert line["ddppo"]["n_gpus"] == 2 and line["ddppo"]["value"] > 0
    assert line["roofline"]["frac"] > 0 and line["config"]["batch_per_gpu"] == 8
    # what a first real SCALE run needs to be diagnosable: per-phase GPU time, the gradient all-reduces (count, payload, time on
    # the stream they ran on), a stand-alone all-reduce of the policy gradient's size, and who ran where
    ph = line["ddppo"]["phases"]
    assert ph["rollout_ms"] > 0 and ph["update_pol_ms"] > 0 and ph["update_sep_ms"] > 0
    # 6 x 4 policy backward passes in two buckets each (recurrent encoder + heads under the encoders' backward, then the encoders) + 6 x 4 memory steps
    assert ph["grad_allreduce_count"] == 72 and ph["grad_allreduce_ms"] > 0 and ph["allreduce_23MB_us"] > 0
    # exposed (enqueued on the compute stream): the encoders' bucket of 3 of every 4 policy epochs, 3 of every 4 memory steps; the rest run on the side stream
    assert ph["grad_allreduce_exposed_count"] == 18 + 18 and 0 < ph["grad_allreduce_exposed_ms"] <= ph["grad_allreduce_ms"]
    pay = ph["grad_allreduce_payload_bytes"]
    assert len(pay) == 3 and pay[0] + pay[1] == ph["policy_grad_bytes"] > 20e6 and pay[2] < 100e3   # two policy buckets, the acoustic memory
    assert line["ddppo"]["roofline"]["phases"]["rollout"]["m2h_kernel_launches_per_cycle"] > 1000
    devs = line["ddppo"]["devices"]
    assert [d["rank"] for d in devs] == [0, 1] and all(d["name"] for d in devs) and line["ddppo"]["distinct_devices"] == 1   # (both ranks share the box's card here)
    rf = line["ddppo"]["roofline"]
    assert rf["algorithmic_gflop_per_env_step"] == 13.8 and 0 < rf["frac"] < 1 and rf["executed_gflop_per_env_step"] > 3


def test_bench_refuses_a_rank_count_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "process group has 1 rank" in r.stderr
