"""CPU: the oracle's restatement of the DD-PPO training loop (oracle/m2h_oracle_trainer.py) against fixtures produced by the
REFERENCE's own ``PPOTrainer.train`` (oracle/gen_trainer_golden.py -> tests/golden/trainer_{near,far,ddp2}.npz).

Rows A15 / A18 / A19 of SURVEY 8a: rollout step (second separator pass on the next observation, reward override and the extra
reward at MAX_EPISODE_STEPS - 2, per-episode statistics, the two storage inserts), the cycle (6x pol then 6x sep at fixture
size 2x/2x, LambdaLR, clip decay, window statistics, checkpoint interval) and DDP gradient averaging over two ranks.
"""
import numpy as np
import pytest
import torch

import m2h_oracle_trainer as OT
from trainer_golden_util import (check_run, check_scalars, check_updates, check_weights, fixture_from_oracle_record, initial_state_dict, load_fixture,
                                 make_env, record_step)


def run_oracle(d, flat, replay, world=1, forced=True, pre=("",), action_noise=None):
    torch.set_num_threads(4)
    sd = initial_state_dict(flat["SEED"], replay["passive_seed"])
    envs = [make_env(flat, replay, r) for r in range(world)]
    fa = None
    if forced:
        fa = [[torch.from_numpy(a) for a in d[p + "step.actions"]] for p in pre]
    store = [[] for _ in range(world)]
    rec = OT.train(flat, envs, sd, forced_actions=fa, distributed=True, on_step=record_step(store), action_noise=action_noise)
    rec["steps"] = store
    return rec


@pytest.mark.parametrize("name", ["trainer_near.npz", "trainer_far.npz", "trainer_unfrozen.npz"])
def test_oracle_training_loop_matches_reference_run(name):
    d, flat, replay = load_fixture(name)
    rec = run_oracle(d, flat, replay)
    check_run(d, rec, 0, "")
    check_updates(d, rec, "", 0, 1)
    check_scalars(d, rec)
    check_weights(d, rec, "")
    if name == "trainer_unfrozen.npz":
        # RL.PPO.train_passive_separators = True in the reference run: the separators are frozen all the same (ppo_trainer.py:637-638,
        # :557-577): their BatchNorm buffers and weights after training are the loaded checkpoint's
        assert flat["train_passive_separators"] is True
        sd0 = initial_state_dict(flat["SEED"], replay["passive_seed"])
        n = 0
        for k in d.files:
            if k.startswith(("bn.", "sepw.")):
                name_ = k.split(".", 1)[1]
                assert np.array_equal(d[k], sd0[name_].numpy()) and np.array_equal(rec["state_dict"][name_].numpy(), d[k]), k
                n += 1
        assert n == 13
    if name == "trainer_near.npz":
        # the extra-reward step (episode step MAX_EPISODE_STEPS - 2 = 3, global steps 3, 8, 13) carries 2 x 10 x util(next)
        r = d["step.rewards"].reshape(len(d["step.rewards"]), -1)
        assert (np.abs(r[[3, 8, 13]]) > 5).all() and (np.abs(r[[0, 1, 2, 5, 6, 7]]) < 2).all() and (r[[4, 9, 14]] == 0).all()


def test_oracle_samples_the_reference_actions_unforced():
    """Same seed, same CPU generator stream (policy init, then one multinomial per step, one randperm per epoch): the oracle's own
    draws reproduce the reference's actions."""
    d, flat, replay = load_fixture("trainer_near.npz")
    rec = run_oracle(d, flat, replay, forced=False)
    assert np.array_equal(np.stack([s["actions"] for s in rec["steps"][0]]), d["step.actions"])


def test_oracle_with_the_generators_noise_handed_in_is_the_unforced_run():
    """OT.train(action_noise=...) -- the entry the fused-sampling parity test (tests/test_gpu_trainer_golden.py) feeds with the noise the
    heads kernel recorded: handed the Exp(1) noise the CPU generator would have drawn at each step, it reproduces the reference's actions
    and, re-keyed like a fixture (fixture_from_oracle_record), its own record passes the checks the reference-run fixture passes."""
    d, flat, replay = load_fixture("trainer_near.npz")
    # the generator's stream of the reference run: policy init, then per step one exponential_ draw of [N, 3]; per epoch one randperm.
    # Replay it once unforced to harvest the per-step noise at the right stream positions.
    noise = []
    orig = OT.draw_actions

    def harvesting(probs, q=None):
        q = torch.empty_like(probs).exponential_(1)
        noise.append(q.clone())
        return orig(probs, q)
    OT.draw_actions = harvesting
    try:
        rec0 = run_oracle(d, flat, replay, forced=False)
    finally:
        OT.draw_actions = orig
    assert np.array_equal(np.stack([s["actions"] for s in rec0["steps"][0]]), d["step.actions"])
    # the same run with that noise handed in (the generator now only serves randperm, whose values one mini-batch never uses)
    rec = run_oracle(d, flat, replay, forced=False, action_noise=[noise])
    check_run(d, rec, 0, "")
    check_updates(d, rec, "", 0, 1)
    check_weights(d, rec, "")
    d2 = fixture_from_oracle_record(rec)
    check_run(d2, rec, 0, "")
    check_updates(d2, rec, "", 0, 1)
    check_scalars(d2, rec)
    check_weights(d2, rec, "")


def test_oracle_two_rank_ddp_matches_reference_run():
    d, flat, replay = load_fixture("trainer_ddp2.npz")
    rec = run_oracle(d, flat, replay, world=2, pre=("rank0.", "rank1."))
    for r in range(2):
        check_run(d, rec, r, "rank%d." % r)
        check_updates(d, rec, "rank%d." % r, r, 2)
    check_scalars(d, rec, "rank0.")
    check_weights(d, rec, "rank0.")
    check_weights(d, rec, "rank1.")


@pytest.mark.parametrize("tag,forced", [("single", True), ("switch", False)])
def test_oracle_eval_loop_matches_reference_eval_run(tag, forced):
    """_eval_checkpoint (one policy / two policies with the switch at step 3) against tests/golden/trainer_eval.npz."""
    import json
    import os
    from m2h import synthetic
    from m2h.envs.replay_env import ReplayHostVecEnv
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trainer_eval.npz"))
    cfg = json.loads(str(d[tag + ".config"]))
    sd = lambda seed: {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), seed).items()}  # noqa: E731
    env = ReplayHostVecEnv(1, seed=cfg["SEED"], episode_len=cfg["MAX_EPISODE_STEPS"], pool=cfg["REPLAY"]["pool"])
    torch.set_num_threads(4)
    fa = [torch.from_numpy(a).reshape(1, 1) for a in d[tag + ".actions"]] if forced else None
    steps, agg = OT.eval_loop(dict(cfg["PPO"]), env, [sd(7), sd(8)] if tag == "switch" else [sd(7)], 4, cfg["PPO"]["deterministic_eval"],
                              switch_thres=cfg["PPO"]["time_thres_for_pol_switch"] if tag == "switch" else None, forced_actions=fa)
    assert np.array_equal(np.array([s[0].reshape(-1).numpy() for s in steps]), d[tag + ".actions"])
    assert np.abs(np.array([s[1] for s in steps]) - d[tag + ".mono_l2"]).max() < 1e-5
    assert np.abs(np.array([s[2] for s in steps]) - d[tag + ".mem_l2"]).max() < 1e-5
    for k, (mean, std) in agg.items():
        want = d[tag + ".agg." + k]
        assert abs(mean - want[0]) < 2e-6 + 1e-5 * abs(want[0]) and abs(std - want[1]) < 2e-6 + 1e-5, (k, mean, std, want)
