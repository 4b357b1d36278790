"""Shared helpers of the trainer-golden tests (CPU oracle and GPU product against the reference's own training run).

CPU: the oracle's restatement of the DD-PPO training loop (oracle/m2h_oracle_trainer.py) against fixtures produced by the
REFERENCE's own ``PPOTrainer.train`` (oracle/gen_trainer_golden.py -> tests/golden/trainer_{near,far,ddp2}.npz).

Rows A15 / A18 / A19 of SURVEY 8a: rollout step (second separator pass on the next observation, reward override and the extra
reward at MAX_EPISODE_STEPS - 2, per-episode statistics, the two storage inserts), the cycle (6x pol then 6x sep at fixture
size 2x/2x, LambdaLR, clip decay, window statistics, checkpoint interval) and DDP gradient averaging over two ranks.
"""
import json
import os

import numpy as np
import torch

import m2h_oracle_trainer as OT  # noqa: F401  (STAT_NAMES, stats11)
from m2h import synthetic
from m2h.envs.replay_env import ReplayHostVecEnv

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    d = np.load(os.path.join(GOLD, name), allow_pickle=False)
    cfg = json.loads(str(d["config"]))
    flat = dict(cfg["PPO"], NUM_UPDATES=cfg["NUM_UPDATES"], CHECKPOINT_INTERVAL=cfg["CHECKPOINT_INTERVAL"], MAX_EPISODE_STEPS=cfg["MAX_EPISODE_STEPS"],
                SEED=cfg["SEED"], NUM_PROCESSES=cfg["NUM_PROCESSES"])
    return d, flat, cfg["REPLAY"]


def initial_state_dict(seed, passive_seed):
    """The weights the reference trainer starts from: default init of the policy under torch.manual_seed(SEED)
    (ppo_trainer.py:622-624, 168-177; bit-identical construction pinned by tests/golden/rl_init_seed0.json) with the
    pre-trained passive separators loaded over it (:542-577)."""
    from m2h.common.spaces import Discrete, move2hear_observation_space
    from m2h.rl.ppo.policy import Move2HearPolicy
    torch.manual_seed(seed)
    pol = Move2HearPolicy(move2hear_observation_space(), Discrete(3), "spectrogram", 512, False, True, use_ddppo=True)
    sd = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), passive_seed).items():
        sd[k] = torch.from_numpy(np.asarray(v))
    return sd


def make_env(flat, replay, rank=0):
    return ReplayHostVecEnv(flat["NUM_PROCESSES"], seed=flat["SEED"] + rank * flat["NUM_PROCESSES"], episode_len=flat["MAX_EPISODE_STEPS"],
                            pool=replay["pool"], ragged=replay["ragged"], env_rewards=replay["env_rewards"])


def check_run(d, rec, r, pre, step_tol=2e-5, skip=(), stat_tol=1e-4):
    steps = rec["steps"][r]
    n = len(steps)
    assert n == d[pre + "step.rewards"].shape[0]
    for key, tol in (("rewards", step_tol), ("values", step_tol), ("logp", step_tol), ("probs", step_tol), ("h", step_tol), ("masks", 0)):
        if key in skip:
            continue
        got = np.stack([s[key] for s in steps]).reshape(d[pre + "step." + key].shape)
        want = d[pre + "step." + key]
        assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), (key, np.abs(got - want).max())
    assert np.array_equal(np.stack([s["actions"] for s in steps]).reshape(d[pre + "step.actions"].shape), d[pre + "step.actions"])
    assert np.array_equal(np.stack([s["env_state"] for s in steps]), d[pre + "step.env_state"])
    for name in OT.STAT_NAMES:
        got = np.stack([s["stat." + name] for s in steps])
        want = d[pre + "step.stat." + name]
        assert np.abs(got - want).max() <= stat_tol * max(1.0, np.abs(want).max()), (name, np.abs(got - want).max())
    for key in ("pm_stats", "mono_stats", "mem_stats"):
        got = np.stack([s[key] for s in steps])
        assert np.abs(got - d[pre + "step." + key]).max() <= 1e-4, key


def check_updates(d, rec, pre, r, R):
    for key, tol in (("losses", 2e-4), ("lr", 1e-12), ("clip", 1e-12)):
        got = np.array([(u[key][r] if key == "losses" else u[key]) for u in rec["pol"]])
        want = d[pre + "pol." + key]
        assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), ("pol", key, got, want)
    got = np.stack([u["returns"][r].numpy() for u in rec["pol"]])
    assert np.abs(got - d[pre + "pol.returns"]).max() <= 2e-4 * max(1.0, np.abs(d[pre + "pol.returns"]).max())
    for key, tol in (("losses", 2e-5), ("lr", 1e-12)):
        got = np.array([(u[key][r] if key == "losses" else u[key]) for u in rec["sep"]])
        assert np.abs(got - d[pre + "sep." + key]).max() <= tol, ("sep", key)
    if r == 0:   # only world rank 0 writes checkpoints (ppo_trainer.py:995, :1007-1009)
        assert [c[0] for c in rec["ckpts"]] == [str(x) for x in d[pre + "ckpt_names"]]
        assert [c[1] for c in rec["ckpts"]] == d[pre + "ckpt_after_sep_updates"].tolist()
    else:
        assert d[pre + "ckpt_names"].size == 0


def check_weights(d, rec, pre, tol=3e-5):
    n = 0
    for k in d.files:
        if k.startswith(pre + "post."):
            name = k[len(pre) + 5:]
            got, want = rec["state_dict"][name].numpy(), d[k]
            assert np.abs(got - want).max() <= tol, (name, np.abs(got - want).max())
            n += 1
        elif k.startswith(pre + "postsample."):
            name = k[len(pre) + 11:]
            flat = rec["state_dict"][name].reshape(-1)
            idx = torch.linspace(0, flat.numel() - 1, 64).long()
            assert np.abs(flat[idx].numpy() - d[k]).max() <= tol, name
            n += 1
    assert n >= 30


def check_scalars(d, rec, pre=""):
    for tag in [str(t) for t in d[pre + "scalar_tags"]]:
        want = d[pre + "scalar." + tag]
        got = np.array([[sc[tag], cs] for cs, sc in rec["scalars"]])
        assert got.shape == want.shape, tag
        assert np.abs(got - want).max() <= 2e-4 * max(1.0, np.abs(want).max()), (tag, got, want)




def record_step(store):
    """on_step hook of OT.train: one record per (rank, step), shaped like the fixtures' ``step.*`` arrays."""
    def on_step(r, k, rk):
        values, actions, logp, h, probs = rk.last_act
        ro = rk.ro
        s = (ro.step - 1) % ro.num_steps
        stats11 = OT.stats11
        rec = {"rewards": ro.rewards[s].numpy().copy(), "values": values.numpy().copy(), "logp": logp.numpy().copy(), "probs": probs.numpy().copy(),
               "h": h.numpy().copy(), "masks": ro.masks[s + 1].numpy().copy(), "actions": actions.numpy().copy(), "env_state": rk.envs.s.copy(),
               "pm_stats": stats11(ro.pred_binSepMasks[s]), "mono_stats": stats11(ro.pred_mono[s]), "mem_stats": stats11(ro.prev_pred_monoFromMem[s + 1])}
        for n in OT.STAT_NAMES:
            rec["stat." + n] = rk.stats[n].numpy().copy()
        store[r].append(rec)
    return on_step


def fixture_from_oracle_record(rec, r=0):
    """An oracle run (OT.train's record, steps from ``record_step``) as a dict keyed like the reference-run fixtures
    (oracle/gen_trainer_golden.py run_reference), so that check_run / check_updates / check_scalars / check_weights compare a product
    run with it: used where the expected values depend on inputs made at test time (the fused sampler's recorded noise)."""
    out = {}
    steps = rec["steps"][r]
    for k in steps[0]:
        out["step." + k] = np.stack([np.asarray(s[k]) for s in steps])
    for k in ("losses", "lr", "clip"):
        out["pol." + k] = np.stack([np.asarray(u[k][r] if k == "losses" else u[k]) for u in rec["pol"]])
    out["pol.returns"] = np.stack([u["returns"][r].numpy() for u in rec["pol"]])
    for k in ("losses", "lr"):
        out["sep." + k] = np.stack([np.asarray(u[k][r] if k == "losses" else u[k]) for u in rec["sep"]])
    tags = sorted(rec["scalars"][0][1])
    out["scalar_tags"] = np.array(tags)
    for t in tags:
        out["scalar." + t] = np.array([[float(sc[t]), cs] for cs, sc in rec["scalars"]])
    out["ckpt_names"] = np.array([c[0] for c in rec["ckpts"]])
    out["ckpt_after_sep_updates"] = np.array([c[1] for c in rec["ckpts"]])
    for k, t in rec["state_dict"].items():
        if k.startswith(("pol_net", "action_dist", "critic", "acoustic_mem")):
            if t.numel() <= 100000:
                out["post." + k] = t.numpy().copy()
            else:
                flat = t.reshape(-1)
                out["postsample." + k] = flat[torch.linspace(0, flat.numel() - 1, 64).long()].numpy().copy()

    class _D(dict):
        files = property(lambda self: list(self))
    return _D(out)
