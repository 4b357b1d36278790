"""GPU: the m2h DD-PPO trainer against the REFERENCE's own training run (tests/golden/trainer_{near,far,ddp2}.npz, produced by
oracle/gen_trainer_golden.py from the reference's ``PPOTrainer.train``).  Rows A15 / A18 / A19 / N4 of SURVEY 8.

The same table-driven replay world runs on both sides and NOTHING of the reference's trajectory is fed in: the trainer samples
its own actions in ``action_sampling="cpu_generator"`` mode -- the Exp(1) noise of torch.multinomial's single-draw path drawn on
the CPU default generator (mt19937) at the reference's stream position (policy init, one draw per step, one randperm per
epoch), divided and arg-maxed on the device (m2h_sample_actions) -- so from the seed alone the sampled actions must equal the
reference-CPU run's, bit for bit, and with them everything downstream: per-step rewards (incl. the 2 x 10 x extra reward at
MAX_EPISODE_STEPS - 2 and the zero at episode ends), values, log-probs, hidden states, stored separator outputs, all 17
per-episode statistics, per-update losses / learning rates / clip ranges / returns, the window statistics the reference logs,
the checkpoint schedule and the weights after two cycles.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import m2h_oracle_trainer as OT
from trainer_golden_util import (check_run, check_scalars, check_updates, check_weights, fixture_from_oracle_record, initial_state_dict, load_fixture,
                                 make_env, record_step)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _passive_ckpt(seed):
    from m2h import synthetic
    return {"actor_critic." + k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), seed).items()}


def run_m2h(d, flat, replay, env_kind, graphs, rank=0, world=1, pre="", sampling="cpu_generator"):
    """Runs the product trainer over the fixture's schedule (its own sampled actions); returns a record shaped like the oracle's.
    sampling="fused": the trainers' default draw; every step's Exp(1) noise, as the heads kernel drew it, is kept in rec["noise"]."""
    from m2h.envs.replay_env import ReplayVecEnv
    from m2h.envs.vector_env_adapter import HostVectorEnvAdapter
    from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
    dev = torch.device("cuda", 0)
    keys = {k: flat[k] for k in ("num_updates_per_cycle", "hidden_size", "value_loss_coef", "entropy_coef", "lr_pol", "lr_sep", "clip_param", "ppo_epoch",
                                 "num_mini_batch", "eps", "max_grad_norm", "num_steps", "use_gae", "gamma", "tau", "use_linear_clip_decay",
                                 "use_linear_lr_decay", "sep_reward_weight", "nav_reward_weight", "extra_reward_multiplier", "reward_window_size",
                                 "use_ddppo", "NUM_UPDATES", "CHECKPOINT_INTERVAL", "MAX_EPISODE_STEPS", "SEED", "NUM_PROCESSES", "train_passive_separators")}
    cfg = near_target_config(use_hip_graphs=graphs, action_sampling=sampling, record_action_noise=sampling == "fused", **keys)
    seed = flat["SEED"] + rank * flat["NUM_PROCESSES"]
    if env_kind == "device":
        envs = ReplayVecEnv(flat["NUM_PROCESSES"], dev, seed=seed, episode_len=flat["MAX_EPISODE_STEPS"], pool=replay["pool"],
                            env_rewards=replay["env_rewards"])
        state = lambda: envs.s.cpu().numpy().copy()  # noqa: E731
    else:
        host = make_env(flat, replay, rank)
        envs = HostVectorEnvAdapter(host, dev)
        state = lambda: host.s.copy()  # noqa: E731
    tr = PPOTrainer(cfg, dev, world_rank=rank, world_size=world, envs=envs)
    tr.setup(passive_state_dict=_passive_ckpt(replay["passive_seed"]))
    steps, saved, noise = [], [], []
    orig_step, orig_pol, orig_sep = tr._collect_rollout_step, tr._update_pol, tr._update_sep

    def step():
        n = orig_step()
        ro = tr.rollouts_pol
        s = (ro.step - 1) % ro.num_steps
        rec = {"rewards": ro.rewards[s], "values": ro.value_preds[s], "logp": ro.action_log_probs[s], "h": ro.recurrent_hidden_states_pol[s + 1],
               "masks": ro.masks[s + 1], "actions": ro.actions[s]}
        rec = {k: v.cpu().numpy().copy() for k, v in rec.items()}
        rec.update(env_state=state(), pm_stats=OT.stats11(ro.pred_binSepMasks[s].cpu()), mono_stats=OT.stats11(ro.pred_mono[s].cpu()),
                   mem_stats=OT.stats11(ro.prev_pred_monoFromMem[s + 1].cpu()))
        for name in OT.STAT_NAMES:
            rec["stat." + name] = getattr(tr.stats, name).cpu().numpy().copy()
        steps.append(rec)
        if sampling == "fused":
            noise.append(tr.actor_critic.last_action_noise.cpu().clone())
        return n
    pol, sep = [], []

    def update_pol(as_tensor=False):
        lr, clip = tr.agent.optimizer_pol.param_groups[0]["lr"], tr.agent.clip_param
        out = orig_pol(as_tensor=as_tensor)
        pol.append({"losses": np.array([tuple(out.tolist()) if torch.is_tensor(out) else out]), "lr": lr, "clip": clip,
                    "returns": [tr.rollouts_pol.returns.cpu().clone()]})
        return out

    def update_sep(as_tensor=False):   # (train_cycle keeps the losses on the device until the cycle's last update is enqueued)
        lr = tr.agent.optimizer_sep.param_groups[0]["lr"]
        out = orig_sep(as_tensor=as_tensor)
        sep.append({"losses": np.array([tuple(out.tolist()) if torch.is_tensor(out) else out]), "lr": lr})
        return out
    tr._collect_rollout_step, tr._update_pol, tr._update_sep = step, update_pol, update_sep
    tr.save_checkpoint = lambda name: saved.append((name, len(sep)))
    cfg.CHECKPOINT_FOLDER = "unused"
    tr.train()
    graph_replays = 0 if tr._graph_state is None else len(tr._graph_state.graphs)
    return {"steps": [steps], "pol": pol, "sep": sep, "ckpts": saved, "scalars": tr.scalars, "graphs": graph_replays, "trainer": tr, "noise": noise,
            "state_dict": {k: v.detach().cpu() for k, v in tr.actor_critic.state_dict().items()}}


def check_all(d, rec, pre="", rank=0):
    check_run(d, rec, 0, pre, step_tol=5e-5, skip=("probs",), stat_tol=2e-4)
    check_updates(d, rec, pre, 0, 1)
    if rank == 0:
        check_scalars(d, rec, pre)
    check_weights(d, rec, pre, tol=1e-4)


@pytest.mark.parametrize("env_kind,graphs", [("device", True), ("device", False), ("host", False)])
def test_near_target_training_matches_the_reference_run(env_kind, graphs):
    d, flat, replay = load_fixture("trainer_near.npz")
    rec = run_m2h(d, flat, replay, env_kind, graphs)
    check_all(d, rec)
    if graphs:   # the steps really were replayed: one graph per (extra-reward, episode-end) flag pair that occurred
        assert rec["graphs"] == 3 and rec["trainer"].agent._pol_graph is not None
    # the frozen separators' BatchNorm statistics are untouched by training
    assert np.array_equal(rec["state_dict"]["binSep_enc.passive_sep_encoder.cnn.0.1.running_mean"].numpy(), d["frozen_bn_running_mean0"])


@pytest.mark.parametrize("graphs", [True, False])
def test_near_target_training_in_the_default_fused_sampling_mode_matches_the_oracle_on_the_recorded_noise(graphs):
    """The mode the trainers (and bench.py's DD-PPO legs) run by default, ``action_sampling="fused"``: the Exp(1) noise of
    torch.multinomial's single draw (common/utils.py:16-24, rl/ppo/policy.py:198-225) is made inside the heads kernel, so no seed of the
    reference's CPU generator reproduces it and the reference-run fixture's trajectory is not this run's.  The contract of SURVEY 8(d) --
    actions bit-exact GIVEN probs and the draw's noise -- is pinned end to end instead: the kernel writes the noise it drew at every step
    (m2h_policy_heads_act_rng noise_out), the CPU oracle's training loop (oracle/m2h_oracle_trainer.py, itself pinned to the reference's
    own PPOTrainer.train run by tests/test_oracle_trainer_golden.py, its noise-fed draw to the reference's sample() by
    tests/test_oracle_rl_golden.py) runs the same schedule from the same seeds with that noise in place of its generator's draw, and the
    product's whole run must equal the oracle's: actions and env states bit for bit, rewards / values / log-probs / hidden states /
    stored separator outputs / 17 statistics per step, losses, learning rates, clip ranges, returns, window scalars, the checkpoint
    schedule and the weights after two cycles, at the tolerances of the reference-run tests."""
    d, flat, replay = load_fixture("trainer_near.npz")
    rec = run_m2h(d, flat, replay, "device", graphs, sampling="fused")
    tr = rec["trainer"]
    n_steps = len(rec["steps"][0])
    assert len(rec["noise"]) == n_steps == d["step.actions"].shape[0]
    noise = torch.stack(rec["noise"])
    N, A = flat["NUM_PROCESSES"], 3
    assert noise.shape == (n_steps, N, A) and bool((noise > 0).all()) and bool(torch.isfinite(noise).all())
    # every step drew fresh noise (the device counter moved on by N x A per step, inside the replayed graphs too) ...
    assert len({tuple(q.reshape(-1).tolist()) for q in rec["noise"]}) == n_steps
    seed, ctr = tr.actor_critic.sampler_state()
    assert ctr == n_steps * N * A and seed == 0x5eed0000 + flat["SEED"]
    # ... and it is the Philox stream of the documented (seed, counter) layout
    from kernel_model import philox_exp1 as _philox_exp1
    want_noise, _u = _philox_exp1(seed, np.arange(n_steps * N * A, dtype=np.uint64))
    assert np.abs(noise.numpy().reshape(-1) - want_noise).max() <= 3e-7 * np.abs(want_noise).max()
    if graphs:
        assert rec["graphs"] == 3
    # the oracle's loop on the recorded noise
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    store = [[]]
    want = OT.train(flat, [make_env(flat, replay, 0)], initial_state_dict(flat["SEED"], replay["passive_seed"]), distributed=True,
                    on_step=record_step(store), action_noise=[list(rec["noise"])])
    want["steps"] = store
    d2 = fixture_from_oracle_record(want)
    assert not np.array_equal(d2["step.actions"], d["step.actions"])      # (another noise stream than the reference run's: another trajectory)
    check_run(d2, rec, 0, "", step_tol=5e-5, skip=("probs",), stat_tol=2e-4)
    check_updates(d2, rec, "", 0, 1)
    check_scalars(d2, rec, "")
    check_weights(d2, rec, "", tol=1e-4)
    # the draw itself, on the oracle's own CPU probabilities: argmax(probs / recorded noise) at every step, exactly
    for k, s in enumerate(store[0]):
        assert np.array_equal((s["probs"] / rec["noise"][k].numpy()).argmax(1).reshape(-1, 1), rec["steps"][0][k]["actions"])


def test_train_passive_separators_key_changes_nothing_as_in_the_reference():
    """RL.PPO.train_passive_separators = True (ppo_trainer.py:72-73): the reference's train() freezes the separators all the same
    (:637-638, :557-577) and PPO never reads the flag it stores (ppo.py:46) -- the reference run with the key set is the fixture; the
    product with the key set reproduces it: same trajectory, losses and weights, separator BatchNorm buffers and weights untouched,
    the step still replayed from HIP graphs."""
    d, flat, replay = load_fixture("trainer_unfrozen.npz")
    assert flat["train_passive_separators"] is True
    rec = run_m2h(d, flat, replay, "device", True)
    check_all(d, rec)
    assert rec["graphs"] >= 2 and rec["trainer"].agent.freeze_passive_separators is False
    for k in d.files:
        if k.startswith(("bn.", "sepw.")):
            assert np.array_equal(rec["state_dict"][k.split(".", 1)[1]].numpy(), d[k]), k


def test_far_target_training_with_ragged_episodes_matches_the_reference_run():
    """farTarget.yaml's reward wiring (the env's own reward, no override), episodes ending at different steps per env, non-zero
    distance infos, through the host vector-env adapter (row N4)."""
    d, flat, replay = load_fixture("trainer_far.npz")
    rec = run_m2h(d, flat, replay, "host", False)
    check_all(d, rec)
    assert float(np.abs(d["step.stat.episode_ndgs"][-1]).sum()) > 0


@pytest.mark.parametrize("tag", ["single", "switch"])
def test_evaluation_loop_matches_the_reference_eval_run(tag, tmp_path):
    """PPOTrainer.eval against the reference's own ``_eval_checkpoint`` (ppo_trainer.py:1015-1551) on the replay env, one process:
    ``single`` = one policy with sampled actions (cpu_generator mode: seed, policy construction, then one draw per step, as
    :1031-1089 -- the reference's actions must come out of the seed alone), ``switch`` = the far-target evaluation with two
    policies and deterministic actions (the argmax actions must coincide).  Per-step STFT-L2 distances and the four
    per-episode aggregates the reference logs."""
    import json
    from m2h import synthetic
    from m2h.envs.replay_env import ReplayHostVecEnv
    from m2h.envs.vector_env_adapter import HostVectorEnvAdapter
    from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
    d = np.load(os.path.join(ROOT, "tests", "golden", "trainer_eval.npz"))
    cfg_rec = json.loads(str(d[tag + ".config"]))
    dev = torch.device("cuda", 0)
    cfg = near_target_config(NUM_PROCESSES=1, MAX_EPISODE_STEPS=cfg_rec["MAX_EPISODE_STEPS"], SEED=cfg_rec["SEED"], use_hip_graphs=False,
                             deterministic_eval=cfg_rec["PPO"]["deterministic_eval"], action_sampling="cpu_generator",
                             time_thres_for_pol_switch=cfg_rec["PPO"]["time_thres_for_pol_switch"])
    host = ReplayHostVecEnv(1, seed=cfg.SEED, episode_len=cfg.MAX_EPISODE_STEPS, pool=cfg_rec["REPLAY"]["pool"])
    tr = PPOTrainer(cfg, dev, envs=HostVectorEnvAdapter(host, dev))
    tr.setup()
    ck = lambda seed: {"actor_critic." + k: torch.from_numpy(np.asarray(v)) for k, v in  # noqa: E731
                       synthetic.make_state_dict(synthetic.policy_shapes(), seed).items()}
    trace = []
    want_actions = d[tag + ".actions"]
    if tag == "switch":
        path = str(tmp_path / "switch.pth")
        PPOTrainer.save_switch_checkpoint(path, {"state_dict": ck(7), "config": {}}, {"state_dict": ck(8), "config": {}})
        agg = tr.eval(num_episodes=4, switch_checkpoint_path=path, waveform_metrics=(), trace=trace)
    else:
        path = str(tmp_path / "single.pth")
        torch.save({"state_dict": ck(7), "config": {}}, path)
        agg = tr.eval(num_episodes=4, checkpoint_path=path, waveform_metrics=(), trace=trace)
    assert len(trace) == len(want_actions) == 24 and agg["num_episodes"] == 4
    assert np.array_equal(np.stack([t[0].numpy().reshape(-1) for t in trace]), want_actions)
    assert host.actions_seen == want_actions.tolist()
    mono = np.array([float(t[1]) for t in trace])
    mem = np.array([float(t[2]) for t in trace])
    assert np.abs(mono - d[tag + ".mono_l2"]).max() <= 2e-5 * np.abs(d[tag + ".mono_l2"]).max()
    assert np.abs(mem - d[tag + ".mem_l2"]).max() <= 2e-5 * np.abs(d[tag + ".mem_l2"]).max()
    for key in ("mono_loss_last_step", "mono_loss_all_steps", "monoFromMem_loss_last_step", "monoFromMem_loss_all_steps"):
        want = d[tag + ".agg." + key]              # (mean, std) as the reference logs them: six decimals
        assert abs(agg[key]["mean"] - want[0]) < 2e-5 * max(1, abs(want[0])) + 1e-6 and abs(agg[key]["std"] - want[1]) < 2e-5 + 1e-6, (key, agg[key], want)


TWO_RANK = r'''
import os, sys
root = %(root)r
for p in ("move2hear-active-av-separation_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(root, p))
import torch, torch.distributed as dist
rank = int(sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%(port)d), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=2)   # both ranks share the box's one GPU; collectives over gloo
from trainer_golden_util import load_fixture
from test_gpu_trainer_golden import run_m2h, check_all
d, flat, replay = load_fixture("trainer_ddp2.npz")
pre = "rank%%d." %% rank
rec = run_m2h(d, flat, replay, "device", True, rank=rank, world=2, pre=pre)
check_all(d, rec, pre, rank)
assert rec["trainer"].agent._world == 2 and rec["trainer"].agent._reducers["pol"].deferred_steps == flat["NUM_UPDATES"]
dist.barrier()
dist.destroy_process_group()
print("RANK_OK", rank)
'''


def test_two_rank_ddppo_matches_the_reference_ddp_run():
    """Two DD-PPO ranks against the reference's two-rank DistributedDataParallel run (gloo, CPU): per-rank seeds and
    environments, rank-0 broadcast, flat-gradient all-reduce averaged before clip + Adam (deferred last step on the side
    stream), distributed advantage statistics, summed window statistics; per-rank losses and the weights of both replicas."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", TWO_RANK % {"root": ROOT, "port": port}, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK_OK %d" % r) in o, o[-4000:]
