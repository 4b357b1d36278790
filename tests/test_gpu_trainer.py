"""GPU: the DD-PPO training cycle end to end on the synthetic env (small schedule), and result-preservation of the
trainer-level re-use of separator outputs."""
import numpy as np
import pytest
import torch

from m2h import synthetic

pytestmark = pytest.mark.gpu


def _trainer(**over):
    from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
    cfg = near_target_config(**dict(dict(NUM_PROCESSES=3, num_steps=4, num_updates_per_cycle=2, ppo_epoch=2, MAX_EPISODE_STEPS=4, use_ddppo=True), **over))
    tr = PPOTrainer(cfg, torch.device("cuda", 0))
    tr.setup()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 5).items()}
    tr.actor_critic.load_state_dict(sd)
    return tr, sd


def test_training_cycle_runs_and_updates_only_trainable_parts():
    tr, sd = _trainer()
    res = tr.train_cycle()
    assert res["env_steps"] == 2 * 4 * 3
    assert all(np.isfinite(res["pol_losses"])) and all(np.isfinite(res["sep_losses"]))
    post = tr.actor_critic.state_dict()
    changed = lambda k: not torch.equal(post[k].cpu(), sd[k])  # noqa: E731
    assert changed("pol_net.visual_encoder.cnn.0.weight") and changed("pol_net.state_encoder.rnn.weight_hh_l0")
    assert changed("action_dist.linear.weight") and changed("critic.fc.bias") and changed("acoustic_mem.cnn.0.weight")
    for k in post:
        if "Sep_" in k or "bin2mono_" in k:
            assert torch.equal(post[k].cpu(), sd[k]), k  # frozen separators untouched
    t = tr.all_reduce_stats()
    assert t.shape == (6,) and float(t[1]) == 2 * 3  # two episodes of 4 steps finished per env
    res2 = tr.train_cycle()
    assert all(np.isfinite(res2["pol_losses"]))


def test_overlapped_grad_reduce_schedule_gives_the_synchronous_weights():
    """running the last all-reduce + clip + Adam of every update on the side stream, fenced at the next reader of
    those parameters, must give bit-identical weights, losses and rollout contents to the synchronous order."""
    runs = []
    for overlap in (False, True):
        tr, _ = _trainer(overlap_grad_reduce=overlap)
        losses = []
        for c in range(2):
            torch.manual_seed(500 + c)  # action sampling draws from the global device generator
            res = tr.train_cycle()
            losses.append((res["pol_losses"], res["sep_losses"]))
        red = tr.agent._reducers
        assert (red["pol"].deferred_steps, red["mem"].deferred_steps) == ((4, 4) if overlap else (0, 0))
        assert red["pol"].pending() == overlap  # the last policy step is still fenced until someone reads the policy
        sd = {k: v.detach().cpu().clone() for k, v in tr.actor_critic.state_dict().items()}  # state_dict() fences
        assert not red["pol"].pending() and not red["mem"].pending()
        runs.append((losses, sd, tr.rollouts_pol.value_preds.cpu().clone(), tr.rollouts_sep.prev_pred_monoFromMem.cpu().clone()))
    (la, sa, va, ma), (lb, sb, vb, mb) = runs
    assert la == lb
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(va, vb) and torch.equal(ma, mb)


@pytest.mark.parametrize("graphs", [False, True])
def test_bucketed_grad_reduce_schedule_gives_the_flat_schedule_weights(graphs):
    """The policy gradient's all-reduce in two buckets (ppo.py:286-319 is DDP's bucketed reducer): the backward stops at the encoders'
    features, the recurrent encoder's and the heads' slice of the flat gradient goes to the side stream, the encoders' backward follows
    -- as two HIP graphs per epoch when graphs are on.  Forced on at world size 1 (no collective: the schedule, the split backward and
    the partial gathers are what is tested): bit-identical weights, losses and storages to the one-backward / one-buffer schedule."""
    runs = []
    for bucketed in (False, True):
        tr, _ = _trainer(bucketed_grad_reduce=bucketed, overlap_grad_reduce=True, use_hip_graphs=graphs)
        losses = []
        for c in range(3):
            torch.manual_seed(700 + c)
            res = tr.train_cycle()
            losses.append((res["pol_losses"], res["sep_losses"]))
        red = tr.agent._reducers["pol"]
        assert red.early_buckets == (3 * 2 * 2 if bucketed else 0)     # cycles x updates x epochs
        if graphs and bucketed:
            assert tr.agent._pol_graph.graph is not None and tr.agent._pol_graph.graph_rest is not None
        sd = {k: v.detach().cpu().clone() for k, v in tr.actor_critic.state_dict().items()}
        runs.append((losses, sd, tr.rollouts_pol.value_preds.cpu().clone()))
    (la, sa, va), (lb, sb, vb) = runs
    assert la == lb
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(va, vb)


def test_hip_graph_rollout_equals_the_kernel_by_kernel_rollout():
    """The rollout step replayed from a HIP graph (device-indexed storage rows, static separator-output buffers, in-place
    packed weights) must leave bit-identical storages, statistics, sampled actions and -- after the updates -- weights."""
    runs = []
    for graphs in (False, True):
        tr, _ = _trainer(use_hip_graphs=graphs, MAX_EPISODE_STEPS=3)  # episodes of 3 against rollouts of 4: every (extra, done) variant
        snaps = []
        for c in range(2):
            torch.manual_seed(900 + c)
            tr.train_cycle()
            snaps.append((tr.rollouts_pol.actions.cpu().clone(), tr.rollouts_pol.rewards.cpu().clone(),
                          tr.rollouts_pol.observations["rgb"].cpu().clone(), tr.rollouts_sep.prev_pred_monoFromMem.cpu().clone(),
                          tr.rollouts_sep.observations["mixed_bin_audio_mag"].cpu().clone(), tr.stats.episode_rewards.cpu().clone(),
                          tr.stats.episode_counts.cpu().clone(), tr.rollouts_pol.recurrent_hidden_states_pol.cpu().clone()))
        gs = tr._graph_state
        assert (gs is not None and len(gs.graphs) == 3) if graphs else gs is None  # (extra, done) in {(0,0), (1,0), (0,1)}
        runs.append((snaps, {k: v.detach().cpu().clone() for k, v in tr.actor_critic.state_dict().items()}))
    (sa, wa), (sb, wb) = runs
    for ca, cb in zip(sa, sb):
        for ta, tb in zip(ca, cb):
            assert torch.equal(ta, tb)
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k


def test_host_vector_env_adapter_reproduces_the_on_device_env_run():
    """Row N4: the reference's host-side vector-env protocol (lists of per-env numpy observation dicts, python actions, done
    flags) through HostVectorEnvAdapter drives the same training run as the on-device env, bit for bit."""
    from m2h.envs.synthetic_env import SyntheticHostVecEnv
    from m2h.envs.vector_env_adapter import HostVectorEnvAdapter
    from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
    dev = torch.device("cuda", 0)
    over = dict(NUM_PROCESSES=3, num_steps=4, num_updates_per_cycle=2, ppo_epoch=2, MAX_EPISODE_STEPS=3, use_ddppo=True, use_hip_graphs=False)
    runs = []
    for host in (False, True):
        cfg = near_target_config(**over)
        envs = HostVectorEnvAdapter(SyntheticHostVecEnv(cfg.NUM_PROCESSES, dev, seed=cfg.SEED, episode_len=cfg.MAX_EPISODE_STEPS), dev) if host else None
        tr = PPOTrainer(cfg, dev, envs=envs)
        tr.setup()
        tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 5).items()})
        for c in range(2):
            torch.manual_seed(700 + c)
            res = tr.train_cycle()
        runs.append((res, tr.rollouts_pol.actions.cpu().clone(), tr.rollouts_pol.rewards.cpu().clone(), tr.rollouts_pol.masks.cpu().clone(),
                     tr.rollouts_sep.observations["gt_bin_comps"].cpu().clone(), tr.stats.episode_counts.cpu().clone(),
                     {k: v.detach().cpu().clone() for k, v in tr.actor_critic.state_dict().items()}))
        if host:
            tr.envs.close()
    a, b = runs
    assert a[0]["pol_losses"] == b[0]["pol_losses"] and a[0]["sep_losses"] == b[0]["sep_losses"]
    for x, y in zip(a[1:6], b[1:6]):
        assert torch.equal(x, y)
    assert float(a[5].sum()) > 0  # episodes did finish (done flags travelled through the host protocol)
    for k in a[6]:
        assert torch.equal(a[6][k], b[6][k]), k


def test_next_step_cache_preserves_rollout_contents():
    """Re-using the next-observation separator outputs as the following step's current outputs stores exactly what a
    from-scratch recomputation stores (frozen eval-mode networks are deterministic per observation)."""
    tr_a, _ = _trainer()
    tr_b, _ = _trainer()
    for i in range(3):
        torch.manual_seed(100 + i)  # action sampling draws from the global device generator: align the two trainers
        tr_a._collect_rollout_step()
        tr_b._next_cache = None  # force recomputation every step
        torch.manual_seed(100 + i)
        tr_b._collect_rollout_step()
    for name in ("pred_binSepMasks", "pred_mono", "prev_pred_monoFromMem", "rewards", "value_preds", "actions"):
        assert torch.equal(getattr(tr_a.rollouts_pol, name), getattr(tr_b.rollouts_pol, name)), name


def test_checkpoint_round_trip_and_eval(tmp_path):
    """save_checkpoint -> load_checkpoint -> load_state_dict reproduces the weights ({"state_dict": actor_critic.*, "config"}
    format of ppo_trainer.py:223-251) and eval() returns the reference's aggregated statistics incl. waveform SI-SDR."""
    tr, sd = _trainer(CHECKPOINT_FOLDER=str(tmp_path))
    tr.train_cycle()
    tr.save_checkpoint("ckpt.0.pth")
    ck = tr.load_checkpoint(str(tmp_path / "ckpt.0.pth"))
    assert set(ck) == {"state_dict", "config"} and all(k.startswith("actor_critic.") for k in ck["state_dict"])
    assert "actor_critic.pol_net.state_encoder.rnn.weight_hh_l0" in ck["state_dict"]
    trained = {k: v.detach().cpu().clone() for k, v in tr.actor_critic.state_dict().items()}
    # the fused action sampler's [seed, counter] rides in the config: a run continued from the file draws on, it does not replay the first steps' noise
    seed_ctr = ck["config"]["m2h_sampler_state"]
    assert seed_ctr == tr.actor_critic.sampler_state() and seed_ctr[1] == 2 * 4 * 3 * 3
    tr2, _ = _trainer(CHECKPOINT_FOLDER=str(tmp_path))
    assert tr2.actor_critic.sampler_state()[1] == 0
    state_ptr = tr2.actor_critic._rng_state.data_ptr()
    tr2.load_state_dict(ck["state_dict"], sampler_state=seed_ctr)
    assert tr2.actor_critic.sampler_state() == seed_ctr and tr2.actor_critic._rng_state.data_ptr() == state_ptr   # (in place: graphs hold the address)
    for k, v in tr2.actor_critic.state_dict().items():
        assert torch.equal(v.cpu(), trained[k]), k
    stats = tr2.eval(num_episodes=3, waveform_metrics=("si_sdr", "si_sdri"), deterministic=True)
    assert stats["num_episodes"] == 3
    for k in ("mono_loss_last_step", "mono_loss_all_steps", "monoFromMem_loss_last_step", "monoFromMem_loss_all_steps",
              "mono_si_sdr", "monoFromMem_si_sdr", "mono_si_sdri"):
        assert np.isfinite(stats[k]["mean"]) and np.isfinite(stats[k]["std"]), k
    # deterministic eval from the same weights and env seed is repeatable
    tr3, _ = _trainer(CHECKPOINT_FOLDER=str(tmp_path))
    stats3 = tr3.eval(num_episodes=3, checkpoint_path=str(tmp_path / "ckpt.0.pth"), waveform_metrics=("si_sdr", "si_sdri"), deterministic=True)
    assert abs(stats3["mono_loss_all_steps"]["mean"] - stats["mono_loss_all_steps"]["mean"]) < 1e-6
    assert abs(stats3["mono_si_sdr"]["mean"] - stats["mono_si_sdr"]["mean"]) < 1e-3


def test_switch_policy_evaluation_with_a_two_policy_checkpoint(tmp_path):
    """Far-target evaluation (RL.PPO.switch_policy, ppo_trainer.py:1093-1130, 1231-1312): the navigation policy acts for the first
    `time_thres_for_pol_switch` steps of an episode, the quality-improvement policy afterwards.  With the threshold beyond the
    episode length the run is the plain evaluation of the navigation checkpoint, with threshold 0 that of the other one."""
    from m2h.rl.ppo.ppo_trainer import PPOTrainer
    paths = {}
    for name, seed in (("nav", 11), ("qual", 12)):
        tr, _ = _trainer(CHECKPOINT_FOLDER=str(tmp_path / name))
        tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), seed).items()})
        tr.save_checkpoint("ckpt.0.pth")
        paths[name] = str(tmp_path / name / "ckpt.0.pth")
    sw = str(tmp_path / "ckpt_polSwitch.pth")
    PPOTrainer.save_switch_checkpoint(sw, paths["nav"], paths["qual"])
    ck = torch.load(sw, map_location="cpu", weights_only=False)
    assert set(ck) == {"state_dict_nav", "config_nav", "state_dict_qualImprov", "config_qualImprov"}

    def run(**kw):
        tr, _ = _trainer()  # episodes of 4 steps; a fresh trainer = the same env seed every time
        return tr.eval(num_episodes=6, waveform_metrics=(), deterministic=True, **kw)
    plain_nav, plain_qual = run(checkpoint_path=paths["nav"]), run(checkpoint_path=paths["qual"])
    late, never_nav, mixed = (run(switch_checkpoint_path=sw, time_thres_for_pol_switch=t) for t in (99, 0, 2))
    key = "monoFromMem_loss_all_steps"
    assert late == plain_nav and never_nav == plain_qual
    assert plain_nav[key]["mean"] != plain_qual[key]["mean"]
    assert mixed["num_episodes"] == 6 and np.isfinite(mixed[key]["mean"]) and mixed != plain_nav and mixed != plain_qual
    with pytest.raises(RuntimeError):
        run(switch_checkpoint_path=paths["nav"])  # not a two-policy file


def test_far_target_schedule_runs_with_env_rewards():
    """farTarget.yaml (BASELINE config 5): nav_reward_weight 1 / sep_reward_weight 0 keeps the environment's
    rewards (no quality-improvement override) and episodes are longer than the rollout; the cycle must run and update."""
    from m2h.rl.ppo.ppo_trainer import PPOTrainer, far_target_config
    cfg = far_target_config(NUM_PROCESSES=2, num_steps=4, num_updates_per_cycle=2, ppo_epoch=2, MAX_EPISODE_STEPS=6, use_ddppo=True)
    assert cfg.nav_reward_weight == 1.0 and cfg.sep_reward_weight == 0.0
    tr = PPOTrainer(cfg, torch.device("cuda", 0))
    tr.setup()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 6).items()}
    tr.actor_critic.load_state_dict(sd)
    res = tr.train_cycle()
    assert res["env_steps"] == 2 * 4 * 2
    assert all(np.isfinite(res["pol_losses"])) and all(np.isfinite(res["sep_losses"]))
    assert float(tr.rollouts_pol.rewards.abs().sum()) == 0.0           # the synthetic env's nav reward is zero and is NOT overridden
    assert float(tr.stats.episode_counts.sum()) == 2                    # 8 steps per env, episodes of 6 -> one finished episode each
    assert not torch.equal(tr.actor_critic.state_dict()["critic.fc.weight"].cpu(), sd["critic.fc.weight"])


def test_eval_between_training_cycles_leaves_frozen_separators_untouched():
    """eval() on a training trainer must restore the per-module training flags -- the frozen separators stay in eval
    mode (ppo_trainer.py:557-577), so a later train_cycle neither takes the train-mode BatchNorm path nor updates their running
    statistics, and the captured rollout graphs stay valid."""
    tr, _ = _trainer()
    torch.manual_seed(11)
    tr.train_cycle()
    ac = tr.actor_critic
    flags = {n: m.training for n, m in ac.named_modules()}
    bufs = {k: v.detach().cpu().clone() for k, v in ac.state_dict().items() if "Sep_" in k or "bin2mono_" in k}
    tr.eval(num_episodes=3, waveform_metrics=())
    assert {n: m.training for n, m in ac.named_modules()} == flags
    assert ac.training and not ac.binSep_enc.training and not ac.bin2mono_dec.training
    torch.manual_seed(12)
    res = tr.train_cycle()
    assert all(np.isfinite(res["pol_losses"]))
    post = ac.state_dict()
    for k, v in bufs.items():
        assert torch.equal(post[k].cpu(), v), k   # weights, BN running statistics and num_batches_tracked of the frozen separators


def test_loading_weights_after_graph_capture_invalidates_the_graphs():
    """load_state_dict() after the rollout / update graphs were captured: the graphs hold addresses of packed weights
    and folded-BN buffers that the load replaces, so they must be dropped.  A trainer that trained, then loaded checkpoint W,
    must continue exactly like a fresh trainer that loaded W."""
    w = {"actor_critic." + k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 9).items()}

    def run(pretrain):
        tr, _ = _trainer(use_hip_graphs=True)
        if pretrain:
            for c in range(2):
                torch.manual_seed(70 + c)
                tr.train_cycle()
            assert tr._graph_state is not None and tr.agent._pol_graph is not None
        tr.load_state_dict(w)
        assert tr._graph_state is None and tr.agent._pol_graph is None
        return tr

    tr_a = run(True)
    # the frozen separators are a function of the weights alone: feed both trainers the same observation
    obs = {k: v[0].clone() for k, v in tr_a.rollouts_pol.observations.items()}
    outs = []
    for tr in (tr_a, run(False)):
        with torch.no_grad():
            pm = tr.actor_critic.get_binSepMasks(obs)
            outs.append((pm.cpu(), tr.actor_critic.convert_bin2mono(pm, mixed_audio=obs["mixed_bin_audio_mag"]).cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # and the trainer keeps training from the loaded weights through freshly captured graphs
    torch.manual_seed(5)
    tr_a.train_cycle()
    torch.manual_seed(6)
    res = tr_a.train_cycle()
    assert tr_a._graph_state is not None and all(np.isfinite(res["pol_losses"]))


def test_short_rollouts_when_enough_other_ranks_wait_at_the_update():
    """Straggler pre-emption (ppo_trainer.py:769-782; ddppo_utils.RolloutTracker): with short_rollout_threshold 0.5 a rank that sees more than
    sync_frac of the ranks done stops its rollout after half of its steps; the update runs on the storage as it stands (the reference does the
    same: rollout_storage.py has no notion of a short rollout) and the HIP-graph replay picks the device step indices up again.  At the
    shipped threshold 1.0 no tracker exists at all (no store traffic)."""
    tr, _sd = _trainer()
    assert tr.rollout_tracker is None
    res = tr.train_cycle()
    assert res["env_steps"] == 2 * 4 * 3
    tr2, _sd = _trainer(short_rollout_threshold=0.5, sync_frac=0.6)
    assert tr2.rollout_tracker is not None and tr2.rollout_tracker.world_size == 1
    res = tr2.train_cycle()                     # alone in the job: nobody is ever waiting, full rollouts
    assert res["env_steps"] == 2 * 4 * 3 and tr2.rollout_tracker.num_done() == 0
    tr2.rollout_tracker.num_done = lambda: 1    # one (other) rank done: 1 > 0.6 x 1
    for _ in range(2):
        res = tr2.train_cycle()
        assert res["env_steps"] == 2 * 3 * 3    # steps 0, 1, 2 of 4: the check after step 2 (>= 0.5 x 4) ends the rollout
        assert all(np.isfinite(res["pol_losses"])) and all(np.isfinite(res["sep_losses"]))
