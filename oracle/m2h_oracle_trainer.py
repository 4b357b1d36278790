"""CPU oracle for the DD-PPO training loop -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as m2h_oracle.py).

A plain PyTorch-CPU restatement of the reference's rollout step, storages, updates and training cycle on top of the
function-level oracle in ``m2h_oracle.py``; paths are relative to the reference root ``audio_separation/``:

  collect_rollout_step   rl/ppo/ppo_trainer.py:253-478
  PolStorage/SepStorage  common/rollout_storage.py:6-312, :315-471
  update_pol/update_sep  rl/ppo/ppo.py:82-246 (+ DDP gradient averaging :286-319, distributed advantages :275-284)
  train                  rl/ppo/ppo_trainer.py:579-1013 (schedule, LambdaLR, clip decay, window statistics, checkpoint interval)

Pinned against the reference's own ``PPOTrainer.train`` run in the build container (oracle/gen_trainer_golden.py ->
tests/golden/trainer_{near,far,ddp2}.npz; tests/test_oracle_trainer_golden.py).

Several ranks are emulated in ONE process: every rank has its own env, storages and statistics, all share one set of
parameters (DDP keeps replicas identical), gradients are averaged over ranks before clip + Adam (what DDP's reducer does),
advantage statistics and logging sums follow ddppo_utils.py:168-190 / ppo_trainer.py:790-860.
"""
import warnings
from collections import deque

import numpy as np
import torch
import torch.nn.functional as F

import m2h_oracle as O

POL_PREFIXES = ("pol_net.", "action_dist.", "critic.")
MEM_PREFIXES = ("acoustic_mem.",)
INFO_KEYS = ("normalized_geo_distance_to_target_audio_source", "geo_distance_to_target_audio_source")
STAT_SHAPES = {"episode_dist_probs": 3, "current_episode_dist_probs": 3}
STAT_NAMES = ("current_episode_reward", "current_episode_step", "current_episode_dist_probs", "current_episode_bin_losses",
              "current_episode_mono_losses", "current_episode_monoFromMem_losses", "episode_rewards", "episode_counts",
              "episode_steps", "episode_dist_probs", "episode_bin_losses_allSteps", "episode_mono_losses_lastStep",
              "episode_mono_losses_allSteps", "episode_monoFromMem_losses_lastStep", "episode_monoFromMem_losses_allSteps",
              "episode_ndgs", "episode_dgs")


def stats11(t):
    """Fixture digest of a large tensor: mean, mean |.|, std and 8 evenly spaced elements (float64)."""
    t = t.detach().double()
    flat = t.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 8).long()
    return np.concatenate([[t.mean().item(), t.abs().mean().item(), t.std().item()], flat[idx].numpy()])


def batch_obs(observations):
    """common/utils.py:75-97: list of per-env dicts -> dict of stacked float tensors."""
    return {k: torch.stack([torch.as_tensor(np.asarray(o[k])).float() for o in observations], 0) for k in observations[0]}


class PolStorage:
    """common/rollout_storage.py:6-312."""

    def __init__(self, T, N, obs_shapes, hidden):
        self.observations = {k: torch.zeros(T + 1, N, *s) for k, s in obs_shapes.items()}
        self.recurrent_hidden_states_pol = torch.zeros(T + 1, 1, N, hidden)
        f, t = obs_shapes["gt_mono_comps"][:2]
        self.pred_binSepMasks = torch.zeros(T, N, f, t, 2)
        self.pred_mono = torch.zeros(T, N, f, t, 1)
        self.prev_pred_monoFromMem = torch.zeros(T + 1, N, f, t, 1)
        self.rewards = torch.zeros(T, N, 1)
        self.value_preds = torch.zeros(T + 1, N, 1)
        self.returns = torch.zeros(T + 1, N, 1)
        self.action_log_probs = torch.zeros(T, N, 1)
        self.actions = torch.zeros(T, N, 1).long()
        self.masks = torch.ones(T + 1, N, 1)
        self.num_steps, self.step = T, 0

    def insert(self, obs, h, actions, logp, values, rewards, masks, pm, mono, mem):   # :103-148
        s = self.step
        for k in obs:
            self.observations[k][s + 1].copy_(obs[k])
        self.recurrent_hidden_states_pol[s + 1].copy_(h)
        self.pred_binSepMasks[s].copy_(pm)
        self.pred_mono[s].copy_(mono)
        self.prev_pred_monoFromMem[s + 1].copy_(mem)
        self.rewards[s].copy_(rewards)
        self.value_preds[s].copy_(values)
        self.actions[s].copy_(actions)
        self.action_log_probs[s].copy_(logp)
        self.masks[s + 1].copy_(masks)
        self.step = (s + 1) % self.num_steps

    def after_update(self):   # :150-157
        for k in self.observations:
            self.observations[k][0].copy_(self.observations[k][-1])
        self.recurrent_hidden_states_pol[0].copy_(self.recurrent_hidden_states_pol[-1])
        self.prev_pred_monoFromMem[0].copy_(self.prev_pred_monoFromMem[-1])
        self.masks[0].copy_(self.masks[-1])

    def compute_returns(self, next_value, use_gae, gamma, tau):   # :159-180
        self.returns, self.value_preds = O.compute_returns(self.rewards, self.value_preds, self.masks, next_value, use_gae, gamma, tau)

    def batches(self, advantages, num_mini_batch):
        """recurrent_generator (:182-312): torch.randperm on the CPU generator, envs gathered in that order, [T, n] -> [T*n]."""
        N = self.rewards.size(1)
        per = N // num_mini_batch
        perm = torch.randperm(N)
        for start in range(0, N, per):
            idx = perm[start:start + per]
            flat = lambda t: t[:, idx].reshape(t.size(0) * idx.numel(), *t.shape[2:])  # noqa: E731
            yield ({k: flat(v[:-1]) for k, v in self.observations.items()}, self.recurrent_hidden_states_pol[0][:, idx],
                   flat(self.pred_binSepMasks), flat(self.pred_mono), flat(self.prev_pred_monoFromMem[1:]), flat(self.value_preds[:-1]),
                   flat(self.returns[:-1]), flat(advantages), flat(self.actions), flat(self.action_log_probs), flat(self.masks[:-1]))


class SepStorage:
    """common/rollout_storage.py:315-471."""

    def __init__(self, T, N, obs_shapes):
        self.observations = {k: torch.zeros(T + 1, N, *s) for k, s in obs_shapes.items()}
        f, t = obs_shapes["gt_mono_comps"][:2]
        self.prev_pred_monoFromMem = torch.zeros(T + 1, N, f, t, 1)
        self.masks = torch.ones(T + 1, N, 1)
        self.num_steps, self.step = T, 0

    def insert(self, obs, masks, mem):   # :359-382
        s = self.step
        for k in obs:
            self.observations[k][s + 1].copy_(obs[k])
        self.prev_pred_monoFromMem[s + 1].copy_(mem)
        self.masks[s + 1].copy_(masks)
        self.step = (s + 1) % self.num_steps

    def after_update(self):   # :384-390
        for k in self.observations:
            self.observations[k][0].copy_(self.observations[k][-1])
        self.prev_pred_monoFromMem[0].copy_(self.prev_pred_monoFromMem[-1])
        self.masks[0].copy_(self.masks[-1])

    def batches(self, num_mini_batch):   # :392-455
        N = self.masks.size(1)
        per = N // num_mini_batch
        perm = torch.randperm(N)
        for start in range(0, N, per):
            idx = perm[start:start + per]
            flat = lambda t: t[:, idx].reshape(t.size(0) * idx.numel(), *t.shape[2:])  # noqa: E731
            yield ({k: flat(v[:-1]) for k, v in self.observations.items()}, flat(self.prev_pred_monoFromMem[1:]),
                   flat(self.prev_pred_monoFromMem[:-1]), flat(self.masks[:-1]))


class Rank:
    """What one DD-PPO worker owns: env, storages, statistics (ppo_trainer.py:643-697)."""

    def __init__(self, envs, cfg):
        self.envs = envs
        N = envs.num_envs
        shapes = {k: tuple(v.shape) for k, v in envs.observation_spaces[0].spaces.items()}
        self.ro = PolStorage(cfg["num_steps"], N, shapes, cfg["hidden_size"])
        self.rs = SepStorage(cfg["num_steps"] * cfg["num_updates_per_cycle"], N, shapes)
        batch = batch_obs(envs.reset())
        for k in self.ro.observations:
            self.ro.observations[k][0].copy_(batch[k])
            self.rs.observations[k][0].copy_(batch[k])
        self.stats = {n: torch.zeros(N, STAT_SHAPES.get(n, 1)) for n in STAT_NAMES}
        self.windows = {}
        self.last_act = None


def separate(sd, cfg, mix, target_class):
    """One pass of the separator pair as the trainer's call sites make it (ppo_trainer.py:297-304, :358-366; ppo.py:184-195), under
    no_grad with eval-mode BatchNorm -- whatever RL.PPO.train_passive_separators says: train() always calls
    _load_pretrained_passive_separators (:637-638), which puts the four separator modules in eval mode and clears requires_grad
    unconditionally (:557-577); the freeze_passive_separators flag derived from the key (:72-73) is stored by PPO.__init__
    (ppo.py:46) and read nowhere.  Pinned by tests/golden/trainer_unfrozen.npz (the reference run with the key set to True)."""
    pm = O.get_binSepMasks(sd, mix, target_class)
    return pm, O.convert_bin2mono(sd, pm, mix)


def draw_actions(probs, noise=None):
    """CustomFixedCategorical.sample (common/utils.py:16-24): torch.multinomial(probs, 1, True).  Its single-draw path is
    argmax(probs / q), q ~ Exp(1) from the tensor's generator (pinned against the reference's own sample() by tests/golden/rl_forward.npz sample_*, tests/test_oracle_rl_golden.py); with
    ``noise`` [N, A] given, q is that tensor instead of the generator's draw -- "same probs + same noise => same actions"."""
    if noise is None:
        return torch.multinomial(probs, 1, True)
    return (probs / noise).argmax(dim=1, keepdim=True)


def collect_rollout_step(sd, cfg, rk, forced_actions=None, action_noise=None):
    """ppo_trainer.py:253-478 for one rank.  forced_actions: [N,1] int64 to take instead of sampling (the action's log-prob
    is still this policy's).  action_noise: [N, A] Exp(1) noise for the draw in place of the generator's (draw_actions)."""
    ro, rs, st = rk.ro, rk.rs, rk.stats
    with torch.no_grad():
        obs = {k: v[ro.step] for k, v in ro.observations.items()}                                       # :292-294
        pm, mono = separate(sd, cfg, obs["mixed_bin_audio_mag"], obs["target_class"])                   # :297-304
        mem = O.acoustic_mem(sd, mono, O.mask_prev_mem(ro.prev_pred_monoFromMem[ro.step], ro.masks[ro.step]))   # :307-318
        feats, h, _ = O.policy_net(sd, obs, ro.recurrent_hidden_states_pol[ro.step], ro.masks[ro.step], pm, mono, mem)
        values, logp_all, probs = O.heads(sd, feats)                                                    # :321-335 (Policy.act)
        actions = forced_actions if forced_actions is not None else draw_actions(probs, action_noise)
        logp = logp_all.gather(1, actions)
    rk.last_act = (values, actions, logp, h, probs)
    outputs = rk.envs.step([a[0].item() for a in actions])                                              # :340
    observations, rewards, dones, infos = [list(x) for x in zip(*outputs)]
    batch = batch_obs(observations)
    masks = torch.tensor([[0.0] if d else [1.0] for d in dones])
    ndgs = torch.tensor([[i[INFO_KEYS[0]]] for i in infos])
    dgs = torch.tensor([[i[INFO_KEYS[1]]] for i in infos])
    with torch.no_grad():                                                                               # :357-373
        npm, nmono = separate(sd, cfg, batch["mixed_bin_audio_mag"], batch["target_class"])
        nmem = O.acoustic_mem(sd, nmono, O.mask_prev_mem(mem, masks))
    gt_mono = obs["gt_mono_comps"][..., 0::2][..., :1]
    ngt_mono = batch["gt_mono_comps"][..., 0::2][..., :1]
    if cfg["sep_reward_weight"] == 1.0 and cfg["nav_reward_weight"] == 0.0:                             # :385-405
        rewards = O.override_rewards(rewards, dones, nmem, ngt_mono, "quality_improvement", mem, gt_mono)
        if st["current_episode_step"][0].item() == cfg["MAX_EPISODE_STEPS"] - 2:
            # the reference's override_rewards writes into the list it is given and returns it (env_utils.py:692-706), so the
            # "extra" call overwrites the quality-improvement rewards and the sum of the two lists is TWICE the extra reward
            extra = O.override_rewards(rewards, dones, nmem, ngt_mono, "extra", extra_reward_multiplier=cfg["extra_reward_multiplier"])
            rewards = (np.array(extra) + np.array(extra)).tolist()
    _, mem_losses = O.stft_l2_distance(obs["mixed_bin_audio_mag"], pm, obs["gt_bin_comps"], mem, obs["gt_mono_comps"])   # :407-420
    bin_losses, mono_losses = O.stft_l2_distance(obs["mixed_bin_audio_mag"], pm, obs["gt_bin_comps"], mono, obs["gt_mono_comps"])
    rewards = torch.tensor(rewards, dtype=torch.float).unsqueeze(1)
    st["current_episode_reward"] += rewards                                                             # :426-455
    st["current_episode_step"] += 1
    st["current_episode_dist_probs"] += probs
    st["current_episode_bin_losses"] += bin_losses
    st["current_episode_mono_losses"] += mono_losses
    st["current_episode_monoFromMem_losses"] += mem_losses
    nd = 1 - masks
    st["episode_rewards"] += nd * st["current_episode_reward"]
    st["episode_ndgs"] += nd * ndgs
    st["episode_dgs"] += nd * dgs
    st["episode_steps"] += nd * st["current_episode_step"]
    st["episode_counts"] += nd
    st["episode_dist_probs"] += nd * (st["current_episode_dist_probs"] / st["current_episode_step"])
    st["episode_bin_losses_allSteps"] += nd * (st["current_episode_bin_losses"] / st["current_episode_step"])
    st["episode_mono_losses_lastStep"] += nd * mono_losses
    st["episode_mono_losses_allSteps"] += nd * (st["current_episode_mono_losses"] / st["current_episode_step"])
    st["episode_monoFromMem_losses_lastStep"] += nd * mem_losses
    st["episode_monoFromMem_losses_allSteps"] += nd * (st["current_episode_monoFromMem_losses"] / st["current_episode_step"])
    for n in STAT_NAMES[:6]:
        st[n] *= masks
    ro.insert(batch, h, actions, logp, values, rewards, masks, pm, mono, mem)                            # :457-474
    rs.insert(batch, masks, mem)


def _trainable(sd, prefixes):
    return [v for k, v in sd.items() if k.startswith(prefixes)]


def update_pol(sd, opt, ranks, cfg, clip_param, distributed):
    """PPO.update_pol (ppo.py:82-177) on every rank with DDP's gradient averaging; returns per-rank (value, action, entropy) means."""
    R = len(ranks)
    raw = [rk.ro.returns[:-1] - rk.ro.value_preds[:-1] for rk in ranks]
    if distributed:
        advs = O.get_advantages_distributed(raw)                                                         # :275-284
    else:
        advs = [(a - a.mean()) / (a.std() + O.EPS_PPO) for a in raw]                                     # :75-80
    params = _trainable(sd, POL_PREFIXES)
    acc = np.zeros((R, 3))
    for _e in range(cfg["ppo_epoch"]):
        gens = [rk.ro.batches(adv, cfg["num_mini_batch"]) for rk, adv in zip(ranks, advs)]
        for samples in zip(*gens):
            opt.zero_grad()
            for r, (obs_b, h_b, pm_b, mono_b, mem_b, vp_b, ret_b, adv_b, act_b, olp_b, m_b) in enumerate(samples):
                values, logp, ent, _ = O.evaluate_actions(sd, obs_b, h_b, m_b, act_b, pm_b, mono_b, mem_b)
                v_loss, a_loss, total = O.ppo_losses(values, logp, ent, vp_b, ret_b, adv_b, olp_b, clip_param, cfg["value_loss_coef"],
                                                     cfg["entropy_coef"])
                (total / R).backward()                                                                    # DDP: mean over ranks
                acc[r] += [v_loss.item(), a_loss.item(), ent.item()]
            torch.nn.utils.clip_grad_norm_(params, cfg["max_grad_norm"])                                 # before_step_pol
            opt.step()
    return acc / (cfg["ppo_epoch"] * cfg["num_mini_batch"])


def update_sep(sd, opt, ranks, cfg):
    """PPO.update_sep (ppo.py:179-246): only the acoustic memory's L1 loss is back-propagated (:226)."""
    R = len(ranks)
    params = _trainable(sd, MEM_PREFIXES)
    acc = np.zeros((R, 3))
    for _e in range(cfg["ppo_epoch"]):
        gens = [rk.rs.batches(cfg["num_mini_batch"]) for rk in ranks]
        for samples in zip(*gens):
            opt.zero_grad()
            for r, (obs_b, _mem_b, prev_b, m_b) in enumerate(samples):
                with torch.no_grad():
                    pm, mono = separate(sd, cfg, obs_b["mixed_bin_audio_mag"], obs_b["target_class"])
                mem = O.acoustic_mem(sd, mono, O.mask_prev_mem(prev_b, m_b))
                gt_bin, gt_mono = O.gt_mags(obs_b)
                mem_loss = F.l1_loss(mem, gt_mono)
                mono_loss = F.l1_loss(mono, gt_mono)
                bin_loss = F.l1_loss((torch.exp(obs_b["mixed_bin_audio_mag"]) - 1) * pm, gt_bin)
                (mem_loss / R).backward()
                acc[r] += [bin_loss.item(), mono_loss.item(), mem_loss.item()]
            torch.nn.utils.clip_grad_norm_(params, cfg["max_grad_norm"])                                 # before_step_sep
            opt.step()
    return acc / (cfg["ppo_epoch"] * cfg["num_mini_batch"])


WINDOW_KEYS = (("count", "episode_counts"), ("reward", "episode_rewards"), ("step", "episode_steps"), ("dist_probs", "episode_dist_probs"),
               ("avg_bin_loss_allSteps", "episode_bin_losses_allSteps"), ("mono_loss_lastStep", "episode_mono_losses_lastStep"),
               ("mono_loss_allSteps", "episode_mono_losses_allSteps"), ("monoFromMem_loss_lastStep", "episode_monoFromMem_losses_lastStep"),
               ("monoFromMem_loss_allSteps", "episode_monoFromMem_losses_allSteps"), (INFO_KEYS[0], "episode_ndgs"), (INFO_KEYS[1], "episode_dgs"))
SCALAR_TAGS = (("Environment/Reward", "reward"), ("Environment/Episode_length", "step"),
               ("Environment/STFT_L2_loss/mono_lastStep", "mono_loss_lastStep"), ("Environment/STFT_L2_loss/mono_avgAllSteps", "mono_loss_allSteps"),
               ("Environment/STFT_L2_loss/monoFromMem_lastStep", "monoFromMem_loss_lastStep"),
               ("Environment/STFT_L2_loss/monoFromMem_avgAllSteps", "monoFromMem_loss_allSteps"),
               ("Environment/Normalized_geo_distance_to_target_audio_source", INFO_KEYS[0]),
               ("Environment/Geo_distance_to_target_audio_source", INFO_KEYS[1]))


def window_scalars(windows, summed_stats, window_size):
    """The window-of-N statistics the reference logs after every policy update (ppo_trainer.py:826-960): every statistic,
    summed over ranks, is pushed into a deque; the logged value is (newest - oldest) summed over envs, divided by the episode
    count of the window (at least 1)."""
    deltas = {}
    for key, name in WINDOW_KEYS:
        w = windows.setdefault(key, deque(maxlen=window_size))
        w.append(summed_stats[name].clone())
        d = (w[-1] - w[0]) if len(w) > 1 else w[0]
        deltas[key] = d.sum(dim=0) if key == "dist_probs" else d.sum().item()
    deltas["count"] = max(deltas["count"], 1.0)
    out = {tag: deltas[key] / deltas["count"] for tag, key in SCALAR_TAGS}
    for i in range(3):
        out["Policy/Action_prob_%d" % i] = (deltas["dist_probs"] / deltas["count"])[i].item()
    return out


def train(cfg, envs_per_rank, sd, forced_actions=None, distributed=True, on_step=None, action_noise=None):
    """PPOTrainer.train (ppo_trainer.py:579-1013) for len(envs_per_rank) emulated ranks.
    cfg: dict with the RL.PPO keys + NUM_UPDATES, CHECKPOINT_INTERVAL, MAX_EPISODE_STEPS.
    sd: full policy state dict (reference keys, no "actor_critic." root); trainable tensors are replaced by leaf copies.
    forced_actions: optional [rank][global step] -> [N,1] int64.  action_noise: optional [rank][global step] -> [N, A] Exp(1) noise the
    step's draw uses in place of the generator's (the CPU generator then only serves the epochs' randperm).
    Returns a record dict shaped like the golden fixtures."""
    # the reference steps its LR schedulers at the START of each sub-update (:733-735, :981-982); torch warns about that order
    warnings.filterwarnings("ignore", message="Detected call of `lr_scheduler.step\\(\\)` before `optimizer.step\\(\\)`")
    sd = {k: v.clone() for k, v in sd.items()}
    for k in sd:
        if k.startswith(POL_PREFIXES + MEM_PREFIXES):
            sd[k].requires_grad_(True)
    opt_pol = torch.optim.Adam(_trainable(sd, POL_PREFIXES), lr=cfg["lr_pol"], eps=cfg["eps"])           # ppo.py:48-55
    opt_sep = torch.optim.Adam(_trainable(sd, MEM_PREFIXES), lr=cfg["lr_sep"], eps=cfg["eps"])
    decay = lambda x: 1 - x / float(cfg["NUM_UPDATES"])                                                  # noqa: E731  common/utils.py:53-63
    sched_pol = torch.optim.lr_scheduler.LambdaLR(opt_pol, lr_lambda=decay)                              # :711-718
    sched_sep = torch.optim.lr_scheduler.LambdaLR(opt_sep, lr_lambda=decay)
    ranks = [Rank(e, cfg) for e in envs_per_rank]
    R = len(ranks)
    rec = {"steps": [[] for _ in ranks], "pol": [], "sep": [], "scalars": [], "ckpts": []}
    count_steps, count_ckpt, k = 0, 0, 0
    windows = {}
    for update in range(int(cfg["NUM_UPDATES"] / cfg["num_updates_per_cycle"])):                         # :730
        for sub in range(cfg["num_updates_per_cycle"]):
            actual = update * cfg["num_updates_per_cycle"] + sub
            if cfg["use_linear_lr_decay"]:
                sched_pol.step()                                                                         # :733-735
            clip = cfg["clip_param"] * decay(actual) if cfg["use_linear_clip_decay"] else cfg["clip_param"]   # :736-739
            for _step in range(cfg["num_steps"]):
                for r, rk in enumerate(ranks):
                    collect_rollout_step(sd, cfg, rk, None if forced_actions is None else forced_actions[r][k],
                                         None if action_noise is None else action_noise[r][k])
                    if on_step is not None:
                        on_step(r, k, rk)
                    count_steps += rk.envs.num_envs
                k += 1
            lr = opt_pol.param_groups[0]["lr"]
            with torch.no_grad():                                                                        # _update_pol :480-520
                for rk in ranks:
                    ro = rk.ro
                    last = {kk: v[-1] for kk, v in ro.observations.items()}
                    feats, _, _ = O.policy_net(sd, last, ro.recurrent_hidden_states_pol[-1], ro.masks[-1], ro.pred_binSepMasks[-1],
                                               ro.pred_mono[-1], ro.prev_pred_monoFromMem[-1])
                    ro.compute_returns(O.heads(sd, feats)[0], cfg["use_gae"], cfg["gamma"], cfg["tau"])
            losses = update_pol(sd, opt_pol, ranks, cfg, clip, distributed)
            rec["pol"].append({"losses": losses, "lr": lr, "clip": clip, "returns": [rk.ro.returns.clone() for rk in ranks]})
            for rk in ranks:
                rk.ro.after_update()
            summed = {n: sum(rk.stats[n] for rk in ranks) for n in STAT_NAMES}                           # :790-838 (all_reduce = sum)
            sc = window_scalars(windows, summed, cfg["reward_window_size"])
            sc.update({"Policy/Value_Loss": losses[:, 0].mean(), "Policy/Action_Loss": losses[:, 1].mean(), "Policy/Entropy": losses[:, 2].mean(),
                       "Policy/Learning_Rate": sched_pol.get_last_lr()[0]})
            rec["scalars"].append((count_steps, sc))
        for sub in range(cfg["num_updates_per_cycle"]):                                                  # :979-1009
            actual = update * cfg["num_updates_per_cycle"] + sub
            if cfg["use_linear_lr_decay"]:
                sched_sep.step()
            lr = opt_sep.param_groups[0]["lr"]
            losses = update_sep(sd, opt_sep, ranks, cfg)
            for rk in ranks:
                rk.rs.after_update()
            rec["sep"].append({"losses": losses, "lr": lr})
            if actual % cfg["CHECKPOINT_INTERVAL"] == 0:
                rec["ckpts"].append(("ckpt.%d.pth" % count_ckpt, len(rec["sep"])))
                count_ckpt += 1
    rec["state_dict"] = {kk: v.detach() for kk, v in sd.items()}
    rec["ranks"] = ranks
    return rec


def eval_loop(cfg, envs, sds, num_episodes, deterministic, switch_thres=None, forced_actions=None):
    """PPOTrainer._eval_checkpoint (ppo_trainer.py:1015-1551) for ONE process (the reference asserts NUM_PROCESSES == 1, :1052).
    sds: [state dict] for one policy, or [navigation, quality-improvement] with ``switch_thres`` = RL.PPO.time_thres_for_pol_switch
    (:1231-1312): the navigation policy acts for the first ``switch_thres`` steps of an episode, the quality-improvement policy
    afterwards; the previous memory is masked with the NAVIGATION policy's not-done flags in both phases (:1276-1281) and the
    quality-improvement policy's flags follow the env only while it acts (:1348-1360).
    Returns per-step (actions, mono, monoFromMem STFT-L2) and the four aggregates the reference logs (:1484-1504)."""
    N = envs.num_envs
    assert N == 1
    batch = batch_obs(envs.reset())
    hs = [torch.zeros(1, N, cfg["hidden_size"]) for _ in sds]
    nd = [torch.ones(N, 1) for _ in sds]
    prev_mem = torch.zeros(N, 512, 32, 1)
    steps, step_count, done_eps = [], 0, 0
    last, allsteps = {"mono": [], "mem": []}, {"mono": [], "mem": []}
    acc = {"mono": 0.0, "mem": 0.0}
    while done_eps < num_episodes:
        w = 0 if (switch_thres is None or step_count < switch_thres) else 1
        sd = sds[w]
        with torch.no_grad():
            pm = O.get_binSepMasks(sd, batch["mixed_bin_audio_mag"], batch["target_class"])
            mono = O.convert_bin2mono(sd, pm, batch["mixed_bin_audio_mag"])
            mem = O.acoustic_mem(sd, mono, O.mask_prev_mem(prev_mem, nd[0]))
            feats, h, _ = O.policy_net(sd, batch, hs[w], nd[w], pm, mono, mem)
            _v, _lp_all, probs = O.heads(sd, feats)
            if forced_actions is not None:
                actions = forced_actions[len(steps)]
            else:
                actions = probs.argmax(dim=-1, keepdim=True) if deterministic else torch.multinomial(probs, 1, True)
            hs[w] = h
        prev_mem = mem
        outputs = envs.step([a[0].item() for a in actions])
        observations, _rewards, dones, _infos = [list(x) for x in zip(*outputs)]
        masks = torch.tensor([[0.0] if d else [1.0] for d in dones])
        nd[0] = masks
        if w == 1:
            nd[1] = masks
        _b, d_mem = O.stft_l2_distance(batch["mixed_bin_audio_mag"], pm, batch["gt_bin_comps"], mem, batch["gt_mono_comps"])
        _b, d_mono = O.stft_l2_distance(batch["mixed_bin_audio_mag"], pm, batch["gt_bin_comps"], mono, batch["gt_mono_comps"])
        acc["mono"] += d_mono[0][0].item()
        acc["mem"] += d_mem[0][0].item()
        steps.append((actions.clone(), d_mono[0][0].item(), d_mem[0][0].item()))
        batch = batch_obs(observations)
        step_count += 1
        if dones[0]:
            last["mono"].append(d_mono[0][0].item())
            last["mem"].append(d_mem[0][0].item())
            allsteps["mono"].append(acc["mono"] / step_count)
            allsteps["mem"].append(acc["mem"] / step_count)
            acc = {"mono": 0.0, "mem": 0.0}
            done_eps += 1
            step_count = 0
    agg = {"mono_loss_last_step": last["mono"], "mono_loss_all_steps": allsteps["mono"], "monoFromMem_loss_last_step": last["mem"],
           "monoFromMem_loss_all_steps": allsteps["mem"]}
    return steps, {k: (float(np.mean(v)), float(np.std(v))) for k, v in agg.items()}
