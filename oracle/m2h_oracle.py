"""CPU oracle for the Move2Hear hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-CPU fp32 restatement of the reference's algorithm, function by function, each
citing the reference file:line it follows (paths relative to the reference root).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this module, and
only as the checker / the reported CPU baseline; the product path (``m2h``) never routes through it.

Pinning: the reference holds no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, produced in the build container by
``oracle/gen_golden.py`` (which imports the reference through ``oracle/_ref_import.py``) and
committed under ``tests/golden/``; ``tests/test_oracle_golden.py``, ``test_oracle_rl_golden.py``,
``test_oracle_passive_train.py`` and ``test_oracle_eval_metrics.py`` check every function here against them
(the training loop on top of these functions: ``m2h_oracle_trainer.py`` / ``test_oracle_trainer_golden.py``).
Unpinned: the librosa-based STFT / iSTFT restatements (``np_stft``, ``np_istft``, ``np_compute_audiospects``):
librosa is not importable here and the reference holds no fixtures for them.

All functions take a flat ``state_dict`` (name -> torch.Tensor, reference key names without the
``actor_critic.`` root) and plain tensors; no nn.Module, so nothing here can be mistaken for the
product's modules.
"""
import math

import torch
import torch.nn.functional as F

SLICE = 16  # separator_cnn.py:43, memory_nets.py:8, audio_cnn.py:24
BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, used by separator_cnn.py:9,21


# ----------------------------------------------------------------------------------------------
# layout helpers
# ----------------------------------------------------------------------------------------------
def slice_freq(x_bhwc):
    """[B,F,T,C] -> [B,C*16,F/16,T]   (separator_cnn.py:85-90; memory_nets.py:41-59; audio_cnn.py:129-133)"""
    x = x_bhwc.permute(0, 3, 1, 2)
    b, c, f, t = x.shape
    x = x.reshape(b, c, SLICE, f // SLICE, t)
    return x.reshape(b, c * SLICE, f // SLICE, t)


def deslice_freq(x_bchw):
    """[B,c*16,H,T] -> [B,16*H,T,c]   (separator_cnn.py:163-168; memory_nets.py:62-67)"""
    b, cs, h, t = x_bchw.shape
    x = x_bchw.reshape(b, cs // SLICE, SLICE, h, t)
    x = x.reshape(b, cs // SLICE, SLICE * h, t)
    return x.permute(0, 2, 3, 1)


def _bn_eval(x, sd, pre):
    """BatchNorm2d in eval mode (running statistics)."""
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"],
                        sd[pre + "bias"], training=False, momentum=0.1, eps=BN_EPS)


def _bn_train(x, sd, pre, stats_out=None):
    """BatchNorm2d in train mode: batch statistics (biased var) for normalisation; the running-stat
    update (momentum 0.1, unbiased var) is returned through ``stats_out`` instead of mutating sd."""
    mean = x.mean(dim=(0, 2, 3))
    var_b = x.var(dim=(0, 2, 3), unbiased=False)
    y = (x - mean[None, :, None, None]) / torch.sqrt(var_b[None, :, None, None] + BN_EPS)
    y = y * sd[pre + "weight"][None, :, None, None] + sd[pre + "bias"][None, :, None, None]
    if stats_out is not None:
        n = x.numel() / x.size(1)
        var_u = var_b * (n / max(n - 1.0, 1.0))
        stats_out[pre + "running_mean"] = 0.9 * sd[pre + "running_mean"] + 0.1 * mean
        stats_out[pre + "running_var"] = 0.9 * sd[pre + "running_var"] + 0.1 * var_u
    return y


# ----------------------------------------------------------------------------------------------
# A1: PassiveSepEncCNN.forward  (separator_cnn.py:70-108)
# ----------------------------------------------------------------------------------------------
def sep_enc_input(mixed_bin_audio_mag, target_class=None, pred_masks=None):
    """Builds the conv-stack input.  binSep: slice(mix) ++ (target_class+1) plane (:82-99);
    bin2mono: slice(log1p(clamp0(mask * (exp(mix)-1)))) (:73-79)."""
    if pred_masks is not None:
        x = pred_masks * (torch.exp(mixed_bin_audio_mag) - 1)
        x = torch.log1p(torch.clamp(x, min=0))
        return slice_freq(x)
    x = slice_freq(mixed_bin_audio_mag)
    tc = target_class.reshape(-1, 1, 1, 1).float() + 1  # :96
    plane = tc.expand(x.size(0), 1, x.size(2), x.size(3))
    return torch.cat((x, plane), dim=1)


def sep_enc_stack(x, sd, pre, train_bn=False, stats_out=None):
    """5x {conv4x4 s2 p1 no-bias -> BN -> LeakyReLU(0.2)}  (separator_cnn.py:5-12,46-52,101-105).
    Returns the list of the 5 stage outputs (NCHW)."""
    feats = []
    out = x
    for i in range(5):
        out = F.conv2d(out, sd[pre + "%d.0.weight" % i], None, stride=2, padding=1)
        out = _bn_train(out, sd, pre + "%d.1." % i, stats_out) if train_bn else _bn_eval(out, sd, pre + "%d.1." % i)
        out = F.leaky_relu(out, 0.2)
        feats.append(out)
    return feats


# ----------------------------------------------------------------------------------------------
# A2: PassiveSepDecCNN.forward  (separator_cnn.py:153-170), fully-convolutional generalisation
# ----------------------------------------------------------------------------------------------
def sep_dec_stack(feats, sd, pre, train_bn=False, stats_out=None):
    """5x {convT4x4 s2 p1 no-bias -> BN -> ReLU} with skip concat on stages 1..4 (:156-161) and the
    biased 1x1 conv (:134).  ``feats`` = encoder stage outputs [e1..e5]; the bottleneck is e5 kept as
    [B,512,1,Tm/32] (the reference's ``view(B,-1,1,1)`` at :154 is the Tm=32 special case).
    Returns BHWC [B,512,Tm,c]."""
    out = feats[4]
    skips = feats[:4][::-1]  # separator_cnn.py:108
    for i in range(5):
        if i > 0:
            out = torch.cat((out, skips[i - 1]), dim=1)
        out = F.conv_transpose2d(out, sd[pre + "%d.0.weight" % i], None, stride=2, padding=1)
        out = _bn_train(out, sd, pre + "%d.1." % i, stats_out) if train_bn else _bn_eval(out, sd, pre + "%d.1." % i)
        out = F.relu(out)
    out = F.conv2d(out, sd[pre + "5.0.weight"], sd[pre + "5.0.bias"])
    return deslice_freq(out)


# ----------------------------------------------------------------------------------------------
# A3: Policy.get_binSepMasks / convert_bin2mono  (pretrain/passive/policy.py:61-71; rl/ppo/policy.py:183-193)
# ----------------------------------------------------------------------------------------------
ENC_B = "binSep_enc.passive_sep_encoder.cnn."
DEC_B = "binSep_dec.passive_sep_decoder.cnn."
ENC_M = "bin2mono_enc.passive_sep_encoder.cnn."
DEC_M = "bin2mono_dec.passive_sep_decoder.cnn."


def get_binSepMasks(sd, mixed_bin_audio_mag, target_class, train_bn=False, stats_out=None, return_feats=False):
    x = sep_enc_input(mixed_bin_audio_mag, target_class=target_class)
    feats = sep_enc_stack(x, sd, ENC_B, train_bn, stats_out)
    out = sep_dec_stack(feats, sd, DEC_B, train_bn, stats_out)
    return (out, feats) if return_feats else out


def convert_bin2mono(sd, pred_masks, mixed_bin_audio_mag, train_bn=False, stats_out=None, return_feats=False):
    x = sep_enc_input(mixed_bin_audio_mag, pred_masks=pred_masks)
    feats = sep_enc_stack(x, sd, ENC_M, train_bn, stats_out)
    out = sep_dec_stack(feats, sd, DEC_M, train_bn, stats_out)
    return (out, feats) if return_feats else out


def passive_pair(sd, mixed_bin_audio_mag, target_class):
    """The headline unit of work: one spectrogram through both U-Nets (eval-mode BN)."""
    masks = get_binSepMasks(sd, mixed_bin_audio_mag, target_class)
    mono = convert_bin2mono(sd, masks, mixed_bin_audio_mag)
    return masks, mono


def rel_l1(a, b):
    """Parity metric of SURVEY.md section 8d: sum|a-b| / sum|b|."""
    return float((a.double() - b.double()).abs().sum() / b.double().abs().sum().clamp_min(1e-30))


def pred_bin(masks, mixed_bin_audio_mag):
    """mask * (exp(mix) - 1): the separated binaural magnitude (passive_trainer.py:271-272)."""
    return masks * (torch.exp(mixed_bin_audio_mag) - 1)


# ==============================================================================================
# RL path (forward).  State-dict keys are the reference's, without the "actor_critic." root.
# ==============================================================================================
MEM = "acoustic_mem.cnn."
VIS = "pol_net.visual_encoder.cnn."
BIN = "pol_net.bin_encoder.cnn."
MNM = "pol_net.monoNmonoFromMem_encoder.cnn."
GRU = "pol_net.state_encoder.rnn."


# A5: AcousticMem.forward (rl/models/memory_nets.py:40-69), ddppo variant (no BN, :11-16)
def acoustic_mem(sd, pred_mono, prev_pred_monoFromMem_masked):
    x = torch.cat((slice_freq(pred_mono), slice_freq(prev_pred_monoFromMem_masked)), dim=1)
    x = F.relu(F.conv2d(x, sd[MEM + "0.weight"], None, padding=1))
    x = F.conv2d(x, sd[MEM + "2.weight"], None, padding=1)
    return deslice_freq(x)


def mask_prev_mem(prev_pred_monoFromMem, masks):
    """prev * masks.unsqueeze(1).unsqueeze(2).repeat(...)  (ppo_trainer.py:310-314; ppo.py:206-209)"""
    return prev_pred_monoFromMem * masks.reshape(-1, 1, 1, 1)


# A6: VisualCNN.forward (rl/models/visual_cnn.py:135-152): /255, NHWC->NCHW, conv8x8s4+ReLU, conv4x4s2+ReLU,
# conv3x3s1 (no ReLU), flatten (NCHW order), FC+ReLU
def visual_cnn(sd, rgb, depth=None):
    x = rgb.permute(0, 3, 1, 2) / 255.0
    if depth is not None:
        x = torch.cat((x, depth.permute(0, 3, 1, 2)), dim=1)
    x = F.relu(F.conv2d(x, sd[VIS + "0.weight"], sd[VIS + "0.bias"], stride=4))
    x = F.relu(F.conv2d(x, sd[VIS + "2.weight"], sd[VIS + "2.bias"], stride=2))
    x = F.conv2d(x, sd[VIS + "4.weight"], sd[VIS + "4.bias"], stride=1)
    x = x.reshape(x.size(0), -1)
    return F.relu(F.linear(x, sd[VIS + "6.weight"], sd[VIS + "6.bias"]))


# A7: AudioCNN.forward (rl/models/audio_cnn.py:117-140)
def audio_cnn(sd, pre, mixed_bin_audio_mag=None, pred_binSepMasks=None, pred_monoNmonoFromMem=None):
    if pred_monoNmonoFromMem is not None:
        x = torch.log1p(torch.clamp(pred_monoNmonoFromMem, min=0))  # :121-122
    else:
        x = (torch.exp(mixed_bin_audio_mag) - 1) * pred_binSepMasks  # :125-127
        x = torch.log1p(torch.clamp(x, min=0))
    x = slice_freq(x)
    x = F.relu(F.conv2d(x, sd[pre + "0.weight"], sd[pre + "0.bias"], stride=4))
    x = F.relu(F.conv2d(x, sd[pre + "2.weight"], sd[pre + "2.bias"], stride=2))
    x = F.relu(F.conv2d(x, sd[pre + "4.weight"], sd[pre + "4.bias"], stride=1))
    x = x.reshape(x.size(0), -1)
    return F.relu(F.linear(x, sd[pre + "7.weight"], sd[pre + "7.bias"]))


# A8: RNNStateEncoder (rl/models/rnn_state_encoder.py:74-143), GRU(1536 -> 512), gate order r,z,n
def gru_cell(sd, x, h):
    gi = F.linear(x, sd[GRU + "weight_ih_l0"], sd[GRU + "bias_ih_l0"])
    gh = F.linear(h, sd[GRU + "weight_hh_l0"], sd[GRU + "bias_hh_l0"])
    H = h.size(1)
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def rnn_forward(sd, x, hidden_states, masks):
    """single_forward when x.size(0) == N (:74-84), else seq_forward (:86-137).  seq_forward's segment
    trick (run the GRU over stretches without resets, mask h at every stretch start) equals masking h
    with masks[t] before EVERY step only when masks are 1 inside stretches -- which is the definition of a
    stretch -- except at t=0 of each segment where the reference applies masks[start]; identical."""
    n = hidden_states.size(1)
    h = hidden_states[0]
    if x.size(0) == n:
        h = gru_cell(sd, x, h * masks)
        return h, h.unsqueeze(0)
    t = x.size(0) // n
    xs = x.reshape(t, n, -1)
    ms = masks.reshape(t, n, 1)
    outs = []
    for i in range(t):
        h = gru_cell(sd, xs[i], h * ms[i])
        outs.append(h)
    return torch.cat(outs, 0), h.unsqueeze(0)


# PolicyNet.forward (rl/ppo/policy.py:98-118)
def policy_net(sd, obs, hidden_states, masks, pred_binSepMasks, pred_mono, pred_monoFromMem, use_depth=False):
    feats = [
        visual_cnn(sd, obs["rgb"], obs["depth"] if use_depth else None),
        audio_cnn(sd, BIN, mixed_bin_audio_mag=obs["mixed_bin_audio_mag"], pred_binSepMasks=pred_binSepMasks),
        audio_cnn(sd, MNM, pred_monoNmonoFromMem=torch.cat((pred_mono, pred_monoFromMem), dim=3)),
    ]
    x = torch.cat(feats, dim=1)
    return rnn_forward(sd, x, hidden_states, masks) + (x,)


# A9: CategoricalNet / CustomFixedCategorical / CriticHead (common/utils.py:16-50; rl/ppo/policy.py:15-23)
def heads(sd, feats):
    logits = F.linear(feats, sd["action_dist.linear.weight"], sd["action_dist.linear.bias"])
    value = F.linear(feats, sd["critic.fc.weight"], sd["critic.fc.bias"])
    # torch.distributions.Categorical(logits=x): logits normalised by logsumexp, probs = softmax
    logp_all = logits - logits.logsumexp(dim=-1, keepdim=True)
    probs = F.softmax(logits, dim=-1)
    return value, logp_all, probs


def categorical_entropy(logp_all, probs):
    """Categorical.entropy(): -(clamped logits * probs).sum(-1)"""
    min_real = torch.finfo(logp_all.dtype).min
    return -(torch.clamp(logp_all, min=min_real) * probs).sum(-1)


def act(sd, obs, hidden_states, masks, pred_binSepMasks, pred_mono, pred_monoFromMem, deterministic=False, generator=None):
    """Policy.act (rl/ppo/policy.py:198-225).  sample() = torch.multinomial(probs, 1, True) on the probs tensor."""
    feats, h, _ = policy_net(sd, obs, hidden_states, masks, pred_binSepMasks, pred_mono, pred_monoFromMem)
    value, logp_all, probs = heads(sd, feats)
    if deterministic:
        action = probs.argmax(dim=-1, keepdim=True)
    else:
        action = torch.multinomial(probs, 1, True, generator=generator)
    logp = logp_all.gather(1, action)
    return value, action, logp, h, probs


def evaluate_actions(sd, obs, hidden_states, masks, action, pred_binSepMasks, pred_mono, pred_monoFromMem):
    """Policy.evaluate_actions (rl/ppo/policy.py:248-273)."""
    feats, h, _ = policy_net(sd, obs, hidden_states, masks, pred_binSepMasks, pred_mono, pred_monoFromMem)
    value, logp_all, probs = heads(sd, feats)
    logp = logp_all.gather(1, action)
    return value, logp, categorical_entropy(logp_all, probs).mean(), h


# A10: RolloutStoragePol.compute_returns (common/rollout_storage.py:155-180)
def compute_returns(rewards, value_preds, masks, next_value, use_gae, gamma, tau):
    """rewards [T,N,1], value_preds [T+1,N,1] (last row overwritten by next_value when use_gae), masks [T+1,N,1].
    Returns (returns [T+1,N,1], value_preds)."""
    T = rewards.size(0)
    value_preds = value_preds.clone()
    returns = torch.zeros_like(value_preds)
    if use_gae:
        value_preds[-1] = next_value
        gae = 0
        for step in reversed(range(T)):
            delta = rewards[step] + gamma * value_preds[step + 1] * masks[step + 1] - value_preds[step]
            gae = delta + gamma * tau * masks[step + 1] * gae
            returns[step] = gae + value_preds[step]
    else:
        returns[-1] = next_value
        for step in reversed(range(T)):
            returns[step] = returns[step + 1] * gamma * masks[step + 1] + rewards[step]
    return returns, value_preds


# A14: PPO.get_advantages (rl/ppo/ppo.py:75-80) and the distributed variant (:275-284; ddppo_utils.py:168-190)
EPS_PPO = 1e-5


def get_advantages(returns, value_preds, normalized=True):
    adv = returns[:-1] - value_preds[:-1]
    if not normalized:
        return adv
    return (adv - adv.mean()) / (adv.std() + EPS_PPO)


def get_advantages_distributed(per_rank_adv):
    """Emulates _get_advantages_distributed over a list of per-rank advantage tensors: global mean, then the
    mean over ranks of per-rank mean((A - mean)^2) (biased), (A - mean) / (sqrt(var) + eps)."""
    w = len(per_rank_adv)
    mean = sum(a.mean() for a in per_rank_adv) / w
    var = sum((a - mean).pow(2).mean() for a in per_rank_adv) / w
    return [(a - mean) / (var.sqrt() + EPS_PPO) for a in per_rank_adv]


# K15: PPO losses (rl/ppo/ppo.py:125-157)
def ppo_losses(values, action_log_probs, dist_entropy, value_preds_batch, return_batch, adv_targ, old_action_log_probs,
               clip_param, value_loss_coef, entropy_coef, use_clipped_value_loss=True):
    ratio = torch.exp(action_log_probs - old_action_log_probs)
    surr1 = ratio * adv_targ
    surr2 = torch.clamp(ratio, 1.0 - clip_param, 1.0 + clip_param) * adv_targ
    action_loss = -torch.min(surr1, surr2).mean()
    if use_clipped_value_loss:
        value_pred_clipped = value_preds_batch + (values - value_preds_batch).clamp(-clip_param, clip_param)
        value_loss = 0.5 * torch.max((values - return_batch).pow(2), (value_pred_clipped - return_batch).pow(2)).mean()
    else:
        value_loss = 0.5 * (return_batch - values).pow(2).mean()
    total = value_loss * value_loss_coef + action_loss - dist_entropy * entropy_coef
    return value_loss, action_loss, total


# A16: reward_util / override_rewards (common/env_utils.py:690-713)
def reward_util(pred_monoFromMem, gt_mono_mag):
    loss = F.mse_loss(pred_monoFromMem, gt_mono_mag)
    return -loss.item() / torch.mean(torch.pow(gt_mono_mag, 2.0)).item()


def override_rewards(rewards, dones, next_pred, next_gt, reward_type=None, pred=None, gt=None, extra_reward_multiplier=10.0):
    rewards = list(rewards)
    for idx in range(len(rewards)):
        if not dones[idx]:
            rewards[idx] = reward_util(next_pred[idx].unsqueeze(0), next_gt[idx].unsqueeze(0))
            if reward_type == "quality_improvement":
                rewards[idx] -= reward_util(pred[idx].unsqueeze(0), gt[idx].unsqueeze(0))
            else:
                rewards[idx] *= extra_reward_multiplier
        else:
            rewards[idx] = 0.0
    return rewards


# A17: STFT_L2_distance (common/eval_metrics.py:306-366): squared distance of (mag*cos(phi), mag*sin(phi)) using the
# GT phase for both; mean over (re/im, F, T) per env; binaural = left + right.
def stft_l2_distance(mixed_audio, pred_binSepMasks, gt_bin_comps, pred_mono, gt_mono_comps):
    def ri(mag, phase):
        return torch.stack((mag * torch.cos(phase), mag * torch.sin(phase)), dim=1).reshape(mag.size(0), 1, -1)

    pred_bin_ = (torch.exp(mixed_audio) - 1) * pred_binSepMasks
    d = 0
    for ch in range(2):
        gt_mag, gt_ph = gt_bin_comps[..., 2 * ch], gt_bin_comps[..., 2 * ch + 1]
        d = d + torch.mean(torch.pow(ri(gt_mag, gt_ph) - ri(pred_bin_[..., ch], gt_ph), 2), dim=2)
    gm, gp = gt_mono_comps[..., 0], gt_mono_comps[..., 1]
    dm = torch.mean(torch.pow(ri(gm, gp) - ri(pred_mono[..., 0], gp), 2), dim=2)
    return d, dm


def gt_mags(obs):
    """gt_mono_mag / gt_bin_mag selections used by the trainers (ppo.py:212,219; ppo_trainer.py:376-383)."""
    gt_mono_mag = obs["gt_mono_comps"][..., 0::2][..., :1]
    gt_bin_mag = obs["gt_bin_comps"][..., 0::2][..., :2]
    return gt_bin_mag, gt_mono_mag


# ==============================================================================================
# A4: passive pre-training step (pretrain/passive/passive_trainer.py:218-249, 269-286)
# ==============================================================================================
def passive_losses(sd, mixed_bin_audio_mag, target_class, gt_bin_mag, gt_mono_mag, train_bn=True, stats_out=None):
    """Forward of one passive batch: masks from binSep, mono from bin2mono on the DETACHED masks (:228-230), the two L1 losses
    (:270-275).  With train_bn the BatchNorms use batch statistics (the trainer calls actor_critic.train(), :211-212)."""
    masks = get_binSepMasks(sd, mixed_bin_audio_mag, target_class, train_bn, stats_out)
    mono = convert_bin2mono(sd, masks.detach(), mixed_bin_audio_mag, train_bn, stats_out)
    bin_loss = F.l1_loss(masks * (torch.exp(mixed_bin_audio_mag) - 1), gt_bin_mag)
    mono_loss = F.l1_loss(mono, gt_mono_mag)
    return bin_loss, mono_loss, masks, mono


def passive_train_step(params, buffers, batch, lr=5e-4, eps=1e-5, opt_state=None):
    """optimize_supervised_loss (:269-286): loss = bin + mono, zero_grad, clip_grad_norm_ BEFORE backward (clips nothing, SURVEY
    D11), backward, Adam(lr, eps).  params: dict name -> leaf tensor requiring grad; buffers: BN running stats (updated in
    place with momentum 0.1).  Returns (bin_loss, mono_loss, optimizer)."""
    sd = dict(params)
    sd.update(buffers)
    stats = {}
    bin_loss, mono_loss, _, _ = passive_losses(sd, batch["mixed_bin_audio_mag"], batch["target_class"], batch["gt_bin_mag"],
                                               batch["gt_mono_mag"], True, stats)
    opt = opt_state if opt_state is not None else torch.optim.Adam(list(params.values()), lr=lr, eps=eps)
    opt.zero_grad()
    (bin_loss + mono_loss).backward()
    opt.step()
    with torch.no_grad():
        for k, v in stats.items():
            buffers[k].copy_(v)
    return bin_loss.detach(), mono_loss.detach(), opt


# ==============================================================================================
# A20 / A21: feeder STFT and eval iSTFT -- librosa 0.8.0 semantics restated in numpy.
# librosa is a third-party dependency of the reference (requirements.txt:85) that is NOT installed here: parity for these two
# functions is pinned against torch.stft / torch.istft (an independent implementation) in tests/test_oracle_stft.py, not
# against the reference itself ("parity unpinned" by the reference).
# ==============================================================================================
def np_hann_periodic(n):
    import numpy as np
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def np_stft(y, n_fft=1023, hop=512):
    """librosa.stft(y, n_fft, hop_length=hop): win_length = n_fft, periodic Hann, center=True, pad_mode='reflect';
    returns complex64 [1 + n_fft//2, 1 + (len(y_padded) - n_fft)//hop]  (dataset.py:190-226; simulator_train.py:425-481)."""
    import numpy as np
    w = np_hann_periodic(n_fft).astype(np.float32)
    yp = np.pad(y.astype(np.float32), n_fft // 2, mode="reflect")
    T = 1 + (len(yp) - n_fft) // hop
    frames = np.stack([yp[t * hop:t * hop + n_fft] * w for t in range(T)], axis=1)
    return np.fft.rfft(frames, axis=0).astype(np.complex64)


def np_istft(stft_matrix, hop=512, length=16000):
    """librosa.istft(stft_matrix, hop_length=hop, length=length): n_fft = 2*(bins-1), periodic Hann, overlap-add, division by
    the window sum-of-squares where > tiny, trim n_fft//2, fix length  (eval_metrics.py:232-251)."""
    import numpy as np
    nb, T = stft_matrix.shape
    n_fft = 2 * (nb - 1)
    w = np_hann_periodic(n_fft).astype(np.float32)
    full = n_fft + hop * (T - 1)
    y = np.zeros(full, np.float32)
    wss = np.zeros(full, np.float32)
    frames = np.fft.irfft(stft_matrix, n=n_fft, axis=0).astype(np.float32)
    for t in range(T):
        y[t * hop:t * hop + n_fft] += frames[:, t] * w
        wss[t * hop:t * hop + n_fft] += w * w
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    y = y[n_fft // 2:]
    if len(y) >= length:
        return y[:length]
    return np.pad(y, (0, length - len(y)))


def np_stft_features(wave_bcl, fp16_round=False):
    """[B,C,L] -> (log1p|STFT| BHWC [B,512,T,C], phase BHWC): dataset.py:228 / simulator_train.py:437-441,483-486."""
    import numpy as np
    B, C, L = wave_bcl.shape
    mags, phs = [], []
    for b in range(B):
        mc, pc = [], []
        for c in range(C):
            X = np_stft(wave_bcl[b, c])
            m = np.abs(X)
            if fp16_round:
                m = m.astype(np.float16).astype(np.float32)
            mc.append(np.log1p(m))
            pc.append(np.angle(X))
        mags.append(np.stack(mc, -1))
        phs.append(np.stack(pc, -1))
    return np.stack(mags).astype(np.float32), np.stack(phs).astype(np.float32)


# ==============================================================================================
# N2: waveform quality metrics (common/eval_metrics.py:12-196), restated in numpy following the reference line by line
# ==============================================================================================
BSS_EPS = 1e-13
BSS_METRIC_ORDER = ("si_sdr", "si_sir", "si_sar", "sd_sdr", "snr", "srr", "si_sdri", "sd_sdri", "snri", "si_siri", "si_sari")


def np_scale_bss_eval_helper(references, estimate, idx, compute_sir_sar=True):
    """eval_metrics.py:12-58.  references [n_samples, n_sources], estimate [n_samples]."""
    import numpy as np
    source = references[..., idx]
    source_energy = (source ** 2).sum()
    alpha = source @ estimate / source_energy
    e_true = source
    e_res = estimate - e_true
    signal = (e_true ** 2).sum()
    noise = (e_res ** 2).sum()
    snr = 10 * np.log10(signal / noise)
    e_true = source * alpha
    e_res = estimate - e_true
    signal = (e_true ** 2).sum()
    noise = (e_res ** 2).sum()
    si_sdr = 10 * np.log10(signal / noise)
    srr = -10 * np.log10((1 - (1 / alpha)) ** 2)
    sd_sdr = snr + 10 * np.log10(alpha ** 2)
    si_sir = si_sar = np.nan
    if compute_sir_sar:
        references_projection = references.T @ references
        references_onto_residual = np.dot(references.transpose(), e_res)
        b = np.linalg.solve(references_projection, references_onto_residual) + BSS_EPS
        e_interf = np.dot(references, b)
        e_artif = e_res - e_interf + BSS_EPS
        si_sir = 10 * np.log10(signal / (e_interf ** 2).sum())
        si_sar = 10 * np.log10(signal / (e_artif ** 2).sum())
    return si_sdr, si_sir, si_sar, sd_sdr, snr, srr


def np_waveform_metrics(gt_wave, est_wave, mix_lr, dtype=None):
    """evaluate (:199-229) = preprocess (:170-196) + evaluate_helper (:124-167) + scale_bss_eval (:61-122) for one clip.
    gt_wave, est_wave [L]; mix_lr [2, L].  Returns the 11 metrics in BSS_METRIC_ORDER.  dtype: compute dtype (the reference
    works in the float32 librosa returns; float64 gives the well-conditioned reference values for tolerance checks)."""
    import numpy as np
    dt = dtype or np.float32
    true_signal, estimated_signal, mixed_signal = [np.asarray(gt_wave, dt)[None]], [np.asarray(est_wave, dt)[None]], [np.asarray(mix_lr, dt)]
    references = np.stack([x for x in true_signal], axis=-1).transpose(1, 0, 2)   # time x channels x sources
    references = references - references.mean(axis=0)
    estimates = np.stack([x for x in estimated_signal], axis=-1).transpose(1, 0, 2)
    estimates = estimates - estimates.mean(axis=0)
    mixture = mixed_signal[0].transpose(1, 0) - mixed_signal[0].transpose(1, 0).mean(axis=0)
    mixture = np.mean(mixture, axis=1, keepdims=True)
    est = np_scale_bss_eval_helper(references[..., 0, :], estimates[..., 0, 0], 0)
    mix = np_scale_bss_eval_helper(references[..., 0, :], mixture[..., 0], 0)
    si_sdr, si_sir, si_sar, sd_sdr, snr, srr = est
    return np.array([si_sdr, si_sir, si_sar, sd_sdr, snr, srr, si_sdr - mix[0], sd_sdr - mix[3], snr - mix[4],
                     si_sir - mix[1], si_sar - mix[2]], np.float64)


# ==============================================================================================
# N1: RIR-convolution feeder (pretrain/datasets/dataset.py:162-228), numpy/scipy restatement
# ==============================================================================================
def np_compute_audiospects(mono_sources, rirs, gt_mono_mag_norm=0.0):
    """mono_sources [S][L] int16-valued, rirs [S][Lr][2] float32 -> (log1p mixed mag [512,T,2], gt_bin_mag [512,T,2],
    gt_mono_mag [512,T,1], mixed waveform [2,L], per-source int16 waveforms) following compute_audiospects line by line
    (scipy.signal.fftconvolve mode "same"; np.round -> int16 -> float32 / 32768; librosa.stft restated by np_stft)."""
    import numpy as np
    from scipy.signal import fftconvolve
    gt_mono_mag, gt_bin_mag = None, None
    mixed = 0
    per_source = []
    for idx in range(len(mono_sources)):
        mono_audio = np.asarray(mono_sources[idx])
        binaural_rir = np.asarray(rirs[idx])
        conv = np.array([fftconvolve(mono_audio, binaural_rir[:, ch], mode="same") for ch in range(2)])
        conv = np.round(conv).astype("int16").astype("float32")
        conv *= (1 / 32768)
        per_source.append(conv)
        if idx == 0:
            gt_bin_mag = np.stack([np.abs(np_stft(conv[0])), np.abs(np_stft(conv[1]))], axis=-1).astype("float32")
            m = np.abs(np_stft(mono_audio.astype("float32") / 32768))
            rms = np.power(np.mean(np.power(m, 2)), 0.5)
            if gt_mono_mag_norm != 0.0 and rms != 0.:
                m = m * gt_mono_mag_norm / rms
            gt_mono_mag = m[..., None].astype("float32")
        mixed = mixed + conv
    mixed = mixed / len(mono_sources)
    mixed_mag = np.stack([np.abs(np_stft(mixed[0])), np.abs(np_stft(mixed[1]))], axis=-1).astype("float32")
    return np.log1p(mixed_mag), gt_bin_mag, gt_mono_mag, mixed, per_source
