"""GPU tests of the training path: HIP backward kernels vs torch autograd (CPU), FlatAdam vs torch.optim.Adam, and one full
PPO.update_pol / PPO.update_sep against the reference-generated fixture (tests/golden/rl_updates.npz)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import m2h_oracle as O
from m2h import synthetic

pytestmark = pytest.mark.gpu
TOL = 3e-5


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _rel(a, b):
    return O.rel_l1(torch.as_tensor(a).detach().cpu(), torch.as_tensor(b).detach())


@pytest.mark.parametrize("B,H,W,Ci,Co,k,s,p,slope", [
    (3, 32, 32, 32, 32, 3, 1, 1, 0.0),     # AcousticMem conv0
    (2, 32, 32, 32, 32, 8, 4, 0, 0.0),     # AudioCNN conv0
    (4, 7, 7, 32, 64, 4, 2, 0, 0.0),       # AudioCNN conv1
    (5, 2, 2, 64, 32, 2, 1, 0, 0.0),       # AudioCNN conv2
    (2, 31, 31, 32, 64, 4, 2, 0, 0.0),     # VisualCNN conv1 (odd input, last row unused)
    (2, 14, 14, 64, 32, 3, 1, 0, 1.0),     # VisualCNN conv2 (no activation)
    (6, 12, 12, 32, 512, 12, 1, 0, 0.0),   # VisualCNN FC as a 12x12 conv
    (9, 1, 1, 1536, 512, 1, 1, 0, 1.0),    # Linear
    (3, 16, 8, 64, 128, 4, 2, 1, 0.2),     # U-Net encoder stage: the input gradient runs as ONE transposed-conv launch
    (2, 2, 16, 128, 256, 4, 2, 1, 0.2),    # ... on a two-row image (one output row: half of the window in the padding)
])
def test_conv2d_backward_matches_torch(B, H, W, Ci, Co, k, s, p, slope):
    from m2h import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(B * 7 + k)
    x = torch.randn(B, Ci, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5).requires_grad_(True)
    b = (torch.randn(Co, generator=g) * 0.1).requires_grad_(True)
    y = F.conv2d(x, w, b, stride=s, padding=p)
    y = y if slope == 1.0 else F.leaky_relu(y, slope)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wd, bd = w.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    yd = MF.conv2d(xd, wd, bd, s, p, slope=slope)
    assert _rel(yd.permute(0, 3, 1, 2), y) < TOL
    yd.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    assert _rel(wd.grad, w.grad) < TOL
    assert _rel(bd.grad, b.grad) < TOL
    assert _rel(xd.grad.permute(0, 3, 1, 2), x.grad) < TOL


def test_visual_conv0_padded_channels_and_deslice_backward():
    from m2h import functional as MF, ops
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    # 3-channel weight on a 4-channel (zero padded) input: gradient only for the 3 real channels
    x = torch.rand(2, 3, 128, 128, generator=g)
    w = (torch.randn(32, 3, 8, 8, generator=g) * 0.07).requires_grad_(True)
    b = torch.zeros(32, requires_grad=True)
    y = F.relu(F.conv2d(x, w, b, stride=4))
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    x4 = torch.cat((x, torch.zeros(2, 1, 128, 128)), 1).permute(0, 2, 3, 1).contiguous().to(dev)
    wd, bd = w.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    yd = MF.conv2d(x4, wd, bd, 4, 0, slope=0.0)
    yd.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    assert _rel(wd.grad, w.grad) < TOL and _rel(bd.grad, b.grad) < TOL
    # de-sliced output (AcousticMem conv1): gradient arrives in BHWC layout
    xin = torch.randn(2, 32, 32, 32, generator=g)
    w2 = (torch.randn(16, 32, 3, 3, generator=g) * 0.08).requires_grad_(True)
    out = O.deslice_freq(F.conv2d(xin, w2, None, padding=1))
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    w2d = w2.detach().to(dev).requires_grad_(True)
    od = MF.conv2d(xin.permute(0, 2, 3, 1).contiguous().to(dev), w2d, None, 1, 1, slope=1.0, deslice=True)
    assert _rel(od, out) < TOL
    od.backward(go.to(dev))
    assert _rel(w2d.grad, w2.grad) < TOL


@pytest.mark.parametrize("N", [4, 16, 20])  # <= 16 rows: one fused launch per step (m2h_gru_step); above: GEMM + gate kernel
def test_gru_sequence_backward_matches_torch(N):
    from m2h import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(8)
    T, I, H = 5, 1536, 512
    sd = {O.GRU + "weight_ih_l0": (torch.randn(3 * H, I, generator=g) * I ** -0.5).requires_grad_(True),
          O.GRU + "weight_hh_l0": (torch.randn(3 * H, H, generator=g) * H ** -0.5).requires_grad_(True),
          O.GRU + "bias_ih_l0": (torch.randn(3 * H, generator=g) * 0.05).requires_grad_(True),
          O.GRU + "bias_hh_l0": (torch.randn(3 * H, generator=g) * 0.05).requires_grad_(True)}
    x = torch.randn(T * N, I, generator=g, requires_grad=True)
    h0 = torch.randn(1, N, H, generator=g) * 0.5
    masks = (torch.rand(T * N, 1, generator=g) > 0.25).float()
    out, hT = O.rnn_forward(sd, x, h0, masks)
    gout = torch.randn(out.shape, generator=g)
    (out * gout).sum().backward()
    ps = [sd[O.GRU + k].detach().to(dev).requires_grad_(True) for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    xd = x.detach().to(dev).requires_grad_(True)
    od, hd = MF.GRUSequence.apply(xd, h0[0].to(dev), masks.to(dev), ps[0], ps[1], ps[2], ps[3], T)
    assert _rel(od, out) < TOL and _rel(hd, hT[0]) < TOL
    (od * gout.to(dev)).sum().backward()
    for p_, k in zip(ps, ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")):
        assert _rel(p_.grad, sd[O.GRU + k].grad) < 1e-4, k
    assert _rel(xd.grad, x.grad) < 1e-4


@pytest.mark.parametrize("T", [1, 4])
def test_two_layer_gru_state_encoder_matches_torch(T):
    """RNNStateEncoder(num_layers=2) (rnn_state_encoder.py:10-32,63-137): nn.GRU's stacked layers, every layer's hidden state masked at
    the reset steps -- forward (single step and sequence) and backward against torch's nn.GRU stepped on the CPU."""
    from m2h.rl.models.rnn_state_encoder import RNNStateEncoder
    dev = _dev()
    torch.manual_seed(5)
    N, I, H = 14, 96, 64
    enc = RNNStateEncoder(I, H, num_layers=2)
    ref = torch.nn.GRU(I, H, num_layers=2)
    ref.load_state_dict(enc.rnn.state_dict())
    assert enc.num_recurrent_layers == 2
    g = torch.Generator().manual_seed(9)
    x = torch.randn(T * N, I, generator=g)
    h0 = torch.randn(2, N, H, generator=g) * 0.5
    masks = (torch.rand(T * N, 1, generator=g) > 0.3).float()
    # reference: one step at a time, the hidden state of BOTH layers multiplied by the step's mask first (_mask_hidden)
    xr = x.clone().requires_grad_(True)
    h, outs = h0.clone(), []
    for t in range(T):
        o, h = ref(xr[t * N:(t + 1) * N].unsqueeze(0), h * masks[t * N:(t + 1) * N].view(1, N, 1))
        outs.append(o[0])
    want = torch.cat(outs, 0)
    gout = torch.randn(want.shape, generator=g)
    ((want * gout).sum() + h.sum()).backward()
    enc = enc.to(dev)
    xd = x.to(dev).requires_grad_(True)
    got, hT = enc(xd, h0.to(dev), masks.to(dev))
    assert hT.shape == (2, N, H) and _rel(got, want) < TOL and _rel(hT, h) < TOL
    ((got * gout.to(dev)).sum() + hT.sum()).backward()
    assert _rel(xd.grad, xr.grad) < 1e-4
    for (k, p_), (_k2, q) in zip(enc.rnn.named_parameters(), ref.named_parameters()):
        assert _rel(p_.grad, q.grad) < 1e-4, k
    with torch.no_grad():                       # the no-grad path of a rollout step takes the same stacked route
        got2, hT2 = enc(x.to(dev), h0.to(dev), masks.to(dev))
    assert _rel(got2, want) < TOL and _rel(hT2, h) < TOL


@pytest.mark.parametrize("tag,kw", [("gru2", dict(num_layers=2, rnn_type="GRU")), ("lstm1", dict(num_layers=1, rnn_type="LSTM")),
                                    ("lstm2", dict(num_layers=2, rnn_type="LSTM"))])
def test_rnn_state_encoder_variants_match_the_reference_fixture(tag, kw):
    """RNNStateEncoder's configurations that policy.py never selects (rnn_state_encoder.py:10-61): outputs, final hidden states (h and c
    packed for the LSTM) and gradients of the reference class itself (tests/golden/rnn_variants.npz, oracle/gen_golden.py --only rnn_variants)."""
    from m2h.rl.models.rnn_state_encoder import RNNStateEncoder
    dev = _dev()
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "rnn_variants.npz"))
    N, I, H = 14, 96, 64
    enc = RNNStateEncoder(I, H, **kw)
    enc.load_state_dict({k[len(tag) + 3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(tag + ".w.")})
    enc = enc.to(dev)
    assert enc.num_recurrent_layers == kw["num_layers"] * (2 if kw["rnn_type"] == "LSTM" else 1)
    for T in (1, 4):
        pre = "%s.T%d." % (tag, T)
        t = lambda k: torch.from_numpy(gold[pre + k]).to(dev)  # noqa: E731
        x = t("x").requires_grad_(True)
        enc.zero_grad()
        y, hT = enc(x, t("h0"), t("masks"))
        assert _rel(y, gold[pre + "y"]) < TOL and _rel(hT, gold[pre + "hT"]) < TOL
        ((y * t("gout")).sum() + hT.sum()).backward()
        assert _rel(x.grad, gold[pre + "dx"]) < 1e-4
        for k, p_ in enc.named_parameters():
            assert _rel(p_.grad, gold[pre + "d." + k]) < 1e-4, (T, k)
        with torch.no_grad():
            y2, hT2 = enc(t("x"), t("h0"), t("masks"))
        assert _rel(y2, gold[pre + "y"]) < TOL and _rel(hT2, gold[pre + "hT"]) < TOL


@pytest.mark.parametrize("M,H,A", [(280, 512, 3), (280, 512, 4), (37, 128, 4), (280, 512, 8)])
def test_policy_heads_and_ppo_loss_backward_match_torch(M, H, A):
    """(A = 3 / 4: dz rows of 4 / 8 floats through m2h_policy_heads_wgrad, 37 rows: a ragged last round; A = 8: dz rows of 12 floats, the tiled engine's route)"""
    from m2h import functional as MF
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(M, H, generator=g, requires_grad=True)
    sd = {"action_dist.linear.weight": (torch.randn(A, H, generator=g) * 0.05).requires_grad_(True),
          "action_dist.linear.bias": (torch.randn(A, generator=g) * 0.1).requires_grad_(True),
          "critic.fc.weight": (torch.randn(1, H, generator=g) * 0.05).requires_grad_(True),
          "critic.fc.bias": (torch.randn(1, generator=g) * 0.1).requires_grad_(True)}
    actions = torch.randint(0, A, (M, 1), generator=g)
    old_v, ret, adv = (torch.randn(M, 1, generator=g) for _ in range(3))
    value, logp_all, probs = O.heads(sd, feats)
    logp = logp_all.gather(1, actions)
    old_logp = logp.detach() + torch.randn(M, 1, generator=g) * 0.2
    ent = O.categorical_entropy(logp_all, probs).mean()
    vl, al, total = O.ppo_losses(value, logp, ent, old_v, ret, adv, old_logp, 0.1, 0.5, 0.2)
    total.backward()
    fd = feats.detach().to(dev).requires_grad_(True)
    ps = {k: v.detach().to(dev).requires_grad_(True) for k, v in sd.items()}
    v_, lp_, ent_rows, probs_, _ = MF.PolicyHeads.apply(fd, ps["action_dist.linear.weight"], ps["action_dist.linear.bias"],
                                                        ps["critic.fc.weight"], ps["critic.fc.bias"], actions.reshape(-1).to(dev))
    tot, stats = MF.PPOLoss.apply(v_, lp_, ent_rows, old_v.to(dev), ret.to(dev), adv.to(dev), old_logp.to(dev), 0.1, 0.5, 0.2, True)
    assert abs(tot.item() - total.item()) < 1e-5 and abs(stats[0].item() - vl.item()) < 1e-5 and abs(stats[2].item() - ent.item()) < 1e-5
    tot.backward()
    assert _rel(fd.grad, feats.grad) < 1e-4
    for k in sd:
        assert _rel(ps[k].grad, sd[k].grad) < 1e-4, k


def test_l1_loss_and_flat_adam_match_torch():
    from m2h import functional as MF
    from m2h.optim import FlatAdam
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    pred = torch.randn(6, 512, 32, 1, generator=g, requires_grad=True)
    gt = torch.randn(6, 512, 32, 4, generator=g)
    ref = F.l1_loss(pred, gt[..., 0::2][..., :1])
    ref.backward()
    pd = pred.detach().to(dev).requires_grad_(True)
    ls = MF.l1_loss(pd, gt.to(dev), 0)
    ls.backward()
    assert abs(ls.item() - ref.item()) < 1e-6 and torch.allclose(pd.grad.cpu(), pred.grad, atol=1e-9)
    # FlatAdam == clip_grad_norm_ + torch.optim.Adam(eps=1e-5)
    ws = [torch.randn(64, 33, generator=g), torch.randn(17, generator=g)]
    pt = [w.clone().requires_grad_(True) for w in ws]
    pm = [w.clone().to(dev).requires_grad_(True) for w in ws]
    ot = torch.optim.Adam(pt, lr=1e-2, eps=1e-5)
    om = FlatAdam(pm, lr=1e-2, eps=1e-5)
    for step in range(4):
        gs = [torch.randn(w.shape, generator=g) * (3.0 if step % 2 else 0.01) for w in ws]
        ot.zero_grad()
        om.zero_grad()
        for p_, q_, gg in zip(pt, pm, gs):
            p_.grad = gg.clone()
            q_.grad = gg.to(dev)     # (FlatAdam gathers the gradients autograd leaves in .grad; every third step below: none)
        torch.nn.utils.clip_grad_norm_(pt, 0.5)
        ot.step()
        om.step(max_grad_norm=0.5)
    for p_, q_ in zip(pt, pm):
        assert torch.allclose(q_.detach().cpu(), p_.detach(), atol=2e-6, rtol=1e-5)


def _fill_pol_storage(ro, obs_all, T, N, g):
    # identical sequence of generator draws as oracle/gen_golden.py::_fill_pol_storage
    for k in ro.observations:
        ro.observations[k].copy_(obs_all[k].reshape(T + 1, N, *obs_all[k].shape[1:]))
    ro.recurrent_hidden_states_pol.copy_(torch.randn(T + 1, 1, N, 512, generator=g) * 0.3)
    ro.pred_binSepMasks.copy_(torch.randn(T, N, 512, 32, 2, generator=g))
    ro.pred_mono.copy_(torch.rand(T, N, 512, 32, 1, generator=g))
    ro.prev_pred_monoFromMem.copy_(torch.rand(T + 1, N, 512, 32, 1, generator=g))
    ro.rewards.copy_(torch.randn(T, N, 1, generator=g) * 0.1)
    ro.value_preds.copy_(torch.randn(T + 1, N, 1, generator=g) * 0.2)
    ro.returns.copy_(torch.randn(T + 1, N, 1, generator=g) * 0.2)
    ro.action_log_probs.copy_(-1.1 + 0.1 * torch.randn(T, N, 1, generator=g))
    ro.actions.copy_(torch.randint(0, 3, (T, N, 1), generator=g))
    ro.masks.copy_((torch.rand(T + 1, N, 1, generator=g) > 0.2).float())


def _agent(seed, dev, cache=True):
    from m2h.common.spaces import Discrete, move2hear_observation_space
    from m2h.rl.ppo.policy import Move2HearPolicy
    from m2h.rl.ppo.ppo import PPO
    pol = Move2HearPolicy(move2hear_observation_space(), Discrete(3), "spectrogram", 512, False, True, use_ddppo=True)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), seed).items()}
    pol.load_state_dict(sd, strict=True)
    pol = pol.to(dev)
    pol.train()
    agent = PPO(actor_critic=pol, clip_param=0.1, ppo_epoch=2, num_mini_batch=1, value_loss_coef=0.5, bin_separation_loss_coef=1.0,
                mono_conversion_loss_coef=1.0, entropy_coef=0.2, lr_pol=1e-4, lr_sep=5e-4, eps=1e-5, max_grad_norm=0.5,
                freeze_passive_separators=True, cache_separator_outputs=cache)
    for m in (pol.binSep_enc, pol.binSep_dec, pol.bin2mono_enc, pol.bin2mono_dec):  # ppo_trainer.py:557-577
        m.eval()
        for p in m.parameters():
            p.requires_grad_(False)
    return agent, pol, sd


@pytest.mark.parametrize("views", [False, True])
def test_update_pol_matches_reference_fixture(golden_dir, views):
    """views=True: the one-mini-batch update reads the storage in place instead of gathering a permuted copy (the trainer's
    setting); both must reproduce the reference's losses and post-step weights."""
    from m2h.common.rollout_storage import RolloutStoragePol
    from m2h.common.spaces import move2hear_observation_space
    dev = _dev()
    gold = np.load(os.path.join(golden_dir, "rl_updates.npz"))
    agent, pol, sd = _agent(int(gold["seed_w"]), dev)
    T, N = int(gold["pol_T"]), int(gold["pol_N"])
    obs_all = {k: torch.from_numpy(v).float() for k, v in synthetic.make_rl_observations((T + 1) * N, int(gold["pol_obs_seed"])).items()}
    ro = RolloutStoragePol(T, N, move2hear_observation_space(), 512)
    _fill_pol_storage(ro, obs_all, T, N, torch.Generator().manual_seed(int(gold["pol_fill_seed"])))
    ro.to(dev)
    ro.full_batch_views = views
    torch.manual_seed(int(gold["pol_perm_seed"]))
    v, a, h = agent.update_pol(ro)
    ref = gold["pol_losses"]
    assert abs(v - ref[0]) < 1e-4 * max(1, abs(ref[0])) and abs(a - ref[1]) < 2e-5 and abs(h - ref[2]) < 1e-4
    # parameter deltas after 2 Adam steps (lr 1e-4): element-wise vs the reference for every small tensor
    post = pol.state_dict()
    checked = 0
    for key in gold.files:
        if not key.startswith("polpost."):
            continue
        k = key[len("polpost."):]
        d_ref = torch.from_numpy(gold[key]) - sd[k]
        d_mine = post[k].cpu() - sd[k]
        assert d_ref.abs().max() > 0
        # Adam's first steps are +-lr per element (sign of the gradient): compare deltas, allowing sign flips only where the
        # gradient is ~0 (tiny fraction)
        bad = ((d_ref - d_mine).abs() > 2e-5).float().mean().item()
        assert bad < 0.01, (k, bad)
        checked += 1
    assert checked >= 15


@pytest.mark.parametrize("cache,views", [(True, False), (False, False), (True, True), (False, True)])
def test_update_sep_matches_reference_fixture(golden_dir, cache, views):
    from m2h.common.rollout_storage import RolloutStorageSep
    from m2h.common.spaces import move2hear_observation_space
    dev = _dev()
    gold = np.load(os.path.join(golden_dir, "rl_updates.npz"))
    agent, pol, sd = _agent(int(gold["seed_w"]), dev, cache)
    T, N = int(gold["sep_T"]), int(gold["sep_N"])
    obs_s = {k: torch.from_numpy(v).float() for k, v in synthetic.make_rl_observations((T + 1) * N, int(gold["sep_obs_seed"])).items()}
    rs = RolloutStorageSep(T, N, move2hear_observation_space())
    g2 = torch.Generator().manual_seed(int(gold["sep_fill_seed"]))
    for k in rs.observations:
        rs.observations[k].copy_(obs_s[k].reshape(T + 1, N, *obs_s[k].shape[1:]))
    rs.prev_pred_monoFromMem.copy_(torch.rand(T + 1, N, 512, 32, 1, generator=g2))
    rs.masks.copy_((torch.rand(T + 1, N, 1, generator=g2) > 0.3).float())
    rs.to(dev)
    rs.full_batch_views = views
    torch.manual_seed(int(gold["sep_perm_seed"]))
    b, m, mm = agent.update_sep(rs)
    ref = gold["sep_losses"]
    assert abs(b - ref[0]) < 1e-4 * ref[0] and abs(m - ref[1]) < 1e-4 * ref[1] and abs(mm - ref[2]) < 1e-4 * ref[2]
    post = pol.state_dict()
    for k in ("acoustic_mem.cnn.0.weight", "acoustic_mem.cnn.2.weight"):
        d_ref = torch.from_numpy(gold["seppost." + k]) - sd[k]
        d_mine = post[k].cpu() - sd[k]
        bad = ((d_ref - d_mine).abs() > 1e-4).float().mean().item()
        assert bad < 0.01, (k, bad)


def test_update_sep_cache_follows_after_update_by_refreshing_row_zero(golden_dir):
    """Between the six sub-updates of a cycle RolloutStorageSep.after_update() copies the last stored observation into row 0
    (rollout_storage.py:386-390); the separator-output cache then refreshes that row only.  Three update_sep calls with
    after_update() in between must give the losses of the uncached schedule (which re-runs the separators over the whole
    buffer every epoch), and the cached outputs must equal a from-scratch evaluation of the buffer as it stands."""
    from m2h.common.rollout_storage import RolloutStorageSep
    from m2h.common.spaces import move2hear_observation_space
    dev = _dev()
    gold = np.load(os.path.join(golden_dir, "rl_updates.npz"))
    T, N = int(gold["sep_T"]), int(gold["sep_N"])
    obs_s = {k: torch.from_numpy(v).float() for k, v in synthetic.make_rl_observations((T + 1) * N, int(gold["sep_obs_seed"])).items()}
    out = []
    for cache in (True, False):
        agent, pol, _sd = _agent(int(gold["seed_w"]), dev, cache)
        rs = RolloutStorageSep(T, N, move2hear_observation_space())
        g2 = torch.Generator().manual_seed(int(gold["sep_fill_seed"]))
        for k in rs.observations:
            rs.observations[k].copy_(obs_s[k].reshape(T + 1, N, *obs_s[k].shape[1:]))
        rs.prev_pred_monoFromMem.copy_(torch.rand(T + 1, N, 512, 32, 1, generator=g2))
        rs.masks.copy_((torch.rand(T + 1, N, 1, generator=g2) > 0.3).float())
        rs.to(dev)
        rs.full_batch_views = True
        losses = []
        for i in range(3):
            torch.manual_seed(77 + i)
            losses.append(agent.update_sep(rs))
            rs.after_update()
        out.append(losses)
        if cache:
            pm_c, mono_c = agent._separator_outputs(rs)[:2]   # refreshed for the state after the last after_update()
            mix = rs.observations["mixed_bin_audio_mag"][:-1]
            obs = {"mixed_bin_audio_mag": mix.reshape(T * N, *mix.shape[2:]), "target_class": rs.observations["target_class"][:-1].reshape(T * N, -1)}
            with torch.no_grad():
                pm_f = pol.get_binSepMasks(obs)
                mono_f = pol.convert_bin2mono(pm_f, mixed_audio=obs["mixed_bin_audio_mag"])
            assert _rel(pm_c.reshape(pm_f.shape), pm_f.cpu()) < 1e-6 and _rel(mono_c.reshape(mono_f.shape), mono_f.cpu()) < 1e-6
            assert not torch.equal(pm_c[0], pm_c[1])
    for la, lb in zip(*out):
        for x, y in zip(la, lb):
            assert abs(x - y) < 1e-5 * max(1.0, abs(y)), (out[0], out[1])


@pytest.mark.parametrize("N,B,H", [(32, 3, 32), (16, 3, 32), (16, 5, 8), (8, 2, 32)])
def test_image_row_wgrad_kernel_matches_the_general_kernel_and_torch(N, B, H):
    """The 3x3 / 32-channel / 32-wide weight-gradient kernel (one image row per reduction chunk, taps as shifts of one staged
    patch, 16-wide MFMA for N <= 16) against the general gather kernel (knob 21 = -1) and torch autograd on the CPU."""
    from m2h import functional as MF
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, H, 32, 32, generator=g)
    dy = torch.randn(B, H, 32, N, generator=g)
    w = torch.randn(N, 32, 3, 3, generator=g, requires_grad=True)
    out = F.conv2d(x.permute(0, 3, 1, 2), w, None, 1, 1)
    out.backward(dy.permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(N, 9 * 32)        # packed [n][(kh, kw, c)]
    got = {}
    for knob in (0, -1):
        ops.debug_set(21, knob)
        try:
            got[knob] = MF.conv_wgrad(x.to(dev), None, dy.to(dev), N, 3, 3, 1, 1).cpu()
        finally:
            ops.debug_set(21, 0)
    assert _rel(got[0], ref) < 2e-5 and _rel(got[-1], ref) < 2e-5
    assert _rel(got[0], got[-1]) < 1e-5


@pytest.mark.parametrize("N,C,B,deslice,grad", [(32, 32, 20, False, False), (16, 32, 20, True, False), (32, 16, 20, False, True), (16, 16, 17, False, False),
                                                (8, 32, 16, False, False)])
def test_image_row_conv3x3_kernel_matches_the_engine_and_torch(N, C, B, deslice, grad):
    """The 3x3 / stride 1 image-row kernel (whole weight matrix in LDS, four image rows per chunk staged as one padded patch)
    against the general engine (knob 22 = -1) and torch on the CPU: forward with ReLU, de-sliced store, and the input gradient
    (taps walked backwards, 16 input channels)."""
    from m2h import functional as MF
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(23)
    H = 32
    if grad:   # input gradient of Conv2d(N -> C): dy [B,H,W,C] -> dx [B,H,W,N]
        w = torch.randn(C, N, 3, 3, generator=g) * 0.1
        dy = torch.randn(B, H, 32, C, generator=g)
        ref = F.conv_transpose2d(dy.permute(0, 3, 1, 2), w, None, 1, 1).permute(0, 2, 3, 1)
        run = lambda: MF.conv_dgrad(dy.to(dev), w.to(dev), (H, 32), 1, 1)   # noqa: E731
    else:
        x = torch.randn(B, H, 32, C, generator=g)
        w = torch.randn(N, C, 3, 3, generator=g) * 0.1
        slope = 1.0 if deslice else 0.0
        y = F.conv2d(x.permute(0, 3, 1, 2), w, None, 1, 1)
        y = y if deslice else torch.relu(y)
        if deslice:   # memory_nets.py:62-67: channel c*16+s, row h -> frequency row s*H + h, channel c
            ref = y.view(B, N // 16, 16, H, 32).permute(0, 2, 3, 4, 1).reshape(B, 16 * H, 32, N // 16)
        else:
            ref = y.permute(0, 2, 3, 1)
        wp = ops.pack_conv_weight_ex(w.to(dev).contiguous(), C, C)
        run = lambda: ops.conv2d_nhwc(x.to(dev), wp, N, 3, 3, stride=1, pad=1, slope=slope, deslice=deslice)   # noqa: E731
    got = {}
    for knob in (0, -1):
        ops.debug_set(22, knob)
        try:
            got[knob] = run().cpu()
        finally:
            ops.debug_set(22, 0)
    assert got[0].shape == ref.shape
    assert _rel(got[0], ref) < 2e-5 and _rel(got[-1], ref) < 2e-5 and _rel(got[0], got[-1]) < 1e-5


def test_batched_pack_equals_the_single_tensor_packs():
    """m2h_pack_batch (one launch, LDS-tiled transposes) against the single-tensor pack entry points, bit for bit, for every
    kind and the shapes the models use (incl. channel padding, 8x8 / 12x12 taps, stride-4 / stride-2 / stride-1 input gradients)."""
    from m2h import _lib, functional as MF, ops
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    items, want = [], []
    for (Co, Ci, KH, KW, cpad) in [(64, 33, 4, 4, 36), (512, 256, 4, 4, 256), (32, 3, 8, 8, 4), (64, 32, 4, 4, 32), (512, 32, 12, 12, 32), (32, 64, 2, 2, 64),
                                 (1536, 1536, 1, 1, 1536), (1536, 512, 1, 1, 512), (5, 512, 1, 1, 512), (130, 70, 1, 1, 72)]:   # (1 x 1: nn.Linear -- a block per 64 x 64 tile)
        w = torch.randn(Co, Ci, KH, KW, generator=g).to(dev)
        dst = torch.empty(Co, KH * KW * cpad, device=dev)
        items.append((_lib.PACK_CONV, w, dst, (Co, Ci, KH, KW, Ci, cpad)))
        want.append(ops.pack_conv_weight_ex(w, Ci, cpad))
    for (Ci, Co) in [(512, 512), (128, 32), (1024, 256), (64, 16)]:
        w = torch.randn(Ci, Co, 4, 4, generator=g).to(dev)
        dst = torch.empty(4, Co, 4 * Ci, device=dev)
        items.append((_lib.PACK_CONVT, w, dst, (Ci, Co, 0, 0, 0, 0)))
        want.append(ops.pack_convT_weight(w))
    for (Co, Ci, K, st, pad) in [(64, 32, 4, 2, 1), (32, 32, 8, 4, 0), (32, 32, 3, 1, 1), (64, 32, 4, 2, 0), (32, 64, 2, 1, 0), (512, 64, 4, 2, 1),
                                (1536, 1536, 1, 1, 0), (1536, 512, 1, 1, 0), (5, 512, 1, 1, 0), (130, 70, 1, 1, 0)]:
        w = torch.randn(Co, Ci, K, K, generator=g).to(dev)
        dst = torch.empty(st * st, Ci, (K // st) * (K // st) * Co, device=dev)
        items.append((_lib.PACK_DGRAD, w, dst, (Co, Ci, K, K, st, pad)))
        want.append(MF.pack_dgrad_weight(w, st, pad))
    for (Co, Ci, K, cpad) in [(512, 32, 12, 32), (512, 32, 1, 32), (128, 3, 2, 4), (1536, 1536, 1, 1536), (130, 70, 1, 72)]:
        w = torch.randn(Co, Ci, K, K, generator=g).to(dev)
        dst = torch.empty(K * K * cpad, Co, device=dev)
        items.append((_lib.PACK_FC_DGRAD, w, dst, (Co, Ci, K, K, Ci, cpad)))
        wp = ops.pack_conv_weight_ex(w, Ci, cpad)
        want.append(MF.pack_dgrad_weight(wp.view(Co, K * K * cpad, 1, 1), 1, 0).view(K * K * cpad, Co))
    ops.pack_batch(items)
    for (kind, _w, dst, prm), ref in zip(items, want):
        assert torch.equal(dst.reshape(-1), ref.reshape(-1)), (kind, prm)
