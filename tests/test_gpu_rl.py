"""GPU parity tests of the RL forward path (AcousticMem, encoders, GRU, heads/act, storage, GAE, advantages, PPO loss,
rewards, STFT-L2) through the C-ABI, against the reference-generated fixtures and the CPU oracle.
Tolerances: fp32 path, rel-L1 <= 2e-5 on feature tensors (contract 1e-3); index/permutation work bit-exact."""
import os

import numpy as np
import pytest
import torch

import m2h_oracle as O
from m2h import synthetic

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _policy(seed, dev):
    from m2h.common.spaces import Discrete, move2hear_observation_space
    from m2h.rl.ppo.policy import Move2HearPolicy
    pol = Move2HearPolicy(move2hear_observation_space(), Discrete(3), "spectrogram", 512, False, True, use_ddppo=True)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), seed).items()}
    pol.load_state_dict(sd, strict=True)
    pol = pol.to(dev)
    pol.train()
    for m in (pol.binSep_enc, pol.binSep_dec, pol.bin2mono_enc, pol.bin2mono_dec):  # ppo_trainer.py:557-577
        m.eval()
        for p in m.parameters():
            p.requires_grad_(False)
    return pol, sd


def _obs(n, seed, dev):
    return {k: torch.from_numpy(v).float().to(dev) for k, v in synthetic.make_rl_observations(n, seed).items()}


def _rel(a, b):
    return O.rel_l1(torch.as_tensor(a).cpu(), torch.as_tensor(b))


def test_rl_forward_matches_reference_fixture(golden_dir):
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "rl_forward.npz"))
    pol, sd = _policy(int(g["seed_w"]), dev)
    N = int(g["N"])
    obs = _obs(N, int(g["seed_x"]), dev)
    masks, h0, prev = (torch.from_numpy(g[k]).to(dev) for k in ("masks", "h0", "prev_mem"))
    with torch.no_grad():
        pm = pol.get_binSepMasks(obs)
        mono = pol.convert_bin2mono(pm, mixed_audio=obs["mixed_bin_audio_mag"])
        assert _rel(pm, g["pred_binSepMasks"]) < TOL and _rel(mono, g["pred_mono"]) < TOL
        # AcousticMem: reference-style (pre-masked input) and fused-mask entry points agree with the fixture
        prev_masked = prev * masks.reshape(-1, 1, 1, 1)
        mem = pol.get_monoFromMem(mono, prev_masked)
        mem2 = pol.get_monoFromMem_masked(mono, prev, masks)
        assert mem.shape == (N, 512, 32, 1)
        assert _rel(mem, g["pred_monoFromMem"]) < TOL and torch.equal(mem, mem2)
        assert _rel(pol.pol_net.visual_encoder(obs), g["visual_feats"]) < TOL
        assert _rel(pol.pol_net.bin_encoder(obs, pred_binSepMasks=pm), g["bin_feats"]) < TOL
        cat = torch.cat((mono, mem), dim=3)
        assert _rel(pol.pol_net.monoNmonoFromMem_encoder(obs, pred_monoNmonoFromMem=cat), g["mnm_feats"]) < TOL
        feats, h1 = pol.pol_net(obs, h0, masks, pred_binSepMasks=pm, pred_mono=mono, pred_monoFromMem=mem)
        assert _rel(feats, g["gru_out"]) < TOL and _rel(h1, g["h1"]) < TOL
        # act: values/probs match; deterministic action is bit-exact; sampled action is multinomial on the same probs
        torch.manual_seed(5)
        v, a, lp, hh, probs = pol.act(obs, h0, masks, deterministic=False, pred_binSepMasks=pm, pred_mono=mono, pred_monoFromMem=mem)
        torch.manual_seed(5)
        a_ref = torch.multinomial(probs, 1, True)
        assert torch.equal(a, a_ref) and a.dtype == torch.int64 and a.shape == (N, 1)
        assert _rel(v, g["act_value"]) < TOL and _rel(probs, g["act_probs"]) < TOL
        assert torch.allclose(lp.cpu(), torch.log(probs.cpu()).gather(1, a.cpu()), atol=1e-6)
        v2, a2, lp2, _, _ = pol.act(obs, h0, masks, deterministic=True, pred_binSepMasks=pm, pred_mono=mono, pred_monoFromMem=mem)
        assert torch.equal(a2.cpu(), torch.from_numpy(g["det_action"])) and _rel(lp2, g["det_logp"]) < TOL
        assert _rel(pol.get_value(obs, h0, masks, pred_binSepMasks=pm, pred_mono=mono, pred_monoFromMem=mem), g["get_value"]) < TOL
        # evaluate_actions on a T=3 x N=2 flattened sequence with a mid-sequence reset
        T, n = 3, 2
        obs_seq = {k: v_[:T * n].contiguous() for k, v_ in obs.items()}
        ev, elp, eent, eh = pol.evaluate_actions(obs_seq, h0[:, :n].contiguous(), torch.from_numpy(g["eval_masks"]).to(dev),
                                                 torch.from_numpy(g["eval_actions"]).to(dev), pred_binSepMasks=pm[:T * n].contiguous(),
                                                 pred_mono=mono[:T * n].contiguous(), pred_monoFromMem=mem[:T * n].contiguous())
        assert _rel(ev, g["eval_value"]) < TOL and _rel(elp, g["eval_logp"]) < TOL and _rel(eh, g["eval_h"]) < TOL
        assert abs(eent.item() - float(g["eval_entropy"])) < 1e-5


def test_unbuilt_gradient_paths_fail_loudly():
    """What is not built raises instead of silently returning graph-less tensors: gradients into the (frozen) separator
    outputs through AcousticMem, and autograd through the U-Nets themselves."""
    dev = _dev()
    pol, _ = _policy(2, dev)
    mono = torch.rand(2, 512, 32, 1, device=dev, requires_grad=True)
    with pytest.raises(NotImplementedError):
        pol.get_monoFromMem(mono, torch.rand(2, 512, 32, 1, device=dev))
    for p in pol.binSep_enc.parameters():
        p.requires_grad_(True)
    with pytest.raises(NotImplementedError):
        pol.get_binSepMasks(_obs(2, 3, dev))


def test_acoustic_mem_batchnorm_variant_matches_the_reference_layers():
    """AcousticMem(use_ddppo=False) (memory_nets.py:17-23: conv3x3 -> BatchNorm2d -> ReLU -> conv3x3, the single-process PPO
    variant) against the reference's own layers -- it is a plain nn.Sequential of torch layers over the sliced concatenation
    (:40-69): train mode (batch statistics, running statistics and num_batches_tracked updated, gradients of both convs and the
    BatchNorm affine through an L1 loss as update_sep applies it), then eval mode (running statistics)."""
    import torch.nn.functional as F
    import m2h_oracle as O
    from m2h.rl.models.memory_nets import AcousticMem
    dev = _dev()
    torch.manual_seed(7)
    mem = AcousticMem(use_ddppo=False)
    with torch.no_grad():
        mem.cnn[1].weight.uniform_(0.5, 1.5)
        mem.cnn[1].bias.uniform_(-0.2, 0.2)
    ref = torch.nn.Sequential(torch.nn.Conv2d(32, 32, 3, padding=1, bias=False), torch.nn.BatchNorm2d(32), torch.nn.ReLU(),
                              torch.nn.Conv2d(32, 16, 3, padding=1, bias=False))
    ref.load_state_dict(mem.cnn.state_dict())
    mem = mem.to(dev)
    g = torch.Generator().manual_seed(8)
    B = 6
    mono, prev = torch.rand(B, 512, 32, 1, generator=g), torch.rand(B, 512, 32, 1, generator=g)
    gt = torch.rand(B, 512, 32, 1, generator=g)
    ref_fwd = lambda: O.deslice_freq(ref(torch.cat((O.slice_freq(mono), O.slice_freq(prev)), dim=1)))  # noqa: E731
    mem.train()
    ref.train()
    out = mem(mono.to(dev), prev.to(dev))
    want = ref_fwd()
    assert _rel(out.detach().cpu(), want.detach()) < TOL
    F.l1_loss(want, gt).backward()
    from m2h import functional as MF
    comps = torch.zeros(B, 512, 32, 4)
    comps[..., 0:1] = gt
    MF.l1_loss(out, comps.to(dev), 0).backward()
    for (n, p), (_n2, q) in zip(mem.cnn.named_parameters(), ref.named_parameters()):
        assert _rel(p.grad.cpu(), q.grad) < 2e-4, n
    assert _rel(mem.cnn[1].running_mean.cpu(), ref[1].running_mean) < 1e-5 and _rel(mem.cnn[1].running_var.cpu(), ref[1].running_var) < 1e-5
    assert int(mem.cnn[1].num_batches_tracked) == int(ref[1].num_batches_tracked) == 1
    mem.eval()
    ref.eval()
    with torch.no_grad():
        assert _rel(mem(mono.to(dev), prev.to(dev)).cpu(), ref_fwd()) < TOL
    with pytest.raises(NotImplementedError):
        mem(mono.to(dev), prev.to(dev))          # eval mode + autograd: not silently graph-less


def test_returns_advantages_and_generators_match_fixture(golden_dir):
    from m2h import ops
    from m2h.common.rollout_storage import RolloutStoragePol, RolloutStorageSep
    from m2h.common.spaces import Box, DictSpace
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "rl_scalars.npz"))
    T, N = int(g["T"]), int(g["N"])
    space = DictSpace({"gt_mono_comps": Box((512, 32, 4)), "target_class": Box((1,))})
    ro = RolloutStoragePol(T, N, space, 512)
    ro.to(dev)
    ro.rewards.copy_(torch.from_numpy(g["rewards"]))
    ro.value_preds.copy_(torch.from_numpy(g["value_preds"]))
    ro.masks.copy_(torch.from_numpy(g["masks"]))
    nv = torch.from_numpy(g["next_value"]).to(dev)
    ro.compute_returns(nv, True, 0.99, 0.95)
    assert torch.allclose(ro.returns.cpu(), torch.from_numpy(g["returns_gae"]), atol=1e-6)  # <= 1e-6 abs
    adv, _ = ops.advantages(ro.returns, ro.value_preds, 1)
    assert torch.allclose(adv.cpu(), torch.from_numpy(g["advantages"]), atol=2e-6)
    # distributed recipe with world = 1: biased variance around the global mean
    raw, stats = ops.advantages(ro.returns, ro.value_preds, 2)
    gmean = stats[0:1].clone()
    gvar = ops.adv_sqdiff(raw, gmean)
    d = ops.adv_apply(raw.clone(), gmean, gvar).cpu()
    a = torch.from_numpy(g["returns_gae"])[:-1] - ro.value_preds.cpu()[:-1]
    assert torch.allclose(d, (a - a.mean()) / (a.var(unbiased=False).sqrt() + 1e-5), atol=2e-6)
    ro2 = RolloutStoragePol(T, N, space, 512)
    ro2.to(dev)
    ro2.rewards.copy_(ro.rewards)
    ro2.masks.copy_(ro.masks)
    ro2.compute_returns(nv, False, 0.99, 0.95)
    assert torch.allclose(ro2.returns.cpu(), torch.from_numpy(g["returns_nogae"]), atol=1e-6)
    # generator order: bit-exact with the reference for the same CPU seed
    ro.actions.copy_(torch.arange(T * N).reshape(T, N, 1))
    torch.manual_seed(int(g["gen_seed"]))
    batch = next(iter(ro.recurrent_generator(adv, 1)))
    assert torch.equal(batch[8].cpu(), torch.from_numpy(g["gen_actions_flat"]))
    assert batch[0]["gt_mono_comps"].shape == (T * N, 512, 32, 4) and batch[1].shape == (1, N, 512)
    rs = RolloutStorageSep(6, 5, space)
    rs.to(dev)
    rs.masks.copy_(torch.arange(7 * 5).reshape(7, 5, 1).float())
    torch.manual_seed(int(g["gen_sep_seed"]))
    sb = next(iter(rs.recurrent_generator(1)))
    assert torch.equal(sb[3].cpu(), torch.from_numpy(g["gen_sep_masks_flat"]))


def test_stft_l2_and_rewards_match_fixture(golden_dir):
    from m2h import ops
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "rl_scalars.npz"))
    obs = _obs(5, int(g["l2_seed_x"]), dev)
    pm, pmono = torch.from_numpy(g["l2_masks"]).to(dev), torch.from_numpy(g["l2_mono"]).to(dev)
    d_bin = ops.stft_l2(pm, obs["gt_bin_comps"], 2, mix=obs["mixed_bin_audio_mag"])
    d_mono = ops.stft_l2(pmono, obs["gt_mono_comps"], 1)
    assert torch.allclose(d_bin.cpu(), torch.from_numpy(g["stft_l2_bin"]), rtol=2e-5)
    assert torch.allclose(d_mono.cpu(), torch.from_numpy(g["stft_l2_mono"]), rtol=2e-5)
    nxt, cur = torch.from_numpy(g["rew_mem_next"]).to(dev), torch.from_numpy(g["rew_mem_cur"]).to(dev)
    not_done = torch.from_numpy(1.0 - g["rew_dones"].astype(np.float32)).to(dev)
    L = 512 * 32
    ns, cs = ops.sq_stats(nxt, obs["gt_mono_comps"], 0), ops.sq_stats(cur, obs["gt_mono_comps"], 0)
    r1 = ops.rewards_from_stats(ns, cs, not_done, L, True).cpu().double().reshape(-1).numpy()
    r2 = ops.rewards_from_stats(ns, None, not_done, L, False, 10.0).cpu().double().reshape(-1).numpy()
    assert np.allclose(r1, g["rew_quality_improvement"], rtol=2e-5, atol=1e-7)
    assert np.allclose(r2, g["rew_extra"], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("clipped", [True, False])
def test_ppo_loss_forward_and_gradients_match_autograd(clipped):
    from m2h import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(3)
    n = 280
    values = torch.randn(n, 1, generator=gen, requires_grad=True)
    logp = (torch.randn(n, 1, generator=gen) * 0.3 - 1.0).requires_grad_(True)
    old_values, returns, adv = (torch.randn(n, 1, generator=gen) for _ in range(3))
    old_logp = logp.detach() + torch.randn(n, 1, generator=gen) * 0.2
    vl, al, total = O.ppo_losses(values, logp, torch.tensor(0.0), old_values, returns, adv, old_logp, 0.1, 0.5, 0.2, clipped)
    total.backward()
    out, gv, gl = ops.ppo_loss(values.detach().to(dev), logp.detach().to(dev), old_values.to(dev), returns.to(dev), adv.to(dev),
                               old_logp.to(dev), 0.1, value_loss_coef=0.5, use_clipped_value_loss=clipped, want_grads=True)
    assert abs(out[0].item() - vl.item()) < 1e-5 * max(1, abs(vl.item())) and abs(out[1].item() - al.item()) < 1e-5
    assert torch.allclose(gv.cpu(), values.grad, atol=1e-7, rtol=1e-4)
    assert torch.allclose(gl.cpu(), logp.grad, atol=1e-7, rtol=1e-4)


def test_generic_conv_engine_matches_torch():
    """The igemm engine on the policy-net shapes (8x8 s4, 4x4 s2, 3x3, 2x2, FC-as-conv), ragged M, odd spatial sizes."""
    import torch.nn.functional as F
    from m2h import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(11)
    for (B, H, W, Ci, Co, k, s, p) in [(3, 128, 128, 4, 32, 8, 4, 0), (3, 31, 31, 32, 64, 4, 2, 0), (2, 14, 14, 64, 32, 3, 1, 0),
                                       (5, 32, 32, 32, 32, 3, 1, 1), (7, 2, 2, 64, 32, 2, 1, 0), (9, 12, 12, 32, 512, 12, 1, 0)]:
        x = torch.randn(B, Ci, H, W, generator=gen)
        w = torch.randn(Co, Ci, k, k, generator=gen) * (2.0 / (Ci * k * k)) ** 0.5
        b = torch.randn(Co, generator=gen) * 0.1
        ref = F.relu(F.conv2d(x, w, b, stride=s, padding=p))
        y = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), ops.pack_conv_weight(w.to(dev)), Co, k, k, stride=s, pad=p,
                            bias=b.to(dev), slope=0.0)
        assert _rel(y.permute(0, 3, 1, 2), ref) < TOL, (B, H, W, Ci, Co, k, s, p)


def test_episode_stats_update_matches_the_elementwise_bookkeeping():
    """m2h_episode_stats_update == the reference's per-step bookkeeping (ppo_trainer.py:421-478) written as elementwise torch
    ops, over several steps with episode ends on some envs, bit for bit.  (The loop as a whole, on the reference's own run:
    tests/test_gpu_trainer_golden.py.)"""
    from types import SimpleNamespace
    from m2h import _lib, ops
    dev = _dev()
    N, A = 14, 3
    g = torch.Generator().manual_seed(3)
    mk = lambda: SimpleNamespace(**{n: torch.zeros(N, A if "dist_probs" in n else 1) for n in _lib.EPISODE_STATS_FIELDS})  # noqa: E731
    ref, st = mk(), mk()
    for n in _lib.EPISODE_STATS_FIELDS:
        setattr(st, n, getattr(st, n).to(dev))
    for step in range(7):
        rewards, bl, ml, fl, ndg, dg = (torch.randn(N, 1, generator=g) for _ in range(6))
        probs = torch.softmax(torch.randn(N, A, generator=g), dim=1)
        masks = (torch.rand(N, 1, generator=g) > (0.4 if step in (2, 5) else 2.0)).float()
        if step == 6:
            masks.zero_()
        r = ref
        r.current_episode_reward += rewards
        r.current_episode_step += 1
        r.current_episode_dist_probs += probs
        r.current_episode_bin_losses += bl
        r.current_episode_mono_losses += ml
        r.current_episode_monoFromMem_losses += fl
        nd = 1 - masks
        r.episode_rewards += nd * r.current_episode_reward
        r.episode_ndgs += nd * ndg
        r.episode_dgs += nd * dg
        r.episode_steps += nd * r.current_episode_step
        r.episode_counts += nd
        r.episode_dist_probs += nd * (r.current_episode_dist_probs / r.current_episode_step)
        r.episode_bin_losses_allSteps += nd * (r.current_episode_bin_losses / r.current_episode_step)
        r.episode_mono_losses_lastStep += nd * ml
        r.episode_mono_losses_allSteps += nd * (r.current_episode_mono_losses / r.current_episode_step)
        r.episode_monoFromMem_losses_lastStep += nd * fl
        r.episode_monoFromMem_losses_allSteps += nd * (r.current_episode_monoFromMem_losses / r.current_episode_step)
        for name in ("current_episode_reward", "current_episode_step", "current_episode_bin_losses", "current_episode_mono_losses",
                     "current_episode_monoFromMem_losses", "current_episode_dist_probs"):
            getattr(r, name).mul_(masks)
        ops.episode_stats_update(st, rewards.to(dev), probs.to(dev), bl.to(dev), ml.to(dev), fl.to(dev), masks.to(dev), ndg.to(dev), dg.to(dev))
    for n in _lib.EPISODE_STATS_FIELDS:
        assert torch.equal(getattr(st, n).cpu(), getattr(ref, n)), n


def test_rows_copy_moves_rows_by_device_indices():
    """m2h_rows_copy: batched gathers / scatters of storage rows addressed by a device-resident index tensor, all dtypes and
    row sizes the rollout storages hold (16-byte and 4-byte paths)."""
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    idx = torch.tensor([3, 4, 0], dtype=torch.int64, device=dev)
    big = torch.randn(6, 14, 64, 8, generator=g).to(dev)          # 16-byte path
    small = torch.randn(6, 14, 1, generator=g).to(dev)            # 56-byte rows: 4-byte path
    acts = torch.randint(0, 3, (6, 14, 1), generator=g).to(dev)   # int64
    v_big, v_small, v_acts = torch.randn(14, 64, 8, generator=g).to(dev), torch.randn(14, 1, generator=g).to(dev), torch.randint(0, 3, (14, 1), generator=g).to(dev)
    exp_big, exp_small, exp_acts = big.clone(), small.clone(), acts.clone()
    exp_big[4], exp_small[3], exp_acts[0] = v_big, v_small, v_acts
    got_row = torch.empty_like(big[0])
    want_row = big[3].clone()
    ops.rows_copy([(big, got_row, 0, -1)], idx)
    ops.rows_copy([(v_big, big, -1, 1), (v_small, small, -1, 0), (v_acts, acts, -1, 2)], idx)
    assert torch.equal(got_row, want_row)
    assert torch.equal(big, exp_big) and torch.equal(small, exp_small) and torch.equal(acts, exp_acts)
    many = [(v_small, small, -1, 0)] * 40                            # more than one launch's worth of items
    ops.rows_copy(many, idx)
    assert torch.equal(small, exp_small)
    with pytest.raises(RuntimeError):
        ops.rows_copy([(v_small, big, -1, 0)], idx)


def test_sampling_is_torch_multinomial_bit_for_bit():
    """CustomFixedCategorical.sample (argmax(probs / Exp(1)) on the device generator) draws exactly torch.multinomial(probs, 1,
    True) -- the reference's Categorical.sample (common/utils.py:16-24) -- for the same generator state, call after call, and
    leaves the generator in the same state."""
    from m2h.common.utils import CustomFixedCategorical
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    for M in (1, 14, 280):
        probs = torch.softmax(torch.randn(M, 3, generator=g) * 2.0, dim=1).to(dev)
        d = CustomFixedCategorical(torch.log(probs), probs, torch.zeros(M, device=dev))
        torch.manual_seed(1234 + M)
        ref = [torch.multinomial(probs, 1, True) for _ in range(5)]
        ref_after = torch.rand(4, device=dev)
        torch.manual_seed(1234 + M)
        got = [d.sample() for _ in range(5)]
        got_after = torch.rand(4, device=dev)
        for a, b in zip(got, ref):
            assert a.dtype == torch.int64 and a.shape == (M, 1) and torch.equal(a, b)
        assert torch.equal(ref_after, got_after)


def test_cpu_generator_sampling_is_the_reference_cpu_draw_bit_for_bit():
    """cpu_generator mode (north-star: action sampling bit-exact against the reference PyTorch-CPU path): for probabilities held
    on the device, CustomFixedCategorical.sample with a HostNoise source returns exactly what ``torch.multinomial(probs.cpu(), 1,
    True)`` -- the reference's Categorical.sample on a CPU policy (common/utils.py:16-24) -- returns for the same CPU generator
    state, call after call, interleaved with torch.randperm as the training loop interleaves them, and leaves the CPU
    generator in the same state.  Includes rows with exact ties and a row count that wraps the pinned ring."""
    from m2h.common.utils import CustomFixedCategorical, HostNoise
    dev = _dev()
    g = torch.Generator().manual_seed(22)
    for M in (1, 3, 14, 280):
        probs = torch.softmax(torch.randn(M, 3, generator=g) * 2.0, dim=1)
        if M >= 3:
            probs[1] = torch.tensor([0.25, 0.5, 0.25])
        noise = HostNoise(dev)
        d = CustomFixedCategorical(torch.log(probs).to(dev), probs.to(dev), torch.zeros(M, device=dev), noise)
        torch.manual_seed(4321 + M)
        ref = []
        for i in range(2 * HostNoise.RING + 3):
            ref.append(torch.multinomial(probs, 1, True))
            if i % 4 == 3:
                torch.randperm(14)
        ref_after = torch.rand(4)
        torch.manual_seed(4321 + M)
        got = []
        for i in range(2 * HostNoise.RING + 3):
            got.append(d.sample())
            if i % 4 == 3:
                torch.randperm(14)
        got_after = torch.rand(4)
        for a, b in zip(got, ref):
            assert a.dtype == torch.int64 and a.shape == (M, 1) and torch.equal(a.cpu(), b)
        assert torch.equal(ref_after, got_after)
    # known answer of the host draw (ATen's default exponential path: -log1p(-u), u a 53-bit uniform of the mt19937 stream): the
    # fixtures under tests/golden were sampled with it, so a platform where this differs cannot reproduce them
    torch.manual_seed(0)
    q = torch.empty(2, 3).exponential_(1)
    assert torch.allclose(q, torch.tensor([[3.5083, 1.2304, 0.6150], [2.5351, 1.0357, 1.5661]]), atol=5e-5), q
    # ties go to the lowest index, a zero-probability class is never drawn
    p = torch.tensor([[0.5, 0.5, 0.0], [0.0, 0.0, 1.0]], device=dev)
    from m2h import ops
    assert ops.sample_actions(p, torch.ones(2, 3, device=dev)).view(-1).tolist() == [0, 2]


@pytest.mark.parametrize("M,K,N,slope", [(14, 1536, 1536, 1.0), (1, 1536, 1536, 1.0), (16, 512, 512, 0.0), (5, 4608, 512, 0.0)])
def test_skinny_linear_kernel_matches_the_engine_and_torch(M, K, N, slope):
    """M <= 16 dense rows (Linear at the rollout width, full-spatial conv as Linear): both operands straight into the 16x16x4
    fp32 MFMA, against the tiled engine (knob 23 = -1) and torch on the CPU."""
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g) * 0.1
    ref = x @ w.t() + b
    ref = torch.relu(ref) if slope == 0.0 else ref
    got = {}
    for knob in (0, -1):
        ops.debug_set(23, knob)
        try:
            if K == 4608:   # VisualCNN's fc: a 12x12x32 map under a 12x12 kernel (visual_cnn.py:140-141)
                got[knob] = ops.conv2d_nhwc(x.view(M, 12, 12, 32).to(dev), w.to(dev), N, 12, 12, bias=b.to(dev), slope=slope).reshape(M, N).cpu()
            else:
                got[knob] = ops.linear(x.to(dev), w.to(dev), b.to(dev), slope=slope).cpu()
        finally:
            ops.debug_set(23, 0)
    assert _rel(got[0], ref) < 2e-5 and _rel(got[-1], ref) < 2e-5 and _rel(got[0], got[-1]) < 1e-5


@pytest.mark.parametrize("override,extra", [(True, False), (True, True), (False, False)])
def test_fused_rollout_step_stats_matches_the_separate_kernels(override, extra):
    """m2h_rollout_step_stats (one launch: rewards, three STFT-L2 distances, episode statistics) against the separate kernels it
    replaces in the trainer (sq_stats, rewards_from_stats, stft_l2, episode_stats_update -- themselves pinned to the reference
    fixtures above): rewards / losses to fp32 summation order, statistics accordingly; repeated launches are bit-identical (the
    partial sums are added in slice order whatever the arrival order) and the ticket counters return to zero."""
    from types import SimpleNamespace
    from m2h import _lib, ops
    dev = _dev()
    N, A, L = 14, 3, 512 * 32
    g = torch.Generator().manual_seed(31)
    obs, nobs = _obs(N, 71, dev), _obs(N, 72, dev)
    pm = torch.randn(N, 512, 32, 2, generator=g).to(dev)
    mono, mem, nmem = (torch.rand(N, 512, 32, 1, generator=g).to(dev) for _ in range(3))
    probs = torch.softmax(torch.randn(N, A, generator=g), 1).to(dev)
    masks = (torch.rand(N, 1, generator=g) > 0.3).float().to(dev)
    env_r, ndg, dg = (torch.randn(N, 1, generator=g).to(dev) for _ in range(3))
    mk = lambda: SimpleNamespace(**{n: (torch.rand(N, A if "dist_probs" in n else 1, generator=torch.Generator().manual_seed(5)) + 1).to(dev)  # noqa: E731
                                    for n in _lib.EPISODE_STATS_FIELDS})
    ref, fused, fused2 = mk(), mk(), mk()
    if override:
        nxt = ops.sq_stats(nmem, nobs["gt_mono_comps"], 0)
        if extra:
            r_ref = ops.rewards_from_stats(nxt, None, masks, L, False, 20.0)
        else:
            r_ref = ops.rewards_from_stats(nxt, ops.sq_stats(mem, obs["gt_mono_comps"], 0), masks, L, True)
    else:
        r_ref = env_r
    bl = ops.stft_l2(pm, obs["gt_bin_comps"], 2, mix=obs["mixed_bin_audio_mag"])
    ml = ops.stft_l2(mono, obs["gt_mono_comps"], 1)
    fl = ops.stft_l2(mem, obs["gt_mono_comps"], 1)
    ops.episode_stats_update(ref, r_ref, probs, bl, ml, fl, masks, ndg, dg)
    outs = []
    for st in (fused, fused2):
        outs.append(ops.rollout_step_stats(st, nmem, nobs["gt_mono_comps"], mem, obs["gt_mono_comps"], pm, obs["mixed_bin_audio_mag"],
                                           obs["gt_bin_comps"], mono, masks, probs, env_rewards=env_r, ndgs=ndg, dgs=dg, override=override,
                                           extra=extra, extra_mult=20.0))
    (r, losses), (r2, losses2) = outs
    assert torch.equal(r, r2) and torch.equal(losses, losses2)
    for n in _lib.EPISODE_STATS_FIELDS:
        assert torch.equal(getattr(fused, n), getattr(fused2, n)), n
    close = lambda x, y: torch.allclose(x.cpu().reshape(-1), y.cpu().reshape(-1), rtol=2e-6, atol=1e-7)  # noqa: E731
    # (the quality-improvement reward is a difference of two near-equal ratios: absolute tolerance of a few ulps of those ratios)
    assert torch.allclose(r.cpu(), r_ref.cpu(), rtol=2e-6, atol=2e-6) and close(losses[0], bl) and close(losses[1], ml) and close(losses[2], fl)
    if not override:
        assert torch.equal(r, env_r)
    assert float(r[masks == 0].abs().sum()) == 0 or not override
    for n in _lib.EPISODE_STATS_FIELDS:
        assert torch.allclose(getattr(fused, n).cpu(), getattr(ref, n).cpu(), rtol=3e-6, atol=3e-6), n
    # the tickets reset themselves; the scratch belongs to (device, stream, N): launches on one stream are ordered, two streams never share it
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, N)
    assert int(ops._step_stats_scratch[key][1].abs().sum()) == 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        ops.rollout_step_stats(mk(), nmem, nobs["gt_mono_comps"], mem, obs["gt_mono_comps"], pm, obs["mixed_bin_audio_mag"], obs["gt_bin_comps"], mono,
                               masks, probs, env_rewards=env_r, ndgs=ndg, dgs=dg, override=override, extra=extra, extra_mult=20.0)
        assert (dev.index, side.cuda_stream, N) in ops._step_stats_scratch and ops._step_stats_scratch[(dev.index, side.cuda_stream, N)][0].data_ptr() != \
            ops._step_stats_scratch[key][0].data_ptr()
    torch.cuda.current_stream(dev).wait_stream(side)


from kernel_model import philox_exp1 as _philox_exp1  # noqa: E402


def test_fused_action_draw_is_the_multinomial_draw_on_philox_noise():
    """"fused" sampling (Policy.set_action_sampling): the heads kernel draws argmax(probs / Exp(1) noise) -- torch.multinomial's single-draw
    path, common/utils.py:16-24 -- with the noise from Philox4x32-10 at (seed, counter + row A + a): equal to the same draw on the host's
    restatement of that generator, distributed as the probabilities, and a different counter gives other draws."""
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    M, H, A = 4096, 512, 3
    feats = torch.randn(M, H, generator=g).to(dev)
    Wa, ba = (torch.randn(A, H, generator=g) * 0.05).to(dev), (torch.randn(A, generator=g) * 0.1).to(dev)
    Wc, bc = (torch.randn(1, H, generator=g) * 0.05).to(dev), torch.zeros(1, device=dev)
    seed, ctr = 0x5eed0007, 12345
    rng = torch.tensor([seed, ctr], dtype=torch.int64, device=dev)
    drawn = torch.zeros(M, A, device=dev)
    value, logp_all, probs, ent, action, logp_act = ops.policy_heads_act(feats, Wa, ba, Wc, bc, rng=rng, noise_out=drawn)
    assert int(rng[1]) == ctr                                     # the kernel reads the counter; the caller advances it
    # (1) the noise the kernel drew IS the generator's: -log(u) of the host restatement's u, to the device logf's last bits, and u -- 23
    # random bits + 0.5, exact in fp32 -- never reaches 0 or 1 (0xFFFFFFFF included: the largest word gives u = 1 - 2^-24, noise 6e-8 > 0)
    noise, u = _philox_exp1(seed, ctr + np.arange(M * A, dtype=np.uint64))
    got = drawn.cpu().numpy().reshape(-1)
    assert np.abs(got - noise).max() <= 4e-7 * np.abs(noise).max() and (np.abs(got - noise) <= 3e-7 * np.maximum(noise, 1e-3)).all()
    assert u.min() >= 2.0 ** -24 and u.max() <= 1 - 2.0 ** -24 and got.min() > 0 and np.isfinite(got).all()
    top = (np.float32(0x7FFFFF) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    assert top < 1 and -np.log(top) > 0
    # (2) the action IS torch.multinomial's single draw on that noise, exactly: argmax(probs / noise) in correctly rounded fp32 division
    # on both sides, ties to the lowest index -- every one of the 4096 rows, no tolerance
    want = (probs.cpu().numpy() / drawn.cpu().numpy()).argmax(1)
    assert np.array_equal(action.cpu().numpy().reshape(-1), want)
    # without the record the same draw
    action_b = ops.policy_heads_act(feats, Wa, ba, Wc, bc, rng=rng)[4]
    assert torch.equal(action_b, action)
    assert torch.equal(logp_act, logp_all.gather(1, action))
    v0, lp0, p0, e0, a_mode, _ = ops.policy_heads_act(feats, Wa, ba, Wc, bc)       # the mode: same heads, no draw
    assert torch.equal(p0, probs) and torch.equal(v0, value)
    # distribution: per-action frequency against the mean probability (M = 4096 rows: 4 sigma of a binomial is ~0.03)
    freq = torch.bincount(action.reshape(-1), minlength=A).float() / M
    assert (freq.cpu() - probs.mean(0).cpu()).abs().max() < 0.035, (freq, probs.mean(0))
    rng2 = torch.tensor([seed, ctr + M * A], dtype=torch.int64, device=dev)
    action2 = ops.policy_heads_act(feats, Wa, ba, Wc, bc, rng=rng2)[4]
    assert (action2 != action).float().mean() > 0.2
    ops.step_index_advance(torch.zeros(3, dtype=torch.int64, device=dev), 4, 4, rng=rng, rng_inc=M * A)
    assert int(rng[1]) == ctr + M * A and int(rng[0]) == seed
