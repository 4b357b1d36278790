"""GPU: the one-launch kernels of round 4, each directly against the CPU oracle and against the separate kernels it replaces --
AcousticMem's rollout forward (memory_nets.py:40-69), the no-grad GRU cell (rnn_state_encoder.py:74-84), Policy.act's heads + draw +
log-probability (rl/ppo/policy.py:217-225), the BPTT step with the previous step's gate backward (nn.GRU autograd), the activation
backward folded into the image-row weight gradient (ppo.py:228-230), and update_sep on the separator outputs the rollout stored
(ppo.py:184-195 without the recompute), and the image-row 3x3 kernels of update_sep in bf16x3 arithmetic against the fp32 ones and torch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import m2h_oracle as O
from m2h import synthetic

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _rel(a, b):
    return O.rel_l1(torch.as_tensor(a).cpu(), torch.as_tensor(b).cpu())


def _policy(seed, dev):
    from m2h.common.spaces import Discrete, move2hear_observation_space
    from m2h.rl.ppo.policy import Move2HearPolicy
    pol = Move2HearPolicy(move2hear_observation_space(), Discrete(3), "spectrogram", 512, False, True, use_ddppo=True)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), seed).items()}
    pol.load_state_dict(sd, strict=True)
    return pol.to(dev), sd


@pytest.mark.parametrize("B", [1, 3, 14, 33])
def test_acoustic_mem_one_launch_matches_the_oracle_and_the_tiled_path(B):
    from m2h import ops
    dev = _dev()
    pol, sd = _policy(6, dev)
    g = torch.Generator().manual_seed(B)
    mono, prev = torch.rand(B, 512, 32, 1, generator=g) * 2, torch.rand(B, 512, 32, 1, generator=g) * 2
    nd = (torch.rand(B, 1, generator=g) > 0.3).float()
    want = O.acoustic_mem(sd, mono, O.mask_prev_mem(prev, nd))
    with torch.no_grad():
        got = pol.get_monoFromMem_masked(mono.to(dev), prev.to(dev), nd.to(dev))
        assert ops.last_kernel() == "acoustic_mem_small"
        sliced = pol.acoustic_mem.slice_inputs(mono.to(dev), prev.to(dev), nd.to(dev))
        tiled = pol.get_monoFromMem_masked(mono.to(dev), prev.to(dev), nd.to(dev), sliced=sliced)   # the update batch's route
        assert ops.last_kernel() != "acoustic_mem_small"
        unmasked = pol.get_monoFromMem(mono.to(dev), (prev * nd.view(-1, 1, 1, 1)).to(dev))
    assert got.shape == (B, 512, 32, 1)
    assert _rel(got, want) < 2e-6 and _rel(got, tiled) < 2e-6 and _rel(unmasked, want) < 2e-6


@pytest.mark.parametrize("N", [1, 7, 14, 16])
def test_gru_cell_one_launch_matches_the_oracle_and_the_two_kernel_path(N):
    from m2h import functional as MF
    from m2h import ops
    dev = _dev()
    pol, sd = _policy(2, dev)
    enc = pol.pol_net.state_encoder
    g = torch.Generator().manual_seed(N)
    x, h = torch.randn(N, 1536, generator=g), torch.randn(1, N, 512, generator=g)
    m = (torch.rand(N, 1, generator=g) > 0.4).float()
    want = O.gru_cell(sd, x, h[0] * m)
    with torch.no_grad():
        out, hn = enc(x.to(dev), h.to(dev), m.to(dev))
        assert ops.last_kernel() == "gru_cell"
        r = enc.rnn
        two, _ = MF.GRUSequence.apply(x.to(dev), h[0].to(dev), m.to(dev), r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0, 1)
    assert out.shape == (N, 512) and hn.shape == (1, N, 512) and torch.equal(out, hn[0])
    assert _rel(out, want) < 2e-6 and _rel(out, two) < 2e-6


def test_heads_act_one_launch_is_the_separate_ops_bit_for_bit():
    """values, probabilities, the multinomial draw for given noise, the mode, and the log-probability of the action taken."""
    from m2h import ops
    dev = _dev()
    pol, sd = _policy(5, dev)
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(14, 512, generator=g).to(dev) * 4
    a, c = pol.action_dist.linear, pol.critic.fc
    noise = torch.empty(14, 3).exponential_(1, generator=g).to(dev)
    with torch.no_grad():
        v0, lpa0, p0, e0, _ = ops.policy_heads(feats, a.weight, a.bias, c.weight, c.bias)
        act0 = ops.sample_actions(p0, noise)
        lp0 = ops.gather_logp(lpa0, act0)
        v1, lpa1, p1, e1, act1, lp1 = ops.policy_heads_act(feats, a.weight, a.bias, c.weight, c.bias, noise)
        assert all(torch.equal(x, y) for x, y in ((v0, v1), (lpa0, lpa1), (p0, p1), (e0, e1), (act0, act1), (lp0, lp1)))
        assert torch.equal(act1, torch.argmax(p1 / noise, dim=-1, keepdim=True)) and act1.dtype == torch.int64
        _, _, p2, _, act2, lp2 = ops.policy_heads_act(feats, a.weight, a.bias, c.weight, c.bias, None)      # the mode
        assert torch.equal(act2, p2.argmax(dim=-1, keepdim=True)) and torch.equal(lp2, ops.gather_logp(lpa1, act2))
    wv, wl, wp = O.heads(sd, feats.cpu())
    assert _rel(v1, wv) < 2e-6 and _rel(p1, wp) < 2e-6
    assert len(set(act1.reshape(-1).tolist())) > 1   # (the noise does move the draw)


def test_bptt_step_with_fused_gate_backward_is_the_two_kernels_bit_for_bit():
    from m2h import _lib, ops
    dev = _dev()
    N, H = 14, 512
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    gi, gh, bhh, hp2, dpre_t, whh_t, a_prev, dhp, nxt_dh = r(N, 3 * H), r(N, 3 * H), r(3 * H), r(N, H), r(N, 3 * H) * 0.1, r(H, 3 * H) * 0.05, r(N, H), r(N, H), None
    mask_t, mask_p = (torch.rand(N, generator=g) > 0.3).float().to(dev), (torch.rand(N, generator=g) > 0.3).float().to(dev)
    lib = _lib.load()
    p, st = ops._ptr, ops._stream(gi)
    # separate: recurrent backward of step t, then the gate backward of step t-1 on its output
    out0, dhp0 = torch.empty(N, H, device=dev), dhp.clone()
    _lib.check(lib.m2h_gru_bwd_rec(p(dpre_t), p(whh_t), p(a_prev), p(dhp0), p(mask_t), p(out0), N, H, st), "rec")
    dgi0, dpre0, hpm0, dhp_n0 = (torch.empty(N, 3 * H, device=dev), torch.empty(N, 3 * H, device=dev), torch.empty(N, H, device=dev), torch.empty(N, H, device=dev))
    _lib.check(lib.m2h_gru_gates_bwd(p(gi), p(gh), p(bhh), p(hp2), p(mask_p), p(out0), p(dgi0), p(dpre0), p(dhp_n0), p(hpm0), N, H, st), "gates")
    # fused
    out1, dhp1 = torch.empty(N, H, device=dev), dhp.clone()
    dgi1, dpre1, hpm1 = torch.empty(N, 3 * H, device=dev), torch.empty(N, 3 * H, device=dev), torch.empty(N, H, device=dev)
    _lib.check(lib.m2h_gru_bwd_step(p(dpre_t), p(whh_t), p(a_prev), p(dhp1), p(mask_t), p(out1), p(gi), p(gh), p(bhh), p(hp2), p(mask_p), p(dgi1),
                                    p(dpre1), p(hpm1), N, H, st), "step")
    for x, y in ((out0, out1), (dgi0, dgi1), (dpre0, dpre1), (hpm0, hpm1), (dhp_n0, dhp1)):
        assert torch.equal(x, y)
    with pytest.raises(RuntimeError, match="must not alias"):
        _lib.check(lib.m2h_gru_bwd_step(p(dpre_t), p(whh_t), p(a_prev), p(dhp1), p(mask_t), p(out1), p(gi), p(gh), p(bhh), p(hp2), p(mask_p), p(dgi1),
                                        p(dpre_t), p(hpm1), N, H, st), "step")


@pytest.mark.parametrize("N,slope", [(32, 0.0), (16, 0.2)])
def test_gated_weight_gradient_is_act_bwd_then_wgrad_bit_for_bit(N, slope):
    from m2h import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(N)
    B, Hh = 20, 32
    x, dy, y = torch.randn(B, Hh, 32, 32, generator=g).to(dev), torch.randn(B, Hh, 32, N, generator=g).to(dev), torch.randn(B, Hh, 32, N, generator=g).to(dev)
    two = MF.conv_wgrad(x, None, MF.act_bwd(dy, y, slope), N, 3, 3, 1, 1)
    one = MF.conv_wgrad(x, None, dy, N, 3, 3, 1, 1, gate=y, gate_slope=slope)
    assert torch.equal(one, two)
    ref = F.conv2d(x.cpu().permute(0, 3, 1, 2), torch.zeros(N, 32, 3, 3, requires_grad=True), None, 1, 1)   # shape check of the layout only
    assert one.shape == (N, 9 * 32) and ref.shape == (B, N, Hh, 32)
    with pytest.raises(RuntimeError, match="image-row 3x3 kernel only"):     # other shapes are refused, not silently ungated
        MF.conv_wgrad(torch.randn(4, 8, 8, 64, device=dev), None, torch.randn(4, 8, 8, 32, device=dev), 32, 3, 3, 1, 1,
                      gate=torch.randn(4, 8, 8, 32, device=dev), gate_slope=0.0)


def test_update_sep_on_stored_separator_outputs_equals_the_recompute():
    """The trainer's rollout leaves every stored observation's separator outputs beside it; update_sep on them gives the losses and the
    post-update memory weights of update_sep recomputing them over the buffer (same networks, same observations: fp32 association only)."""
    from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
    dev = _dev()
    out = {}
    for stored in (True, False):
        cfg = near_target_config(num_updates_per_cycle=1, num_steps=6, NUM_PROCESSES=5, use_hip_graphs=True, action_sampling="cpu_generator")
        tr = PPOTrainer(cfg, dev)
        tr.setup()
        tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
        if not stored:
            tr.rollouts_sep.pred_mono = tr.rollouts_sep.pred_binSepMasks = tr.rollouts_sep._pred_rows_valid = None   # a storage without them
        torch.manual_seed(0)
        for _ in range(cfg.num_steps):
            tr._collect_rollout_step()
        assert (tr.rollouts_sep.stored_separator_outputs() is not None) == stored
        losses = tr._update_sep()
        out[stored] = (losses, {k: v.detach().clone() for k, v in tr.actor_critic.acoustic_mem.state_dict().items()})
    for a, b in zip(out[True][0], out[False][0]):
        assert abs(a - b) < 2e-6 * max(1.0, abs(b))
    for k in out[True][1]:
        assert (out[True][1][k] - out[False][1][k]).abs().max().item() < 2e-6, k


# ---------------------------------------------------------------------------------------------------------------- bf16x3 image-row kernels
def _amem_case(B, seed, dev):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 32, 32, 32, generator=g) * 2            # the sliced, concatenated input (non-negative magnitudes)
    w0 = torch.randn(32, 32, 3, 3, generator=g) * 0.08
    w1 = torch.randn(16, 32, 3, 3, generator=g) * 0.08
    return x.to(dev), w0.to(dev), w1.to(dev)


@pytest.mark.parametrize("B", [64, 97])
def test_image_row_convs_in_bf16x3_match_fp32_and_torch(B):
    """memory_nets.py:11-16 over an update batch: conv3x3 + ReLU (32 -> 32, NHWC), conv3x3 (32 -> 16, de-sliced store) and the second
    conv's input gradient (16 -> 32, taps mirrored), each by the fp32-MFMA image-row kernel and by its bf16x3 twin."""
    from m2h import functional as MF
    from m2h import ops
    dev = _dev()
    x, w0, w1 = _amem_case(B, B, dev)
    dy = (torch.sign(torch.randn(B, 512, 32, 1, generator=torch.Generator().manual_seed(B))) / (B * 16384)).to(dev)   # an L1 loss's gradient
    res = {}
    for mode in (ops.MATH_FP32, ops.MATH_BF16X3):
        with ops.math_scope(mode), torch.no_grad():
            h = MF.conv2d(x, w0, None, 1, 1, slope=0.0)
            k0 = ops.last_kernel()
            y = MF.conv2d(h, w1, None, 1, 1, slope=1.0, deslice=True)
            k1 = ops.last_kernel()
            dh = MF.conv_dgrad(ops.slice_concat_input(dy, op=0), w1, (32, 32), 1, 1)
            k2 = ops.last_kernel()
        want = "conv_igemm_bf16x3 (image-row 3x3)" if mode == ops.MATH_BF16X3 else "conv_igemm_f32 (image-row 3x3)"
        assert (k0, k1, k2) == (want, want, want), (k0, k1, k2)
        res[mode] = (h, y, dh)
    for a, b in zip(res[ops.MATH_BF16X3], res[ops.MATH_FP32]):
        assert a.shape == b.shape and _rel(a, b) < 2e-5
    xc = x.cpu().double().permute(0, 3, 1, 2)
    hr = F.relu(F.conv2d(xc, w0.cpu().double(), None, 1, 1))
    yr = F.conv2d(hr, w1.cpu().double(), None, 1, 1)
    h16, y16, _ = res[ops.MATH_BF16X3]
    assert _rel(h16.cpu().double().permute(0, 3, 1, 2), hr) < 2e-5
    assert _rel(y16.cpu().double(), yr.reshape(B, 512, 32, 1)) < 2e-5


@pytest.mark.parametrize("B,N", [(64, 16), (64, 32), (131, 32), (19, 16)])
def test_image_row_weight_gradient_in_bf16x3_matches_fp32_and_torch(B, N):
    """nn.Conv2d's weight gradient over an update batch (ppo.py:228-230), plain and with the layer's ReLU derivative folded in (the
    SAME gate tensor in both modes: a gate that flips with the arithmetic is a different sub-gradient, not an error of the kernel)."""
    from m2h import functional as MF
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B + N)
    x = (torch.rand(B, 32, 32, 32, generator=g) * 2).to(dev)
    dy = torch.randn(B, 32, 32, N, generator=g).to(dev)
    gate = torch.randn(B, 32, 32, N, generator=g).to(dev)
    out = {}
    for mode in (ops.MATH_FP32, ops.MATH_BF16X3):
        with ops.math_scope(mode):
            out[mode] = (MF.conv_wgrad(x, None, dy, N, 3, 3, 1, 1), MF.conv_wgrad(x, None, dy, N, 3, 3, 1, 1, gate=gate, gate_slope=0.0))
    for a, b in zip(out[ops.MATH_BF16X3], out[ops.MATH_FP32]):
        assert a.shape == (N, 288) and _rel(a, b) < 2e-5
    w = torch.zeros(N, 32, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.cpu().double().permute(0, 3, 1, 2), w, None, 1, 1).backward((dy * (gate > 0)).cpu().double().permute(0, 3, 1, 2))
    want = w.grad.permute(0, 2, 3, 1).reshape(N, 288)              # packed [n][(kh, kw, c)]
    assert _rel(out[ops.MATH_BF16X3][1].cpu().double(), want) < 2e-5 and _rel(out[ops.MATH_FP32][1].cpu().double(), want) < 2e-6


def test_update_sep_in_bf16x3_follows_the_fp32_update():
    """sep_update_math="bf16x3" (build-side config key): the same update_sep -- losses and post-update memory weights -- as in fp32,
    to the arithmetic's 1e-5; the rollout (which produced the storage) is untouched by the key."""
    from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
    from m2h import ops
    dev = _dev()
    out = {}
    for sm in (None, "bf16x3"):
        cfg = near_target_config(num_updates_per_cycle=1, num_steps=8, NUM_PROCESSES=8, use_hip_graphs=True, action_sampling="cpu_generator", sep_update_math=sm)
        tr = PPOTrainer(cfg, dev)
        tr.setup()
        tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 1).items()})
        torch.manual_seed(0)
        for _ in range(cfg.num_steps):
            tr._collect_rollout_step()
        losses = tr._update_sep()
        assert ops.math_mode() == ops.MATH_FP32        # the scope ends with the update
        out[sm] = (losses, {k: v.detach().clone() for k, v in tr.actor_critic.acoustic_mem.state_dict().items()})
    for a, b in zip(out["bf16x3"][0], out[None][0]):
        assert abs(a - b) < 1e-5 * max(1.0, abs(b))
    moved = 0.0
    for k in out[None][1]:
        assert (out["bf16x3"][1][k] - out[None][1][k]).abs().max().item() < 2e-5, k
        moved = max(moved, (out[None][1][k] - torch.from_numpy(np.asarray(synthetic.make_state_dict(synthetic.policy_shapes(), 1)["acoustic_mem." + k])).to(dev)).abs().max().item())
    assert moved > 1e-4                                 # (the update did move the weights)


@pytest.mark.parametrize("M,N", [(1 << 20, 2), (65536, 32), (269080, 32), (1000, 1), (4099, 4), (70000, 16), (5000, 8), (3000, 48), (2049, 64), (700, 2)])
def test_bias_gradient_column_sums_narrow_and_wide(M, N):
    """nn.Conv2d's bias gradient = column sums of dY (ppo.py:228-230 through autograd): the narrow kernel (N dividing 64: a wave reads
    whole rows, 64 floats a step -- the U-Net heads' 2 channels over a million pixels) and the column-walking one, against float64."""
    from m2h import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(M + N)
    dy = torch.randn(M, N, generator=g)
    got = MF.bias_grad(dy.to(dev))
    want = dy.double().sum(0)
    assert got.shape == (N,)
    assert (got.cpu().double() - want).abs().max().item() < 2e-6 * max(1.0, dy.abs().double().sum(0).max().item())
    assert torch.equal(got, MF.bias_grad(dy.to(dev)))          # (ordered reduction: the same bits every time)


def test_cycle_tail_overlap_leaves_the_same_bits():
    """overlap_update_tail (build-side config key): the cycle's six update_sep on a second stream beside the last update_pol.  Same kernels
    on the same data in the same per-stream order: losses, every weight, both storages and the CPU generator's state equal the
    one-stream cycle's bit for bit, over three cycles (a race between the two streams would show as a difference here)."""
    from m2h.rl.ppo.ppo_trainer import PPOTrainer, near_target_config
    dev = _dev()
    out = {}
    for overlap in (False, True):
        cfg = near_target_config(num_updates_per_cycle=3, num_steps=6, NUM_PROCESSES=6, MAX_EPISODE_STEPS=6, use_hip_graphs=True, action_sampling="cpu_generator",
                                 overlap_update_tail=overlap, sep_update_math="bf16x3")
        tr = PPOTrainer(cfg, dev)
        tr.setup()
        tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.policy_shapes(), 3).items()})
        torch.manual_seed(7)
        res = [tr.train_cycle() for _ in range(3)]
        assert (tr._tail_stream is not None) == overlap
        tr.agent.synchronize_updates()
        torch.cuda.synchronize()
        out[overlap] = ([(r["pol_losses"], r["sep_losses"]) for r in res], {k: v.detach().clone() for k, v in tr.actor_critic.state_dict().items()},
                        tr.rollouts_sep.prev_pred_monoFromMem.clone(), tr.rollouts_pol.rewards.clone(), torch.get_rng_state().clone(),
                        tr.num_sep_updates_done, tr.num_updates_done)
    a, b = out[False], out[True]
    assert a[0] == b[0], (a[0], b[0])
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]) and a[5:] == b[5:]


@pytest.mark.parametrize("B,H,W,c_in,Ci,Co,k,st,pad", [(280, 31, 31, 32, 32, 64, 4, 2, 0), (14, 128, 128, 4, 3, 32, 8, 4, 0), (64, 32, 32, 32, 32, 32, 3, 1, 1),
                                                      (3, 16, 16, 64, 64, 128, 4, 2, 1), (280, 14, 14, 64, 64, 32, 3, 1, 0), (2, 8, 8, 16, 16, 16, 3, 1, 1)])
def test_weight_gradient_in_the_torch_layout_is_the_packed_gradient_permuted(B, H, W, c_in, Ci, Co, k, st, pad):
    """m2h_conv_wgrad_torch_f32: split sum + re-layout in one launch == m2h_conv_wgrad_f32 followed by view / permute / contiguous, bit for
    bit (few and many splits, 9 / 16 / 64 taps, a channel-padded input whose padding channels carry no gradient), and == torch."""
    from m2h import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(B + Co)
    x = torch.randn(B, H, W, c_in, generator=g)
    if Ci < c_in:
        x[..., Ci:] = 0
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    dy = torch.randn(B, Ho, Wo, Co, generator=g) / (B * Ho * Wo) ** 0.5
    xd, dyd = x.to(dev), dy.to(dev)
    packed = MF.conv_wgrad(xd, None, dyd, Co, k, k, st, pad)
    two = packed.view(Co, k, k, c_in)[..., :Ci].permute(0, 3, 1, 2).contiguous()
    one = MF.conv_wgrad(xd, None, dyd, Co, k, k, st, pad, torch_ci=Ci)
    assert one.shape == (Co, Ci, k, k) and torch.equal(one, two)
    w = torch.zeros(Co, Ci, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x[..., :Ci].double().permute(0, 3, 1, 2), w, None, st, pad).backward(dy.double().permute(0, 3, 1, 2))
    assert _rel(one.cpu().double(), w.grad) < 2e-6


def test_weight_gradients_are_born_in_the_optimizers_flat_buffer():
    """functional.grad_slot: a conv / transposed-conv weight owned by a FlatAdam gets its gradient written straight into its slice of the
    flat gradient buffer (no gather copy); a second backward before zero_grad() accumulates the usual way; parameters without a
    gradient are zeroed slice by slice without touching the resident ones; the step equals torch.optim.Adam's."""
    from m2h import functional as MF
    from m2h.optim import FlatAdam
    dev = _dev()
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(8, 16, 4, 2, 1, bias=False).to(dev)
    convT = torch.nn.ConvTranspose2d(16, 8, 4, 2, 1, bias=False).to(dev)
    unused = torch.nn.Parameter(torch.randn(5, device=dev))
    ref = [p.detach().clone().requires_grad_(True) for p in (conv.weight, convT.weight, unused)]
    opt = FlatAdam([conv.weight, convT.weight, unused], lr=1e-2, eps=1e-5)
    ropt = torch.optim.Adam(ref, lr=1e-2, eps=1e-5)
    x = torch.randn(3, 16, 16, 8, device=dev)

    def loss_m2h():
        h = MF.conv2d(x, conv.weight, None, 2, 1, slope=0.0)
        return MF.conv_transpose2d(h, convT.weight).square().mean()

    def loss_ref():
        h = F.relu(F.conv2d(x.permute(0, 3, 1, 2), ref[0], None, 2, 1))
        return F.conv_transpose2d(h, ref[1], None, 2, 1).square().mean()

    for step in range(3):
        opt.zero_grad()
        loss_m2h().backward()
        for p, off in zip(opt._ps[:2], opt._offsets[:2]):
            assert p.grad.data_ptr() == opt.flat_g[off:off + p.numel()].data_ptr(), "gradient not resident in the flat buffer"
        if step == 1:                              # a second backward pass: accumulated by autograd, not overwritten
            g0 = [p.grad.clone() for p in opt._ps[:2]]
            loss_m2h().backward()
            for p, g in zip(opt._ps[:2], g0):
                assert _rel(p.grad, 2 * g) < 1e-6
            ropt.zero_grad()
            (2 * loss_ref()).backward()
        else:
            ropt.zero_grad()
            loss_ref().backward()
        assert unused.grad is None
        opt.step(max_grad_norm=0.5)
        torch.nn.utils.clip_grad_norm_(ref[:2], 0.5)
        ropt.step()
        for p, r in zip(opt._ps, ref):
            assert _rel(p.detach(), r.detach()) < 2e-5, step


@pytest.mark.parametrize("B,T", [(3, 32), (5, 20)])
def test_l1_loss_in_the_convs_layout_matches_torch(B, T):
    """m2h_l1_loss_nhwc16: F.l1_loss(deslice(y), gt[..., off]) and its gradient for y in NHWC [B, 32, T, 16] (memory_nets.py:62-67, ppo.py:212-216)."""
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * T)
    y = torch.randn(B, 32, T, 16, generator=g)
    gt = torch.randn(B, 512, T, 2, generator=g)
    y[0, 0, 0, 0] = gt[0, 0, 0, 1]                                           # an exact tie: sign(0) = 0 like torch
    yr = y.clone().double().requires_grad_(True)
    pred = yr.permute(0, 3, 1, 2).reshape(B, 512, T, 1)                      # band-major frequency axis: f = band * 32 + row
    want = F.l1_loss(pred, gt[..., 1:2].double())
    want.backward()
    loss, dy = ops.l1_loss_nhwc16(y.to(dev), gt.to(dev), 1)
    assert abs(loss.item() - want.item()) < 1e-6 * max(1.0, abs(want.item()))
    assert torch.equal(dy.cpu(), yr.grad.float()) and dy[0, 0, 0, 0].item() == 0.0
    loss2, none = ops.l1_loss_nhwc16(y.to(dev), gt.to(dev), 1, want_grad=False)
    assert none is None and loss2.item() == loss.item()


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_memory_loss_without_the_desliced_output_equals_the_two_step_form(mode):
    """AcousticMem.l1_loss_masked (ConvL1NHWC16) == F.l1_loss(forward_masked(...), gt[..., 0:1]): the loss to summation order, both weight
    gradients bit for bit (the same conv values, the same gradient signs, the same weight-gradient kernels)."""
    from m2h import functional as MF
    from m2h import ops
    dev = _dev()
    pol, sd = _policy(8, dev)
    mem = pol.acoustic_mem
    B = 70
    g = torch.Generator().manual_seed(2)
    mono, prev = (torch.rand(B, 512, 32, 1, generator=g) * 2).to(dev), (torch.rand(B, 512, 32, 1, generator=g) * 2).to(dev)
    nd = (torch.rand(B, 1, generator=g) > 0.3).float().to(dev)
    gt = torch.rand(B, 512, 32, 2, generator=g).to(dev)
    out = {}
    with ops.math_scope(ops.MATH_BF16X3 if mode == "bf16x3" else ops.MATH_FP32):
        for fused in (False, True):
            for p in mem.parameters():
                p.grad = None
            if fused:
                loss = pol.monoFromMem_l1_masked(mono, prev, nd, gt, 0)
            else:
                loss = MF.l1_loss(pol.get_monoFromMem_masked(mono, prev, nd), gt, 0)
            loss.backward(MF.unit_grad(dev))
            out[fused] = (loss.item(), [p.grad.clone() for p in mem.parameters()])
    assert abs(out[True][0] - out[False][0]) < 1e-6 * max(1.0, abs(out[False][0]))
    for a, b in zip(out[True][1], out[False][1]):
        assert torch.equal(a, b)
    want = O.acoustic_mem(sd, mono.cpu(), O.mask_prev_mem(prev.cpu(), nd.cpu()))
    assert abs(out[True][0] - F.l1_loss(want, gt.cpu()[..., 0:1]).item()) < (2e-5 if mode == "bf16x3" else 2e-6)


@pytest.mark.parametrize("M,N,slope", [(269080, 32, 0.0), (54880, 64, 0.0), (280, 512, 0.0), (70001, 16, 0.2), (3000, 48, 0.0), (1 << 18, 2, 0.2), (5, 8, 0.0)])
def test_activation_backward_and_bias_gradient_in_one_pass(M, N, slope):
    """m2h_act_bwd_bias == m2h_act_bwd followed by m2h_bias_grad, bit for bit (same partition, same summation order), for the column-walking,
    the narrow and the single-split forms; and == torch."""
    from m2h import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(M % 1000 + N)
    dy, y = torch.randn(M, N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    two = MF.act_bwd(dy, y, slope)
    db2 = MF.bias_grad(two)
    one, db1 = MF.act_bwd_bias(dy, y, slope)
    assert torch.equal(one, two) and torch.equal(db1, db2)
    want = torch.where(y > 0, dy, dy * slope).double().sum(0)
    assert (db1.double() - want).abs().max().item() < 2e-6 * max(1.0, torch.where(y > 0, dy, dy * slope).abs().double().sum(0).max().item())
