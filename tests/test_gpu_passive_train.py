"""GPU: passive pre-training path (train-mode BN, transposed-conv / conv / BN backward, Adam) against two training steps
of the reference (tests/golden/passive_train.npz) and against torch autograd for the new kernels."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import m2h_oracle as O
from m2h import synthetic

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


def _rel(a, b):
    return O.rel_l1(torch.as_tensor(a).detach().cpu(), torch.as_tensor(b).detach())


@pytest.mark.parametrize("slope,shape", [(0.2, (6, 64, 8, 16)), (0.0, (6, 64, 8, 16)), (0.0, (3, 512, 1, 2)), (0.2, (64, 64, 16, 16)), (0.2, (5, 128, 8, 8)),
                                         (0.0, (7, 256, 4, 4)), (0.2, (64, 512, 2, 2)), (0.2, (3, 12, 4, 4)), (0.0, (5, 64, 20, 20)),
                                         (0.2, (17, 64, 16, 16)), (0.0, (16, 128, 16, 16)), (0.2, (65, 64, 16, 16)), (0.0, (64, 32, 32, 32))])
@pytest.mark.parametrize("path", ["auto", "three_launches"])
def test_bn_act_train_forward_backward_match_torch(slope, shape, path):
    """Both routes of csrc/bn.hip: one launch per direction for layers of at most 4 096 rows (1, 4 or 16 rows per thread: the shapes here sit on
    both sides of every limit and include ragged row counts; the default limit is 256 rows, knob 37 raises it), three launches otherwise / with
    knob 37 = -1."""
    from m2h import functional as MF
    from m2h import ops
    dev = _dev()
    M = shape[0] * shape[2] * shape[3]
    if path == "three_launches" and M > 4096:
        pytest.skip("already the three-launch route")
    MF.carry_tuning(True)
    ops.debug_set(37, -1 if path == "three_launches" else 4096)    # (the one-launch kernels wherever they can run: 1, 4 and 16 rows per thread)
    try:
        _bn_case(MF, dev, slope, shape)
        label = ops.last_kernel()
    finally:
        ops.debug_set(37, 0)
        MF.carry_tuning(False)
    assert ("one launch" in label) == (path == "auto" and M <= 4096), label


def _bn_case(MF, dev, slope, shape):
    g = torch.Generator().manual_seed(3)
    Cn = shape[1]
    x = (torch.randn(*shape, generator=g) * 1.7 + 0.4).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(Cn)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cn, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cn, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(Cn, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(Cn, generator=g) + 0.5)
    bn2 = torch.nn.BatchNorm2d(Cn)
    bn2.load_state_dict(bn.state_dict())
    bn2 = bn2.to(dev)
    bn.train()
    y = F.leaky_relu(bn(x), slope) if slope else F.relu(bn(x))
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    yd = MF.bn_act_train(xd, bn2, slope)
    assert _rel(yd.permute(0, 3, 1, 2), y) < 2e-5
    yd.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    assert _rel(xd.grad.permute(0, 3, 1, 2), x.grad) < 1e-4
    assert _rel(bn2.weight.grad, bn.weight.grad) < 1e-4 and _rel(bn2.bias.grad, bn.bias.grad) < 1e-4
    assert torch.allclose(bn2.running_mean.cpu(), bn.running_mean, atol=1e-6) and torch.allclose(bn2.running_var.cpu(), bn.running_var, rtol=1e-5)
    assert int(bn2.num_batches_tracked) == 1


@pytest.mark.parametrize("B,H,W,C0,C1,Co", [(3, 1, 1, 512, 0, 512), (2, 2, 2, 512, 512, 256), (2, 8, 8, 128, 128, 64), (2, 16, 16, 64, 64, 16)])
def test_conv_transpose_backward_matches_torch(B, H, W, C0, C1, Co):
    from m2h import functional as MF
    dev = _dev()
    g = torch.Generator().manual_seed(B + H + Co)
    x = torch.randn(B, C0, H, W, generator=g, requires_grad=True)
    s = torch.randn(B, C1, H, W, generator=g, requires_grad=True) if C1 else None
    w = (torch.randn(C0 + C1, Co, 4, 4, generator=g) * (0.5 / (C0 + C1)) ** 0.5).requires_grad_(True)
    xin = x if s is None else torch.cat((x, s), 1)
    z = F.conv_transpose2d(xin, w, None, 2, 1)
    gz = torch.randn(z.shape, generator=g)
    z.backward(gz)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    sd_ = s.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True) if s is not None else None
    wd = w.detach().to(dev).requires_grad_(True)
    zd = MF.conv_transpose2d(xd, wd, sd_)
    assert _rel(zd.permute(0, 3, 1, 2), z) < 2e-5
    zd.backward(gz.permute(0, 2, 3, 1).contiguous().to(dev))
    assert _rel(wd.grad, w.grad) < 1e-4
    assert _rel(xd.grad.permute(0, 3, 1, 2), x.grad) < 1e-4
    if s is not None:
        assert _rel(sd_.grad.permute(0, 3, 1, 2), s.grad) < 1e-4


def test_passive_training_steps_match_reference_fixture(golden_dir):
    from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "passive_train.npz"))
    tr = PassiveTrainer(passive_config(), dev)
    tr.setup()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), int(g["seed_w"])).items()}
    tr.actor_critic.load_state_dict(sd)
    tr.actor_critic.train()
    B = int(g["B"])
    mixed, tc = synthetic.make_passive_inputs(B, 32, int(g["seed_x"]))
    gen = torch.Generator().manual_seed(int(g["gt_seed"]))
    gt_bin = (torch.rand(B, 512, 32, 2, generator=gen) * 2).to(dev)
    gt_mono = (torch.rand(B, 512, 32, 1, generator=gen) * 2).to(dev)
    mix, tct = torch.from_numpy(mixed).to(dev), torch.from_numpy(tc).to(dev)
    # step 0 forward (train-mode BN) matches the reference's outputs
    obs = {"mixed_bin_audio_mag": mix, "target_class": tct}
    for step in range(2):
        if step == 0:
            masks = tr.actor_critic.get_binSepMasks(obs)
            mono = tr.actor_critic.convert_bin2mono(masks.detach(), mixed_audio=mix)
            assert _rel(masks, g["masks_step0"]) < 5e-5 and _rel(mono, g["mono_step0"]) < 5e-5
            b, m = tr.optimize_supervised_loss(mix, masks, gt_bin, mono, gt_mono, "train")
        else:
            b, m = tr.train_batch(mix, gt_bin, gt_mono, tct, "train")
        assert abs(b.item() - g["losses"][step][0]) < 1e-4 and abs(m.item() - g["losses"][step][1]) < 1e-4, (step, b.item(), m.item())
    post = tr.actor_critic.state_dict()
    n = 0
    for key in g.files:
        if not key.startswith("post."):
            continue
        k = key[5:]
        ref = torch.from_numpy(g[key])
        mine = post[k].cpu()
        if "running_" in k:
            assert torch.allclose(mine, ref, rtol=2e-3, atol=1e-4), k  # step-2 stats see weights after a sign-sensitive Adam step
        else:
            bad = ((mine - sd[k]) - (ref - sd[k])).abs().gt(2e-4).float().mean().item()  # two Adam steps of lr 5e-4
            assert bad < 0.02, (k, bad)
        n += 1
    assert n > 40
    # eval after training still runs the fused inference path
    tr.actor_critic.eval()
    with torch.no_grad():
        assert tr.actor_critic.get_binSepMasks(obs).shape == (B, 512, 32, 2)


def test_graphed_training_batches_equal_the_kernel_by_kernel_batches_and_eval_sees_the_new_weights():
    """A training batch replayed from a HIP graph (forward with train-mode BN, losses, backward; Adam outside) leaves the same
    weights, BN statistics and losses as the kernel-by-kernel batch; and an eval-mode forward between training batches is
    computed from the CURRENT weights (the fused inference path re-packs after optimizer steps of trainable separators)."""
    from m2h.common.spaces import move2hear_observation_space
    from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config
    from m2h.pretrain.passive.policy import Move2HearPassiveWoMemoryPolicy
    dev = _dev()
    outs = []
    for graphs in (False, True):
        tr = PassiveTrainer(passive_config(BATCH_SIZE=4, use_hip_graphs=graphs), dev)
        tr.setup()
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), 4).items()}
        tr.actor_critic.load_state_dict(sd)
        losses, evals = [], []
        for i in range(4):
            tr.actor_critic.train()
            b, m = tr.train_batch(*tr.feeders["train"].batch())
            losses.append((b.item(), m.item()))
            tr.actor_critic.eval()
            mix, _gb, _gm, tc = tr.feeders["val"].batch()
            with torch.no_grad():
                evals.append(tr.actor_critic.get_binSepMasks({"mixed_bin_audio_mag": mix, "target_class": tc}).cpu())
        assert (tr._train_graph is not None) == graphs
        outs.append((losses, evals, {k: v.detach().cpu().clone() for k, v in tr.actor_critic.state_dict().items()}, mix, tc))
    (la, ea, wa, _, _), (lb, eb, wb, mix, tc) = outs
    assert la == lb
    for x, y in zip(ea, eb):
        assert torch.equal(x, y)
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k
    assert not torch.equal(ea[0], ea[1])  # the weights moved between the two eval passes and the eval path noticed
    fresh = Move2HearPassiveWoMemoryPolicy(move2hear_observation_space(32))
    fresh.load_state_dict(wb)
    fresh = fresh.to(dev).eval()
    with torch.no_grad():
        ref = fresh.get_binSepMasks({"mixed_bin_audio_mag": mix, "target_class": tc}).cpu()
    assert torch.equal(ref, eb[-1])


def test_passive_trainer_epoch_runs():
    from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config
    tr = PassiveTrainer(passive_config(BATCH_SIZE=8, BATCHES_PER_EPOCH=3, VAL_BATCHES=1), _dev())
    log = tr.train(num_epochs=2)
    assert len(log) == 2 and all(np.isfinite(v) for r in log for v in r["train"] + r["val"])
    assert log[1]["train"][0] < log[0]["train"][0] * 1.5  # sanity: not diverging
