"""GPU: round-6 kernels.  The fused conv + L1-loss launch of update_sep (m2h_conv3x3_l1_nhwc16: AcousticMem's last conv, its de-slice and
F.l1_loss against gt_mono_comps[..., 0], rl/models/memory_nets.py:16,62-67 with rl/ppo/ppo.py:206-216) against the CPU oracle's layer and
against the two launches it replaces, in both arithmetic modes; the update itself (losses, gradients, weights after the step) stays pinned by
tests/test_gpu_rl.py / test_gpu_trainer_golden.py through the same code path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import m2h_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda", 0)


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_fused_conv_l1_matches_the_oracle_and_the_two_launches_it_replaces(mode):
    from m2h import functional as MF
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    B = 72                                                  # (>= 64 samples: the image-row kernels' shapes)
    h = torch.randn(B, 32, 32, 32, generator=g).relu()      # NHWC: the first conv's activated output
    w = torch.randn(16, 32, 3, 3, generator=g) * 0.06
    gt_comps = torch.rand(B, 512, 32, 4, generator=g) * 2
    # oracle: conv2d -> de-slice [B, 16, 32, 32] -> [B, 512, 32, 1] (band n, row q -> n * 32 + q) -> mean |.|
    y_ref = F.conv2d(h.permute(0, 3, 1, 2), w, None, 1, 1)                    # [B, 16, 32, 32]
    pred = y_ref.reshape(B, 512, 32, 1)
    want = F.l1_loss(pred, gt_comps[..., 0:1])
    hd, wd, gd = h.to(dev), w.to(dev), gt_comps.to(dev)
    plane = gd[..., 0:1].contiguous()
    ops.set_math_mode(ops.MATH_BF16X3 if mode == "bf16x3" else ops.MATH_FP32)
    try:
        wp = ops.pack_conv_weight_ex(wd, 32, 32)
        assert ops.conv3x3_l1_supported(hd) and not ops.conv3x3_l1_supported(hd[:8])
        loss, dy = ops.conv3x3_l1_nhwc16(hd, wp, plane)
        assert "L1 loss" in ops.last_kernel(), ops.last_kernel()
        y = ops.conv2d_nhwc(hd, wp, 16, 3, 3, stride=1, pad=1, slope=1.0)
        loss2, dy2 = ops.l1_loss_nhwc16(y, gd, 0, want_grad=True)
        loss3, dy3 = ops.l1_loss_nhwc16(y, plane, 0, want_grad=True)
        # through the autograd function update_sep uses: a plane takes the fused launch, interleaved components the two launches
        hw = hd.clone().requires_grad_(False)
        wq = wd.clone().requires_grad_(True)
        lf = MF.conv_l1_nhwc16(hw, wq, plane, 0)
        lf.backward()
        gw_fused = wq.grad.clone()
        wq.grad = None
        lt = MF.conv_l1_nhwc16(hw, wq, gd, 0)
        lt.backward()
        gw_two = wq.grad.clone()
    finally:
        ops.set_math_mode(ops.MATH_FP32)
    tol = 2e-5 if mode == "bf16x3" else 2e-6
    assert abs(float(loss) - float(want)) <= tol * float(want), (float(loss), float(want))
    assert abs(float(loss) - float(loss2)) <= 2e-6 * float(loss2) and float(loss2) == float(loss3)
    # the gradient is sign(y - g) / n: identical wherever |y - g| is not within rounding of zero
    inv = 1.0 / (B * 512 * 32)
    assert torch.equal(dy2, dy3)
    same = (dy == dy2).float().mean().item()
    assert same == 1.0, same                                 # same conv values (same kernel, same accumulation), same comparison
    assert set(np.unique(dy.cpu().numpy()).tolist()) <= {np.float32(-inv).item(), 0.0, np.float32(inv).item()}
    d_ref = torch.sign(pred - gt_comps[..., 0:1]).reshape(B, 16, 32, 32).permute(0, 2, 3, 1) * inv      # NHWC
    assert (dy.cpu() != d_ref.float()).float().mean().item() < (2e-3 if mode == "bf16x3" else 2e-4)     # (sign flips only where y ~ g)
    assert abs(float(lf.detach()) - float(loss)) == 0.0 and abs(float(lt.detach()) - float(loss2)) == 0.0
    assert O.rel_l1(gw_fused.cpu(), gw_two.cpu()) < 1e-6


def test_acoustic_mem_update_path_with_both_fusions_matches_torch_autograd():
    """update_sep's differentiable path in bf16x3 arithmetic (functional.AcousticMemL1: conv0 + ReLU | conv1 + L1 loss + its gradient ||
    conv1's weight gradient | conv0's weight gradient with conv1's input gradient and the ReLU gate made inside the kernel,
    m2h_conv_wgrad_dgrad_fused_f32) against torch autograd on the CPU through the reference's ops (memory_nets.py:11-16,62-67; ppo.py:206-226),
    and against the per-layer Functions it replaces (the unfused launches: the same values to fp32 summation order).  Batches whose row
    ranges end inside an image, at an image edge and on a single row per block are all in the split."""
    from m2h import functional as MF
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(23)
    for B in (64, 75):
        x = torch.randn(B, 32, 32, 32, generator=g) * 0.7          # NHWC sliced input
        w0 = torch.randn(32, 32, 3, 3, generator=g) * 0.06
        w1 = torch.randn(16, 32, 3, 3, generator=g) * 0.08
        gt = torch.rand(B, 512, 32, 1, generator=g) * 2
        # CPU reference with torch autograd
        w0c, w1c = w0.clone().requires_grad_(True), w1.clone().requires_grad_(True)
        h = F.relu(F.conv2d(x.permute(0, 3, 1, 2), w0c, None, 1, 1))
        y = F.conv2d(h, w1c, None, 1, 1)                           # [B, 16, 32, 32]: band n, row q -> de-sliced row n * 32 + q
        want = F.l1_loss(y.reshape(B, 512, 32, 1), gt)
        want.backward()
        xd, gtd = x.to(dev), gt.to(dev)
        ops.set_math_mode(ops.MATH_BF16X3)
        try:
            assert MF.acoustic_mem_l1_supported(xd)
            w0d, w1d = w0.to(dev).requires_grad_(True), w1.to(dev).requires_grad_(True)
            loss = MF.acoustic_mem_l1(xd, w0d, w1d, gtd)
            loss.backward()
            g0, g1 = w0d.grad.clone(), w1d.grad.clone()
            # the per-layer Functions (no fused input gradient: dgrad launch + gated weight gradient)
            w0e, w1e = w0.to(dev).requires_grad_(True), w1.to(dev).requires_grad_(True)
            he = MF.conv2d(xd, w0e, None, 1, 1, slope=0.0)
            le = MF.conv_l1_nhwc16(he, w1e, gtd, 0)
            le.backward()
        finally:
            ops.set_math_mode(ops.MATH_FP32)
        assert abs(float(loss.detach()) - float(want.detach())) <= 3e-5 * float(want.detach())
        assert float(loss.detach()) == float(le.detach())
        assert torch.equal(g1, w1e.grad)                                       # same launches for conv1's weight gradient
        assert O.rel_l1(g0.cpu(), w0e.grad.cpu()) < 3e-5, (B, O.rel_l1(g0.cpu(), w0e.grad.cpu()))
        # against torch autograd: the gradients of an L1 loss are sums of +-1/n signs pushed through the convs -- a sign flip where y ~ gt moves
        # them by O(1 / n) per flipped element; 2e-3 of the gradient's L1 mass is what the unfused bf16x3 path shows too
        e0, e1 = O.rel_l1(g0.cpu(), w0c.grad), O.rel_l1(g1.cpu(), w1c.grad)
        u0 = O.rel_l1(w0e.grad.cpu(), w0c.grad)
        assert e1 < 2e-3 and e0 < 2e-3 and e0 < 1.5 * u0 + 1e-5, (B, e0, e1, u0)


def _passive_run(side, steps=4):
    """`steps` training batches of the passive trainer from one seeded state (first one kernel by kernel, the rest replayed from the
    step's HIP graph); returns (losses, weights, was anything deferred)."""
    import m2h.functional as MF
    from m2h import synthetic
    from m2h.pretrain.passive.passive_trainer import PassiveTrainer, passive_config
    dev = _dev()
    tr = PassiveTrainer(passive_config(BATCH_SIZE=4, wgrad_side_branches=side), dev)
    tr.setup()
    tr.actor_critic.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), 4).items()})
    seen = []
    orig = MF.flush_deferred_wgrads

    def spy(device=None, side=False):
        cur = torch.cuda.current_stream(device)
        seen.append((bool(side), len(MF._wgrad_deferred.get((cur.device.index, cur.cuda_stream), []))))
        return orig(device, side)

    MF.flush_deferred_wgrads = spy
    try:
        tr.actor_critic.train()
        losses = []
        for _ in range(steps):
            b, m = tr.train_batch(*tr.feeders["train"].batch())
            losses.append((b.item(), m.item()))
    finally:
        MF.flush_deferred_wgrads = orig
    assert not MF._wgrad_deferred and not MF._wgrad_pending and not MF._wgrad_side["on"]
    return losses, {k: v.detach().cpu().clone() for k, v in tr.actor_critic.state_dict().items()}, seen


def test_deferred_weight_gradients_of_the_passive_step_leave_the_same_weights():
    """functional.wgrad_side_branches: inside the captured training step each U-Net's decoder weight gradients are launched from the
    bottleneck's flush point on a side stream, the encoder's behind the backward -- same kernels, so losses and weights after four
    batches equal the step captured with the switch off BIT FOR BIT; and the flushes really carried launches (5 + 5 decoder / encoder layers
    per network + the head's 1x1 conv)."""
    la, wa, seen_a = _passive_run(True)
    lb, wb, seen_b = _passive_run(False)
    assert la == lb
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k
    assert all(n == 0 for _s, n in seen_b)                            # switch off: nothing is ever deferred
    side = [n for s, n in seen_a if s]
    tail = [n for s, n in seen_a if not s]
    assert sorted(side) == [6, 6] and sorted(tail) == [5, 5], seen_a  # per network: head + 5 transposed convs at the bottleneck, 5 encoder convs at the end


def test_deferred_weight_gradients_must_be_flushed():
    """A capture that defers weight gradients and never launches them is an error of the caller, reported at the end of the block (and
    by the optimizer's join, had it been reached) -- not a silently missing gradient."""
    import m2h.functional as MF
    from m2h import graphs
    from m2h.optim import FlatAdam
    dev = _dev()
    w = torch.nn.Parameter(torch.randn(8, 8, 3, 3, device=dev) * 0.1)
    opt = FlatAdam([w], lr=1e-3)
    opt.build()
    x = torch.randn(2, 8, 8, 8, device=dev)
    memo = MF._PackMemo()
    opt.zero_grad()
    MF.conv2d(x, w, None, 1, 1, memo=memo).sum().backward()   # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="deferred"):
        with graphs.capture(g), MF.wgrad_side_branches():
            opt.zero_grad()
            MF.conv2d(x, w, None, 1, 1, memo=memo).sum().backward()
    assert not MF._wgrad_deferred and not MF._wgrad_side["on"]
    # the same capture with the flush: the gradient equals the eager one
    opt.zero_grad()
    MF.conv2d(x, w, None, 1, 1, memo=memo).sum().backward()
    want = w.grad.detach().clone()
    g = torch.cuda.CUDAGraph()
    with graphs.capture(g), MF.wgrad_side_branches():
        opt.zero_grad()
        MF.conv2d(x, w, None, 1, 1, memo=memo).sum().backward()
        MF.flush_deferred_wgrads(dev)
    opt.flat_g.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(opt.flat_g.view_as(w), want)
