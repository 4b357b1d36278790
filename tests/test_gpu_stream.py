"""GPU: the weight-streaming split-K kernel (csrc/conv_stream.hip) -- at most 64 GEMM rows per phase against >= 1 MB of weights: the
U-Net stages around the bottleneck at the rollout batch of 14 environments (separator_cnn.py:46-52, 128-135) and the policy's 14-row
Linear layers (visual_cnn.py:140-141) -- against the CPU oracle's layer (torch conv2d / conv_transpose2d + eval BatchNorm +
activation, what m2h_oracle.py composes the U-Net from), against the 16-row kernels it replaces, and as a cross-workgroup hand-off:
bit-reproducible whatever the arrival order, no stale slab ever read, ticket words left zero."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import m2h_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5
STREAM = "conv_igemm_f32 (stream split-K)"


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda", 0)


def _bn(Co, g):
    return (torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1, torch.randn(Co, generator=g) * 0.1, torch.rand(Co, generator=g) + 0.5)


def _down_case(B, H, W, Ci, Co, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 4, 4, generator=g) * (2.0 / (Ci * 16)) ** 0.5
    gamma, beta, mean, var = _bn(Co, g)
    ref = F.leaky_relu(F.batch_norm(F.conv2d(x, w, None, 2, 1), mean, var, gamma, beta, False, 0.1, 1e-5), 0.2)   # separator_cnn.py:5-12
    return x, w, (gamma, beta, mean, var), ref


def _up_case(B, H, W, C0, C1, Co, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C0, H, W, generator=g)
    s = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(C0 + C1, Co, 4, 4, generator=g) * (2.0 / ((C0 + C1) * 4)) ** 0.5
    gamma, beta, mean, var = _bn(Co, g)
    xin = x if s is None else torch.cat((x, s), 1)
    ref = F.relu(F.batch_norm(F.conv_transpose2d(xin, w, None, 2, 1), mean, var, gamma, beta, False, 0.1, 1e-5))   # :15-24
    return x, s, w, (gamma, beta, mean, var), ref


def _run_down(ops, dev, x, w, bn, Co):
    wp = ops.pack_conv_weight(w.to(dev))
    sc, sh = ops.fold_bn(*(t.to(dev) for t in bn), 1e-5)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    return lambda: ops.unet_down_fwd(xd, wp, sc, sh, Co)


def _run_up(ops, dev, x, s, w, bn, Co):
    wp = ops.pack_convT_weight(w.to(dev))
    sc, sh = ops.fold_bn(*(t.to(dev) for t in bn), 1e-5)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    sd = None if s is None else s.permute(0, 2, 3, 1).contiguous().to(dev)
    return lambda: ops.unet_up_fwd(xd, sd, wp, sc, sh, Co)


# (B, H, W, Ci, Co): the fourth and fifth encoder stage at 14 / 1 / 3 / 16 environments (16 x 2 x 2 = 64 rows: the kernel's limit)
DOWN = [(14, 4, 4, 256, 512), (14, 2, 2, 512, 512), (1, 4, 4, 256, 512), (3, 2, 2, 512, 512), (16, 4, 4, 256, 512), (16, 2, 2, 512, 512)]
# (B, H, W, C0, C1, Co): the first two decoder stages (1 x 1 -> 2 x 2 without a skip, 2 x 2 -> 4 x 4 with one)
UP = [(14, 1, 1, 512, 0, 512), (14, 2, 2, 512, 512, 256), (1, 2, 2, 512, 512, 256), (16, 2, 2, 512, 512, 256), (5, 1, 1, 512, 0, 512)]


@pytest.mark.parametrize("shape", DOWN)
def test_stream_kernel_encoder_stages_match_the_oracle_layer(shape):
    from m2h import ops
    dev = _dev()
    B, H, W, Ci, Co = shape
    x, w, bn, ref = _down_case(B, H, W, Ci, Co, 1000 * B + H)
    run = _run_down(ops, dev, x, w, bn, Co)
    y = run()
    assert ops.last_kernel() == STREAM, ops.last_kernel()
    assert y.shape == (B, H // 2, W // 2, Co) and O.rel_l1(y.cpu().permute(0, 3, 1, 2), ref) < TOL
    ops.debug_set(5, -1)          # the 16-row kernels it replaces
    try:
        y_old = run()
        assert ops.last_kernel() != STREAM
    finally:
        ops.debug_set(5, 0)
    assert O.rel_l1(y.cpu(), y_old.cpu()) < 1e-5


@pytest.mark.parametrize("shape", UP)
def test_stream_kernel_decoder_stages_match_the_oracle_layer(shape):
    from m2h import ops
    dev = _dev()
    B, H, W, C0, C1, Co = shape
    x, s, w, bn, ref = _up_case(B, H, W, C0, C1, Co, 100 * B + H + C1)
    run = _run_up(ops, dev, x, s, w, bn, Co)
    y = run()
    assert ops.last_kernel() == STREAM, ops.last_kernel()
    assert y.shape == (B, 2 * H, 2 * W, Co) and O.rel_l1(y.cpu().permute(0, 3, 1, 2), ref) < TOL
    ops.debug_set(5, -1)
    try:
        y_old = run()
        assert ops.last_kernel() != STREAM
    finally:
        ops.debug_set(5, 0)
    assert O.rel_l1(y.cpu(), y_old.cpu()) < 1e-5


def test_stream_kernel_takes_the_14_row_linear_layers_and_leaves_the_rest():
    """nn.Linear at the rollout batch (VisualCNN's 4608 -> 512: 9.4 MB of weights against 14 rows) through ops.linear, bias and ReLU in
    the epilogue, into a column block of a wider matrix (the policy's concatenated features); 65 rows, small weights or a fused head
    stay on the other kernels."""
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    for M, K, N in ((14, 4608, 512), (1, 4608, 512), (16, 1536, 1536), (64, 2048, 512)):
        x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g) * 0.1
        ref = F.relu(F.linear(x, w, b))
        wide = torch.zeros(M, N + 512, device=dev)
        y = ops.linear(x.to(dev), w.to(dev), b.to(dev), slope=0.0, out=wide[:, 512:])
        assert ops.last_kernel() == STREAM, (M, K, N, ops.last_kernel())
        assert O.rel_l1(wide[:, 512:].cpu(), ref) < TOL and float(wide[:, :512].abs().max()) == 0.0 and y is not None
    for M, K, N in ((65, 2048, 512), (14, 512, 256)):     # too many rows / 0.5 MB of weights
        x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5
        y = ops.linear(x.to(dev), w.to(dev))
        assert ops.last_kernel() != STREAM
        assert O.rel_l1(y.cpu(), F.linear(x, w)) < TOL


def test_stream_kernel_is_bit_reproducible_and_never_reads_a_stale_slab():
    """The K-slices of a tile meet through scratch in SLICE order behind an agent-scope release / ticket / acquire (conv_stream.hip):
    (1) the same inputs give the same bits on every launch, alone on the chip or beside a second stream that keeps every CU busy with a
    copy (uneven arrival); (2) launches with DIFFERENT inputs back to back re-use the same slab and ticket addresses -- each result must
    be its own input's (a stale L1 line of the previous launch's slab, or a ticket word not back at zero, would show at once);
    (3) another block-count target (more, fewer slices) gives the same values to fp32 summation order."""
    from m2h import ops
    dev = _dev()
    B, H, W, C0, C1, Co = 14, 2, 2, 512, 512, 256
    cases = [_up_case(B, H, W, C0, C1, Co, 7 + k) for k in range(3)]
    runs = [_run_up(ops, dev, x, s, w, bn, Co) for x, s, w, bn, _ref in cases]
    first = [r().clone() for r in runs]
    for (x, s, w, bn, ref), y in zip(cases, first):
        assert O.rel_l1(y.cpu().permute(0, 3, 1, 2), ref) < TOL
    side = torch.cuda.Stream(dev)
    big_a, big_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
    for rep in range(30):
        if rep % 2 == 1:
            with torch.cuda.stream(side):
                for _ in range(4):
                    big_b.copy_(big_a)
        for k in (rep % 3, (rep + 1) % 3):          # alternate the inputs: every launch follows one with other values at the same scratch addresses
            y = runs[k]()
            assert ops.last_kernel() == STREAM
            assert torch.equal(y, first[k]), (rep, k, float((y - first[k]).abs().max()))
    torch.cuda.synchronize()
    for target in (64, 128, 512, 1024):
        ops.debug_set(6, target)
        try:
            y = runs[0]()
            assert ops.last_kernel() == STREAM
        finally:
            ops.debug_set(6, 0)
        assert O.rel_l1(y.cpu(), first[0].cpu()) < 1e-5, target


def test_unet_pair_at_the_rollout_batch_runs_its_deep_stages_on_the_stream_kernel_and_matches_the_oracle():
    """The whole-network runner at 14 environments (the rollout step's call, rl/ppo/ppo_trainer.py:295-373): stages 4-7 of its 11 (fourth
    and fifth encoder stage, first and second decoder stage) run on the stream kernel with the runner's own ticket words (cleared by its
    first kernel); masks and mono against the CPU oracle, the same bits from a HIP-graph replay and with the stream kernel off to 1e-5."""
    from m2h import ops, synthetic
    from m2h.common.spaces import move2hear_observation_space
    from m2h.pretrain.passive.policy import Move2HearPassiveWoMemoryPolicy
    dev = _dev()
    pol = Move2HearPassiveWoMemoryPolicy(move2hear_observation_space())
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), 3).items()}
    pol.load_state_dict(sd)
    pol = pol.to(dev).eval()
    mixed, tc = synthetic.make_passive_inputs(14, 32, 9)
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}
    with torch.no_grad():
        want_m, want_mono = O.passive_pair(sd, torch.from_numpy(mixed), torch.from_numpy(tc))
        masks = pol.get_binSepMasks(obs)
        labels = ops.unet_stage_kernels()
        mono = pol.convert_bin2mono(masks, mixed_audio=obs["mixed_bin_audio_mag"])
    assert [labels[i] for i in (4, 5, 6, 7)] == [STREAM] * 4, labels
    assert O.rel_l1(masks.cpu(), want_m) < TOL and O.rel_l1(mono.cpu(), want_mono) < TOL
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s), torch.no_grad():
        pol.get_binSepMasks(obs)                    # warm-up on the capture stream
        with torch.cuda.graph(g, stream=s):
            gm = pol.get_binSepMasks(obs)
            gmono = pol.convert_bin2mono(gm, mixed_audio=obs["mixed_bin_audio_mag"])
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(gm, masks) and torch.equal(gmono, mono)
    ops.debug_set(5, -1)
    try:
        with torch.no_grad():
            m_old = pol.get_binSepMasks(obs)
            assert STREAM not in ops.unet_stage_kernels()
            mono_old = pol.convert_bin2mono(m_old, mixed_audio=obs["mixed_bin_audio_mag"])
    finally:
        ops.debug_set(5, 0)
    assert O.rel_l1(masks.cpu(), m_old.cpu()) < 1e-5 and O.rel_l1(mono.cpu(), mono_old.cpu()) < 1e-5
