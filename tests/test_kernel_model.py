"""CPU: the numpy model of the HIP kernels' index arithmetic (tests/kernel_model.py) reproduces the oracle.
This checks the algorithm the kernels implement (phase-decomposed transposed conv, in-place concat, class
plane as bias, slice/de-slice) without a GPU."""
import numpy as np
import torch

import kernel_model as KM
import m2h_oracle as O
from m2h import synthetic


def test_kernel_model_unet_pair_matches_oracle():
    sd_np = synthetic.make_state_dict(synthetic.passive_shapes(), 3)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
    mixed, tc = synthetic.make_passive_inputs(1, 32, 5)
    mix, tct = torch.from_numpy(mixed), torch.from_numpy(tc)
    with torch.no_grad():
        masks_o, feats_o = O.get_binSepMasks(sd, mix, tct, return_feats=True)
        mono_o = O.convert_bin2mono(sd, masks_o, mix)
    masks_k, feats_k = KM.unet_forward(sd_np, O.ENC_B, O.DEC_B, mixed, target_class=tc)
    for fo, fk in zip(feats_o, feats_k):
        assert O.rel_l1(torch.from_numpy(fk).permute(0, 3, 1, 2), fo) < 1e-5
    assert O.rel_l1(torch.from_numpy(masks_k), masks_o) < 1e-5
    mono_k, _ = KM.unet_forward(sd_np, O.ENC_M, O.DEC_M, mixed, masks=masks_o.numpy())
    assert O.rel_l1(torch.from_numpy(mono_k), mono_o) < 1e-5


def test_kernel_model_fully_convolutional_tm64():
    sd_np = synthetic.make_state_dict(synthetic.passive_shapes(), 4)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
    mixed, tc = synthetic.make_passive_inputs(1, 64, 6)
    with torch.no_grad():
        masks_o = O.get_binSepMasks(sd, torch.from_numpy(mixed), torch.from_numpy(tc))
    masks_k, _ = KM.unet_forward(sd_np, O.ENC_B, O.DEC_B, mixed, target_class=tc)
    assert O.rel_l1(torch.from_numpy(masks_k), masks_o) < 1e-5


def test_split32_layout_and_bf16x3_product_error():
    """x = hi + lo to 2^-16 relative; the three-product formation is within 2^-15 relative of the exact fp32 product sum
    (the dropped lo*lo term and the two roundings), three orders of magnitude inside the 1e-3 contract."""
    import kernel_model as KM
    r = np.random.default_rng(0)
    x = (r.standard_normal((5, 64)) * np.exp(r.uniform(-6, 6, (5, 64)))).astype(np.float32)
    hi, lo = KM.split_hi_lo(x)
    assert np.all(np.abs((hi.astype(np.float64) + lo) - x) <= np.abs(x) * 2.0 ** -16)
    img = KM.split32(x)
    assert img.shape == (5, 2, 2, 32) and img.dtype == np.uint16
    back = (img[:, :, 0, :].astype(np.uint32) << 16).view(np.float32) + (img[:, :, 1, :].astype(np.uint32) << 16).view(np.float32)
    assert np.array_equal(back.reshape(5, 64), (hi + lo).astype(np.float32))
    a = r.standard_normal((33, 256)).astype(np.float32)
    w = (r.standard_normal((17, 256)) * 0.1).astype(np.float32)
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    got = KM.bf16x3_matmul(a, w)
    scale = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T      # sum of |products|
    assert np.all(np.abs(got - exact) <= scale * 2.0 ** -15)
    plain = KM.bf16_rne(a).astype(np.float64) @ KM.bf16_rne(w).astype(np.float64).T
    assert np.abs(plain - exact).max() > 30 * np.abs(got - exact).max()          # what the split buys over plain bf16 operands


def test_tap_sharing_kernel_indexing_matches_conv_transpose():
    """The staged-image / shifted-window indexing of convT_tap_kernel (128- and 256-output tiles, 1/2/4 image rows per tile)
    reproduces torch's ConvTranspose2d(4, 2, 1) phase by phase."""
    import kernel_model as KM
    torch.manual_seed(0)
    for (B, H, W, C, N, bm) in ((2, 4, 32, 8, 5, 128), (1, 2, 64, 4, 3, 128), (1, 2, 128, 4, 3, 256), (1, 8, 32, 4, 2, 256)):
        x = torch.randn(B, C, H, W)
        w = torch.randn(C, N, 4, 4) * 0.3
        ref = torch.nn.functional.conv_transpose2d(x, w, stride=2, padding=1).permute(0, 2, 3, 1).double().numpy()
        wp = KM.pack_convT_weight(w.numpy())
        xn = x.permute(0, 2, 3, 1).contiguous().numpy()
        for phase in range(4):
            ph, pw = phase >> 1, phase & 1
            got = KM.convT_tap_phase(xn, wp[phase], ph, pw, bm)
            assert np.allclose(got, ref[:, ph::2, pw::2, :], atol=1e-5), (B, H, W, bm, phase)


def test_tap_window_and_image_row_indexing_match_torch():
    """CPU: the host-side tap window (rows / columns of the kernel that lie in the padding for every output pixel) and the patch
    indexing of the image-row 3x3 kernels, restated in numpy, against torch convolutions."""
    import torch
    import torch.nn.functional as F
    g = np.random.default_rng(2)
    # tap window: deepest encoder stage shapes (2-row input -> 1 output row) and a shape where nothing can be skipped
    assert KM.tap_range(4, 1, 2, -1, 1, 2) == (1, 2) and KM.tap_range(4, 2, 2, -1, 1, 4) == (0, 4)
    assert KM.tap_range(2, 1, 1, 0, -1, 1) == (0, 1) and KM.tap_range(2, 1, 1, 0, 1, 1) == (0, 1)      # transposed conv over a 1-row image
    for (B, Hi, Wi, C, N) in ((2, 2, 2, 8, 5), (2, 2, 16, 8, 5), (1, 4, 4, 8, 5)):
        x = g.standard_normal((B, Hi, Wi, C)).astype(np.float32)
        w = (g.standard_normal((N, C, 4, 4)) * 0.1).astype(np.float32)
        ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), None, 2, 1).permute(0, 2, 3, 1).numpy()
        got, win = KM.conv_with_tap_window(x, KM.pack_conv_weight(w), N, 2, 4, 4, -1, Hi // 2, Wi // 2)
        assert np.abs(got - ref).max() < 1e-4, (B, Hi, Wi)
        assert win[1] == (2 if Hi == 2 else 4) and win[3] == (2 if Wi == 2 else 4)
    # image-row kernels: forward (mul 1), input gradient (mul -1, taps walked backwards), weight gradient
    B, H, C, N = 2, 8, 16, 8
    x = g.standard_normal((B, H, 32, C)).astype(np.float32)
    w = (g.standard_normal((N, C, 3, 3)) * 0.1).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.from_numpy(w).requires_grad_(True)
    y = F.conv2d(xt, wt, None, 1, 1)
    fwd = KM.conv3x3_row(x, KM.pack_conv_weight(w), N, 1)
    assert np.abs(fwd - y.detach().permute(0, 2, 3, 1).numpy()).max() < 1e-4
    dy = g.standard_normal((B, H, 32, N)).astype(np.float32)
    y.backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    # input gradient = conv of dy with the (ci <-> co)-transposed weights, taps th -> offset 1 - th (m2h_pack_dgrad_weight order)
    wd = np.ascontiguousarray(w.transpose(1, 2, 3, 0)).reshape(C, 9 * N)       # [ci][(th, tw, co)]
    dx = KM.conv3x3_row(dy, wd, C, -1)
    assert np.abs(dx - xt.grad.permute(0, 2, 3, 1).numpy()).max() < 1e-4
    dw = KM.wgrad3x3_row(x, dy)
    assert np.abs(dw - wt.grad.permute(0, 2, 3, 1).reshape(N, 9 * C).numpy()).max() < 2e-3


def test_strip_kernel_lds_map_is_conflict_free_and_the_channel_order_is_a_permutation():
    """csrc/conv_strip.hip: the A-fragment reads of the strip-walker kernels hit every 16-byte bank slot exactly once per LDS
    lane group at EVERY pixel shift (taps shift the window by one record, fragments start at multiples of 16 records, the halo
    record adds one), records of one plane never overlap, and the first stage's in-kernel channel order covers all 32 channels."""
    for i0 in range(0, 64):
        assert KM.strip_read_conflict_degree(i0) == 1, i0
    seen = set()
    for i in range(34):
        for j in range(4):
            a = KM.strip_px_addr(i, j)
            assert a % 16 == 0 and i * 64 <= a < (i + 1) * 64 and a not in seen
            seen.add(a)
    assert sorted(KM.strip_conv1_k_order()) == list(range(32))


def test_patch_engine_lds_map_is_conflict_free_at_every_shift():
    """csrc/conv_patch.hip: the pixel-fragment reads hit every 16-byte bank slot exactly once per LDS lane group whatever patch row the
    fragment starts at (the four taps of a patch read it at row shifts 0, 1, W + 1, W + 2 and patch lines are W + 1 rows long, so
    every alignment occurs), for the hi and the lo pieces; every row holds each piece once.  The XOR map of conv_dma.hip, kept for
    the weight rows (always read at multiples of 16 rows), is conflict-free there and only there."""
    for row0 in range(0, 200):
        assert KM.patch_read_conflict_degree(row0, 0) == 1 and KM.patch_read_conflict_degree(row0, 1) == 1, row0
    for row in range(16):
        assert sorted((KM.patch_row_addr(row, p) - row * 128) // 16 for p in range(8)) == list(range(8))
        assert KM.patch_row_addr(row, 5) == KM.patch_row_addr(row, 1) ^ 64     # lo piece = hi piece's address ^ 64
    assert all(KM.patch_read_conflict_degree(r, 0, xor_map=True) == 1 for r in range(0, 64, 16))
    assert max(KM.patch_read_conflict_degree(r, 0, xor_map=True) for r in range(1, 16)) == 2


def test_patch_engine_tap_classes_match_torch():
    """csrc/conv_patch.hip's decomposition -- one staged patch per parity class of the 4x4/s2 window / per transposed-conv phase,
    four taps reading it at a shift, the whole-image form's zero rows selected by edge / kill bits -- equals Conv2d(4, 2, 1) and
    ConvTranspose2d(4, 2, 1) (separator_cnn.py:5-24) on every output pixel, for both patch forms, including 1-row images."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    for (H, W, Ci, Co) in ((4, 8, 3, 5), (2, 4, 4, 2), (8, 4, 2, 3)):
        x = torch.randn(1, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, 4, 4, generator=g)
        want = F.conv2d(x, w, None, stride=2, padding=1)[0].permute(1, 2, 0).numpy()
        for whole in (True, False):
            got = KM.patch_engine_layer(x[0].permute(1, 2, 0).numpy(), w.numpy(), False, whole)
            assert np.abs(got - want).max() < 1e-4, (H, W, whole)
    for (H, W, Ci, Co) in ((2, 4, 3, 4), (1, 8, 2, 3), (4, 2, 5, 2)):
        x = torch.randn(1, Ci, H, W, generator=g)
        w = torch.randn(Ci, Co, 4, 4, generator=g)
        want = F.conv_transpose2d(x, w, None, stride=2, padding=1)[0].permute(1, 2, 0).numpy()
        for whole in (True, False):
            got = KM.patch_engine_layer(x[0].permute(1, 2, 0).numpy(), w.numpy(), True, whole)
            assert np.abs(got - want).max() < 1e-4, (H, W, whole)


def test_bf16x3_weight_gradient_row_ring_and_column_shift_match_torch():
    """CPU: the index arithmetic of wgrad3x3_row_bf16x3_kernel (ring of transposed rows, column shift on dY, rows outside the image
    skipped, splits that start and end mid-image) and of conv_wgrad_reduce_torch_kernel (split-sum order, re-layout, dropped padding
    channels), restated in numpy, against torch's weight gradient; the split products' error stays at the bf16x3 level."""
    import torch
    import torch.nn.functional as F
    g = np.random.default_rng(7)
    B, H, C, N = 3, 5, 32, 16
    x = g.standard_normal((B, H, 32, C)).astype(np.float32)
    x[..., 30:] = 0                                           # two padding channels: no gradient, dropped by the re-layout
    dy = g.standard_normal((B, H, 32, N)).astype(np.float32)
    w = torch.zeros(N, 30, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(torch.from_numpy(x[..., :30]).double().permute(0, 3, 1, 2), w, None, 1, 1).backward(torch.from_numpy(dy).double().permute(0, 3, 1, 2))
    want = w.grad.numpy()
    for splits in (1, 4, 7, 15, 16, 23):                      # 15 rows in all: splits of one row and of none; both split-sum orders
        slabs = KM.wgrad3x3_row_bf16x3(x, dy, splits)
        got = KM.wgrad_reduce_torch(slabs, 30, 3, 3)
        assert got.shape == (N, 30, 3, 3)
        err = np.abs(got - want).sum() / np.abs(want).sum()
        assert err < 2e-5, (splits, err)
    exact = KM.wgrad3x3_row(x, dy)                            # the fp32 kernel's model: the same gradient
    assert np.abs(KM.wgrad3x3_row_bf16x3(x, dy, 3).sum(0) - exact).max() < 2e-3
