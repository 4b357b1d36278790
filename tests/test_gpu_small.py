"""GPU: the small-batch conv engine (csrc/conv_small.hip, m2h_conv_small_fwd) -- single layers against torch's convolutions on the
CPU at the shapes of the separator U-Nets at the rollout batch (separator_cnn.py:46-52,128-135 at 14 envs), every feature of the
engine on its own: channel-group slabs summed by the consumer with the producer's epilogue, two concatenated sources, the tap
window of tiny images, ragged image tiles, the four sub-pixel phases of the transposed conv, the first stage's fused slice /
pre-op / class plane, the last stage's fused head + de-slice.  Tolerance: fp32 sums in another association, rel-L1 <= 2e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import m2h_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _rel(a, b):
    return O.rel_l1(torch.as_tensor(a).cpu(), torch.as_tensor(b).cpu())


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def _slabs(x_nhwc, S, g):
    """x as S random partial sums (x = sum of the slabs)."""
    parts = [torch.randn(x_nhwc.shape, generator=g) for _ in range(S - 1)]
    last = x_nhwc - sum(parts) if parts else x_nhwc
    return torch.stack(parts + [last], 0).contiguous()


@pytest.mark.parametrize("B,H,C,N,tiling,S", [
    (14, 8, 128, 256, (4, 4, 32, 16, 4), 1),      # down2: four images per tile, ragged last tile (2 images)
    (14, 4, 256, 512, (14, 2, 32, 16, 4), 4),     # down3: one tile, four slabs in
    (14, 2, 512, 512, (14, 1, 64, 16, 4), 8),     # down4: 2 x 2 image, tap window 2 x 2
    (5, 16, 64, 128, (1, 8, 32, 32, 2), 2),       # down1: two column groups per block
    (3, 8, 128, 256, (4, 4, 64, 64, 4), 1),       # sixteen waves
])
def test_conv_layers_match_torch(B, H, C, N, tiling, S):
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + H)
    raw = torch.randn(B, C, H, H, generator=g)
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    x = F.leaky_relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.2)        # what the consumer must see
    w = torch.randn(N, C, 4, 4, generator=g) / (4 * C ** 0.5)
    want = F.conv2d(x, w, stride=2, padding=1)
    src = _slabs(_nhwc(raw), S, g).to(dev)
    wp = ops.pack_conv_weight(w.to(dev))
    out = ops.conv_small([(src, scale.to(dev), shift.to(dev), 0.2)], wp, N, B, H, H, tiling=tiling)
    assert out.shape == (C // tiling[2], B, H // 2, H // 2, N)
    got = out.sum(0).cpu().permute(0, 3, 1, 2)
    assert _rel(got, want) < 2e-6, _rel(got, want)


@pytest.mark.parametrize("B,H,C0,C1,N,tiling", [
    (14, 1, 512, 0, 512, (14, 1, 64, 16, 1)),      # up0: 1 x 1 image, one tap per phase
    (14, 2, 512, 512, 256, (14, 2, 64, 16, 1)),    # up1: two sources
    (14, 2, 512, 512, 256, (14, 2, 128, 16, 2)),   # ... eight waves
    (6, 4, 256, 256, 128, (4, 4, 64, 32, 1)),      # up2: ragged tile, two column groups
    (3, 8, 128, 128, 64, (1, 4, 64, 16, 4)),       # up3: half-image tiles, sixteen waves
])
def test_transposed_conv_layers_match_torch(B, H, C0, C1, N, tiling):
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 77 + H)
    x0 = torch.randn(B, C0, H, H, generator=g)
    srcs = [(_slabs(_nhwc(x0), 2, g).to(dev), None, None, 1.0)]
    x = x0
    if C1:
        raw1 = torch.randn(B, C1, H, H, generator=g)
        sc, sh = torch.rand(C1, generator=g) + 0.5, torch.randn(C1, generator=g) * 0.1
        x = torch.cat((x0, F.relu(raw1 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))), 1)
        srcs.append((_slabs(_nhwc(raw1), 3, g).to(dev), sc.to(dev), sh.to(dev), 0.0))
    w = torch.randn(C0 + C1, N, 4, 4, generator=g) / (2 * (C0 + C1) ** 0.5)
    want = F.conv_transpose2d(x, w, stride=2, padding=1)
    out = ops.conv_small(srcs, ops.pack_convT_weight(w.to(dev)), N, B, H, H, conv_transpose=True, tiling=tiling)
    got = out.sum(0).cpu().permute(0, 3, 1, 2)
    assert _rel(got, want) < 2e-6, _rel(got, want)


@pytest.mark.parametrize("with_masks", [False, True])
def test_first_stage_slice_preop_class_plane(with_masks):
    """separator_cnn.py:73-105: pre-op, 16-way slice, (target_class + 1) plane, Conv2d(4, 2, 1), BatchNorm(eval), LeakyReLU."""
    from m2h import ops
    dev = _dev()
    B, T = 3, 32
    g = torch.Generator().manual_seed(9)
    mix = torch.rand(B, 512, T, 2, generator=g) * 2
    masks = torch.randn(B, 512, T, 2, generator=g) if with_masks else None
    tc = torch.randint(0, 11, (B, 1), generator=g)
    x = O.sep_enc_input(mix, None if with_masks else tc, masks)            # NCHW [B, 33 | 32, 32, T]
    Ci = x.shape[1]
    w = torch.randn(64, Ci, 4, 4, generator=g) / (4 * Ci ** 0.5)
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    want = F.leaky_relu(F.conv2d(x, w, stride=2, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.2)
    wp = ops.pack_conv_weight(w.to(dev), ci_used=32)
    kw = {}
    if not with_masks:
        kw = dict(cls_table=ops.unet_class_table(w.to(dev), 32), cls_val=(tc.float() + 1).reshape(-1).to(dev))
    out = ops.conv_small(None, wp, 64, B, 32, T, tiling=(1, 4, 32, 64, 2), finish=1, scale=scale.to(dev), shift=shift.to(dev), slope=0.2,
                         mix=mix.to(dev), masks=masks.to(dev) if with_masks else None, **kw)
    assert _rel(out.cpu().permute(0, 3, 1, 2), want) < 2e-6


@pytest.mark.parametrize("N,tiling", [(32, (1, 2, 128, 32, 1)), (16, (1, 4, 128, 16, 2)), (32, (1, 1, 128, 32, 2))])
def test_last_stage_head_deslice(N, tiling):
    """separator_cnn.py:128-135, :156-168: cat, ConvTranspose2d(4, 2, 1), BatchNorm(eval), ReLU, Conv2d(1x1, bias), de-slice to BHWC."""
    from m2h import ops
    dev = _dev()
    B, H = 3, 16
    g = torch.Generator().manual_seed(N)
    x0, x1 = torch.randn(B, 64, H, H, generator=g), torch.randn(B, 64, H, H, generator=g)
    w = torch.randn(128, N, 4, 4, generator=g) / 22.0
    scale, shift = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    hw, hb = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g)
    y = F.relu(F.conv_transpose2d(torch.cat((x0, x1), 1), w, stride=2, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    y = F.conv2d(y, hw.view(N, N, 1, 1), hb)
    want = O.deslice_freq(y)                                                 # BHWC [B, 512, 32, N/16]
    out = ops.conv_small([(_nhwc(x0).to(dev), None, None, 1.0), (_nhwc(x1).to(dev), None, None, 1.0)], ops.pack_convT_weight(w.to(dev)), N, B, H, H,
                         conv_transpose=True, tiling=tiling, finish=2, scale=scale.to(dev), shift=shift.to(dev), slope=0.0,
                         head_w=hw.to(dev).contiguous(), head_b=hb.to(dev))
    assert out.shape == (B, 512, 32, N // 16)
    assert _rel(out.cpu(), want) < 2e-6


def test_small_engine_refuses_what_it_cannot_tile():
    from m2h import ops
    dev = _dev()
    x = torch.zeros(2, 8, 8, 64, device=dev)
    wp = torch.zeros(64, 16 * 64, device=dev)
    with pytest.raises(RuntimeError, match="exceeds 64 GEMM rows"):
        ops.conv_small([(x, None, None, 1.0)], wp, 64, 2, 8, 8, tiling=(4, 8, 32, 16, 4))
    with pytest.raises(RuntimeError, match="power of two"):
        ops.conv_small([(x, None, None, 1.0)], wp, 64, 2, 8, 8, tiling=(1, 4, 48, 16, 4))
    with pytest.raises(RuntimeError, match="finishing layer"):
        ops.conv_small([(x, None, None, 1.0)], wp, 64, 2, 8, 8, tiling=(1, 4, 32, 16, 4), finish=1)
