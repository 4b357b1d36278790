"""GPU: the shared-patch LDS-DMA engine (csrc/conv_patch.hip: the pixel operand of a 4x4/s2 conv / a transposed-conv phase staged
once per four taps) -- single layers against torch on the CPU (separator_cnn.py:5-24: Conv2d(4, 2, 1) / ConvTranspose2d(4, 2, 1)
+ BatchNorm(eval) + LeakyReLU / ReLU) and against the LDS-DMA engine it replaces at the benchmark batch, then the whole runner
pair with it forced on every layer it takes, through the C-ABI.  bf16x3 arithmetic on split32 operands."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import m2h_oracle as O
from m2h import synthetic

pytestmark = pytest.mark.gpu

LABEL = "igemm_patch<256,128>"


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda", 0)


def unsplit32(t):
    b = t.contiguous().view(torch.bfloat16).reshape(-1, 64)
    return (b[:, :32].float() + b[:, 32:].float()).reshape(t.shape)


def _layer(x, x2, wp, Co, transposed, scale, shift, slope):
    """One split32 layer through m2h_conv_igemm_f32; returns (NHWC fp32 values, label of the kernel that ran)."""
    from m2h import _lib, ops
    B, H, W, C0 = x.shape
    C1 = x2.shape[3] if x2 is not None else 0
    Ho, Wo = (2 * H, 2 * W) if transposed else (H // 2, W // 2)
    out = torch.empty((B, Ho, Wo, Co), device=x.device, dtype=torch.float32)
    a = _lib.ConvArgs()
    a.src0, a.src1, a.C0, a.C1 = x.data_ptr(), (x2.data_ptr() if x2 is not None else None), C0, C1
    if transposed:
        a.B, a.Hi, a.Wi, a.Hq, a.Wq = B, H, W, H, W
        a.stride, a.nth, a.ntw, a.mulh, a.offh, a.mulw, a.offw = 1, 2, 2, 0, 0, 0, 0
        a.conv_transpose, a.os = 1, 2
    else:
        a.B, a.Hi, a.Wi, a.Hq, a.Wq = B, H, W, Ho, Wo
        a.stride, a.nth, a.ntw, a.mulh, a.offh, a.mulw, a.offw = 2, 4, 4, 1, -1, 1, -1
        a.conv_transpose, a.os = 0, 1
    a.wp, a.N = wp.data_ptr(), Co
    a.scale, a.shift, a.slope, a.cls_table, a.cls_val = scale.data_ptr(), shift.data_ptr(), float(slope), None, None
    a.dst, a.Ho, a.Wo, a.ph, a.pw, a.ldc, a.out_mode = out.data_ptr(), Ho, Wo, 0, 0, Co, ops.OUT_NHWC
    a.operand_format = ops.FMT_SRC_SPLIT | ops.FMT_W_SPLIT | ops.FMT_DST_SPLIT
    lib = _lib.load()
    with torch.cuda.device(x.device):
        ws, wsb = ops._workspace(lib.m2h_conv_igemm_workspace_bytes(ctypes.byref(a)), x.device)
        a.workspace, a.workspace_bytes = (ws.data_ptr() if ws is not None else None), wsb
        _lib.check(lib.m2h_conv_igemm_f32(ctypes.byref(a), ops._stream(x)), "m2h_conv_igemm_f32")
    return unsplit32(out), ops.last_kernel()


# (B, H, W of the input, C0, C1, Co, transposed): the four wide stages of the U-Net at 256 / 128 / 64 frames and odd batches --
# one segment per tile (5 x 65, 9 x 33 and 17 x 17 patch rows), two and four images per tile (rows past M, a segment past the batch)
CASES = [
    (2, 16, 128, 64, 0, 128, False),     # down1 at 256 frames: 8 x 64 outputs, 4 rows per tile
    (3, 8, 64, 128, 0, 256, False),      # down2 at 256 frames: 4 x 32 outputs, two images per tile, odd batch, two n-tiles
    (5, 8, 32, 128, 0, 256, False),      # down2 at 128 frames: 4 x 16 outputs, four images per tile, ragged
    (1, 32, 32, 32, 0, 128, False),      # 16 x 16 outputs, 16 rows per tile
    (3, 4, 32, 256, 256, 256, True),     # up1 at 256 frames: two sources, two images per tile, odd batch
    (1, 8, 64, 128, 128, 128, True),     # up2 at 256 frames
    (2, 8, 16, 64, 32, 128, True),       # 16 wide, two images per tile, unequal sources
    (2, 16, 16, 96, 0, 128, True),       # one source, 16 rows per tile
]


@pytest.mark.parametrize("B,H,W,C0,C1,Co,transposed", CASES)
def test_patch_engine_layer_matches_torch_and_the_dma_engine(B, H, W, C0, C1, Co, transposed):
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W + C0)
    x = torch.randn(B, C0, H, W, generator=g)
    x2 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    Ci = C0 + C1
    if transposed:
        w = torch.randn(Ci, Co, 4, 4, generator=g) * (1.0 / (4 * Ci) ** 0.5)
    else:
        w = torch.randn(Co, Ci, 4, 4, generator=g) * (1.0 / (16 * Ci) ** 0.5)
    scale = torch.rand(Co, generator=g) + 0.5
    shift = torch.randn(Co, generator=g) * 0.1
    slope = 0.0 if transposed else 0.2
    xin = torch.cat((x, x2), 1) if C1 else x
    y = F.conv_transpose2d(xin, w, None, stride=2, padding=1) if transposed else F.conv2d(xin, w, None, stride=2, padding=1)
    want = F.leaky_relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), slope)
    nhwc = lambda t: ops.split32(t.permute(0, 2, 3, 1).contiguous().to(dev))  # noqa: E731
    wp = ops.split32(ops.pack_convT_weight(w.to(dev)) if transposed else ops.pack_conv_weight(w.to(dev)))
    args = (nhwc(x), nhwc(x2) if C1 else None, wp, Co, transposed, scale.to(dev), shift.to(dev), slope)
    ops.set_math_mode(ops.MATH_BF16X3)
    try:
        ops.debug_set(36, 2)          # the engine also below its tile-count threshold
        got, label = _layer(*args)
        again, _ = _layer(*args)
        ops.debug_set(36, -1)
        ops.debug_set(27, 2)
        ref, ref_label = _layer(*args)
    finally:
        ops.debug_set(36, 0)
        ops.debug_set(27, 0)
        ops.set_math_mode(ops.MATH_FP32)
    assert label == LABEL and ref_label.startswith("igemm_dma")
    got, ref = got.cpu().permute(0, 3, 1, 2), ref.cpu().permute(0, 3, 1, 2)
    assert got.shape == want.shape
    assert O.rel_l1(got, want) < 1e-5 and (got - want).abs().max() < 2e-4 * want.abs().max()   # every pixel: borders, seams, both sources
    assert O.rel_l1(got, ref) < 3e-6 and (got - ref).abs().max() < 5e-5 * want.abs().max()     # the engine it replaces: summation order only
    assert torch.equal(again.cpu().permute(0, 3, 1, 2), got)


@pytest.mark.parametrize("B,tm", [(3, 256), (5, 128), (2, 64)])
def test_runner_with_the_patch_engine_matches_the_dma_engine_and_the_oracle(B, tm):
    """m2h_unet_fwd (both U-Nets) with the patch engine forced on every stage it takes against the same call without it, and
    against the oracle."""
    from m2h import ops
    dev = _dev()
    from m2h.common.spaces import move2hear_observation_space
    from m2h.pretrain.passive.policy import Move2HearPassiveWoMemoryPolicy
    pol = Move2HearPassiveWoMemoryPolicy(move2hear_observation_space())
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), 4).items()}
    pol.load_state_dict(sd, strict=True)
    pol = pol.to(dev).eval()
    mixed, tc = synthetic.make_passive_inputs(B, tm, 60 + B)
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}

    def run(knob):
        ops.debug_set(36, knob)
        try:
            with torch.no_grad():
                m = pol.get_binSepMasks(obs)
                labels = ops.unet_stage_kernels()
                return m, pol.convert_bin2mono(m, mixed_audio=obs["mixed_bin_audio_mag"]), labels
        finally:
            ops.debug_set(36, 0)

    ops.set_math_mode(ops.MATH_BF16X3)
    try:
        ref = run(-1)
        got = run(2)
        again = run(2)
    finally:
        ops.set_math_mode(ops.MATH_FP32)
    assert LABEL not in ref[2]
    assert got[2].count(LABEL) == (3 if tm >= 128 else 1)   # down1, down2, up2 (up1's 2 x 16 grid needs 408 patch rows; at 64 frames down2's and up2's grids are 8 wide)
    assert O.rel_l1(got[0].cpu(), ref[0].cpu()) < 1e-5 and O.rel_l1(got[1].cpu(), ref[1].cpu()) < 1e-5
    assert not torch.equal(got[0], ref[0])
    assert torch.equal(again[0], got[0]) and torch.equal(again[1], got[1])
    with torch.no_grad():
        want_m, want_mono = O.passive_pair(sd, torch.from_numpy(mixed), torch.from_numpy(tc))
    em = torch.expm1(torch.from_numpy(mixed))
    assert O.rel_l1(got[0].cpu() * em, want_m * em) < 1e-4 and O.rel_l1(got[1].cpu(), want_mono) < 1e-4
