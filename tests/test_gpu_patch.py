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

LABEL = "igemm_patch<256,128>"     # N a multiple of 128
LABEL64 = "igemm_patch<512,64>"    # the 64-wide decoder stage


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


# (B, H, W of the input, C0, C1, Co, transposed): the stages of the U-Net the engine takes at 256 / 128 / 64 frames and odd batches --
# part of an image per tile (halo patch: 5 x 65, 9 x 33 rows) and whole images per tile (1, 2, 4, 8, 16 of them: rows past M, images
# past the batch, every image edge), both tiles
CASES = [
    (2, 16, 128, 64, 0, 128, False),     # down1 at 256 frames: 8 x 64 outputs, 4 rows per tile (halo patch)
    (1, 32, 64, 32, 0, 128, False),      # 16 x 32 outputs, 8 rows per tile (halo patch)
    (3, 8, 64, 128, 0, 256, False),      # down2 at 256 frames: 4 x 32 outputs, two images per tile, odd batch, two n-tiles
    (5, 8, 32, 128, 0, 256, False),      # down2 at 128 frames: 4 x 16 outputs, four images per tile, ragged
    (1, 32, 32, 32, 0, 128, False),      # 16 x 16 outputs, one image per tile
    (5, 4, 16, 64, 0, 128, False),       # 2 x 8 outputs, sixteen images per tile
    (3, 4, 32, 256, 256, 256, True),     # up2 at 256 frames: two sources, two images per tile, odd batch
    (9, 2, 16, 128, 128, 256, True),     # up1 at 256 frames: 2 x 16 grid, eight images per tile, ragged
    (2, 8, 16, 64, 32, 128, True),       # 16 wide, two images per tile, unequal sources
    (2, 16, 16, 96, 0, 128, True),       # one source, one image per tile
    (1, 8, 64, 128, 128, 64, True),      # up3 at 256 frames: the 512 x 64 tile, one image per tile
    (3, 8, 32, 64, 64, 64, True),        # up3 at 128 frames: two images per tile, odd batch
    (5, 4, 8, 32, 32, 64, True),         # 8 wide, sixteen images per tile
]


@pytest.mark.parametrize("knob", [2, 3])   # 2: the engine's own choice of patch form; 3: the whole-image form wherever it fits
@pytest.mark.parametrize("B,H,W,C0,C1,Co,transposed", CASES)
def test_patch_engine_layer_matches_torch_and_the_dma_engine(B, H, W, C0, C1, Co, transposed, knob):
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
        ops.debug_set(36, knob)       # the engine also below its tile-count threshold
        got, label = _layer(*args)
        again, _ = _layer(*args)
        ops.debug_set(36, -1)
        ops.debug_set(27, 2)
        ref, ref_label = _layer(*args)
    finally:
        ops.debug_set(36, 0)
        ops.debug_set(27, 0)
        ops.set_math_mode(ops.MATH_FP32)
    assert label == (LABEL if Co % 128 == 0 else LABEL64) and not ref_label.startswith("igemm_patch")
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
    # stages (1-5 encoder, 6-9 decoder) the engine takes when forced: the two window geometries with the whole window reached
    # (not down4 / up0: 1 x 8 pixel rows), more than 64 GEMM rows
    taken = {i for i, lb in enumerate(got[2]) if lb.startswith("igemm_patch")}
    assert not any(lb.startswith("igemm_patch") for lb in ref[2])
    assert taken == ({2, 3, 4, 7, 8, 9} if tm >= 128 else {2, 9}), got[2]
    assert got[2][9] == LABEL64 and got[2][2] == LABEL
    assert O.rel_l1(got[0].cpu(), ref[0].cpu()) < 1e-5 and O.rel_l1(got[1].cpu(), ref[1].cpu()) < 1e-5
    assert not torch.equal(got[0], ref[0])
    assert torch.equal(again[0], got[0]) and torch.equal(again[1], got[1])
    with torch.no_grad():
        want_m, want_mono = O.passive_pair(sd, torch.from_numpy(mixed), torch.from_numpy(tc))
    em = torch.expm1(torch.from_numpy(mixed))
    assert O.rel_l1(got[0].cpu() * em, want_m * em) < 1e-4 and O.rel_l1(got[1].cpu(), want_mono) < 1e-4


@pytest.mark.parametrize("B,H,W,C0,C1,Co,transposed", [
    (16, 4, 32, 64, 64, 256, True),     # transposed conv, two n-tiles x four phases x 8 m-tiles = 64 tiles
    (16, 8, 64, 64, 0, 128, False),     # conv, halo patch form, 16 m-tiles
    (32, 8, 16, 64, 0, 64, True),       # 64-wide transposed conv on the 512 x 64 tile: 8 m-tiles x four phases
])
def test_patch_engine_result_does_not_depend_on_the_persistent_grid(B, H, W, C0, C1, Co, transposed):
    """One workgroup per CU walks tiles L, L + G, ...: with G = 8 / 24 / 40 workgroups (tuning knob 10) a workgroup's consecutive tiles differ in
    phase and n-tile (G % 32 != 0), it crosses many tile boundaries, and the last workgroups hold one tile fewer -- the values must be those of
    one workgroup per tile, bit for bit (a tile's summation order does not depend on who computes it)."""
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B + H + W + Co)
    x = torch.randn(B, C0, H, W, generator=g)
    x2 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    Ci = C0 + C1
    w = torch.randn(Ci, Co, 4, 4, generator=g) * 0.05 if transposed else torch.randn(Co, Ci, 4, 4, generator=g) * 0.05
    scale, shift = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    nhwc = lambda t: ops.split32(t.permute(0, 2, 3, 1).contiguous().to(dev))  # noqa: E731
    wp = ops.split32(ops.pack_convT_weight(w.to(dev)) if transposed else ops.pack_conv_weight(w.to(dev)))
    args = (nhwc(x), nhwc(x2) if C1 else None, wp, Co, transposed, scale.to(dev), shift.to(dev), 0.0 if transposed else 0.2)
    ops.set_math_mode(ops.MATH_BF16X3)
    outs = {}
    try:
        for grid in (None, 8, 24, 40):
            ops.debug_set(36, 2)
            ops.debug_set(10, 1 << 20 if grid is None else grid)    # (more workgroups than tiles: one workgroup per tile)
            got, label = _layer(*args)
            assert label.startswith("igemm_patch"), label
            outs[grid] = got.cpu()
    finally:
        ops.debug_set(36, 0)
        ops.debug_set(10, 0)
        ops.set_math_mode(ops.MATH_FP32)
    for grid in (8, 24, 40):
        assert torch.equal(outs[grid], outs[None]), grid


def test_patch_engine_two_class_halves_as_split_k():
    """Half a chip's worth of tiles and a long reduction (the fourth encoder stage at the benchmark batch; here 128 images, N = 1024):
    the engine's own dispatch (no knob) splits the window's classes into two K-halves + the ordered reduce kernel."""
    from m2h import ops
    dev = _dev()
    B, H, W, Ci, Co = 128, 4, 32, 128, 1024
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 4, 4, generator=g) * (1.0 / (16 * Ci) ** 0.5)
    scale = torch.rand(Co, generator=g) + 0.5
    shift = torch.randn(Co, generator=g) * 0.1
    want = F.leaky_relu(F.conv2d(x, w, None, stride=2, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.2)
    args = (ops.split32(x.permute(0, 2, 3, 1).contiguous().to(dev)), None, ops.split32(ops.pack_conv_weight(w.to(dev))), Co, False,
            scale.to(dev), shift.to(dev), 0.2)
    ops.set_math_mode(ops.MATH_BF16X3)
    try:
        got, label = _layer(*args)
        again, _ = _layer(*args)
        ops.debug_set(36, -1)
        ref, ref_label = _layer(*args)
    finally:
        ops.debug_set(36, 0)
        ops.set_math_mode(ops.MATH_FP32)
    assert label == "igemm_patch<256,128> + split-K reduce" and ref_label == "igemm_dma<256,128> + split-K reduce"
    got, ref = got.cpu().permute(0, 3, 1, 2), ref.cpu().permute(0, 3, 1, 2)
    assert O.rel_l1(got, want) < 1e-5 and (got - want).abs().max() < 2e-4 * want.abs().max()
    assert O.rel_l1(got, ref) < 3e-6
    assert torch.equal(again.cpu().permute(0, 3, 1, 2), got)
