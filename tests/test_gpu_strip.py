"""GPU: the strip-walker kernels (csrc/conv_strip.hip) -- slice + first encoder stage in one launch, last decoder stage + head
in one launch -- against the CPU oracle's layers (separator_cnn.py:73-105, :128-135, :153-168) and against the tiled engines the
runner used before (m2h_tuning_set 35 = -1), through the C-ABI.  bf16x3 arithmetic: held to 1e-5 rel-L1 like the other
engine-vs-engine tests (contract 1e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import m2h_oracle as O
from m2h import synthetic

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda", 0)


def _policy(seed, dev):
    from m2h.common.spaces import move2hear_observation_space
    from m2h.pretrain.passive.policy import Move2HearPassiveWoMemoryPolicy
    pol = Move2HearPassiveWoMemoryPolicy(move2hear_observation_space())
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), seed).items()}
    pol.load_state_dict(sd, strict=True)
    return pol.to(dev).eval(), sd


def unsplit32(t):
    """split32 tensor (fp32-typed storage; every 32 values = 32 bf16 hi + 32 bf16 lo) -> the fp32 values hi + lo."""
    b = t.contiguous().view(torch.bfloat16).reshape(-1, 64)
    return (b[:, :32].float() + b[:, 32:].float()).reshape(t.shape)


@pytest.mark.parametrize("B,tm,masked", [(3, 64, False), (2, 128, True), (1, 256, False), (5, 64, True), (2, 192, False)])
def test_strip_conv1_matches_the_oracle_first_stage(B, tm, masked):
    """m2h_strip_conv1_fwd == slice (+ pre-op, + class plane) -> conv4x4/s2/p1 -> BN(eval) -> LeakyReLU(0.2) of the oracle, every
    output pixel: image borders (zero padding rows / columns, the class plane's nine border classes), strip seams (T = 128, 192,
    256: two, three, four strips), ragged batch."""
    from m2h import ops
    dev = _dev()
    pol, sd = _policy(5, dev)
    mixed, tc = synthetic.make_passive_inputs(B, tm, 90 + B)
    mix = torch.from_numpy(mixed)
    tcl = torch.from_numpy(tc)
    if masked:
        enc, pre = pol.bin2mono_enc.passive_sep_encoder, "bin2mono_enc.passive_sep_encoder.cnn."
        masks = torch.randn(mix.shape, generator=torch.Generator().manual_seed(3)) * 0.7 + 0.3
        x = O.sep_enc_input(mix, None, masks)
    else:
        enc, pre = pol.binSep_enc.passive_sep_encoder, "binSep_enc.passive_sep_encoder.cnn."
        masks = None
        x = O.sep_enc_input(mix, tcl)
    want = F.leaky_relu(O._bn_eval(F.conv2d(x, sd[pre + "0.0.weight"], None, stride=2, padding=1), sd, pre + "0.1."), 0.2)
    (wp, scale, shift, table, _co) = enc._packed()[0]
    wreg = ops.pack_strip_conv1(enc.cnn[0][0].weight.detach().contiguous())
    cls_val = None if masked else (tcl.reshape(-1).float() + 1.0).to(dev)
    got = ops.strip_conv1_fwd(mix.to(dev), None if masks is None else masks.to(dev), wreg, scale, shift, table, cls_val)
    got = unsplit32(got).cpu().permute(0, 3, 1, 2)
    assert got.shape == want.shape
    assert O.rel_l1(got, want) < 1e-5
    assert (got - want).abs().max() < 2e-4 * want.abs().max()   # no pixel is off: borders, seams, every channel


@pytest.mark.parametrize("B,W,Co", [(3, 32, 32), (2, 64, 16), (1, 128, 32), (5, 32, 16), (2, 96, 32)])
def test_strip_last_stage_matches_the_oracle_layer(B, W, Co):
    """m2h_strip_last_fwd == cat -> ConvTranspose2d(4, 2, 1) -> BN(eval) -> ReLU -> conv1x1 + bias -> de-slice of the oracle, every
    output element: image borders (missing taps), strip seams, both output widths (binSep 2 channels, bin2mono 1), ragged batch."""
    from m2h import ops
    dev = _dev()
    pol, sd = _policy(6, dev)
    H = 16
    dec, pre = (pol.binSep_dec.passive_sep_decoder, "binSep_dec.passive_sep_decoder.cnn.") if Co == 32 else \
        (pol.bin2mono_dec.passive_sep_decoder, "bin2mono_dec.passive_sep_decoder.cnn.")
    g = torch.Generator().manual_seed(100 + B + W)
    x = torch.relu(torch.randn(B, 64, H, W, generator=g)) * 0.8        # previous decoder stage (post-ReLU)
    skip = F.leaky_relu(torch.randn(B, 64, H, W, generator=g), 0.2)    # first encoder stage (post-LeakyReLU)
    out = F.conv_transpose2d(torch.cat((x, skip), dim=1), sd[pre + "4.0.weight"], None, stride=2, padding=1)
    out = F.relu(O._bn_eval(out, sd, pre + "4.1."))
    want = O.deslice_freq(F.conv2d(out, sd[pre + "5.0.weight"], sd[pre + "5.0.bias"]))
    ups, (hw, hb, hco) = dec._packed()
    wp, scale, shift, co = ups[4]
    assert co == Co and hco == Co
    nhwc = lambda t: ops.split32(t.permute(0, 2, 3, 1).contiguous().to(dev))  # noqa: E731
    got = ops.strip_last_fwd(nhwc(x), nhwc(skip), ops.split32(wp), scale, shift, hw, hb, Co).cpu()
    assert got.shape == want.shape == (B, 512, 2 * W, Co // 16)
    assert O.rel_l1(got, want) < 1e-5
    assert (got - want).abs().max() < 2e-4 * want.abs().max()


@pytest.mark.parametrize("B,tm", [(3, 64), (1, 256), (2, 128)])
def test_runner_with_strip_kernels_matches_the_tiled_engines(B, tm):
    """The whole separator pair through m2h_unet_fwd with the strip kernels (default) against the same call with them switched
    off (m2h_tuning_set 35 = -1: slice kernel + tiled first stage, tiled last stage), and against the oracle."""
    from m2h import ops
    dev = _dev()
    pol, sd = _policy(3, dev)
    mixed, tc = synthetic.make_passive_inputs(B, tm, 40 + B)
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}

    def run(knob):
        ops.debug_set(35, knob)
        try:
            with torch.no_grad():
                m = pol.get_binSepMasks(obs)
                return m, pol.convert_bin2mono(m, mixed_audio=obs["mixed_bin_audio_mag"])
        finally:
            ops.debug_set(35, 0)

    ops.set_math_mode(ops.MATH_BF16X3)
    try:
        ref = run(-1)
        got = run(0)
        again = run(0)
    finally:
        ops.set_math_mode(ops.MATH_FP32)
    # (the strip kernel runs the 1x1 head in bf16x3 like the rest of the layer, the tiled kernel in fp32 MFMA: 3e-5 between the two)
    assert O.rel_l1(got[0].cpu(), ref[0].cpu()) < 3e-5 and O.rel_l1(got[1].cpu(), ref[1].cpu()) < 3e-5
    assert not torch.equal(got[0], ref[0])          # the strip kernels really ran (another summation order)
    assert torch.equal(again[0], got[0]) and torch.equal(again[1], got[1])
    with torch.no_grad():
        want_m, want_mono = O.passive_pair(sd, torch.from_numpy(mixed), torch.from_numpy(tc))
    em = torch.expm1(torch.from_numpy(mixed))
    assert O.rel_l1(got[0].cpu() * em, want_m * em) < 1e-4 and O.rel_l1(got[1].cpu(), want_mono) < 1e-4
