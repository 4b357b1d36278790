"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C-ABI, against the
CPU oracle and the committed golden fixtures.  Tolerance: the north-star contract is rel-L1 <= 1e-3 on the
fp32 separated spectrograms; the fp32 MFMA path is an exact fp32 FMA chain, so we hold it to 2e-5."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import kernel_model as KM
import m2h_oracle as O
from m2h import synthetic

pytestmark = pytest.mark.gpu

TOL = 2e-5  # rel-L1, fp32 path (contract: 1e-3)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda", 0)


def _policy(seed, dev):
    from m2h.common.spaces import move2hear_observation_space
    from m2h.pretrain.passive.policy import Move2HearPassiveWoMemoryPolicy
    pol = Move2HearPassiveWoMemoryPolicy(move2hear_observation_space())
    sd_np = synthetic.make_state_dict(synthetic.passive_shapes(), seed)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
    pol.load_state_dict(sd, strict=True)
    return pol.to(dev).eval(), sd


def test_library_loads_and_reports_version():
    from m2h import _lib
    lib = _lib.load()
    assert lib.m2h_version() == 100


def test_slice_input_matches_model():
    from m2h import ops
    dev = _dev()
    mixed, _ = synthetic.make_passive_inputs(2, 32, 21)
    masks = np.random.default_rng(0).standard_normal(mixed.shape).astype(np.float32)
    out = ops.sep_slice_input(torch.from_numpy(mixed).to(dev)).cpu().numpy()
    assert np.array_equal(out, KM.sep_slice_input(mixed))
    out2 = ops.sep_slice_input(torch.from_numpy(mixed).to(dev), torch.from_numpy(masks).to(dev)).cpu()
    ref = O.slice_freq(torch.log1p(torch.clamp(torch.from_numpy(masks) * (torch.exp(torch.from_numpy(mixed)) - 1), min=0)))
    assert torch.allclose(out2.permute(0, 3, 1, 2), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 32, 32, 32, 64), (3, 8, 8, 128, 256), (1, 2, 16, 512, 512), (5, 4, 4, 64, 128)])
def test_down_conv_matches_torch(B, H, W, Ci, Co):
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 4, 4, generator=g) * (2.0 / (Ci * 16)) ** 0.5
    gamma, beta = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    mean, var = torch.randn(Co, generator=g) * 0.1, torch.rand(Co, generator=g) + 0.5
    ref = F.leaky_relu(F.batch_norm(F.conv2d(x, w, None, 2, 1), mean, var, gamma, beta, False, 0.1, 1e-5), 0.2)
    wp = ops.pack_conv_weight(w.to(dev))
    sc, sh = ops.fold_bn(gamma.to(dev), beta.to(dev), mean.to(dev), var.to(dev), 1e-5)
    y = ops.unet_down_fwd(x.permute(0, 2, 3, 1).contiguous().to(dev), wp, sc, sh, Co)
    assert y.shape == (B, H // 2, W // 2, Co)
    assert O.rel_l1(y.cpu().permute(0, 3, 1, 2), ref) < TOL


@pytest.mark.parametrize("B,H,W,C0,C1,Co", [(2, 1, 1, 512, 0, 512), (2, 2, 2, 512, 512, 256), (3, 8, 8, 128, 128, 64),
                                            (1, 16, 16, 64, 64, 32), (2, 16, 16, 64, 64, 16), (1, 4, 32, 256, 256, 128)])
def test_up_conv_matches_torch(B, H, W, C0, C1, Co):
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 100 + H + C1)
    x = torch.randn(B, C0, H, W, generator=g)
    s = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(C0 + C1, Co, 4, 4, generator=g) * (2.0 / ((C0 + C1) * 4)) ** 0.5
    gamma, beta = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    mean, var = torch.randn(Co, generator=g) * 0.1, torch.rand(Co, generator=g) + 0.5
    xin = x if s is None else torch.cat((x, s), 1)
    ref = F.relu(F.batch_norm(F.conv_transpose2d(xin, w, None, 2, 1), mean, var, gamma, beta, False, 0.1, 1e-5))
    wp = ops.pack_convT_weight(w.to(dev))
    sc, sh = ops.fold_bn(gamma.to(dev), beta.to(dev), mean.to(dev), var.to(dev), 1e-5)
    y = ops.unet_up_fwd(x.permute(0, 2, 3, 1).contiguous().to(dev),
                        None if s is None else s.permute(0, 2, 3, 1).contiguous().to(dev), wp, sc, sh, Co)
    assert y.shape == (B, 2 * H, 2 * W, Co)
    assert O.rel_l1(y.cpu().permute(0, 3, 1, 2), ref) < TOL


@pytest.mark.parametrize("Co", [32, 16])
def test_head_matches_torch(Co):
    from m2h import ops
    dev = _dev()
    g = torch.Generator().manual_seed(Co)
    B, H, W = 2, 32, 32
    x = torch.randn(B, Co, H, W, generator=g)
    w = torch.randn(Co, Co, 1, 1, generator=g) * 0.2
    b = torch.randn(Co, generator=g) * 0.1
    ref = O.deslice_freq(F.conv2d(x, w, b))
    out = ops.unet_head_fwd(x.permute(0, 2, 3, 1).contiguous().to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev), Co)
    assert out.shape == (B, 512, W, Co // 16)
    assert O.rel_l1(out.cpu(), ref) < TOL


def test_pair_tm32_matches_golden_and_oracle(golden_dir):
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "unet_tm32.npz"))
    pol, sd = _policy(int(g["seed_w"]), dev)
    mixed, tc = synthetic.make_passive_inputs(int(g["B"]), 32, int(g["seed_x"]))
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}
    with torch.no_grad():
        bott, skips = pol.binSep_enc(obs)
        masks = pol.get_binSepMasks(obs)
        mono = pol.convert_bin2mono(masks, mixed_audio=obs["mixed_bin_audio_mag"])
    assert masks.shape == (2, 512, 32, 2) and mono.shape == (2, 512, 32, 1)
    assert bott.shape == (2, 512) and [tuple(s.shape) for s in skips] == [(2, 512, 2, 2), (2, 256, 4, 4), (2, 128, 8, 8), (2, 64, 16, 16)]
    assert torch.allclose(bott.cpu(), torch.from_numpy(g["bottleneck_binSep"]), atol=2e-5, rtol=1e-4)
    assert O.rel_l1(skips[3].cpu(), torch.from_numpy(g["binSep_skip3_full"])) < TOL
    gm, gmono = torch.from_numpy(g["masks"]), torch.from_numpy(g["mono"])
    assert O.rel_l1(masks.cpu(), gm) < TOL
    assert O.rel_l1(mono.cpu(), gmono) < TOL
    # the contract metric: separated spectrograms
    mix = torch.from_numpy(mixed)
    assert O.rel_l1(O.pred_bin(masks.cpu(), mix), O.pred_bin(gm, mix)) < TOL
    # live oracle on the same inputs
    with torch.no_grad():
        mo, mono_o = O.passive_pair(sd, mix, torch.from_numpy(tc))
    assert O.rel_l1(masks.cpu(), mo) < TOL and O.rel_l1(mono.cpu(), mono_o) < TOL


def test_pair_tm256_fully_convolutional_matches_golden(golden_dir):
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "unet_tm256.npz"))
    pol, _ = _policy(int(g["seed_w"]), dev)
    mixed, tc = synthetic.make_passive_inputs(1, 256, int(g["seed_x"]))
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}
    with torch.no_grad():
        masks = pol.get_binSepMasks(obs)
        mono = pol.convert_bin2mono(masks, mixed_audio=obs["mixed_bin_audio_mag"])
    assert O.rel_l1(masks.cpu(), torch.from_numpy(g["masks"])) < TOL
    assert O.rel_l1(mono.cpu(), torch.from_numpy(g["mono"])) < TOL


@pytest.mark.parametrize("B", [1, 7, 14])
def test_pair_ragged_batches_match_oracle(B):
    dev = _dev()
    pol, sd = _policy(2, dev)
    mixed, tc = synthetic.make_passive_inputs(B, 32, 100 + B)
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}
    with torch.no_grad():
        masks = pol.get_binSepMasks(obs)
        mono = pol.convert_bin2mono(masks, mixed_audio=obs["mixed_bin_audio_mag"])
        mo, mono_o = O.passive_pair(sd, torch.from_numpy(mixed), torch.from_numpy(tc))
    assert O.rel_l1(masks.cpu(), mo) < TOL and O.rel_l1(mono.cpu(), mono_o) < TOL


def test_batch_independence_and_determinism():
    """Size-independent properties at the full benchmark shape: repeated runs are bit-identical (the split-K
    reduction is ordered, no atomics) and a sample's result does not depend on its batch neighbours (eval-mode
    BN) beyond fp32 summation order (the split-K factor depends on the batch size)."""
    dev = _dev()
    pol, _ = _policy(2, dev)
    mixed, tc = synthetic.make_passive_inputs(8, 256, 77)
    mix, tct = torch.from_numpy(mixed).to(dev), torch.from_numpy(tc).to(dev)
    with torch.no_grad():
        big = pol.get_binSepMasks({"mixed_bin_audio_mag": mix, "target_class": tct})
        big2 = pol.get_binSepMasks({"mixed_bin_audio_mag": mix, "target_class": tct})
        one = pol.get_binSepMasks({"mixed_bin_audio_mag": mix[5:6].contiguous(), "target_class": tct[5:6]})
    assert torch.equal(big, big2)
    assert O.rel_l1(big[5:6].cpu(), one.cpu()) < 1e-6


def test_weight_update_invalidates_packed_cache():
    dev = _dev()
    pol, _ = _policy(2, dev)
    mixed, tc = synthetic.make_passive_inputs(1, 32, 9)
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}
    with torch.no_grad():
        a = pol.get_binSepMasks(obs)
        pol.binSep_dec.passive_sep_decoder.cnn[5][0].bias.add_(1.0)
        b = pol.get_binSepMasks(obs)
    assert torch.allclose(b, a + 1.0, atol=1e-5)


def test_error_behaviour():
    from m2h import ops
    dev = _dev()
    with pytest.raises(RuntimeError):
        ops.sep_slice_input(torch.zeros(1, 512, 32, 2))  # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        ops.sep_slice_input(torch.zeros(1, 500, 32, 2, device=dev))  # F % 16 != 0 -> C-ABI argument error
    pol, _ = _policy(2, dev)
    with pytest.raises(NotImplementedError):  # eval-mode (folded BN) path has no autograd: refuses instead of dropping the graph
        pol.get_binSepMasks({"mixed_bin_audio_mag": torch.zeros(1, 512, 32, 2, device=dev), "target_class": torch.zeros(1, 1, device=dev)})
    pol.train()
    with pytest.raises(RuntimeError):  # train-mode BN needs more than one value per channel (torch raises here too)
        pol.get_binSepMasks({"mixed_bin_audio_mag": torch.zeros(1, 512, 32, 2, device=dev), "target_class": torch.zeros(1, 1, device=dev)})


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_whole_network_runner_is_bitwise_the_module_chain(mode):
    """m2h_unet_fwd (one C call per U-Net) gives the results of the encoder/decoder module chain in both arithmetic modes: bit-identical
    wherever the two run the same kernels (fp32; bf16x3 with the LDS-DMA engine switched off), to fp32 summation order otherwise.  In bf16x3 the runner works on split32 weights and keeps its intermediates in split32 (no operand conversion in any
    k-loop) while the module chain converts inside the kernels: hi/lo are the same values either way.  tm 256 reaches the
    tap-sharing kernel and the long-M tiles."""
    from m2h import ops
    from m2h.rl.models.separator_cnn import unet_forward
    dev = _dev()
    pol, _ = _policy(2, dev)
    ops.set_math_mode(ops.MATH_BF16X3 if mode == "bf16x3" else ops.MATH_FP32)
    try:
        for B, tm in ((3, 32), (2, 64), (1, 256)):
            mixed, tc = synthetic.make_passive_inputs(B, tm, 40 + B)
            obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}
            with torch.no_grad():
                fast_m = pol.get_binSepMasks(obs)
                fast_mono = pol.convert_bin2mono(fast_m, mixed_audio=obs["mixed_bin_audio_mag"])
                chain_m = pol.binSep_dec(*pol.binSep_enc(obs))
                chain_mono = pol.bin2mono_dec(*pol.bin2mono_enc(chain_m, mixed_audio=obs["mixed_bin_audio_mag"]))
            if mode == "fp32":
                assert torch.equal(fast_m, chain_m) and torch.equal(fast_mono, chain_mono)
            else:
                # the runner's wide layers run on the LDS-DMA engine (16x16x32 MFMAs: 32 channels per accumulation step), the
                # module chain on the register engine (32x32x16: 16 per step): same products, fp32 sums in a different order
                # (and, for tm % 64 == 0, the first and last stage on the strip-walker kernels, whose head runs in bf16x3 too: 3e-5)
                assert O.rel_l1(fast_m.cpu(), chain_m.cpu()) < 3e-5 and O.rel_l1(fast_mono.cpu(), chain_mono.cpu()) < 3e-5
                ops.debug_set(27, -1)     # without them the two paths are the same kernels on the same values: bit-identical
                ops.debug_set(35, -1)
                try:
                    with torch.no_grad():
                        reg_m = pol.get_binSepMasks(obs)
                        reg_mono = pol.convert_bin2mono(reg_m, mixed_audio=obs["mixed_bin_audio_mag"])
                finally:
                    ops.debug_set(27, 0)
                    ops.debug_set(35, 0)
                assert torch.equal(reg_m, chain_m) and torch.equal(reg_mono, chain_mono)
        # the event-recording entry point: same result, 11 positive kernel durations
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
        for e in evs:
            e.record()
        with torch.no_grad():
            m_ev = unet_forward(pol.binSep_enc.passive_sep_encoder, pol.binSep_dec.passive_sep_decoder, obs["mixed_bin_audio_mag"], None,
                                obs["target_class"], events=evs)
        torch.cuda.synchronize()
        assert torch.equal(m_ev, fast_m)
        assert all(evs[i].elapsed_time(evs[i + 1]) > 0 for i in range(1, 11))   # (interval 0, the slice, is empty when the strip kernel fuses it)
    finally:
        ops.set_math_mode(ops.MATH_FP32)


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_runner_takes_the_raw_target_class_in_every_dtype(mode):
    """The class plane's value "target_class.float() + 1" (reference separator_cnn.py:96) is made inside the runner's first kernel from the
    raw target_class (m2h_unet_weights.cls_kind): int64 (the simulator's dtype), float32 (the rollout storage's) and any other dtype (the
    host-side conversion) give bit-identical masks, on the slice kernel's path (tm 32) and on the strip kernel's (tm 64, bf16x3)."""
    from m2h import ops
    from m2h.rl.models.separator_cnn import unet_forward
    dev = _dev()
    pol, _ = _policy(4, dev)
    enc, dec = pol.binSep_enc.passive_sep_encoder, pol.binSep_dec.passive_sep_decoder
    ops.set_math_mode(ops.MATH_BF16X3 if mode == "bf16x3" else ops.MATH_FP32)
    try:
        for B, tm in ((5, 32), (3, 64)):
            mixed, tc = synthetic.make_passive_inputs(B, tm, 90 + B)
            mix = torch.from_numpy(mixed).to(dev)
            tcl = torch.from_numpy(tc).to(dev)
            with torch.no_grad():
                outs = [unet_forward(enc, dec, mix, None, tcl.to(dt)) for dt in (torch.int64, torch.float32, torch.int32, torch.float64)]
                chain = pol.binSep_dec(*pol.binSep_enc({"mixed_bin_audio_mag": mix, "target_class": tcl}))
            for o in outs[1:]:
                assert torch.equal(o, outs[0])
            if mode == "fp32":
                assert torch.equal(outs[0], chain)
            else:
                assert O.rel_l1(outs[0].cpu(), chain.cpu()) < 3e-5
    finally:
        ops.set_math_mode(ops.MATH_FP32)


def test_strip_kernels_give_the_same_values_in_either_walking_direction():
    """The strip kernels' persistent workgroups walk the images downwards or upwards (knob 39; the last stage's default is downwards so
    that the next U-Net finds its input in the memory-side cache): every (image, strip) job is independent, so the results are
    bit-identical, also with a last group of fewer than 8 images (empty job slots)."""
    from m2h import ops
    dev = _dev()
    pol, _ = _policy(6, dev)
    mixed, tc = synthetic.make_passive_inputs(11, 64, 123)
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}
    ops.set_math_mode(ops.MATH_BF16X3)
    try:
        res = []
        for v in (0, 1, 2, 3, 7):
            ops.debug_set(39, v)
            with torch.no_grad():
                m = pol.get_binSepMasks(obs)
                res.append((m, pol.convert_bin2mono(m, mixed_audio=obs["mixed_bin_audio_mag"])))
        for m, mono in res[1:]:
            assert torch.equal(m, res[0][0]) and torch.equal(mono, res[0][1])
    finally:
        ops.debug_set(39, 0)
        ops.set_math_mode(ops.MATH_FP32)


@pytest.mark.parametrize("B,tm", [(3, 32), (1, 256), (5, 64)])
def test_dma_engine_matches_register_engine(B, tm):
    """The LDS-DMA engine (csrc/conv_dma.hip: split32 operands DMA'd into an LDS ring, fragments read through a row permutation)
    against the register-staged engine on the whole runner pair -- plain and transposed convs, the skip concat's second source,
    zero-padded borders, rows past M, the class plane, split32 output.  Forced on every wide layer (m2h_tuning_set 27 = 2: its
    256 x 128 tile also below the tile-count threshold) since the test batches are too small for the automatic choice.  With
    32x32x16 fragments (knob 28 = 32) the two engines run the same products in the same order: bit-identical; with the default
    16x16x32 fragments the fp32 sums associate differently: equal to summation order.  Then the four-phase transposed-conv kernel
    (csrc/convt_quad.hip) against the per-phase kernels.  The strip-walker kernels are switched off throughout (knob 35 = -1;
    tests/test_gpu_strip.py covers them), so the first and last stages run on the engines compared here."""
    from m2h import ops
    dev = _dev()
    pol, _ = _policy(3, dev)
    mixed, tc = synthetic.make_passive_inputs(B, tm, 70 + B)
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}

    def run(dma, shape, splitk=0, quad=-1):
        ops.debug_set(35, -1)
        ops.debug_set(27, dma)
        ops.debug_set(28, shape)
        ops.debug_set(0, splitk)
        ops.debug_set(30, quad)
        try:
            with torch.no_grad():
                m = pol.get_binSepMasks(obs)
                return m, pol.convert_bin2mono(m, mixed_audio=obs["mixed_bin_audio_mag"])
        finally:
            ops.debug_set(27, 0)
            ops.debug_set(28, 0)
            ops.debug_set(0, 0)
            ops.debug_set(30, 0)
            ops.debug_set(35, 0)

    ops.set_math_mode(ops.MATH_BF16X3)
    try:
        ref = run(-1, 0, -1)           # the 256 x 128 tile never splits K: compare it with the register engine's unsplit sums
        same = run(2, 32, -1)          # the register engine's MFMA shape and k-tile order
        assert torch.equal(same[0], ref[0]) and torch.equal(same[1], ref[1])
        got = run(2, 0, -1)
        assert O.rel_l1(got[0].cpu(), ref[0].cpu()) < 1e-5 and O.rel_l1(got[1].cpu(), ref[1].cpu()) < 1e-5   # contract: 1e-3
        assert not torch.equal(got[0], ref[0])     # the engine really ran (another summation order)
        again = run(2, 0, -1)
        assert torch.equal(again[0], got[0]) and torch.equal(again[1], got[1])
        # the four-phase transposed-conv kernel on every decoder stage it takes (m2h_tuning_set 30 = 1: also below its block-count
        # threshold), the other layers on the register engine: its sums run (chunk, half, tap) instead of (chunk, tap, half)
        ref = run(-1, 0)
        quad = run(-1, 0, 0, 1)
        assert O.rel_l1(quad[0].cpu(), ref[0].cpu()) < 1e-5 and O.rel_l1(quad[1].cpu(), ref[1].cpu()) < 1e-5
        assert torch.equal(quad[0], ref[0]) == (tm < 64)     # its stages are at least 32 pixels wide: tm >= 64
        quad2 = run(-1, 0, 0, 1)
        assert torch.equal(quad2[0], quad[0]) and torch.equal(quad2[1], quad[1])
    finally:
        ops.set_math_mode(ops.MATH_FP32)


@pytest.mark.parametrize("tm,B", [(32, 2), (256, 1)])
def test_pair_bf16x3_math_mode_within_contract(golden_dir, tm, B):
    """MATH_BF16X3 (fp32 operands split into bf16 hi/lo, three bf16 MFMA products, fp32 accumulate) against the reference
    fixtures: the contract is 1e-3 rel-L1 on the separated spectrograms (BASELINE north_star); measured ~1e-5, so the test
    holds it to 1e-4.  Also: the mode really changes the arithmetic (results differ from the fp32 mode) and is deterministic."""
    from m2h import ops
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "unet_tm%d.npz" % tm))
    pol, _ = _policy(int(g["seed_w"]), dev)
    mixed, tc = synthetic.make_passive_inputs(B, tm, int(g["seed_x"]))
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}

    def run():
        with torch.no_grad():
            m = pol.get_binSepMasks(obs)
            return m, pol.convert_bin2mono(m, mixed_audio=obs["mixed_bin_audio_mag"])

    exact = run()
    ops.set_math_mode(ops.MATH_BF16X3)
    try:
        masks, mono = run()
        masks2, mono2 = run()
    finally:
        ops.set_math_mode(ops.MATH_FP32)
    gm, gmono, mix = torch.from_numpy(g["masks"]), torch.from_numpy(g["mono"]), torch.from_numpy(mixed)
    assert O.rel_l1(masks.cpu(), gm) < 1e-4 and O.rel_l1(mono.cpu(), gmono) < 1e-4
    assert O.rel_l1(O.pred_bin(masks.cpu(), mix), O.pred_bin(gm, mix)) < 1e-4
    assert torch.equal(masks, masks2) and torch.equal(mono, mono2)
    assert not torch.equal(masks, exact[0])


@pytest.mark.parametrize("Ci,Co,H,W", [(64, 128, 16, 32), (512, 512, 4, 8), (32, 64, 8, 256)])
def test_split32_operands_are_bitwise_the_on_the_fly_split(Ci, Co, H, W):
    """bf16x3 math with operands converted to the split32 layout beforehand (and the output written in it) must equal the
    in-kernel split bit for bit: hi = bf16(x), lo = bf16(x - hi) are the same values wherever they are computed."""
    from m2h import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(7)
    B = 3
    x = torch.randn(B, H, W, Ci, device=dev, generator=g)
    wp = torch.randn(Co, 16 * Ci, device=dev, generator=g) * 0.05
    sc = torch.rand(Co, device=dev, generator=g) + 0.5
    sh = torch.randn(Co, device=dev, generator=g) * 0.1
    ops.set_math_mode(ops.MATH_BF16X3)
    try:
        ref = ops.conv2d_nhwc(x, wp, Co, 4, 4, stride=2, pad=1, bias=sh, scale=sc, slope=0.2)
        xs, ws = ops.split32(x), ops.split32(wp)
        got = ops.conv2d_nhwc(xs, ws, Co, 4, 4, stride=2, pad=1, bias=sh, scale=sc, slope=0.2,
                              operand_format=ops.FMT_SRC_SPLIT | ops.FMT_W_SPLIT)
        got_split = ops.conv2d_nhwc(xs, ws, Co, 4, 4, stride=2, pad=1, bias=sh, scale=sc, slope=0.2,
                                    operand_format=ops.FMT_SRC_SPLIT | ops.FMT_W_SPLIT | ops.FMT_DST_SPLIT)
        with pytest.raises(RuntimeError):   # sources and weights must be converted together
            ops.conv2d_nhwc(xs, wp, Co, 4, 4, stride=2, pad=1, operand_format=ops.FMT_SRC_SPLIT)
    finally:
        ops.set_math_mode(ops.MATH_FP32)
    assert torch.equal(got, ref)
    assert torch.equal(got_split, ops.split32(ref))
    # the layout itself: hi + lo reconstructs x to 2^-16 relative
    raw = xs.view(torch.int16).view(B, H, W, Ci // 32, 2, 32)
    hi = (raw[..., 0, :].to(torch.int32) << 16).view(torch.float32)
    lo = (raw[..., 1, :].to(torch.int32) << 16).view(torch.float32)
    rec = (hi + lo).reshape(B, H, W, Ci)
    assert ((rec - x).abs() <= x.abs() * 2.0 ** -16 + 1e-30).all()


def test_graphed_pair_replays_the_direct_result():
    """m2h.graphs.GraphedSeparatorPair: the pair replayed from a HIP graph returns bit-identical tensors to the direct calls,
    follows new data copied into the captured input tensors, and re-captures when the arithmetic mode changes."""
    from m2h import ops
    from m2h.graphs import GraphedSeparatorPair
    dev = _dev()
    pol, _ = _policy(1, dev)
    mixed, tc = synthetic.make_passive_inputs(3, 32, 11)
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed).to(dev), "target_class": torch.from_numpy(tc).to(dev)}
    g = GraphedSeparatorPair(pol, obs)

    def direct():
        with torch.no_grad():
            m = pol.get_binSepMasks(obs)
            return m, pol.convert_bin2mono(m, mixed_audio=obs["mixed_bin_audio_mag"])
    for i, mode in enumerate((ops.MATH_FP32, ops.MATH_BF16X3)):
        ops.set_math_mode(mode)
        try:
            obs["mixed_bin_audio_mag"].copy_(torch.from_numpy(synthetic.make_passive_inputs(3, 32, 30 + i)[0]))
            m0, mono0 = direct()
            m1, mono1 = g()
            assert torch.equal(m0, m1) and torch.equal(mono0, mono1)
            mixed2, tc2 = synthetic.make_passive_inputs(3, 32, 12)
            obs["mixed_bin_audio_mag"].copy_(torch.from_numpy(mixed2))
            obs["target_class"].copy_(torch.from_numpy(tc2))
            m2, mono2 = direct()
            assert not g.stale()
            m3, mono3 = g()
            assert torch.equal(m2, m3) and torch.equal(mono2, mono3) and not torch.equal(m2, m0)
        finally:
            ops.set_math_mode(ops.MATH_FP32)


@pytest.mark.parametrize("math", ["fp32", "bf16x3"])
def test_tap_window_skips_only_exact_zeros(math):
    """Tiny images: a 2-row input under the 4x4/s2/p1 conv (the deepest encoder stage) and a 1-row input under the transposed
    conv (the first decoder stage) have whole kernel rows / columns in the zero padding for EVERY output pixel; the engine
    walks only the tap window that reaches the image.  The skipped products are exact zeros: results are bit-identical to the
    full walk (knob 18 = -1), with and without split-K."""
    from m2h import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(3)
    ops.set_math_mode(ops.MATH_BF16X3 if math == "bf16x3" else ops.MATH_FP32)
    ops.debug_set(23, -1)   # keep the small-M shapes on the tiled engine (the skinny kernels split K over their four waves, so a
    ops.debug_set(24, -1)   # shorter walk regroups the sums): this test isolates the tap window
    try:
        for (B, H, W) in ((3, 2, 2), (14, 2, 2), (5, 2, 16), (2, 4, 4)):
            x = torch.randn(B, H, W, 512, device=dev, generator=g)
            wp = torch.randn(512, 16 * 512, device=dev, generator=g) * 0.02
            sc, sh = torch.rand(512, device=dev, generator=g) + 0.5, torch.randn(512, device=dev, generator=g) * 0.1
            outs = []
            for knob in (0, -1):
                ops.debug_set(18, knob)
                outs.append(ops.unet_down_fwd(x, wp, sc, sh, 512).clone())
            assert torch.equal(outs[0], outs[1]), ("down", B, H, W)
        for (B, H, W) in ((3, 1, 1), (14, 1, 1), (4, 1, 8), (2, 2, 2)):
            x = torch.randn(B, H, W, 512, device=dev, generator=g)
            skip = torch.randn(B, H, W, 512, device=dev, generator=g) if H > 1 else None
            c1 = 512 if skip is not None else 0
            wp = torch.randn(4, 512, 4 * (512 + c1), device=dev, generator=g) * 0.02
            sc, sh = torch.rand(512, device=dev, generator=g) + 0.5, torch.randn(512, device=dev, generator=g) * 0.1
            outs = []
            for knob in (0, -1):
                ops.debug_set(18, knob)
                outs.append(ops.unet_up_fwd(x, skip, wp, sc, sh, 512).clone())
            assert torch.equal(outs[0], outs[1]), ("up", B, H, W)
    finally:
        ops.debug_set(18, 0)
        ops.debug_set(23, 0)
        ops.debug_set(24, 0)
        ops.set_math_mode(ops.MATH_FP32)


def test_skinny_rows_kernel_on_the_bottleneck_stages():
    """At the rollout batch the two U-Net stages around the 1 x 1 bottleneck are weight streams against <= 16 rows: the deepest
    encoder conv (2 x 2 input, tap window = the whole image) and the first transposed conv (1 x 1 input, one tap per phase) run
    on the skinny MFMA kernel; against the tiled engine (knob 23 = -1), including BN scale / shift and the activation."""
    from m2h import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(13)
    for B in (1, 5, 14, 16):
        x = torch.randn(B, 2, 2, 512, device=dev, generator=g)
        wp = torch.randn(512, 16 * 512, device=dev, generator=g) * 0.02
        sc, sh = torch.rand(512, device=dev, generator=g) + 0.5, torch.randn(512, device=dev, generator=g) * 0.1
        x1 = torch.randn(B, 1, 1, 512, device=dev, generator=g)
        wu = torch.randn(4, 512, 4 * 512, device=dev, generator=g) * 0.02
        outs = {}
        for knob in (0, -1):
            ops.debug_set(23, knob)
            try:
                outs[knob] = (ops.unet_down_fwd(x, wp, sc, sh, 512).clone(), ops.unet_up_fwd(x1, None, wu, sc, sh, 512).clone())
            finally:
                ops.debug_set(23, 0)
        for a, b in zip(outs[0], outs[-1]):
            assert a.shape == b.shape and O.rel_l1(a.cpu(), b.cpu()) < 1e-5, B


def test_skinny_gather_kernel_on_the_deep_stages_at_the_rollout_batch():
    """16 < M <= 256 pixels per phase (the U-Net's deep stages at 14 envs): 32 x 16 tiles with both operands straight into the
    16x16x4 fp32 MFMA, taps gathered as the engine gathers them, two-source (skip) transposed convs included; against the tiled
    engine (knob 24 = -1) with BN scale / shift and the activations."""
    from m2h import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(29)
    cases = []
    for (B, H, W, Ci, Co) in ((14, 4, 4, 256, 512), (14, 8, 8, 128, 256), (3, 4, 4, 256, 512), (5, 2, 16, 512, 512), (2, 16, 16, 64, 128)):
        x = torch.randn(B, H, W, Ci, device=dev, generator=g)
        wp = torch.randn(Co, 16 * Ci, device=dev, generator=g) * 0.03
        sc, sh = torch.rand(Co, device=dev, generator=g) + 0.5, torch.randn(Co, device=dev, generator=g) * 0.1
        cases.append(("down", (B, H, W, Ci, Co), lambda x=x, wp=wp, sc=sc, sh=sh, Co=Co: ops.unet_down_fwd(x, wp, sc, sh, Co)))
    for (B, H, W, C0, C1, Co) in ((14, 2, 2, 512, 512, 256), (14, 4, 4, 256, 256, 128), (3, 2, 2, 512, 512, 256), (14, 1, 1, 512, 0, 512)):
        x = torch.randn(B, H, W, C0, device=dev, generator=g)
        skip = torch.randn(B, H, W, C1, device=dev, generator=g) if C1 else None
        wp = torch.randn(4, Co, 4 * (C0 + C1), device=dev, generator=g) * 0.03
        sc, sh = torch.rand(Co, device=dev, generator=g) + 0.5, torch.randn(Co, device=dev, generator=g) * 0.1
        cases.append(("up", (B, H, W, C0, C1, Co), lambda x=x, skip=skip, wp=wp, sc=sc, sh=sh, Co=Co: ops.unet_up_fwd(x, skip, wp, sc, sh, Co)))
    for kind, shape, fn in cases:
        outs = {}
        for knob in (0, -1):
            ops.debug_set(24, knob)
            try:
                outs[knob] = fn().clone()
            finally:
                ops.debug_set(24, 0)
        assert outs[0].shape == outs[-1].shape and O.rel_l1(outs[0].cpu(), outs[-1].cpu()) < 1e-5, (kind, shape)
