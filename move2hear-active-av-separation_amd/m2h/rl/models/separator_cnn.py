"""Separator U-Nets on MI355X: drop-in for audio_separation/rl/models/separator_cnn.py.

Same class names, constructor arguments, ``forward`` signatures/return conventions and ``state_dict``
keys as the reference (``cnn.{i}.0.weight``, ``cnn.{i}.1.{weight,bias,running_mean,running_var,
num_batches_tracked}``, decoder ``cnn.5.0.{weight,bias}``; reference :46-52,128-135), so reference
checkpoints load unchanged.  The torch.nn layers below are parameter containers only (they also give
bit-identical default initialisation under the same seed, reference :56-68,139-151); the arithmetic
runs in libm2h.so:

  encoder  (reference :70-108)  K1/K2 slice kernel -> 5 x fused conv4x4s2+BN(eval)+LeakyReLU (MFMA
           implicit GEMM, NHWC); the (target_class + 1) plane enters stage 0 as a border-aware bias.
  decoder  (reference :153-170) 5 x fused [skip-concat]+convT4x4s2+BN(eval)+ReLU (4 sub-pixel phase
           GEMMs, concat read in place) -> fused conv1x1+bias+de-slice straight into BHWC.

Both stacks are fully convolutional in time: Tm = 32 is the reference-native shape, any multiple of 32
works (the reference's ``view(B,-1,1,1)`` at :154 pins Tm = 32).

Feature maps cross the module boundary as NCHW-shaped tensors in channels-last memory (zero-copy views of
the NHWC buffers the kernels use).

Two execution modes, chosen like torch does by ``module.training``:
  eval   (every RL call site: ppo_trainer.py:557-577, ppo.py:184-195)  the fused inference kernels above, BN folded;
         no autograd (raises NotImplementedError if gradients are required rather than silently dropping the graph).
  train  (passive pre-training, passive_trainer.py:211-249)  per stage: raw conv / transposed conv (same MFMA engine) ->
         train-mode BatchNorm + activation kernel (batch statistics, running stats updated) with full autograd through
         m2h.functional (wgrad/dgrad/BN-backward kernels).  The (target_class+1) plane is a real channel in this mode so
         that its weight gradient falls out of the ordinary wgrad.
"""
import torch
import torch.nn as nn

from ... import functional as MF
from ... import ops


def unet_conv(input_nc, output_nc, norm_layer=nn.BatchNorm2d):
    # parameter containers: Conv2d(4,2,1,no bias) + BN + LeakyReLU(0.2)   (reference :5-12)
    return nn.Sequential(nn.Conv2d(input_nc, output_nc, kernel_size=(4, 4), stride=(2, 2), padding=(1, 1), bias=False),
                         norm_layer(output_nc), nn.LeakyReLU(0.2, True))


def unet_upconv(input_nc, output_nc, norm_layer=nn.BatchNorm2d):
    # parameter containers: ConvTranspose2d(4,2,1,no bias) + BN + ReLU    (reference :15-24, outermost=False)
    return nn.Sequential(nn.ConvTranspose2d(input_nc, output_nc, kernel_size=(4, 4), stride=(2, 2), padding=(1, 1), bias=False),
                         norm_layer(output_nc), nn.ReLU(True))


def _init_like_reference(cnn, a):
    # reference :56-68 / :139-151: kaiming_normal_(weight, <gain passed as `a`>), BN weight=1, bias=0
    for module in cnn:
        for layer in module:
            if isinstance(layer, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
                nn.init.kaiming_normal_(layer.weight, a)
                if layer.bias is not None:
                    nn.init.constant_(layer.bias, val=0)
            elif isinstance(layer, (nn.BatchNorm1d, nn.BatchNorm2d)):
                if layer.affine:
                    layer.weight.data.fill_(1)
                    layer.bias.data.zero_()


class _PackedCache:
    """Packed weights + folded BN, rebuilt when any source tensor is modified in place or replaced."""

    def __init__(self):
        self.key = None
        self.val = None

    def get(self, tensors, build):
        # trainable sources: FlatAdam and the train-mode BN kernels write through raw pointers, which torch's version counters
        # do not see, so the key carries the optimizer epoch too (frozen separators -- every RL call site -- keep one key for
        # good: their packed buffers are never rebuilt and HIP graphs may hold their addresses)
        epoch = MF.param_epoch() if any(t.requires_grad for t in tensors) else -1
        key = tuple((t.data_ptr(), t._version, t.device) for t in tensors) + (epoch,)
        if key != self.key:
            self.val = build()
            self.key = key
        return self.val


def _check_inference(module, *tensors):
    if torch.is_grad_enabled() and (any(p.requires_grad for p in module.parameters())
                                    or any(t is not None and t.requires_grad for t in tensors)):
        raise NotImplementedError(
            "m2h %s: eval-mode (folded BatchNorm) forward has no autograd; call .train() for the differentiable path, or wrap "
            "the call in torch.no_grad() / freeze the separator as ppo_trainer.py:557-577 does" % type(module).__name__)


def _as_nhwc(t):
    """NCHW-shaped tensor (any strides) -> contiguous NHWC buffer; free for channels-last inputs."""
    return t.permute(0, 2, 3, 1).contiguous()


class PassiveSepEncCNN(nn.Module):
    r"""U-net encoder for passive separation (reference separator_cnn.py:27-108)."""

    def __init__(self, convert_bin2mono=False):
        super().__init__()
        self._convert_bin2mono = convert_bin2mono
        self._slice_factor = 16
        self._n_input_audio = 2 * self._slice_factor
        if not convert_bin2mono:
            self._n_input_audio += 1
        self.cnn = nn.Sequential(
            unet_conv(self._n_input_audio, 64),
            unet_conv(64, 64 * 2),
            unet_conv(64 * 2, 64 * 4),
            unet_conv(64 * 4, 64 * 8),
            unet_conv(64 * 8, 64 * 8),
        )
        self.layer_init()
        self._cache = _PackedCache()

    def layer_init(self):
        _init_like_reference(self.cnn, nn.init.calculate_gain("leaky_relu", 0.2))

    def _packed(self):
        srcs = [t for m in self.cnn for t in (m[0].weight, m[1].weight, m[1].bias, m[1].running_mean, m[1].running_var)]

        def build():
            out = []
            for i, m in enumerate(self.cnn):
                conv, bn = m[0], m[1]
                w = conv.weight.detach().contiguous()
                ci_used = 32 if (i == 0) else w.shape[1]
                wp = ops.pack_conv_weight(w, ci_used)
                table = ops.unet_class_table(w, 32) if (i == 0 and not self._convert_bin2mono) else None
                scale, shift = ops.fold_bn(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
                out.append((wp, scale, shift, table, w.shape[0]))
            return out

        return self._cache.get(srcs, build)

    def _forward_train(self, observations, mixed_audio):
        if self._convert_bin2mono:
            if observations.requires_grad:
                raise NotImplementedError("m2h PassiveSepEncCNN(train): gradients into pred_binSepMasks are not built (the passive "
                                          "trainer detaches them, passive_trainer.py:229-230)")
            out = ops.sep_slice_input(mixed_audio.contiguous(), observations.detach().contiguous())
        else:
            cls_val = (observations["target_class"].reshape(-1).to(torch.float32) + 1.0).contiguous()
            out = ops.sep_slice_input_plane(observations["mixed_bin_audio_mag"].contiguous(), cls_val, 36)
        if not hasattr(self, "_memo_t"):
            self._memo_t = [MF._PackMemo() for _ in range(5)]
        feats = []
        for i, m in enumerate(self.cnn):
            z = MF.conv2d(out, m[0].weight, None, 2, 1, slope=1.0, memo=self._memo_t[i], name="unet_down.train")
            out = MF.bn_act_train(z, m[1], 0.2)
            feats.append(out.permute(0, 3, 1, 2))
        bottleneck = feats[-1]
        return bottleneck.reshape(bottleneck.size(0), -1), feats[:-1][::-1]

    def forward(self, observations, mixed_audio=None):
        if self.training:
            if self._convert_bin2mono:
                assert mixed_audio is not None
            return self._forward_train(observations, mixed_audio)
        if self._convert_bin2mono:
            assert mixed_audio is not None
            _check_inference(self, observations, mixed_audio)
            x0 = ops.sep_slice_input(mixed_audio.contiguous(), observations.contiguous())
            cls_val = None
        else:
            mix = observations["mixed_bin_audio_mag"]
            _check_inference(self, mix)
            x0 = ops.sep_slice_input(mix.contiguous())
            cls_val = (observations["target_class"].reshape(-1).to(torch.float32) + 1.0).contiguous()  # reference :96
        feats = []
        out = x0
        for (wp, scale, shift, table, co) in self._packed():
            out = ops.unet_down_fwd(out, wp, scale, shift, co, table, cls_val if table is not None else None)
            feats.append(out.permute(0, 3, 1, 2))  # NCHW-shaped view of the NHWC buffer
        bottleneck = feats[-1]
        return bottleneck.reshape(bottleneck.size(0), -1), feats[:-1][::-1]


class PassiveSepDecCNN(nn.Module):
    r"""U-net decoder for passive separation (reference separator_cnn.py:111-170)."""

    def __init__(self, convert_bin2mono=False):
        super().__init__()
        self._slice_factor = 16
        self._n_out_audio = self._slice_factor
        if not convert_bin2mono:
            self._n_out_audio *= 2
        self.cnn = nn.Sequential(
            unet_upconv(64 * 8, 64 * 8),
            unet_upconv(64 * 16, 64 * 4),
            unet_upconv(64 * 8, 64 * 2),
            unet_upconv(64 * 4, 64 * 1),
            unet_upconv(64 * 2, self._n_out_audio),
            nn.Sequential(nn.Conv2d(self._n_out_audio, self._n_out_audio, kernel_size=(1, 1))),
        )
        self.layer_init()
        self._cache = _PackedCache()

    def layer_init(self):
        _init_like_reference(self.cnn, nn.init.calculate_gain("relu"))

    def _packed(self):
        srcs = [t for m in list(self.cnn)[:5] for t in (m[0].weight, m[1].weight, m[1].bias, m[1].running_mean, m[1].running_var)]
        srcs += [self.cnn[5][0].weight, self.cnn[5][0].bias]

        def build():
            ups = []
            for m in list(self.cnn)[:5]:
                convT, bn = m[0], m[1]
                w = convT.weight.detach().contiguous()
                scale, shift = ops.fold_bn(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
                ups.append((ops.pack_convT_weight(w), scale, shift, w.shape[1]))
            head = self.cnn[5][0]
            hw = ops.pack_conv_weight(head.weight.detach().contiguous())
            return ups, (hw, head.bias.detach().contiguous(), head.weight.shape[0])

        return self._cache.get(srcs, build)

    def _forward_train(self, bottleneck_feats, lst_skip_feats):
        B = bottleneck_feats.size(0)
        wb = lst_skip_feats[0].size(3) // 2
        # (the backward pass is half done when it reaches the bottleneck: inside MF.wgrad_side_branches() the decoder's deferred weight
        #  gradients are launched from here on a side stream, under the encoder's backward chain.  ONE such point per network: a second
        #  one inside the encoder made the step 2.79 instead of 2.36 ms, profiles/r06_wgrad_side_ab.txt)
        out = MF.wgrad_flush_point(_as_nhwc(bottleneck_feats.reshape(B, -1, 1, wb)))
        if not hasattr(self, "_memo_t"):
            self._memo_t = [MF._PackMemo() for _ in range(6)]
        for idx in range(5):
            m = self.cnn[idx]
            skip = None if idx == 0 else _as_nhwc(lst_skip_feats[idx - 1])
            z = MF.conv_transpose2d(out, m[0].weight, skip, memo=self._memo_t[idx])
            out = MF.bn_act_train(z, m[1], 0.0)
        head = self.cnn[5][0]
        return MF.conv2d(out, head.weight, head.bias, 1, 0, slope=1.0, deslice=True, memo=self._memo_t[5], name="unet_head.train")

    def forward(self, bottleneck_feats, lst_skip_feats):
        if self.training:
            return self._forward_train(bottleneck_feats, lst_skip_feats)
        _check_inference(self, bottleneck_feats, *lst_skip_feats)
        B = bottleneck_feats.size(0)
        # reference :154 is view(B,-1,1,1) (Tm = 32); in general the bottleneck is [B,512,1,Tm/32]
        wb = lst_skip_feats[0].size(3) // 2
        out = _as_nhwc(bottleneck_feats.reshape(B, -1, 1, wb))
        ups, (hw, hb, hco) = self._packed()
        for idx, (wp, scale, shift, co) in enumerate(ups[:4]):
            skip = None if idx == 0 else _as_nhwc(lst_skip_feats[idx - 1])
            out = ops.unet_up_fwd(out, skip, wp, scale, shift, co)
        wp, scale, shift, co = ups[4]
        # last stage + biased 1x1 conv + de-slice fused in one kernel -> BHWC, contiguous
        return ops.unet_up_head_fwd(out, _as_nhwc(lst_skip_feats[3]), wp, scale, shift, hw, hb, hco)


UNET_KERNEL_NAMES = ("sep_slice_input", "down0", "down1", "down2", "down3", "down4", "up0", "up1", "up2", "up3", "up4+head")


def _split32_of(wp):
    """split32 copy of a packed weight, cached on the tensor object (a re-pack after a weight update makes new tensors)."""
    sp = getattr(wp, "_m2h_split32", None)
    if sp is None:
        sp = ops.split32(wp)
        wp._m2h_split32 = sp
    return sp


def _strip_conv1_of(enc, wp):
    """The first stage's weights in the strip kernel's register image, cached on the packed-weight tensor (rebuilt with it)."""
    sw = getattr(wp, "_m2h_strip", None)
    if sw is None:
        sw = ops.pack_strip_conv1(enc.cnn[0][0].weight.detach().contiguous())
        wp._m2h_strip = sw
    return sw


def unet_forward(enc, dec, mix, masks=None, target_class=None, events=None):
    """Eval-mode forward of one encoder/decoder pair through the whole-network C runner (m2h_unet_fwd): same kernels and
    values as ``dec(*enc(...))``, one host call instead of ~13.  enc: PassiveSepEncCNN, dec: PassiveSepDecCNN.
    In the bf16x3 math mode the runner gets split32 copies of the packed weights and keeps its intermediates in split32
    (bit-identical results, no operand conversion inside the k-loops).
    events: optional list of 12 recorded-once torch.cuda.Event(enable_timing=True): the runner records them around its 11
    kernels (UNET_KERNEL_NAMES), for per-kernel timing without leaving the one-call path."""
    import ctypes

    from ... import _lib
    _check_inference(enc, mix, masks)
    _check_inference(dec)
    downs = enc._packed()
    ups, (hw, hb, hco) = dec._packed()
    mix = mix.contiguous()
    B, F, T, C = mix.shape
    if C != 2 or F != 512 or T % 32 != 0:
        raise RuntimeError("m2h.unet_forward: expected mix [B,512,T,2] with T %% 32 == 0, got %s" % (tuple(mix.shape),))
    split = ops.math_mode() in (ops.MATH_BF16X3, ops.MATH_BF16)   # (MATH_BF16: the reported hi-halves-only mode, same tensors and engines)
    wsel = _split32_of if split else (lambda t: t)
    w = _lib.UnetWeights()
    for i, (wp, scale, shift, table, _co) in enumerate(downs):
        w.down_w[i], w.down_scale[i], w.down_shift[i] = wsel(wp).data_ptr(), scale.data_ptr(), shift.data_ptr()
    table = downs[0][3]
    w.cls_table = table.data_ptr() if table is not None else None
    for i, (wp, scale, shift, _co) in enumerate(ups):
        w.up_w[i], w.up_scale[i], w.up_shift[i] = wsel(wp).data_ptr(), scale.data_ptr(), shift.data_ptr()
    w.head_w, w.head_b, w.n_out = hw.data_ptr(), hb.data_ptr(), hco
    w.weights_split32 = 1 if split else 0
    w.math_mode = 2 if split else 1   # this call's arithmetic, pinned (include/m2h.h)
    w.down0_strip = _strip_conv1_of(enc, downs[0][0]).data_ptr() if (split and T % 64 == 0) else None
    cls_val = None
    if table is not None:
        # the raw target_class goes to the runner, whose first kernel makes the plane's value ".float() + 1" (reference :96) itself
        cls_val = target_class.reshape(-1).contiguous()
        if cls_val.dtype == torch.float32:
            w.cls_kind = 1
        elif cls_val.dtype == torch.int64:
            w.cls_kind = 2
        else:
            cls_val, w.cls_kind = (cls_val.to(torch.float32) + 1.0).contiguous(), 0
        if cls_val.numel() != B or not cls_val.is_cuda:
            raise RuntimeError("m2h.unet_forward: target_class must hold one value per batch row on the GPU")
    if masks is not None:
        masks = masks.contiguous()
    lib = _lib.load()
    out = torch.empty((B, F, T, hco // 16), device=mix.device, dtype=torch.float32)
    with torch.cuda.device(mix.device):
        nbytes = lib.m2h_unet_fwd_workspace_bytes(B, F, T)
        ws = torch.empty(nbytes // 4, device=mix.device, dtype=torch.float32)
        if events is None:
            _lib.check(lib.m2h_unet_fwd(ctypes.byref(w), ops._ptr(mix), ops._ptr(masks), ops._ptr(cls_val), ops._ptr(out), B, F, T,
                                        ops._ptr(ws), nbytes, ops._stream(mix)), "m2h_unet_fwd")
        else:
            if len(events) != 12:
                raise RuntimeError("m2h.unet_forward: events must be 12 torch.cuda.Event objects")
            arr = (ctypes.c_void_p * 12)(*[ctypes.c_void_p(e.cuda_event) for e in events])
            _lib.check(lib.m2h_unet_fwd_events(ctypes.byref(w), ops._ptr(mix), ops._ptr(masks), ops._ptr(cls_val), ops._ptr(out), B, F, T,
                                               ops._ptr(ws), nbytes, arr, 12, ops._stream(mix)), "m2h_unet_fwd_events")
    return out
