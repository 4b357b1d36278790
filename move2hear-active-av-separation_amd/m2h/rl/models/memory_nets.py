"""Acoustic memory on MI355X: drop-in for audio_separation/rl/models/memory_nets.py (AcousticMem, :5-69).

Same constructor, forward signature and state_dict keys (``cnn.0.weight``, ``cnn.2.weight`` for the DD-PPO variant; ``cnn.0.weight``,
``cnn.1.*`` (BatchNorm2d), ``cnn.3.weight`` for the single-process variant, :17-23).
forward = slice both inputs 16-way + concat (one HBM-bound kernel, concat never materialised as NCHW) -> conv3x3+ReLU ->
conv3x3 with the de-slice fused into its store; both convs run on the MFMA implicit-GEMM engine.
``forward_masked`` additionally fuses the not-done masking of the previous memory (ppo_trainer.py:310-314, ppo.py:206-209).
"""
import torch
import torch.nn as nn

from ... import functional as MF
from ... import ops


class AcousticMem(nn.Module):
    def __init__(self, use_ddppo=False):
        super().__init__()
        self._slice_factor = 16
        _n_out_audio = self._slice_factor
        if use_ddppo:
            self.cnn = nn.Sequential(
                nn.Conv2d(_n_out_audio * 2, 32, kernel_size=3, padding=1, bias=False),
                nn.ReLU(inplace=True),
                nn.Conv2d(32, _n_out_audio, kernel_size=3, padding=1, bias=False),
            )
        else:
            self.cnn = nn.Sequential(
                nn.Conv2d(_n_out_audio * 2, 32, kernel_size=3, padding=1, bias=False),
                nn.BatchNorm2d(32),
                nn.ReLU(inplace=True),
                nn.Conv2d(32, _n_out_audio, kernel_size=3, padding=1, bias=False),
            )
        self._use_ddppo = use_ddppo
        self.layer_init()
        self._memo = [MF._PackMemo(), MF._PackMemo()]

    def layer_init(self):
        # reference :26-38
        for layer in self.cnn:
            if isinstance(layer, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
                nn.init.kaiming_normal_(layer.weight, nn.init.calculate_gain("relu"))
                if layer.bias is not None:
                    nn.init.constant_(layer.bias, val=0)
            elif isinstance(layer, (nn.BatchNorm1d, nn.BatchNorm2d)):
                if layer.affine:
                    layer.weight.data.fill_(1)
                    layer.bias.data.zero_()

    SMALL_BATCH = 64   # up to here the no-grad forward is the one-launch kernel (above it the image-row kernels of the update batch)

    def slice_inputs(self, pred_mono, prev_pred_monoFromMem, masks=None, out=None):
        """The convs' input: both tensors sliced 16-way and concatenated (memory_nets.py:40-61), the previous memory scaled by the
        not-done masks; NHWC [B, 32, T, 32].  A function of the inputs alone: update_sep builds it once for its four epochs."""
        bscale = masks.reshape(-1).contiguous() if masks is not None else None
        return ops.slice_concat_input(pred_mono.contiguous(), prev_pred_monoFromMem.contiguous(), bscale=bscale, op=0, out=out)

    def forward_masked(self, pred_mono, prev_pred_monoFromMem, masks=None, sliced=None):
        """sliced: ``slice_inputs`` of the same three arguments, when the caller already holds it."""
        if torch.is_grad_enabled() and (pred_mono.requires_grad or prev_pred_monoFromMem.requires_grad):
            raise NotImplementedError("m2h AcousticMem: gradients w.r.t. the inputs are not built (the separators are frozen in RL, "
                                      "ppo.py:184-195); detach the inputs")
        if (self._use_ddppo and sliced is None and not torch.is_grad_enabled() and pred_mono.shape[0] <= self.SMALL_BATCH
                and tuple(pred_mono.shape[1:]) == (512, 32, 1) and ops.math_mode() == ops.MATH_FP32 and not ops.timing_enabled()):
            # the rollout step's call (14 envs, no autograd): slice, both convs and the de-slice in ONE launch (csrc/acoustic_mem.hip)
            c0, c1 = self.cnn[0], self.cnn[-1]
            return ops.acoustic_mem_small(pred_mono.contiguous(), prev_pred_monoFromMem.contiguous(),
                                          masks.reshape(-1).contiguous() if masks is not None else None,
                                          self._memo[0].get(c0.weight, 32), self._memo[1].get(c1.weight, 32))
        x = sliced if sliced is not None else self.slice_inputs(pred_mono, prev_pred_monoFromMem, masks)
        c0, c1 = self.cnn[0], self.cnn[-1]
        if self._use_ddppo:
            x = MF.conv2d(x, c0.weight, None, 1, 1, slope=0.0, memo=self._memo[0], name="acoustic_mem.conv0")
        elif self.training:
            # single-process PPO variant (memory_nets.py:17-23): conv -> BatchNorm2d -> ReLU; train mode = batch statistics, running
            # statistics updated, full autograd (the train-mode BatchNorm kernels of the passive pre-training path)
            z = MF.conv2d(x, c0.weight, None, 1, 1, slope=1.0, memo=self._memo[0], name="acoustic_mem.conv0")
            x = MF.bn_act_train(z, self.cnn[1], 0.0)
        else:
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                raise NotImplementedError("m2h AcousticMem: the eval-mode (folded BatchNorm) forward has no autograd; call .train() or use no_grad")
            bn = self.cnn[1]
            scale, shift = ops.fold_bn(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
            wp = self._memo[0].get(c0.weight, x.shape[3])
            x = ops.conv2d_nhwc(x, wp, c0.weight.shape[0], 3, 3, stride=1, pad=1, bias=shift, scale=scale, slope=0.0, name="acoustic_mem.conv0")
        return MF.conv2d(x, c1.weight, None, 1, 1, slope=1.0, deslice=True, memo=self._memo[1], name="acoustic_mem.conv1")

    def l1_loss_masked(self, pred_mono, prev_pred_monoFromMem, masks, gt_comps, off=0, sliced=None):
        """F.l1_loss(forward_masked(...), gt_comps[..., off:off+1]) (ppo.py:206-216) without materialising the memory's output: the last
        conv stays in NHWC and the loss kernel reads it there (functional.ConvL1NHWC16).  DD-PPO variant, [B, 512, T, 1] inputs."""
        if not self._use_ddppo or pred_mono.shape[1] != 512:
            return MF.l1_loss(self.forward_masked(pred_mono, prev_pred_monoFromMem, masks, sliced=sliced), gt_comps, off)
        if torch.is_grad_enabled() and (pred_mono.requires_grad or prev_pred_monoFromMem.requires_grad):
            raise NotImplementedError("m2h AcousticMem: gradients w.r.t. the inputs are not built; detach the inputs")
        x = sliced if sliced is not None else self.slice_inputs(pred_mono, prev_pred_monoFromMem, masks)
        c0, c1 = self.cnn[0], self.cnn[-1]
        if (gt_comps.shape[-1] == 1 and off == 0 and torch.is_grad_enabled() and c0.weight.requires_grad and c1.weight.requires_grad
                and tuple(c0.weight.shape) == (32, 32, 3, 3) and MF.acoustic_mem_l1_supported(x)):
            # update_sep at the update batch in bf16x3 arithmetic: the whole differentiable path as one Function (two convs' outputs / gradients never stored)
            return MF.acoustic_mem_l1(x, c0.weight, c1.weight, gt_comps, self._memo[0], self._memo[1])
        x = MF.conv2d(x, c0.weight, None, 1, 1, slope=0.0, memo=self._memo[0], name="acoustic_mem.conv0")
        return MF.conv_l1_nhwc16(x, c1.weight, gt_comps, off, memo=self._memo[1])

    def forward(self, pred_mono, prev_pred_monoFromMem_masked):
        return self.forward_masked(pred_mono, prev_pred_monoFromMem_masked, None)
