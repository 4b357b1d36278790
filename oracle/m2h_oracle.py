"""CPU oracle for the Move2Hear hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-CPU fp32 restatement of the reference's algorithm, function by function, each
citing the reference file:line it follows (paths relative to the reference root).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this module, and
only as the checker / the reported CPU baseline; the product path (``m2h``) never routes through it.

Pinning: the reference holds no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, produced in the build container by
``oracle/gen_golden.py`` (which imports the reference through ``oracle/_ref_import.py``) and
committed under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here against
them, and ``tests/test_oracle_vs_reference.py`` re-checks live when ``/root/reference`` is present.

All functions take a flat ``state_dict`` (name -> torch.Tensor, reference key names without the
``actor_critic.`` root) and plain tensors; no nn.Module, so nothing here can be mistaken for the
product's modules.
"""
import math

import torch
import torch.nn.functional as F

SLICE = 16  # separator_cnn.py:43, memory_nets.py:8, audio_cnn.py:24
BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, used by separator_cnn.py:9,21


# ----------------------------------------------------------------------------------------------
# layout helpers
# ----------------------------------------------------------------------------------------------
def slice_freq(x_bhwc):
    """[B,F,T,C] -> [B,C*16,F/16,T]   (separator_cnn.py:85-90; memory_nets.py:41-59; audio_cnn.py:129-133)"""
    x = x_bhwc.permute(0, 3, 1, 2)
    b, c, f, t = x.shape
    x = x.reshape(b, c, SLICE, f // SLICE, t)
    return x.reshape(b, c * SLICE, f // SLICE, t)


def deslice_freq(x_bchw):
    """[B,c*16,H,T] -> [B,16*H,T,c]   (separator_cnn.py:163-168; memory_nets.py:62-67)"""
    b, cs, h, t = x_bchw.shape
    x = x_bchw.reshape(b, cs // SLICE, SLICE, h, t)
    x = x.reshape(b, cs // SLICE, SLICE * h, t)
    return x.permute(0, 2, 3, 1)


def _bn_eval(x, sd, pre):
    """BatchNorm2d in eval mode (running statistics)."""
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"],
                        sd[pre + "bias"], training=False, momentum=0.1, eps=BN_EPS)


def _bn_train(x, sd, pre, stats_out=None):
    """BatchNorm2d in train mode: batch statistics (biased var) for normalisation; the running-stat
    update (momentum 0.1, unbiased var) is returned through ``stats_out`` instead of mutating sd."""
    mean = x.mean(dim=(0, 2, 3))
    var_b = x.var(dim=(0, 2, 3), unbiased=False)
    y = (x - mean[None, :, None, None]) / torch.sqrt(var_b[None, :, None, None] + BN_EPS)
    y = y * sd[pre + "weight"][None, :, None, None] + sd[pre + "bias"][None, :, None, None]
    if stats_out is not None:
        n = x.numel() / x.size(1)
        var_u = var_b * (n / max(n - 1.0, 1.0))
        stats_out[pre + "running_mean"] = 0.9 * sd[pre + "running_mean"] + 0.1 * mean
        stats_out[pre + "running_var"] = 0.9 * sd[pre + "running_var"] + 0.1 * var_u
    return y


# ----------------------------------------------------------------------------------------------
# A1: PassiveSepEncCNN.forward  (separator_cnn.py:70-108)
# ----------------------------------------------------------------------------------------------
def sep_enc_input(mixed_bin_audio_mag, target_class=None, pred_masks=None):
    """Builds the conv-stack input.  binSep: slice(mix) ++ (target_class+1) plane (:82-99);
    bin2mono: slice(log1p(clamp0(mask * (exp(mix)-1)))) (:73-79)."""
    if pred_masks is not None:
        x = pred_masks * (torch.exp(mixed_bin_audio_mag) - 1)
        x = torch.log1p(torch.clamp(x, min=0))
        return slice_freq(x)
    x = slice_freq(mixed_bin_audio_mag)
    tc = target_class.reshape(-1, 1, 1, 1).float() + 1  # :96
    plane = tc.expand(x.size(0), 1, x.size(2), x.size(3))
    return torch.cat((x, plane), dim=1)


def sep_enc_stack(x, sd, pre, train_bn=False, stats_out=None):
    """5x {conv4x4 s2 p1 no-bias -> BN -> LeakyReLU(0.2)}  (separator_cnn.py:5-12,46-52,101-105).
    Returns the list of the 5 stage outputs (NCHW)."""
    feats = []
    out = x
    for i in range(5):
        out = F.conv2d(out, sd[pre + "%d.0.weight" % i], None, stride=2, padding=1)
        out = _bn_train(out, sd, pre + "%d.1." % i, stats_out) if train_bn else _bn_eval(out, sd, pre + "%d.1." % i)
        out = F.leaky_relu(out, 0.2)
        feats.append(out)
    return feats


# ----------------------------------------------------------------------------------------------
# A2: PassiveSepDecCNN.forward  (separator_cnn.py:153-170), fully-convolutional generalisation
# ----------------------------------------------------------------------------------------------
def sep_dec_stack(feats, sd, pre, train_bn=False, stats_out=None):
    """5x {convT4x4 s2 p1 no-bias -> BN -> ReLU} with skip concat on stages 1..4 (:156-161) and the
    biased 1x1 conv (:134).  ``feats`` = encoder stage outputs [e1..e5]; the bottleneck is e5 kept as
    [B,512,1,Tm/32] (the reference's ``view(B,-1,1,1)`` at :154 is the Tm=32 special case).
    Returns BHWC [B,512,Tm,c]."""
    out = feats[4]
    skips = feats[:4][::-1]  # separator_cnn.py:108
    for i in range(5):
        if i > 0:
            out = torch.cat((out, skips[i - 1]), dim=1)
        out = F.conv_transpose2d(out, sd[pre + "%d.0.weight" % i], None, stride=2, padding=1)
        out = _bn_train(out, sd, pre + "%d.1." % i, stats_out) if train_bn else _bn_eval(out, sd, pre + "%d.1." % i)
        out = F.relu(out)
    out = F.conv2d(out, sd[pre + "5.0.weight"], sd[pre + "5.0.bias"])
    return deslice_freq(out)


# ----------------------------------------------------------------------------------------------
# A3: Policy.get_binSepMasks / convert_bin2mono  (pretrain/passive/policy.py:61-71; rl/ppo/policy.py:183-193)
# ----------------------------------------------------------------------------------------------
ENC_B = "binSep_enc.passive_sep_encoder.cnn."
DEC_B = "binSep_dec.passive_sep_decoder.cnn."
ENC_M = "bin2mono_enc.passive_sep_encoder.cnn."
DEC_M = "bin2mono_dec.passive_sep_decoder.cnn."


def get_binSepMasks(sd, mixed_bin_audio_mag, target_class, train_bn=False, stats_out=None, return_feats=False):
    x = sep_enc_input(mixed_bin_audio_mag, target_class=target_class)
    feats = sep_enc_stack(x, sd, ENC_B, train_bn, stats_out)
    out = sep_dec_stack(feats, sd, DEC_B, train_bn, stats_out)
    return (out, feats) if return_feats else out


def convert_bin2mono(sd, pred_masks, mixed_bin_audio_mag, train_bn=False, stats_out=None, return_feats=False):
    x = sep_enc_input(mixed_bin_audio_mag, pred_masks=pred_masks)
    feats = sep_enc_stack(x, sd, ENC_M, train_bn, stats_out)
    out = sep_dec_stack(feats, sd, DEC_M, train_bn, stats_out)
    return (out, feats) if return_feats else out


def passive_pair(sd, mixed_bin_audio_mag, target_class):
    """The headline unit of work: one spectrogram through both U-Nets (eval-mode BN)."""
    masks = get_binSepMasks(sd, mixed_bin_audio_mag, target_class)
    mono = convert_bin2mono(sd, masks, mixed_bin_audio_mag)
    return masks, mono


def rel_l1(a, b):
    """Parity metric of SURVEY.md section 8d: sum|a-b| / sum|b|."""
    return float((a.double() - b.double()).abs().sum() / b.double().abs().sum().clamp_min(1e-30))


def pred_bin(masks, mixed_bin_audio_mag):
    """mask * (exp(mix) - 1): the separated binaural magnitude (passive_trainer.py:271-272)."""
    return masks * (torch.exp(mixed_bin_audio_mag) - 1)
