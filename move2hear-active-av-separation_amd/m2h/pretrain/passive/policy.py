"""Passive-separation policy wrappers: drop-in for audio_separation/pretrain/passive/policy.py.

Same class names, constructor arguments, method names and state_dict keys
(``binSep_enc.passive_sep_encoder.cnn...``; reference :7-44,47-97).  The torchsummary print-out of the
reference constructor (:20-30) is dropped: it runs the CPU modules once for logging only.
"""
import torch.nn as nn

from ... import ops
from ...rl.models.separator_cnn import PassiveSepEncCNN, PassiveSepDecCNN, unet_forward


class PassiveSepEnc(nn.Module):
    r"""Network which encodes separated bin or mono outputs (reference :7-33)."""

    def __init__(self, observation_space, world_rank=0, convert_bin2mono=False):
        super().__init__()
        assert 'mixed_bin_audio_mag' in observation_space.spaces
        self.passive_sep_encoder = PassiveSepEncCNN(convert_bin2mono=convert_bin2mono)

    def forward(self, observations, mixed_audio=None):
        return self.passive_sep_encoder(observations, mixed_audio=mixed_audio)


class PassiveSepDec(nn.Module):
    r"""Network which decodes separated bin or mono outputs feature embeddings (reference :36-44)."""

    def __init__(self, convert_bin2mono=False):
        super().__init__()
        self.passive_sep_decoder = PassiveSepDecCNN(convert_bin2mono=convert_bin2mono)

    def forward(self, bottleneck_feats, lst_skip_feats):
        return self.passive_sep_decoder(bottleneck_feats, lst_skip_feats)


class Policy(nn.Module):
    r"""Network for the passive separation in Move2Hear pretraining (reference :47-71)."""

    def __init__(self, binSep_enc, binSep_dec, bin2mono_enc, bin2mono_dec):
        super().__init__()
        self.binSep_enc = binSep_enc
        self.binSep_dec = binSep_dec
        self.bin2mono_enc = bin2mono_enc
        self.bin2mono_dec = bin2mono_dec

    def forward(self):
        raise NotImplementedError

    def get_binSepMasks(self, observations):
        enc, dec = self.binSep_enc.passive_sep_encoder, self.binSep_dec.passive_sep_decoder
        if not enc.training and not dec.training and not ops.timing_enabled() and observations["mixed_bin_audio_mag"].shape[1] == 512:
            # eval mode (every RL call site): the whole U-Net is enqueued by one C call (m2h_unet_fwd)
            return unet_forward(enc, dec, observations["mixed_bin_audio_mag"], None, observations["target_class"])
        bottleneck_feats, lst_skip_feats = self.binSep_enc(observations)
        return self.binSep_dec(bottleneck_feats, lst_skip_feats)

    def convert_bin2mono(self, pred_binSepMasks, mixed_audio=None):
        enc, dec = self.bin2mono_enc.passive_sep_encoder, self.bin2mono_dec.passive_sep_decoder
        if not enc.training and not dec.training and not ops.timing_enabled() and mixed_audio.shape[1] == 512:
            return unet_forward(enc, dec, mixed_audio, pred_binSepMasks)
        bottleneck_feats, lst_skip_feats = self.bin2mono_enc(pred_binSepMasks, mixed_audio=mixed_audio)
        return self.bin2mono_dec(bottleneck_feats, lst_skip_feats)


class Move2HearPassiveWoMemoryPolicy(Policy):
    def __init__(self, observation_space):
        binSep_enc = PassiveSepEnc(observation_space=observation_space)
        binSep_dec = PassiveSepDec()
        bin2mono_enc = PassiveSepEnc(observation_space=observation_space, convert_bin2mono=True)
        bin2mono_dec = PassiveSepDec(convert_bin2mono=True)
        super().__init__(binSep_enc, binSep_dec, bin2mono_enc, bin2mono_dec)
