"""GPU audio feeder: the waveform -> spectrogram path of the reference's data loader and simulator (row N1 of SURVEY 8f, K22+K23).

  BinauralFeeder.compute_audiospects(mono, rirs)   ==  PassiveDataset.compute_audiospects            (pretrain/datasets/dataset.py:162-228)
                                                   ==  get_current_mixed_bin_audio_mag_spec (train)  (habitat_audio/simulator_train.py:386-486)

per source: fftconvolve(mono, rir[:, ch], mode="same") for the two ears -> np.round -> int16 -> /32768; the first source gives the
GT binaural magnitude and the RMS-normalised GT mono magnitude; the mean of the sources' binaural waveforms gives
log1p(|STFT|) of the mixture.

Stages: the two length-32768 real FFTs of the linear convolution are rocFFT transforms reached through torch.fft (a plain
library transform, like a library GEMM); the frequency-domain product is a torch complex multiply; the "same" window, integer
rounding, mixing (m2h_feeder_round_mix), the STFTs (DFT-as-GEMM on the igemm engine, m2h.audio.stft) and the RMS
normalisation (m2h_rms_normalize) are HIP kernels of libm2h.
"""
import torch

from .. import _lib, ops
from .stft import STFT


def _next_pow2(n):
    p = 1
    while p < n:
        p *= 2
    return p


class BinauralFeeder:
    def __init__(self, device, gt_mono_mag_norm=0.0):
        self.device = device
        self.stft = STFT(device)
        self.gt_mono_mag_norm = float(gt_mono_mag_norm)   # SIMULATOR.AUDIO.GT_MONO_MAG_NORM (config/default.py:198)

    def convolve_round(self, mono, rirs):
        """mono [B, S, L] (int16-valued fp32), rirs [B, S, Lr, 2] fp32 -> per-source binaural waveforms [S][B, 2, L] after the
        int16 round trip, and their mean [B, 2, L]."""
        if not mono.is_cuda or not rirs.is_cuda or mono.dtype != torch.float32 or rirs.dtype != torch.float32:
            raise RuntimeError("m2h.BinauralFeeder: inputs must be fp32 GPU tensors")
        B, S, L = mono.shape
        Lr = rirs.shape[2]
        nfft = _next_pow2(L + Lr - 1)
        start = (Lr - 1) // 2                                  # scipy.signal.fftconvolve(mode="same"): centred on the first input
        lib = _lib.load()
        mix = torch.empty((B, 2, L), device=mono.device)
        per_source = []
        with torch.cuda.device(mono.device):
            X = torch.fft.rfft(mono, n=nfft, dim=2)                                   # [B, S, nfft/2+1]
            H = torch.fft.rfft(rirs.permute(0, 1, 3, 2).contiguous(), n=nfft, dim=3)  # [B, S, 2, nfft/2+1]
            full = torch.fft.irfft(X.unsqueeze(2) * H, n=nfft, dim=3).contiguous()    # [B, S, 2, nfft]
            for s in range(S):
                fs = full[:, s].contiguous()                                          # [B, 2, nfft]
                out = torch.empty((B, 2, L), device=mono.device)
                _lib.check(lib.m2h_feeder_round_mix(ops._ptr(fs), nfft, start, ops._ptr(out), ops._ptr(mix), B * 2, L, 1 if s == 0 else 0,
                                                    (1.0 / S) if s == S - 1 else 1.0, ops._stream(mono)), "m2h_feeder_round_mix")
                per_source.append(out)
        return per_source, mix

    def compute_audiospects(self, mono, rirs):
        """-> (log1p mixed magnitude [B,512,T,2], gt_bin_mag [B,512,T,2], gt_mono_mag [B,512,T,1]) of dataset.py:228."""
        per_source, mix = self.convolve_round(mono, rirs)
        mixed_mag, _ = self.stft(mix, mode=1)
        gt_bin_mag, _ = self.stft(per_source[0], mode=0)
        gt_mono_mag, _ = self.stft((mono[:, :1] * (1.0 / 32768.0)).contiguous(), mode=0)
        if self.gt_mono_mag_norm != 0.0:
            B, F, T, _c = gt_mono_mag.shape
            with torch.cuda.device(mono.device):
                _lib.check(_lib.load().m2h_rms_normalize(ops._ptr(gt_mono_mag), B, F * T, self.gt_mono_mag_norm, ops._stream(mono)),
                           "m2h_rms_normalize")
        return mixed_mag, gt_bin_mag, gt_mono_mag
