"""GPU audio feeder: the waveform -> spectrogram path of the reference's data loader and simulator (row N1 of SURVEY 8f, K22+K23).

  BinauralFeeder.compute_audiospects(mono, rirs)   ==  PassiveDataset.compute_audiospects            (pretrain/datasets/dataset.py:162-228)
                                                   ==  get_current_mixed_bin_audio_mag_spec (train)  (habitat_audio/simulator_train.py:386-486)

per source: fftconvolve(mono, rir[:, ch], mode="same") for the two ears -> np.round -> int16 -> /32768; the first source gives the
GT binaural magnitude and the RMS-normalised GT mono magnitude; the mean of the sources' binaural waveforms gives
log1p(|STFT|) of the mixture.

Stages, all HIP kernels of libm2h (no library math): the linear convolution is m2h_fftconv_full -- real FFTs of 32 768 points as
packed 16 384-point complex radix-2 transforms held in LDS, one workgroup per (clip, source), spectrum product and inverse in the
same launch (csrc/fftconv.hip; round 1 went through rocFFT via torch.fft); the "same" window, integer rounding and mixing
(m2h_feeder_round_mix), the STFTs (DFT-as-GEMM on the igemm engine, m2h.audio.stft) and the RMS normalisation
(m2h_rms_normalize).
"""
import numpy as np
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
        self._twiddles = {}

    def _twiddle_table(self, nfft):
        """exp(-2 pi i k / nfft), k < nfft/2, as interleaved fp32 (built in float64 on the host once per length: setup data)."""
        if nfft not in self._twiddles:
            k = np.arange(nfft // 2, dtype=np.float64)
            t = np.stack([np.cos(2.0 * np.pi * k / nfft), -np.sin(2.0 * np.pi * k / nfft)], axis=1).astype(np.float32)
            self._twiddles[nfft] = torch.from_numpy(t).to(self.device).contiguous()
        return self._twiddles[nfft]

    def convolve_round(self, mono, rirs):
        """mono [B, S, L] (int16-valued fp32), rirs [B, S, Lr, 2] fp32 -> per-source binaural waveforms [S][B, 2, L] after the
        int16 round trip, and their mean [B, 2, L]."""
        if not mono.is_cuda or not rirs.is_cuda or mono.dtype != torch.float32 or rirs.dtype != torch.float32:
            raise RuntimeError("m2h.BinauralFeeder: inputs must be fp32 GPU tensors")
        B, S, L = mono.shape
        Lr = rirs.shape[2]
        nfft = max(_next_pow2(L + Lr - 1), 2048)
        if nfft > 32768:
            raise NotImplementedError("m2h.BinauralFeeder: a %d + %d - 1 point convolution needs a %d-point transform; the in-LDS FFT "
                                      "of m2h_fftconv_full holds at most 32768 points" % (L, Lr, nfft))
        start = (Lr - 1) // 2                                  # scipy.signal.fftconvolve(mode="same"): centred on the first input
        lib = _lib.load()
        mix = torch.empty((B, 2, L), device=mono.device)
        per_source = []
        mono, rirs = mono.contiguous(), rirs.contiguous()
        with torch.cuda.device(mono.device):
            full = torch.empty((B, S, 2, nfft), device=mono.device)
            xspec = torch.empty((B * S, 2 * nfft), device=mono.device)                # [B*S][2 ears][nfft/2] complex scratch
            _lib.check(lib.m2h_fftconv_full(ops._ptr(mono), ops._ptr(rirs), ops._ptr(self._twiddle_table(nfft)), ops._ptr(xspec), ops._ptr(full),
                                            B * S, L, Lr, nfft.bit_length() - 1, ops._stream(mono)), "m2h_fftconv_full")
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
