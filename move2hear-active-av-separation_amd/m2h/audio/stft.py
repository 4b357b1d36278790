"""GPU STFT feeder / iSTFT: the librosa calls of the reference's data path on the MI355X (rows N1/N2 of SURVEY 8f).

  stft_features(wave)  == log1p(abs(librosa.stft(y, n_fft=1023, hop_length=512)))  (+ np.angle)   per channel, stacked BHWC
                          (audio_separation/pretrain/datasets/dataset.py:190-228; habitat_audio/simulator_train.py:425-486)
  istft(mag, phase)    == librosa.istft(mag * exp(1j*phase), hop_length=512, length=16000)          (common/eval_metrics.py:232-251)

The transform lengths (1023, 1022) are not powers of two; a frame's DFT is one row of a dense fp32 GEMM against a cos/-sin
matrix built here once in float64 (host setup, like weight packing); framing, magnitude/phase/log1p and overlap-add are HIP
kernels (csrc/stft.hip).
"""
import numpy as np
import torch

from .. import _lib, ops

N_FFT = 1023      # config/default.py: n_fft used by dataset.py:190 / simulator_train.py:425
HOP = 512
N_BINS = 512


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True)"""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


class STFT:
    def __init__(self, device, n_fft=N_FFT, hop=HOP):
        self.device, self.n_fft, self.hop = device, n_fft, hop
        self.nb = n_fft // 2 + 1
        self.ld = (max(n_fft, 2 * self.nb) + 31) // 32 * 32   # GEMM K and N, padded (1024 for n_fft 1023)
        n = np.arange(n_fft)[None, :]
        k = np.arange(self.nb)[:, None]
        ang = 2.0 * np.pi * k * n / n_fft
        W = np.zeros((self.ld, self.ld), np.float64)           # Linear weight [N_out][K]: rows [0,nb) cos, [nb,2nb) -sin
        W[:self.nb, :n_fft] = np.cos(ang)
        W[self.nb:2 * self.nb, :n_fft] = -np.sin(ang)
        self.W = torch.from_numpy(W.astype(np.float32)).to(device)
        self.window = torch.from_numpy(hann_periodic(n_fft).astype(np.float32)).to(device)

    def __call__(self, wave, mode=1, want_phase=False):
        """wave [B, C, L] fp32 (device) -> (mag [B, nb, T, C], phase or None); T = 1 + (L + 2*(n_fft//2) - n_fft) // hop frames
        (librosa's frame count on the centre-padded signal: 32 for 16 000 samples)."""
        if not wave.is_cuda or wave.dtype != torch.float32:
            raise RuntimeError("m2h.STFT: wave must be an fp32 GPU tensor")
        wave = wave.contiguous()
        B, C, L = wave.shape
        T = 1 + (L + 2 * (self.n_fft // 2) - self.n_fft) // self.hop
        S = B * C
        frames = torch.empty((S * T, self.ld), device=wave.device)
        lib = _lib.load()
        with torch.cuda.device(wave.device):
            _lib.check(lib.m2h_stft_frames(ops._ptr(wave), ops._ptr(self.window), ops._ptr(frames), S, L, T, self.n_fft, self.hop, self.ld,
                                           ops._stream(wave)), "m2h_stft_frames")
            spec = ops.linear(frames, self.W, None, name="stft.dft")
            mag = torch.empty((B, self.nb, T, C), device=wave.device)
            phase = torch.empty((B, self.nb, T, C), device=wave.device) if want_phase else None
            _lib.check(lib.m2h_stft_post(ops._ptr(spec), ops._ptr(mag), ops._ptr(phase), B, C, T, self.nb, self.ld, mode, ops._stream(wave)),
                       "m2h_stft_post")
        return mag, phase


class ISTFT:
    def __init__(self, device, nb=N_BINS, hop=HOP):
        self.device, self.nb, self.hop = device, nb, hop
        self.n_fft = 2 * (nb - 1)                               # librosa infers n_fft from the number of bins: 1022
        N = self.n_fft
        self.ld = (max(N, 2 * nb) + 31) // 32 * 32
        n = np.arange(N)[:, None]
        k = np.arange(nb)[None, :]
        ang = 2.0 * np.pi * k * n / N
        ck = np.full(nb, 2.0)
        ck[0] = ck[nb - 1] = 1.0                                # DC and Nyquist appear once; their imaginary parts are ignored
        W = np.zeros((self.ld, self.ld), np.float64)            # Linear weight [N_out = sample n][K = (re | im)]
        W[:N, :nb] = ck[None, :] * np.cos(ang) / N
        si = -ck[None, :] * np.sin(ang) / N
        si[:, 0] = 0.0
        si[:, nb - 1] = 0.0
        W[:N, nb:2 * nb] = si
        self.W = torch.from_numpy(W.astype(np.float32)).to(device)
        self.window = torch.from_numpy(hann_periodic(N).astype(np.float32)).to(device)

    def __call__(self, mag, phase, length=16000, channel=0):
        """mag, phase BHWC [B, nb, T, C] -> waveform [B, length] of channel `channel`."""
        for t in (mag, phase):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("m2h.ISTFT: mag/phase must be contiguous fp32 GPU tensors")
        B, nb, T, C = mag.shape
        rows = torch.zeros((B * T, self.ld), device=mag.device)
        y = torch.empty((B, length), device=mag.device)
        lib = _lib.load()
        with torch.cuda.device(mag.device):
            _lib.check(lib.m2h_istft_pre(ops._ptr(mag), ops._ptr(phase), ops._ptr(rows), B, C, channel, T, nb, self.ld, ops._stream(mag)),
                       "m2h_istft_pre")
            frames = ops.linear(rows, self.W, None, name="istft.idft")
            _lib.check(lib.m2h_istft_ola(ops._ptr(frames), ops._ptr(self.window), ops._ptr(y), B, T, self.n_fft, self.hop, self.ld, length,
                                         ops._stream(mag)), "m2h_istft_ola")
        return y
