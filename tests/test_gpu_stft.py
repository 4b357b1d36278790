"""GPU: STFT feeder / iSTFT kernels against the numpy restatement of librosa's semantics (oracle)."""
import numpy as np
import pytest
import torch

import m2h_oracle as O

pytestmark = pytest.mark.gpu


def _wave(B, C, L, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(L) / 16000.0
    w = rng.standard_normal((B, C, L)) * 0.05
    for b in range(B):
        for c in range(C):
            w[b, c] += 0.3 * np.sin(2 * np.pi * rng.uniform(100, 4000) * t + rng.uniform(0, 6))
    return w.astype(np.float32)


@pytest.mark.parametrize("mode", [1, 2])
def test_stft_features_match_oracle(mode):
    from m2h.audio.stft import STFT
    dev = torch.device("cuda", 0)
    wave = _wave(3, 2, 16000, 5)
    mag, phase = STFT(dev)(torch.from_numpy(wave).to(dev), mode=mode, want_phase=True)
    rm, rp = O.np_stft_features(wave, fp16_round=(mode == 2))
    assert mag.shape == (3, 512, 32, 2) and phase.shape == (3, 512, 32, 2)
    tol = 1e-3 if mode == 2 else 2e-5   # fp16 rounding of the magnitude can flip a half-ulp at exact ties
    assert O.rel_l1(mag.cpu(), torch.from_numpy(rm)) < tol
    # phase is ill-conditioned where |X| ~ 0: compare the reconstructed complex values instead
    m = np.expm1(mag.cpu().numpy()) if mode == 1 else None
    if m is not None:
        z = m * np.exp(1j * phase.cpu().numpy())
        zr = np.expm1(rm) * np.exp(1j * rp)
        assert np.abs(z - zr).sum() / np.abs(zr).sum() < 5e-5


def test_long_clip_256_frames():
    from m2h.audio.stft import STFT
    dev = torch.device("cuda", 0)
    wave = _wave(1, 2, 131072, 6)  # 256 frames (the 512x256 throughput shape)
    mag, _ = STFT(dev)(torch.from_numpy(wave).to(dev))
    rm, _ = O.np_stft_features(wave)
    assert mag.shape == (1, 512, 256, 2) and O.rel_l1(mag.cpu(), torch.from_numpy(rm)) < 2e-5


def test_istft_matches_oracle():
    from m2h.audio.stft import ISTFT
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    B = 3
    mag = np.abs(rng.standard_normal((B, 512, 32, 1))).astype(np.float32)
    phase = rng.uniform(-np.pi, np.pi, (B, 512, 32, 1)).astype(np.float32)
    y = ISTFT(dev)(torch.from_numpy(mag).to(dev), torch.from_numpy(phase).to(dev), length=16000, channel=0).cpu().numpy()
    for b in range(B):
        ref = O.np_istft((mag[b, :, :, 0] * np.exp(1j * phase[b, :, :, 0])).astype(np.complex64), 512, 16000)
        assert np.abs(y[b] - ref).sum() / np.abs(ref).sum() < 5e-5
