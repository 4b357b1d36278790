"""CPU: the numpy restatement of librosa's stft/istft (oracle) against torch.stft / torch.istft, an independent
implementation of the same transform (librosa itself is not installed)."""
import numpy as np
import torch

import m2h_oracle as O


def test_np_stft_matches_torch_stft():
    rng = np.random.default_rng(0)
    y = (rng.standard_normal(16000) * 0.1).astype(np.float32)
    X = O.np_stft(y)
    assert X.shape == (512, 32) and X.dtype == np.complex64
    Xt = torch.stft(torch.from_numpy(y), n_fft=1023, hop_length=512, window=torch.hann_window(1023, periodic=True), center=True,
                    pad_mode="reflect", return_complex=True)
    assert Xt.shape == (512, 32)
    err = np.abs(X - Xt.numpy()).sum() / np.abs(Xt.numpy()).sum()
    assert err < 1e-5


def test_np_istft_matches_torch_istft_and_round_trips():
    rng = np.random.default_rng(1)
    y = (rng.standard_normal(16000) * 0.1).astype(np.float32)
    # build a 1022-point STFT (512 bins) so that istft's inferred n_fft is consistent
    Xt = torch.stft(torch.from_numpy(y), n_fft=1022, hop_length=512, window=torch.hann_window(1022, periodic=True), center=True,
                    pad_mode="reflect", return_complex=True)
    yi = O.np_istft(Xt.numpy().astype(np.complex64), 512, 16000)
    yt = torch.istft(Xt, n_fft=1022, hop_length=512, window=torch.hann_window(1022, periodic=True), center=True, length=16000).numpy()
    assert np.abs(yi - yt).max() < 1e-5
    assert np.abs(yi[1000:15000] - y[1000:15000]).max() < 1e-4   # perfect reconstruction away from the edges
