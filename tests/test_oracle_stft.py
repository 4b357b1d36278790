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


def test_feeder_restatement_identity_rir_and_mixing():
    """np_compute_audiospects (dataset.py:162-228): with a unit-impulse RIR centred for mode="same" the convolved ears equal the
    clip, so the GT binaural magnitude equals the mono magnitude of the int16 round trip and the mixture is the sources' mean."""
    import numpy as np
    r = np.random.default_rng(0)
    L = 16000
    mono = [np.round(2000 * r.standard_normal(L)).astype(np.int16) for _ in range(2)]
    rir = np.zeros((L, 2), np.float32)
    rir[(L - 1) // 2] = 1.0          # fftconvolve(x, delta_c, "same") == x
    mixed_mag, gt_bin, gt_mono, mixed, per_source = O.np_compute_audiospects(mono, [rir, rir], 0.0)
    assert mixed_mag.shape == (512, 32, 2) and gt_bin.shape == (512, 32, 2) and gt_mono.shape == (512, 32, 1)
    for s in range(2):
        assert np.array_equal(per_source[s][0] * 32768, mono[s].astype(np.float32))
        assert np.array_equal(per_source[s][1], per_source[s][0])
    assert np.allclose(mixed[0], (mono[0].astype(np.float32) + mono[1]) / 2 / 32768, atol=1e-7)
    assert np.allclose(gt_bin[..., 0], gt_mono[..., 0], rtol=1e-5, atol=1e-6)
    _, _, gt_norm, _, _ = O.np_compute_audiospects(mono, [rir, rir], 2.0)
    assert abs(np.sqrt(np.mean(gt_norm ** 2)) - 2.0) < 1e-4    # GT_MONO_MAG_NORM sets the RMS of the GT mono magnitude
