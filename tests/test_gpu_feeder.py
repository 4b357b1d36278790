"""GPU: RIR-convolution feeder (m2h.audio.feeder) against the scipy/numpy restatement of dataset.py:162-228."""
import numpy as np
import pytest
import torch

import m2h_oracle as O

pytestmark = pytest.mark.gpu


def _inputs(B, S, L, Lr, seed):
    """SURVEY 8d synthetic audio: mono clip = round(3000*N(0,1)) as int16, binaural RIR = N(0,1)*exp(-t/0.05 s), peak-normalised."""
    r = np.random.default_rng(seed)
    mono = np.clip(np.round(3000 * r.standard_normal((B, S, L))), -32768, 32767).astype(np.float32)
    t = np.arange(Lr) / 16000.0
    rir = (r.standard_normal((B, S, Lr, 2)) * np.exp(-t / 0.05)[None, None, :, None]).astype(np.float32)
    rir /= np.abs(rir).max(axis=(2, 3), keepdims=True)
    rir *= 0.05   # keeps the convolved signal inside the int16 range, as real RIR/clip pairs are
    return mono, rir


@pytest.mark.parametrize("norm", [0.0, 1.3])
def test_feeder_matches_scipy_restatement(norm):
    from m2h.audio.feeder import BinauralFeeder
    dev = torch.device("cuda", 0)
    B, S, L, Lr = 2, 2, 16000, 16000
    mono, rir = _inputs(B, S, L, Lr, 4)
    fd = BinauralFeeder(dev, gt_mono_mag_norm=norm)
    per_source, mix = fd.convolve_round(torch.from_numpy(mono).to(dev), torch.from_numpy(rir).to(dev))
    mixed_mag, gt_bin, gt_mono = fd.compute_audiospects(torch.from_numpy(mono).to(dev), torch.from_numpy(rir).to(dev))
    assert mixed_mag.shape == (B, 512, 32, 2) and gt_bin.shape == (B, 512, 32, 2) and gt_mono.shape == (B, 512, 32, 1)
    for b in range(B):
        w_mixed_mag, w_gt_bin, w_gt_mono, w_mixed, w_src = O.np_compute_audiospects(mono[b], rir[b], norm)
        # integer waveforms: the fp32 FFT convolution may land on the other side of a rounding boundary for a few samples
        for s in range(S):
            got = per_source[s][b].cpu().numpy() * 32768
            want = w_src[s] * 32768
            assert np.abs(got - want).max() <= 1.0
            assert (got != want).mean() < 0.02
        assert np.abs(mix[b].cpu().numpy() - w_mixed).max() <= 1.01 / 32768
        assert O.rel_l1(mixed_mag[b].cpu(), torch.from_numpy(w_mixed_mag)) < 1e-4
        assert O.rel_l1(gt_bin[b].cpu(), torch.from_numpy(w_gt_bin)) < 1e-4
        assert O.rel_l1(gt_mono[b].cpu(), torch.from_numpy(w_gt_mono)) < 2e-5


def test_round_half_even_and_int16_wrap():
    """np.round is half-to-even and astype('int16') wraps: the kernel must do the same on exact values."""
    from m2h import _lib, ops
    dev = torch.device("cuda", 0)
    vals = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 3.49, 32767.4, 32768.0, -32769.0, 40000.2], np.float32)
    full = torch.from_numpy(vals[None]).to(dev).contiguous()
    out, mix = torch.empty((1, vals.size), device=dev), torch.empty((1, vals.size), device=dev)
    _lib.check(_lib.load().m2h_feeder_round_mix(ops._ptr(full), vals.size, 0, ops._ptr(out), ops._ptr(mix), 1, vals.size, 1, 1.0, ops._stream(full)), "x")
    want = np.round(vals.astype(np.float64)).astype(np.int64).astype(np.int16).astype(np.float32) / 32768
    assert np.array_equal(out.cpu().numpy()[0], want)
    assert np.array_equal(mix.cpu().numpy()[0], want)


@pytest.mark.parametrize("L,Lr", [(16000, 16000), (900, 1000), (4000, 2049)])
def test_fftconv_full_matches_numpy_convolution(L, Lr):
    """m2h_fftconv_full (hand-written in-LDS FFTs) against the float64 direct convolution, every output sample, for the feeder's
    lengths and two shorter transforms (other template instances)."""
    from m2h import _lib, ops
    from m2h.audio.feeder import BinauralFeeder, _next_pow2
    dev = torch.device("cuda", 0)
    mono, rir = _inputs(3, 2, L, Lr, 9)
    nfft = max(_next_pow2(L + Lr - 1), 2048)
    fd = BinauralFeeder(dev)
    m, r = torch.from_numpy(mono).to(dev), torch.from_numpy(rir).to(dev)
    full = torch.empty((3, 2, 2, nfft), device=dev)
    xspec = torch.empty((6, 2 * nfft), device=dev)
    _lib.check(_lib.load().m2h_fftconv_full(ops._ptr(m), ops._ptr(r), ops._ptr(fd._twiddle_table(nfft)), ops._ptr(xspec), ops._ptr(full), 6, L, Lr,
                                            nfft.bit_length() - 1, ops._stream(m)), "m2h_fftconv_full")
    got = full.cpu().numpy()
    for b in range(3):
        for s in range(2):
            for ear in range(2):
                want = np.convolve(mono[b, s].astype(np.float64), rir[b, s, :, ear].astype(np.float64))
                err = np.abs(got[b, s, ear, :L + Lr - 1] - want).max()
                assert err <= 2e-6 * np.abs(want).max() * np.sqrt(np.log2(nfft)), (b, s, ear, err, np.abs(want).max())
                assert np.abs(got[b, s, ear, L + Lr - 1:]).max() <= 1e-5 * np.abs(want).max()   # the zero tail of the circular result
