"""GPU: waveform quality metrics (m2h_bss_metrics, m2h.common.eval_metrics) against the numpy restatement of
common/eval_metrics.py (oracle.np_waveform_metrics + np_istft)."""
import numpy as np
import pytest
import torch

import m2h_oracle as O

pytestmark = pytest.mark.gpu

WELL = (0, 3, 4, 5, 6, 7, 8)  # si_sdr, sd_sdr, snr, srr, si_sdri, sd_sdri, snri (si_sir / si_sar divide by rounding noise)


def _clips(S, L, seed):
    r = np.random.default_rng(seed)
    t = np.arange(L) / 16000.0
    ref = np.stack([0.2 * np.sin(2 * np.pi * r.uniform(200, 3000) * t) + 0.05 * r.standard_normal(L) + 0.01 for _ in range(S)])
    other = 0.1 * r.standard_normal((S, L))
    est = ref * r.uniform(0.5, 1.5, (S, 1)) + 0.03 * r.standard_normal((S, L)) - 0.02
    ml, mr = ref + other, 0.8 * ref + 1.2 * other + 0.05
    return [a.astype(np.float32) for a in (ref, est, ml, mr)]


@pytest.mark.parametrize("L", [16000, 4097])
def test_bss_metrics_kernel_matches_oracle(L):
    from m2h import ops
    dev = torch.device("cuda", 0)
    ref, est, ml, mr = _clips(5, L, 11)
    out = ops.bss_metrics(*[torch.from_numpy(a).to(dev) for a in (ref, est, ml, mr)]).cpu().numpy()
    assert out.shape == (5, 11)
    for c in range(5):
        want = O.np_waveform_metrics(ref[c], est[c], np.stack([ml[c], mr[c]]), dtype=np.float64)
        for j in WELL:
            assert abs(out[c, j] - want[j]) < 2e-3, (c, O.BSS_METRIC_ORDER[j], out[c, j], want[j])   # dB
        assert out[c, 1] > 60 and np.isfinite(out[c, 2])


@pytest.mark.parametrize("L", [16000, 4097])
def test_bss_metrics_kernel_matches_the_reference_evaluate_fixture(L, golden_dir):
    """m2h_bss_metrics against the scores of the reference's own evaluate() (tests/golden/eval_metrics.npz): the well-conditioned
    metrics to 2e-3 dB of the float64 reference values; the float32 reference run itself sits that far from them."""
    import os
    from m2h import ops
    from test_oracle_eval_metrics import eval_clips
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(golden_dir, "eval_metrics.npz"))
    ref, est, ml, mr = eval_clips(int(g["S"]), L, int(g["seed"]))
    out = ops.bss_metrics(*[torch.from_numpy(a).to(dev) for a in (ref, est, ml, mr)]).cpu().numpy()
    want = g["scores_f64_L%d" % L]
    for j in WELL:
        assert np.abs(out[:, j] - want[:, j]).max() < 2e-3, (O.BSS_METRIC_ORDER[j], out[:, j], want[:, j])
    assert np.abs(g["scores_f32_L%d" % L][:, WELL] - want[:, WELL]).max() < 5e-3


def test_compute_waveform_quality_end_to_end():
    """spectrograms -> iSTFT (GT phase) -> metrics, as eval_metrics.compute_waveform_quality, vs numpy istft + oracle metrics."""
    from m2h.audio.stft import STFT
    from m2h.common import eval_metrics as EM
    dev = torch.device("cuda", 0)
    r = np.random.default_rng(3)
    L = 16000
    t = np.arange(L) / 16000.0
    src = (0.25 * np.sin(2 * np.pi * 440 * t) * (1 + 0.3 * np.sin(2 * np.pi * 3 * t)) + 0.02 * r.standard_normal(L)).astype(np.float32)
    other = (0.15 * r.standard_normal(L)).astype(np.float32)
    mix = np.stack([src + other, 0.7 * src + 1.1 * other]).astype(np.float32)
    stft = STFT(dev)
    gt_mag, gt_ph = stft(torch.from_numpy(src[None, None]).to(dev), mode=0, want_phase=True)        # [1,512,32,1]
    mx_mag, mx_ph = stft(torch.from_numpy(mix[None]).to(dev), mode=0, want_phase=True)              # [1,512,32,2]
    pred_mono = gt_mag * 0.9 + 0.02 * mx_mag[..., :1]
    pred_mem = gt_mag * 0.97 + 0.005 * mx_mag[..., 1:]
    got = EM.compute_waveform_quality({"mixed_bin_audio_mag": mx_mag, "mixed_bin_audio_phase": mx_ph, "gt_mono_mag": gt_mag,
                                       "gt_mono_phase": gt_ph, "pred_mono": pred_mono, "pred_monoFromMem": pred_mem},
                                      ["si_sdr", "si_sdri", "snr"])
    cpu = lambda x: x.cpu().numpy()  # noqa: E731

    def wave(mag, ph, ch=0):
        return O.np_istft(cpu(mag)[0, :, :, ch] * np.exp(1j * cpu(ph)[0, :, :, ch]))

    gt_w = wave(gt_mag, gt_ph)
    mix_w = np.stack([wave(mx_mag, mx_ph, 0), wave(mx_mag, mx_ph, 1)])
    for name, pred in (("mono", pred_mono), ("monoFromMem", pred_mem)):
        want = O.np_waveform_metrics(gt_w, wave(pred, gt_ph), mix_w, dtype=np.float64)
        for metric in ("si_sdr", "si_sdri", "snr"):
            j = O.BSS_METRIC_ORDER.index(metric)
            assert abs(got[name][metric] - want[j]) < 5e-3, (name, metric, got[name][metric], want[j])
    assert got["monoFromMem"]["si_sdr"] > got["mono"]["si_sdr"]   # the cleaner estimate scores higher
    # istft() list convention of the reference (:232-251)
    sig = EM.istft(mx_mag[0, :, :, 0], mx_ph[0, :, :, 0], mag_r=mx_mag[0, :, :, 1], phase_r=mx_ph[0, :, :, 1])
    assert len(sig) == 2 and sig[0].shape == (16000,)
    assert np.abs(cpu(sig[1]) - mix_w[1]).max() < 2e-4


def test_stft_l2_distance_signature(golden_dir):
    import os
    from m2h import synthetic
    from m2h.common import eval_metrics as EM
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(golden_dir, "rl_scalars.npz"))
    obs = {k: torch.from_numpy(v).float().to(dev) for k, v in synthetic.make_rl_observations(5, int(g["l2_seed_x"])).items()}
    pm, pmono = torch.from_numpy(g["l2_masks"]).to(dev), torch.from_numpy(g["l2_mono"]).to(dev)
    d_bin, d_mono = EM.STFT_L2_distance(obs["mixed_bin_audio_mag"], pm, obs["gt_bin_comps"], pmono, obs["gt_mono_comps"])
    assert torch.allclose(d_bin.cpu(), torch.from_numpy(g["stft_l2_bin"]), rtol=2e-5)
    assert torch.allclose(d_mono.cpu(), torch.from_numpy(g["stft_l2_mono"]), rtol=2e-5)
