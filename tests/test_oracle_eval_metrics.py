"""CPU: the waveform-metric restatement (oracle.np_waveform_metrics, following common/eval_metrics.py:12-229) against the
reference's own ``evaluate()`` (tests/golden/eval_metrics.npz, oracle/gen_golden.py::gen_eval_metrics), against closed forms that
do not share its code path, and the drop-in module's constants against the reference's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import m2h_oracle as O  # noqa: E402


def _signals(seed, L=16000):
    r = np.random.default_rng(seed)
    s = r.standard_normal(L) * 0.1 + 0.02
    other = r.standard_normal(L) * 0.08
    est = 0.8 * s + 0.05 * r.standard_normal(L) + 0.01
    mix = np.stack([s + other, 0.9 * s + 1.1 * other - 0.03])
    return s, est, mix


def eval_clips(S, L, seed):
    """The fixture's seeded clips (same generator as oracle/gen_golden.py::eval_clips)."""
    r = np.random.default_rng(seed)
    t = np.arange(L) / 16000.0
    ref_ = np.stack([0.2 * np.sin(2 * np.pi * r.uniform(200, 3000) * t) + 0.05 * r.standard_normal(L) + 0.01 for _ in range(S)])
    other = 0.1 * r.standard_normal((S, L))
    est = ref_ * r.uniform(0.5, 1.5, (S, 1)) + 0.03 * r.standard_normal((S, L)) - 0.02
    ml, mr = ref_ + other, 0.8 * ref_ + 1.2 * other + 0.05
    return [a.astype(np.float32) for a in (ref_, est, ml, mr)]


def test_metrics_match_the_reference_evaluate_fixture():
    """All 11 scores of the reference's evaluate() per clip, float32 (what it gets from librosa.istft) and float64 inputs."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "eval_metrics.npz"))
    assert tuple(str(x) for x in g["order"]) == tuple(O.BSS_METRIC_ORDER)
    for L in (16000, 4097):
        ref_, est, ml, mr = eval_clips(int(g["S"]), L, int(g["seed"]))
        for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
            want = g["scores_%s_L%d" % (tag, L)]
            for c in range(len(ref_)):
                got = O.np_waveform_metrics(ref_[c], est[c], np.stack([ml[c], mr[c]]), dtype=dt)
                # same numpy operations in the same order: equal to the last bits (si_sir / si_sar included)
                assert np.allclose(got, want[c], rtol=1e-6, atol=1e-6), (L, tag, c, got, want[c])


def test_metrics_match_closed_forms_in_float64():
    s, est, mix = _signals(0)
    m = O.np_waveform_metrics(s, est, mix, dtype=np.float64)
    sc, ec = s - s.mean(), est - est.mean()
    mc = ((mix[0] - mix[0].mean()) + (mix[1] - mix[1].mean())) / 2

    def sisdr(x):
        a = sc @ x / (sc @ sc)
        return 10 * np.log10((a * a * (sc @ sc)) / ((x - a * sc) @ (x - a * sc)))

    def snr(x):
        return 10 * np.log10((sc @ sc) / ((x - sc) @ (x - sc)))

    assert abs(m[0] - sisdr(ec)) < 1e-9
    assert abs(m[4] - snr(ec)) < 1e-9
    assert abs(m[6] - (sisdr(ec) - sisdr(mc))) < 1e-9
    assert abs(m[8] - (snr(ec) - snr(mc))) < 1e-9
    a = sc @ ec / (sc @ sc)
    assert abs(m[3] - (snr(ec) + 10 * np.log10(a * a))) < 1e-9
    assert abs(m[5] - (-10 * np.log10((1 - 1 / a) ** 2))) < 1e-9
    assert m[1] > 60 and np.isfinite(m[2])  # single-source SIR divides by rounding noise: only its magnitude is meaningful


def test_scale_invariance_and_float32_agreement():
    s, est, mix = _signals(1)
    m1 = O.np_waveform_metrics(s, est, mix, dtype=np.float64)
    m2 = O.np_waveform_metrics(s, 3.7 * est, mix, dtype=np.float64)
    assert abs(m1[0] - m2[0]) < 1e-9 and abs(m1[6] - m2[6]) < 1e-9      # SI-SDR(i) ignore the estimate's scale
    assert abs(m1[4] - m2[4]) > 1.0                                        # SNR does not
    m32 = O.np_waveform_metrics(s, est, mix, dtype=np.float32)
    for j in (0, 3, 4, 5, 6, 7, 8):
        assert abs(m32[j] - m1[j]) < 2e-3, (O.BSS_METRIC_ORDER[j], m32[j], m1[j])


def test_metric_names_follow_reference_module():
    sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))
    import importlib.util
    spec = importlib.util.find_spec("m2h.common.eval_metrics")
    src = open(spec.origin).read()
    # constants of eval_metrics.py:5-9
    assert "HOP_LENGTH = 512" in src and "RECONSTRUCTED_SIGNAL_LENGTH = 16000" in src and "EPS = 1e-13" in src
    for name in ("STFT_L2_distance", "istft", "compute_waveform_quality"):
        assert "def %s(" % name in src
    assert tuple(O.BSS_METRIC_ORDER) == ("si_sdr", "si_sir", "si_sar", "sd_sdr", "snr", "srr", "si_sdri", "sd_sdri", "snri", "si_siri", "si_sari")
