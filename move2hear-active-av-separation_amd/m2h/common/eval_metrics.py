"""Evaluation metrics: drop-in for audio_separation/common/eval_metrics.py (same function names and return conventions).

  STFT_L2_distance(mixed_audio, pred_binSepMasks, gt_bin_comps, pred_mono, gt_mono_comps)      reference :306-366
  istft(mag_l, phase_l, mag_r=None, phase_r=None)                                              reference :232-251
  compute_waveform_quality(pred_n_gt_spects, eval_metrics_to_compute)                          reference :256-303

The reference runs these on the CPU (librosa istft + numpy BSS-eval helpers copied from nussl); here the inverse STFT is the
DFT-as-GEMM + overlap-add of m2h.audio.stft and the metric sums run in one HIP kernel per clip (m2h_bss_metrics).  Inputs are
GPU tensors; `compute_waveform_quality` returns Python floats like the reference (one host sync at the end).
"""
import torch

from .. import ops
from ..audio.stft import ISTFT

HOP_LENGTH = 512
RECONSTRUCTED_SIGNAL_LENGTH = 16000
EPS = 1e-13
NAME_OF_ALL_QUALITY_METRICS = ['env', 'si_sdr', 'si_sir', 'si_sar', 'sd_sdr', 'snr', 'srr', 'si_sdri', 'sd_sdri', 'snri',
                               "si_siri", "si_sari", "sdr", "sir", "sar"]
# column order of ops.bss_metrics / m2h_bss_metrics
METRIC_ORDER = ("si_sdr", "si_sir", "si_sar", "sd_sdr", "snr", "srr", "si_sdri", "sd_sdri", "snri", "si_siri", "si_sari")

_istft_cache = {}


def _istft_engine(device):
    key = (device.type, device.index)
    if key not in _istft_cache:
        _istft_cache[key] = ISTFT(device)
    return _istft_cache[key]


def STFT_L2_distance(mixed_audio, pred_binSepMasks, gt_bin_comps, pred_mono, gt_mono_comps):
    """-> (bin_stft_l2_dist [N,1], mono_stft_l2_dist [N,1]); device tensors (the reference moves them to the CPU)."""
    d_bin = ops.stft_l2(pred_binSepMasks.contiguous(), gt_bin_comps.contiguous(), 2, mix=mixed_audio.contiguous())
    d_mono = ops.stft_l2(pred_mono.contiguous(), gt_mono_comps.contiguous(), 1)
    return d_bin, d_mono


def istft(mag_l, phase_l, mag_r=None, phase_r=None):
    """Spectrogram(s) [512, T] (magnitude, phase) -> list of waveforms [16000] (one or two channels), as the reference."""
    eng = _istft_engine(mag_l.device)

    def one(mag, phase):
        m = mag.reshape(1, mag.shape[0], mag.shape[1], 1).contiguous().float()
        ph = phase.reshape(1, phase.shape[0], phase.shape[1], 1).contiguous().float()
        return eng(m, ph, length=RECONSTRUCTED_SIGNAL_LENGTH)[0]

    signal = [one(mag_l, phase_l)]
    if mag_r is not None:
        assert phase_r is not None
        signal.append(one(mag_r, phase_r))
    return signal


def waveform_metrics(gt_mono_mag, gt_mono_phase, pred, mixed_bin_audio_mag, mixed_bin_audio_phase):
    """Batched core of compute_waveform_quality: BHWC tensors [B,512,T,C] -> [B, 11] metrics of `pred` (magnitude, the GT phase
    is used for its inverse transform as in the reference :283-290) against the GT mono waveform and the binaural mixture."""
    eng = _istft_engine(gt_mono_mag.device)
    c = lambda t: t.contiguous().float()  # noqa: E731
    gt = eng(c(gt_mono_mag), c(gt_mono_phase), RECONSTRUCTED_SIGNAL_LENGTH, 0)
    est = eng(c(pred), c(gt_mono_phase), RECONSTRUCTED_SIGNAL_LENGTH, 0)
    mm, mp = c(mixed_bin_audio_mag), c(mixed_bin_audio_phase)
    ml = eng(mm, mp, RECONSTRUCTED_SIGNAL_LENGTH, 0)
    mr = eng(mm, mp, RECONSTRUCTED_SIGNAL_LENGTH, 1)
    return ops.bss_metrics(gt, est, ml, mr)


def compute_waveform_quality(pred_n_gt_spects, eval_metrics_to_compute):
    """Waveform-level quality of `pred_mono` and `pred_monoFromMem` for the first element of the batch, like the reference
    (which supports one eval process).  pred_n_gt_spects: dict of [B,512,T,C] tensors with the reference's keys."""
    d = pred_n_gt_spects
    out = {"mono": {}, "monoFromMem": {}}
    sl = lambda t: t[:1]  # noqa: E731
    vals = {}
    for name, key in (("mono", "pred_mono"), ("monoFromMem", "pred_monoFromMem")):
        vals[name] = waveform_metrics(sl(d["gt_mono_mag"]), sl(d["gt_mono_phase"]), sl(d[key]), sl(d["mixed_bin_audio_mag"]),
                                      sl(d["mixed_bin_audio_phase"]))
    host = {k: v[0].tolist() for k, v in vals.items()}
    for metric in eval_metrics_to_compute:
        assert metric in NAME_OF_ALL_QUALITY_METRICS, "doesn't support computation of this metric"
        if metric not in METRIC_ORDER:
            raise NotImplementedError("m2h eval_metrics: %r is listed by the reference but never computed by it either" % metric)
        j = METRIC_ORDER.index(metric)
        out["mono"][metric] = host["mono"][j]
        out["monoFromMem"][metric] = host["monoFromMem"][j]
    return out
