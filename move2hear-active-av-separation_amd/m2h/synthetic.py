"""Seeded synthetic feeder: weights and observations of the reference's shapes.

The Habitat/SoundSpaces simulator and the trained checkpoints are not available (no network, no
data), so every test, fixture and benchmark draws its tensors from here.  Everything is generated
with numpy's PCG64 from an integer seed plus the CRC32 of the tensor's name, so the build container
(where the golden fixtures are produced from the reference) and the GPU box regenerate bit-identical
tensors without shipping 134 MB of weights.

Shapes follow the reference:
  * passive separator state_dict keys/shapes: audio_separation/rl/models/separator_cnn.py:46-52,
    128-135 wrapped by audio_separation/pretrain/passive/policy.py:7-44 (124 entries).
  * observations: audio_separation/config/default.py:130-157 (mixed_bin_audio_mag [512,32,2] =
    log1p(|STFT|) >= 0, target_class in [0, 10]).
"""
import zlib
from collections import OrderedDict

import numpy as np

ENC_CH = [64, 128, 256, 512, 512]
DEC_IN = [512, 1024, 512, 256, 128]


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(name.encode())]))


def unet_shapes(prefix, convert_bin2mono):
    """Ordered (key -> shape) of one encoder+decoder pair, reference key names."""
    shapes = OrderedDict()
    enc = prefix + "_enc.passive_sep_encoder.cnn."
    dec = prefix + "_dec.passive_sep_decoder.cnn."
    cin = 32 if convert_bin2mono else 33
    for i, cout in enumerate(ENC_CH):
        shapes[enc + "%d.0.weight" % i] = (cout, cin, 4, 4)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            shapes[enc + "%d.1.%s" % (i, nm)] = (cout,)
        shapes[enc + "%d.1.num_batches_tracked" % i] = ()
        cin = cout
    nout = 16 if convert_bin2mono else 32
    dec_out = [512, 256, 128, 64, nout]
    for i in range(5):
        shapes[dec + "%d.0.weight" % i] = (DEC_IN[i], dec_out[i], 4, 4)  # ConvTranspose: [Cin,Cout,kh,kw]
        for nm in ("weight", "bias", "running_mean", "running_var"):
            shapes[dec + "%d.1.%s" % (i, nm)] = (dec_out[i],)
        shapes[dec + "%d.1.num_batches_tracked" % i] = ()
    shapes[dec + "5.0.weight"] = (nout, nout, 1, 1)
    shapes[dec + "5.0.bias"] = (nout,)
    return shapes


def passive_shapes():
    """The 124-entry passive-pair state_dict (Move2HearPassiveWoMemoryPolicy)."""
    s = OrderedDict()
    enc_b = unet_shapes("binSep", False)
    enc_m = unet_shapes("bin2mono", True)
    # reference module registration order: binSep_enc, binSep_dec, bin2mono_enc, bin2mono_dec
    for d in (enc_b, enc_m):
        for k, v in d.items():
            if "_enc." in k:
                s[k] = v
        for k, v in d.items():
            if "_dec." in k:
                s[k] = v
    return s


def fill(name, shape, seed):
    """Deterministic, well-conditioned values for one tensor (float32, or int64 for counters)."""
    r = _rng(seed, name)
    if name.endswith("num_batches_tracked"):
        return np.asarray(100, dtype=np.int64)
    if name.endswith("running_var"):
        return r.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if name.endswith("running_mean"):
        return (0.1 * r.standard_normal(size=shape)).astype(np.float32)
    if len(shape) == 1:
        # BN affine weight / bias, conv bias, GRU bias
        if name.endswith("weight"):
            return r.uniform(0.7, 1.3, size=shape).astype(np.float32)
        return (0.05 * r.standard_normal(size=shape)).astype(np.float32)
    # conv / linear / GRU weights: He-style so activations keep O(1) scale through the stack
    if len(shape) == 4:
        if ".passive_sep_decoder.cnn." in name and not name.endswith("5.0.weight"):
            fan_in = shape[0] * 4  # convT 4x4 s2: 4 taps hit each output pixel
        else:
            fan_in = shape[1] * shape[2] * shape[3]
    else:
        fan_in = shape[-1]
    std = np.sqrt(2.0 / fan_in)
    if name.startswith("action_dist") or "state_encoder" in name:
        std = np.sqrt(1.0 / fan_in)  # keep logits / GRU pre-activations O(1): unsaturated softmax and gates
    return (std * r.standard_normal(size=shape)).astype(np.float32)


def make_state_dict(shapes, seed):
    """numpy state dict for an ordered (key -> shape) mapping."""
    return OrderedDict((k, fill(k, shp, seed)) for k, shp in shapes.items())


def make_passive_inputs(batch, tm, seed, n_freq=512):
    """mixed_bin_audio_mag = log1p(|complex normal| * gain) and target_class, like the feeder's
    log1p(abs(STFT)) (audio_separation/pretrain/datasets/dataset.py:228)."""
    r = _rng(seed, "passive_inputs_%d_%d" % (batch, tm))
    re = r.standard_normal(size=(batch, n_freq, tm, 2)).astype(np.float32)
    im = r.standard_normal(size=(batch, n_freq, tm, 2)).astype(np.float32)
    gain = np.exp(r.uniform(-2.0, 1.0, size=(batch, n_freq, 1, 1))).astype(np.float32)
    mag = np.sqrt(re * re + im * im) * gain
    mixed = np.log1p(mag).astype(np.float32)
    target = r.integers(0, 11, size=(batch, 1)).astype(np.int64)
    return mixed, target


# ------------------------------------------------------------------------------------------------
# RL policy (Move2HearPolicy, audio_separation/rl/ppo/policy.py:276-326) -- 158-entry state_dict
# ------------------------------------------------------------------------------------------------
def policy_shapes(extra_rgb=False, extra_depth=True, hidden=512, n_actions=3):
    """Ordered (key -> shape) of the full RL policy in the reference's registration order:
    pol_net (visual, bin, monoNmonoFromMem encoders, GRU), action_dist, critic, 4 separator modules,
    acoustic_mem (ddppo variant: no BN; rl/models/memory_nets.py:11-16)."""
    s = OrderedDict()
    cin_v = (0 if extra_rgb else 3) + (0 if extra_depth else 1)
    v = "pol_net.visual_encoder.cnn."
    s[v + "0.weight"], s[v + "0.bias"] = (32, cin_v, 8, 8), (32,)
    s[v + "2.weight"], s[v + "2.bias"] = (64, 32, 4, 4), (64,)
    s[v + "4.weight"], s[v + "4.bias"] = (32, 64, 3, 3), (32,)
    s[v + "6.weight"], s[v + "6.bias"] = (hidden, 32 * 12 * 12), (hidden,)
    for enc in ("bin_encoder", "monoNmonoFromMem_encoder"):
        a = "pol_net.%s.cnn." % enc
        s[a + "0.weight"], s[a + "0.bias"] = (32, 32, 8, 8), (32,)
        s[a + "2.weight"], s[a + "2.bias"] = (64, 32, 4, 4), (64,)
        s[a + "4.weight"], s[a + "4.bias"] = (32, 64, 2, 2), (32,)
        s[a + "7.weight"], s[a + "7.bias"] = (hidden, 32), (hidden,)
    g = "pol_net.state_encoder.rnn."
    s[g + "weight_ih_l0"], s[g + "weight_hh_l0"] = (3 * hidden, 3 * hidden), (3 * hidden, hidden)
    s[g + "bias_ih_l0"], s[g + "bias_hh_l0"] = (3 * hidden,), (3 * hidden,)
    s["action_dist.linear.weight"], s["action_dist.linear.bias"] = (n_actions, hidden), (n_actions,)
    s["critic.fc.weight"], s["critic.fc.bias"] = (1, hidden), (1,)
    for k, shp in passive_shapes().items():
        s[k] = shp
    s["acoustic_mem.cnn.0.weight"] = (32, 32, 3, 3)
    s["acoustic_mem.cnn.2.weight"] = (16, 32, 3, 3)
    return s


def make_rl_observations(n, seed, tm=32, n_freq=512):
    """One batch of observations with the reference's sensor shapes (config/default.py:130-157):
    rgb uint8-valued float, depth in [0,1], mixed/gt spectrogram tensors, target_class.
    gt_bin_comps = per source [mag_l, phase_l, mag_r, phase_r] x 2 sources, gt_mono_comps = [mag, phase] x 2."""
    r = _rng(seed, "rl_obs_%d_%d" % (n, tm))
    mixed, tc = make_passive_inputs(n, tm, seed, n_freq)
    obs = {
        "rgb": r.integers(0, 256, size=(n, 128, 128, 3)).astype(np.float32),
        "depth": r.uniform(0.0, 1.0, size=(n, 128, 128, 1)).astype(np.float32),
        "mixed_bin_audio_mag": mixed,
        "target_class": tc,
    }
    gb = np.empty((n, n_freq, tm, 8), np.float32)
    gm = np.empty((n, n_freq, tm, 4), np.float32)
    for j in range(0, 8, 2):
        gb[..., j] = np.log1p(np.abs(r.standard_normal(size=(n, n_freq, tm))).astype(np.float32))
        gb[..., j + 1] = r.uniform(-np.pi, np.pi, size=(n, n_freq, tm)).astype(np.float32)
    for j in range(0, 4, 2):
        gm[..., j] = np.log1p(np.abs(r.standard_normal(size=(n, n_freq, tm))).astype(np.float32))
        gm[..., j + 1] = r.uniform(-np.pi, np.pi, size=(n, n_freq, tm)).astype(np.float32)
    obs["gt_bin_comps"] = gb
    obs["gt_mono_comps"] = gm
    return obs
