"""Generates tests/golden/*.npz by running the REFERENCE itself (build container only).

    python oracle/gen_golden.py [--only unet|init|...]

The reference python is imported from /root/reference through oracle/_ref_import.py; nothing of it
is copied.  What is committed are data fixtures only: expected outputs (and small checksums of the
regenerable inputs/weights so RNG drift is detected) for seeded synthetic inputs and weights from
``m2h.synthetic``.  Every fixture records the torch/numpy versions that produced it.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "move2hear-active-av-separation_amd"))

from _ref_import import FakeObsSpace, FakeActionSpace, load_reference  # noqa: E402
from m2h import synthetic  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
META = {"torch": torch.__version__, "numpy": np.__version__}


def _t(sd_np):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}


def _checksum(a):
    a = np.asarray(a, dtype=np.float64)
    return [float(a.sum()), float(np.abs(a).sum())]


def _stats(t):
    t = t.detach().double()
    flat = t.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 8).long()
    return np.concatenate([[t.mean().item(), t.abs().mean().item(), t.std().item()], flat[idx].numpy()])


def build_ref_passive(ref, seed, tm=32):
    pol = ref["passive_policy"].Move2HearPassiveWoMemoryPolicy(FakeObsSpace(tm))
    sd = synthetic.make_state_dict(synthetic.passive_shapes(), seed)
    missing = pol.load_state_dict(_t(sd), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    pol.eval()
    return pol, sd


def gen_unet_tm32(ref):
    """G1/G2: get_binSepMasks / convert_bin2mono at the reference-native 512x32, eval-mode BN."""
    seed_w, seed_x, B = 1, 11, 2
    pol, sd = build_ref_passive(ref, seed_w)
    mixed, tc = synthetic.make_passive_inputs(B, 32, seed_x)
    obs = {"mixed_bin_audio_mag": torch.from_numpy(mixed), "target_class": torch.from_numpy(tc)}
    with torch.no_grad():
        bott, skips = pol.binSep_enc(obs)
        masks = pol.get_binSepMasks(obs)
        mono = pol.convert_bin2mono(masks, mixed_audio=obs["mixed_bin_audio_mag"])
        bott_m, skips_m = pol.bin2mono_enc(masks, mixed_audio=obs["mixed_bin_audio_mag"])
    out = {
        "seed_w": seed_w, "seed_x": seed_x, "B": B, "tm": 32,
        "masks": masks.contiguous().numpy(), "mono": mono.contiguous().numpy(),
        "bottleneck_binSep": bott.numpy(), "bottleneck_bin2mono": bott_m.numpy(),
        "input_checksum": np.array(_checksum(mixed)),
        "weight_checksum": np.array(_checksum(sd["binSep_enc.passive_sep_encoder.cnn.0.0.weight"])),
    }
    # skips come reversed (e4,e3,e2,e1): store stats for each
    for i, s in enumerate(skips):
        out["binSep_skip%d_stats" % i] = _stats(s)
    for i, s in enumerate(skips_m):
        out["bin2mono_skip%d_stats" % i] = _stats(s)
    out["binSep_skip3_full"] = skips[3].numpy()  # e1: [B,64,16,16]
    np.savez_compressed(os.path.join(GOLD, "unet_tm32.npz"), meta=json.dumps(META), **out)
    print("unet_tm32: masks", masks.shape, float(masks.abs().mean()), "mono", mono.shape, float(mono.abs().mean()))


def _ref_fullyconv(pol_enc, pol_dec, x_nchw):
    """Drives the reference's own conv stacks fully-convolutionally: only the reshape
    glue at separator_cnn.py:108/:154 pins Tm=32, the nn.Sequential stages do not."""
    feats = []
    out = x_nchw
    for m in pol_enc.passive_sep_encoder.cnn:
        out = m(out)
        feats.append(out)
    skips = feats[:-1][::-1]
    dec = pol_dec.passive_sep_decoder.cnn
    out = feats[-1]
    for idx, m in enumerate(dec):
        if idx == 0 or idx == len(dec) - 1:
            out = m(out)
        else:
            out = m(torch.cat((out, skips[idx - 1]), dim=1))
    return out, feats


def gen_unet_tm256(ref):
    """512x256 throughput shape: reference conv stacks driven directly, B=1."""
    seed_w, seed_x, B, tm = 1, 12, 1, 256
    pol, sd = build_ref_passive(ref, seed_w)
    mixed, tc = synthetic.make_passive_inputs(B, tm, seed_x)
    mix = torch.from_numpy(mixed)
    with torch.no_grad():
        # input glue exactly as separator_cnn.py:85-99, restated with torch ops on the reference side
        x = mix.permute(0, 3, 1, 2)
        x = x.reshape(B, 2, 16, 32, tm).reshape(B, 32, 32, tm)
        plane = (torch.from_numpy(tc).float() + 1).reshape(B, 1, 1, 1).expand(B, 1, 32, tm)
        out, feats = _ref_fullyconv(pol.binSep_enc, pol.binSep_dec, torch.cat((x, plane), 1))
        masks = out.reshape(B, 2, 16, 32, tm).reshape(B, 2, 512, tm).permute(0, 2, 3, 1).contiguous()
        xm = torch.log1p(torch.clamp(masks * (torch.exp(mix) - 1), min=0))
        xm = xm.permute(0, 3, 1, 2).reshape(B, 2, 16, 32, tm).reshape(B, 32, 32, tm)
        outm, featsm = _ref_fullyconv(pol.bin2mono_enc, pol.bin2mono_dec, xm)
        mono = outm.reshape(B, 1, 16, 32, tm).reshape(B, 1, 512, tm).permute(0, 2, 3, 1).contiguous()
    out = {"seed_w": seed_w, "seed_x": seed_x, "B": B, "tm": tm,
           "masks": masks.numpy().astype(np.float32), "mono": mono.numpy().astype(np.float32),
           "bottleneck_binSep_stats": _stats(feats[-1]), "bottleneck_bin2mono_stats": _stats(featsm[-1])}
    np.savez_compressed(os.path.join(GOLD, "unet_tm256.npz"), meta=json.dumps(META), **out)
    print("unet_tm256: masks", masks.shape, float(masks.abs().mean()), "mono", float(mono.abs().mean()))


def gen_init(ref):
    """Default-init parity: torch.manual_seed(0) (config SEED default, config/default.py:16) then
    construct the passive policy; store per-parameter checksums."""
    torch.manual_seed(0)
    pol = ref["passive_policy"].Move2HearPassiveWoMemoryPolicy(FakeObsSpace(32))
    rec = {}
    for k, v in pol.state_dict().items():
        a = v.detach().double().reshape(-1)
        rec[k] = {"shape": list(v.shape), "sum": float(a.sum()), "abssum": float(a.abs().sum()),
                  "head": [float(z) for z in a[:4]]}
    with open(os.path.join(GOLD, "passive_init_seed0.json"), "w") as f:
        json.dump({"meta": META, "params": rec}, f, indent=0)
    print("init: %d entries" % len(rec))


GENS = {"unet_tm32": gen_unet_tm32, "unet_tm256": gen_unet_tm256, "init": gen_init}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    ref = load_reference()
    torch.set_num_threads(8)
    for name, fn in GENS.items():
        if args.only and args.only not in name:
            continue
        fn(ref)


if __name__ == "__main__":
    main()
