"""CPU: the oracle restatement (oracle/m2h_oracle.py) against fixtures produced by the reference
itself (oracle/gen_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import torch

import m2h_oracle as O
from m2h import synthetic


def _sd(seed):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in
            synthetic.make_state_dict(synthetic.passive_shapes(), seed).items()}


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_unet_tm32_matches_reference(golden_dir):
    g = _load(golden_dir, "unet_tm32.npz")
    sd = _sd(int(g["seed_w"]))
    mixed, tc = synthetic.make_passive_inputs(int(g["B"]), 32, int(g["seed_x"]))
    # regenerable inputs/weights are bit-identical to what the fixture was made from
    assert np.allclose([mixed.astype(np.float64).sum(), np.abs(mixed).astype(np.float64).sum()], g["input_checksum"], rtol=0, atol=0)
    mix, tct = torch.from_numpy(mixed), torch.from_numpy(tc)
    with torch.no_grad():
        masks, feats = O.get_binSepMasks(sd, mix, tct, return_feats=True)
        mono, feats_m = O.convert_bin2mono(sd, masks, mix, return_feats=True)
    # same torch build, same op sequence -> expect agreement far below the 1e-3 rel-L1 contract
    assert O.rel_l1(masks, torch.from_numpy(g["masks"])) < 1e-6
    assert O.rel_l1(mono, torch.from_numpy(g["mono"])) < 1e-6
    assert torch.allclose(feats[4].reshape(2, -1), torch.from_numpy(g["bottleneck_binSep"]), atol=1e-5)
    assert torch.allclose(feats_m[4].reshape(2, -1), torch.from_numpy(g["bottleneck_bin2mono"]), atol=1e-5)
    assert torch.allclose(feats[0], torch.from_numpy(g["binSep_skip3_full"]), atol=1e-5)
    for i in range(4):  # skips are (e4,e3,e2,e1)
        s = feats[3 - i].double()
        st = g["binSep_skip%d_stats" % i]
        assert abs(s.mean().item() - st[0]) < 1e-6 and abs(s.abs().mean().item() - st[1]) < 1e-6


def test_unet_tm256_fully_convolutional(golden_dir):
    g = _load(golden_dir, "unet_tm256.npz")
    sd = _sd(int(g["seed_w"]))
    mixed, tc = synthetic.make_passive_inputs(int(g["B"]), 256, int(g["seed_x"]))
    with torch.no_grad():
        masks, mono = O.passive_pair(sd, torch.from_numpy(mixed), torch.from_numpy(tc))
    assert masks.shape == (1, 512, 256, 2) and mono.shape == (1, 512, 256, 1)
    assert O.rel_l1(masks, torch.from_numpy(g["masks"])) < 1e-6
    assert O.rel_l1(mono, torch.from_numpy(g["mono"])) < 1e-6


def test_slice_deslice_roundtrip():
    x = torch.randn(3, 512, 32, 2)
    assert torch.equal(O.deslice_freq(O.slice_freq(x)), x)
    s = O.slice_freq(x)
    # channel c*16+s, row h  <->  frequency s*32+h (separator_cnn.py:89-90)
    assert s[1, 1 * 16 + 5, 7, 9] == x[1, 5 * 32 + 7, 9, 1]


def test_passive_shapes_match_reference_init(golden_dir):
    with open(os.path.join(golden_dir, "passive_init_seed0.json")) as f:
        rec = json.load(f)["params"]
    shapes = synthetic.passive_shapes()
    assert list(shapes.keys()) == list(rec.keys())
    for k, shp in shapes.items():
        assert list(shp) == rec[k]["shape"], k
    assert len(shapes) == 124
