"""CPU: the numpy model of the HIP kernels' index arithmetic (tests/kernel_model.py) reproduces the oracle.
This checks the algorithm the kernels implement (phase-decomposed transposed conv, in-place concat, class
plane as bias, slice/de-slice) without a GPU."""
import numpy as np
import torch

import kernel_model as KM
import m2h_oracle as O
from m2h import synthetic


def test_kernel_model_unet_pair_matches_oracle():
    sd_np = synthetic.make_state_dict(synthetic.passive_shapes(), 3)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
    mixed, tc = synthetic.make_passive_inputs(1, 32, 5)
    mix, tct = torch.from_numpy(mixed), torch.from_numpy(tc)
    with torch.no_grad():
        masks_o, feats_o = O.get_binSepMasks(sd, mix, tct, return_feats=True)
        mono_o = O.convert_bin2mono(sd, masks_o, mix)
    masks_k, feats_k = KM.unet_forward(sd_np, O.ENC_B, O.DEC_B, mixed, target_class=tc)
    for fo, fk in zip(feats_o, feats_k):
        assert O.rel_l1(torch.from_numpy(fk).permute(0, 3, 1, 2), fo) < 1e-5
    assert O.rel_l1(torch.from_numpy(masks_k), masks_o) < 1e-5
    mono_k, _ = KM.unet_forward(sd_np, O.ENC_M, O.DEC_M, mixed, masks=masks_o.numpy())
    assert O.rel_l1(torch.from_numpy(mono_k), mono_o) < 1e-5


def test_kernel_model_fully_convolutional_tm64():
    sd_np = synthetic.make_state_dict(synthetic.passive_shapes(), 4)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
    mixed, tc = synthetic.make_passive_inputs(1, 64, 6)
    with torch.no_grad():
        masks_o = O.get_binSepMasks(sd, torch.from_numpy(mixed), torch.from_numpy(tc))
    masks_k, _ = KM.unet_forward(sd_np, O.ENC_B, O.DEC_B, mixed, target_class=tc)
    assert O.rel_l1(torch.from_numpy(masks_k), masks_o) < 1e-5
