"""CPU: the oracle's passive training step (train-mode BN, losses, Adam, D11) against two steps of the reference."""
import os

import numpy as np
import torch

import m2h_oracle as O
from m2h import synthetic


def test_passive_train_two_steps_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "passive_train.npz"))
    sd = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in synthetic.make_state_dict(synthetic.passive_shapes(), int(g["seed_w"])).items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    buffers = {k: v for k, v in sd.items() if "running_" in k}
    B = int(g["B"])
    mixed, tc = synthetic.make_passive_inputs(B, 32, int(g["seed_x"]))
    gen = torch.Generator().manual_seed(int(g["gt_seed"]))
    batch = {"mixed_bin_audio_mag": torch.from_numpy(mixed), "target_class": torch.from_numpy(tc),
             "gt_bin_mag": torch.rand(B, 512, 32, 2, generator=gen) * 2, "gt_mono_mag": torch.rand(B, 512, 32, 1, generator=gen) * 2}
    opt = None
    for step in range(2):
        b, m, opt = O.passive_train_step(params, buffers, batch, opt_state=opt)
        assert abs(b.item() - g["losses"][step][0]) < 2e-5 and abs(m.item() - g["losses"][step][1]) < 2e-5
    n = 0
    for key in g.files:
        if key.startswith("post."):
            k = key[5:]
            mine = (params[k] if k in params else buffers[k]).detach()
            ref = torch.from_numpy(g[key])
            if "running_" in k:
                assert torch.allclose(mine, ref, rtol=1e-4, atol=1e-6), k
            else:
                pre = torch.from_numpy(np.asarray(synthetic.fill(k, ref.shape, int(g["seed_w"]))))
                bad = ((mine - pre) - (ref - pre)).abs().gt(2e-4).float().mean().item()  # two Adam steps of lr 5e-4
                assert bad < 0.01, (k, bad)
            n += 1
    assert n > 40
